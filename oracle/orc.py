"""ctypes front-end for the CHECKER libraries (test infrastructure only).

  Orc  -> oracle/liborc.so            plain-C restatement of /root/reference/llcomp.hpp
  Ref  -> oracle/_ref/libllcomp_ref.so  the real reference header compiled in place (may be absent)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORC_PATH = os.path.join(_HERE, "liborc.so")
REF_PATH = os.path.join(_HERE, "_ref", "libllcomp_ref.so")
REF_SMALL_PATH = os.path.join(_HERE, "_ref", "libllcomp_ref_small.so")  # the reference built with LargeModel = false

OK, BAD_MAGIC, BAD_EXPONENT, TRUNCATED, BAD_ARGS, NOMEM = range(6)

_u8p = C.POINTER(C.c_uint8)
_i16p = C.POINTER(C.c_int16)
_u16p = C.POINTER(C.c_uint16)


def _p(a, t):
    return a.ctypes.data_as(t)


# ---- deterministic input generators: llcomp_amd/synth.py (re-exported for the fixtures and tests) ----------
import sys as _sys

_sys.path.insert(0, os.path.dirname(_HERE))
from llcomp_amd.synth import GENERATORS, gen_checker, gen_g1, gen_g2, gen_g3, gen_mid  # noqa: E402,F401


def fnv1a64(b: bytes) -> int:
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


class Orc:
    """The plain-C restatement."""

    def __init__(self, path=ORC_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = self.lib = C.CDLL(path)
        L.orc_compress_image.restype = C.c_long
        L.orc_compress_image.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.POINTER(_u8p)]
        L.orc_compress_sliced.restype = C.c_long
        L.orc_compress_sliced.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_u8p)]
        L.orc_decompress.restype = C.c_int
        L.orc_decompress.argtypes = [_u8p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_encode_rect.restype = C.c_long
        L.orc_encode_rect.argtypes = [_i16p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_u8p)]
        L.orc_decode_rect.restype = C.c_int
        L.orc_decode_rect.argtypes = [_u8p, C.c_size_t, _i16p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_model_rect.restype = None
        L.orc_model_rect.argtypes = [_i16p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _u16p, _i16p]
        L.orc_forward_rct.restype = None
        L.orc_forward_rct.argtypes = [_u8p, C.c_long, C.c_int, _i16p]
        L.orc_inverse_rct.restype = None
        L.orc_inverse_rct.argtypes = [_i16p, C.c_long, C.c_int, _u8p]
        L.orc_slice_count.restype = C.c_long
        L.orc_slice_count.argtypes = [C.c_int] * 6
        L.orc_fnv1a64.restype = C.c_uint64
        L.orc_fnv1a64.argtypes = [_u8p, C.c_size_t]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_set_small_model.restype = None
        L.orc_set_small_model.argtypes = [C.c_int]
        L.orc_carry_stats.restype = None
        L.orc_carry_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_int]
        for f in ("orc_quant11", "orc_quant5", "orc_state_p"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_int]
        L.orc_state_next.restype = C.c_int
        L.orc_state_next.argtypes = [C.c_int, C.c_int]
        L.orc_median.restype = C.c_int
        L.orc_median.argtypes = [C.c_int] * 3

    def set_small_model(self, on):
        """process-wide: code like a reference built with LargeModel = false (llcomp.hpp:21, 427-429)"""
        self.lib.orc_set_small_model(int(bool(on)))

    def carry_stats(self, reset=False):
        """(carries through a run of undecided 0xFF bytes, longest run) seen by the encoder since the last reset"""
        runs, longest = C.c_long(), C.c_long()
        self.lib.orc_carry_stats(C.byref(runs), C.byref(longest), int(reset))
        return runs.value, longest.value

    def _take(self, ptr, n):
        out = C.string_at(ptr, n)
        self.lib.orc_free(ptr)
        return out

    def compress_image(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, c = img.shape
        ptr = _u8p()
        n = self.lib.orc_compress_image(_p(img, _u8p), w, h, c, C.byref(ptr))
        if n < 0:
            raise ValueError("orc_compress_image rejected the arguments")
        return self._take(ptr, n)

    def compress_sliced(self, img, tile_w=0, tile_h=0, planar=False):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, c = img.shape
        ptr = _u8p()
        n = self.lib.orc_compress_sliced(_p(img, _u8p), w, h, c, tile_w, tile_h, int(planar), C.byref(ptr))
        if n < 0:
            raise ValueError("orc_compress_sliced rejected the arguments")
        return self._take(ptr, n)

    def decompress(self, data: bytes):
        """-> (rc, image or None)"""
        buf = np.frombuffer(data, dtype=np.uint8)
        ptr = _u8p()
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        rc = self.lib.orc_decompress(_p(buf, _u8p) if len(data) else None, len(data), C.byref(ptr), C.byref(w), C.byref(h), C.byref(c))
        if rc != OK:
            return rc, None
        n = w.value * h.value * c.value
        px = np.frombuffer(self._take(ptr, max(n, 0)), dtype=np.uint8)[:n].reshape(h.value, w.value, c.value)
        return rc, px

    def forward_rct(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, c = img.shape
        out = np.empty((h, w, c), dtype=np.int16)
        self.lib.orc_forward_rct(_p(img, _u8p), w * h, c, _p(out, _i16p))
        return out

    def inverse_rct(self, s):
        s = np.ascontiguousarray(s, dtype=np.int16)
        h, w, c = s.shape
        out = np.empty((h, w, c), dtype=np.uint8)
        self.lib.orc_inverse_rct(_p(s, _i16p), w * h, c, _p(out, _u8p))
        return out

    def encode_samples(self, s):
        """bare stream of an interleaved int16 (h,w,c) image with fresh state"""
        s = np.ascontiguousarray(s, dtype=np.int16)
        h, w, c = s.shape
        ptr = _u8p()
        n = self.lib.orc_encode_rect(_p(s, _i16p), w * c, c, c, w, h, C.byref(ptr))
        return self._take(ptr, n)

    def decode_samples(self, data: bytes, w, h, c):
        buf = np.frombuffer(data, dtype=np.uint8)
        out = np.zeros((h, w, c), dtype=np.int16)
        rc = self.lib.orc_decode_rect(_p(buf, _u8p) if len(data) else None, len(data), _p(out, _i16p), w * c, c, c, w, h)
        return rc, out

    def model_samples(self, s):
        """(ctx u16, res i16) per sample in coding order for an interleaved int16 (h,w,c) image"""
        s = np.ascontiguousarray(s, dtype=np.int16)
        h, w, c = s.shape
        ctx = np.empty(h * w * c, dtype=np.uint16)
        res = np.empty(h * w * c, dtype=np.int16)
        self.lib.orc_model_rect(_p(s, _i16p), w * c, c, c, w, h, _p(ctx, _u16p), _p(res, _i16p))
        return ctx.reshape(h, w, c), res.reshape(h, w, c)

    def slice_count(self, w, h, c, tile_w, tile_h, planar):
        return self.lib.orc_slice_count(w, h, c, tile_w, tile_h, int(planar))


class Ref:
    """The real reference (llcomp.hpp compiled in place).  available() is False on the GPU box unless
    the prebuilt oracle/_ref travelled with the snapshot."""

    @staticmethod
    def available(path=REF_PATH):
        return os.path.exists(path)

    def __init__(self, path=REF_PATH):
        L = self.lib = C.CDLL(path)
        L.ref_compress_image.restype = C.c_long
        L.ref_compress_image.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_long]
        L.ref_decompress_image.restype = C.c_int
        L.ref_decompress_image.argtypes = [_u8p, C.c_long, _u8p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_o2_forward_rct.restype = None
        L.ref_o2_forward_rct.argtypes = [_u8p, C.c_long, C.c_int, _i16p]
        L.ref_o2_encode_samples.restype = C.c_long
        L.ref_o2_encode_samples.argtypes = [_i16p, C.c_int, C.c_int, C.c_int, _u8p, C.c_long]
        L.ref_o2_compress_image.restype = C.c_long
        L.ref_o2_compress_image.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_long]
        for f in ("ref_quant11", "ref_quant5", "ref_state_p"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_int]
        L.ref_state_next.restype = C.c_int
        L.ref_state_next.argtypes = [C.c_int, C.c_int]
        L.ref_median.restype = C.c_int
        L.ref_median.argtypes = [C.c_int] * 3
        L.ref_states_nb.restype = C.c_int
        L.ref_magic.restype = C.c_int
        L.ref_large_model.restype = C.c_int

    @staticmethod
    def _cap(n):
        return 16 * n + 4096  # > the 13 B/sample adversarial bound

    def o2_compress_image(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, c = img.shape
        out = np.empty(self._cap(img.size), dtype=np.uint8)
        n = self.lib.ref_o2_compress_image(_p(img, _u8p), w, h, c, _p(out, _u8p), out.size)
        assert n >= 0
        return out[:n].tobytes()

    def o2_encode_samples(self, s):
        s = np.ascontiguousarray(s, dtype=np.int16)
        h, w, c = s.shape
        out = np.empty(self._cap(s.size), dtype=np.uint8)
        n = self.lib.ref_o2_encode_samples(_p(s, _i16p), w, h, c, _p(out, _u8p), out.size)
        assert n >= 0
        return out[:n].tobytes()

    def o2_forward_rct(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, c = img.shape
        out = np.empty((h, w, c), dtype=np.int16)
        self.lib.ref_o2_forward_rct(_p(img, _u8p), w * h, c, _p(out, _i16p))
        return out

    def o1_compress_image(self, img, known_len):
        """Unmodified compressImage.  Only legal when the stream fits the reference's own w*h*c buffer
        (D1) -- pass the length learnt from O2; returns None when O1 is undefined for this input."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, c = img.shape
        if known_len > img.size:
            return None
        out = np.empty(img.size + 64, dtype=np.uint8)
        n = self.lib.ref_compress_image(_p(img, _u8p), w, h, c, _p(out, _u8p), out.size)
        assert n >= 0
        return out[:n].tobytes()

    def o1_decompress_image(self, data: bytes):
        """Unmodified decompressImage -> (rc, image or None).  Only defined for channels >= 3 (D2) and
        len >= 6 (D5): guarded here so a test can never walk the reference into UB."""
        if len(data) < 6:
            return 3, None
        c_hdr, w_hdr, h_hdr = data[1], data[2] | (data[3] << 8), data[4] | (data[5] << 8)
        if data[0] == 0x79 and c_hdr < 3:
            raise ValueError("reference decoder is undefined for channels<3 (SURVEY D2)")
        buf = np.frombuffer(data, dtype=np.uint8)
        out = np.empty(max(1, w_hdr * h_hdr * max(c_hdr, 3)), dtype=np.uint8)
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        rc = self.lib.ref_decompress_image(_p(buf, _u8p), len(data), _p(out, _u8p), out.size, C.byref(w), C.byref(h), C.byref(c))
        if rc != 0:
            return rc, None
        return 0, out[: w.value * h.value * c.value].reshape(h.value, w.value, c.value).copy()
