"""llcomp_amd -- host-side mirror (Python, for tests / bench / scripting) of the MI355X-native llcomp coding path.

The product is libllcomp_mi.so (hand-written HIP kernels for gfx950 behind the C ABI of include/llcomp_mi.h);
include/llcomp_mi.hpp is the C++ drop-in with the reference's own signatures.  This module binds the same C ABI
with ctypes and keeps the reference's names and error behaviour:

    compress_image(rgb, width, height, channels)  <->  llcomp::compressImage    (/root/reference/llcomp.hpp:358)
    decompress_image(data) -> RawImage            <->  llcomp::decompressImage  (/root/reference/llcomp.hpp:461)
    RawImage(pixels, width, height, channels)     <->  llcomp::RawImage         (/root/reference/llcomp.hpp:454-459)
    EXT = ".llcomp"                               <->  llcomp::ext              (/root/reference/llcomp.hpp:18)

Errors: the reference throws std::runtime_error("Invalid magic number") / ("Invalid exponent"); here LlcompError
carries the same message text plus the C status code.  There is no CPU fallback anywhere in this package.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _lib
from ._lib import Info, Opts

EXT = ".llcomp"
FORMAT_LEGACY, FORMAT_SLICED = 0, 1
(OK, BAD_MAGIC, BAD_EXPONENT, TRUNCATED, BAD_ARGS, OUT_OF_RANGE, OUTPUT_OVERFLOW, HIP_ERROR, NO_DEVICE, NOMEM, BUSY, DEVICE_FAILED) = range(12)
JOB_ENCODE, JOB_DECODE = 0, 1

RawImage = namedtuple("RawImage", "pixels width height channels")


class LlcompError(RuntimeError):
    def __init__(self, status, detail=None):
        self.status = int(status)
        msg = _lib.load().llcomp_mi_strerror(int(status)).decode()
        # a call over a device list: (HIP ordinal, position in the list, that device's own status)
        self.device_error = last_device_error() if self.status == DEVICE_FAILED else None
        super().__init__(msg if not detail else f"{msg} ({detail})")


def _check(rc):
    if rc != OK:
        raise LlcompError(rc)


def last_device_error():
    """(HIP ordinal, position in the device list, that device's own status) behind this thread's last DEVICE_FAILED, else None"""
    dev, idx, st = C.c_int32(), C.c_uint32(), C.c_int()
    if not _lib.load().llcomp_mi_last_device_error(C.byref(dev), C.byref(idx), C.byref(st)):
        return None
    return dev.value, idx.value, st.value


def _opts(format, tile_w, tile_h, planar, device, small_model, devices=None, chunks_per_device=0):
    """llcomp_mi_opts (+ the int32 array its `devices` points at: keep it alive for the call)"""
    o = Opts(C.sizeof(Opts), format, tile_w, tile_h, int(bool(planar)), device, int(bool(small_model)), 0, None, int(chunks_per_device), 0)
    arr = None
    if devices is not None:
        arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        o.n_devices, o.devices = len(devices), C.cast(arr, C.POINTER(C.c_int32))
    return o, arr


def plan_chunks(height, tile_h, n_parts, chunks_per_part=4):
    """[(tile_row0, tile_row1, owner)] -- llcomp_mi_plan_chunks, the one work split of every multi-GPU path (device lists in the
    library, ranks in llcomp_amd.sharding)"""
    L = _lib.load()
    n = C.c_uint32()
    _check(L.llcomp_mi_plan_chunks(height, max(0, tile_h), n_parts, max(1, chunks_per_part), None, 0, C.byref(n)))
    tri = (C.c_uint32 * (3 * n.value))()
    _check(L.llcomp_mi_plan_chunks(height, max(0, tile_h), n_parts, max(1, chunks_per_part), tri, n.value, C.byref(n)))
    return [(tri[3 * i], tri[3 * i + 1], tri[3 * i + 2]) for i in range(n.value)]


def device_count():
    return _lib.load().llcomp_mi_device_count()


def compress_image(rgb, width, height, channels, *, format=FORMAT_LEGACY, tile_w=0, tile_h=0, planar=False, device=-1, small_model=False,
                   devices=None, chunks_per_device=0):
    """bytes of an llcomp stream.  Default = the reference's own whole-image format (magic 0x79), byte-identical to
    llcomp::compressImage; format=FORMAT_SLICED produces the parallel container (magic 0x9C).  small_model=True codes like
    a reference built with LargeModel = false (llcomp.hpp:21); a legacy stream does not record that, so it must be passed
    to decompress_image as well.  devices=[ordinals]: the image's tile rows are dealt over these GPUs inside this process
    (llcomp_mi_opts.devices; the container is byte-identical to the one-device container)."""
    L = _lib.load()
    buf = np.ascontiguousarray(np.frombuffer(rgb, dtype=np.uint8) if isinstance(rgb, (bytes, bytearray, memoryview)) else rgb, dtype=np.uint8).reshape(-1)
    if buf.size != width * height * channels:  # the reference only asserts this (llcomp.hpp:361)
        raise LlcompError(BAD_ARGS)
    o, _keep = _opts(format, tile_w, tile_h, planar, device, small_model, devices, chunks_per_device)
    out, n = _lib.u8p(), C.c_size_t()
    _check(L.llcomp_mi_encode(buf.ctypes.data_as(_lib.u8p), width, height, channels, C.byref(o), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value)
    finally:
        L.llcomp_mi_free(out)


def decompress_image(data, *, device=-1, small_model=False, devices=None, chunks_per_device=0):
    """RawImage(pixels: np.uint8[h,w,c], width, height, channels) from either wire format.  devices=[ordinals]: decoded over
    these GPUs inside this process (llcomp_mi_decode_devices)."""
    L = _lib.load()
    data = bytes(data)  # no copy when it already is bytes
    src = C.cast(C.c_char_p(data or b"\0"), _lib.u8p)  # borrows the bytes object's buffer for the call
    px, w, h, c = _lib.u8p(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    if devices is not None:
        arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        _check(L.llcomp_mi_decode_devices(src, len(data), arr, len(devices), int(chunks_per_device), 1 if small_model else 0, C.byref(px), C.byref(w),
                                          C.byref(h), C.byref(c)))
    else:
        _check(L.llcomp_mi_decode_flags(src, len(data), device, 1 if small_model else 0, C.byref(px), C.byref(w), C.byref(h), C.byref(c)))
    try:
        n = w.value * h.value * c.value
        pixels = np.ctypeslib.as_array(px, shape=(max(n, 1),))[:n].copy().reshape(h.value, w.value, c.value)
    finally:
        L.llcomp_mi_free(px)
    return RawImage(pixels, w.value, h.value, c.value)


def trim():
    """Release the idle coding lanes the host-buffer calls keep for the next call of the same shape (GBs of HBM) and
    the device memory the library parks for reuse instead of returning it to the driver (csrc/devmem.hip)."""
    _lib.load().llcomp_mi_trim()


def set_pool_limit(bytes_per_device):
    """Device memory the library may keep parked per device (csrc/devmem.hip); 0 = return every block to the driver."""
    _lib.load().llcomp_mi_set_pool_limit(int(bytes_per_device))


def pool_idle_bytes():
    return int(_lib.load().llcomp_mi_pool_idle_bytes())


def reload_tuning():
    """Have the library read its test / tuning hooks (LLCOMP_MI_*) from the environment again."""
    _lib.load().llcomp_mi_reload_tuning()


class PinnedBuffer:
    """Pinned host memory from llcomp_mi_host_alloc, exposed as a numpy uint8 array (`.array`): copies between it and
    the GPU are plain DMA."""

    def __init__(self, nbytes):
        self._L = _lib.load()
        self.nbytes = int(nbytes)
        self.ptr = self._L.llcomp_mi_host_alloc(self.nbytes)
        if not self.ptr:
            raise LlcompError(NOMEM)
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, _lib.u8p), shape=(max(self.nbytes, 1),))[: self.nbytes]

    def close(self):
        if self.ptr:
            self.array = None
            self._L.llcomp_mi_host_free(self.ptr)
            self.ptr = None

    __del__ = close


def compress_image_into(rgb, width, height, channels, out, *, format=FORMAT_LEGACY, tile_w=0, tile_h=0, planar=False, device=-1, small_model=False,
                        devices=None, chunks_per_device=0):
    """llcomp_mi_encode_into: `rgb` and `out` are numpy uint8 arrays owned by the caller (pinned: PinnedBuffer.array);
    returns the container length.  Raises LlcompError(OUTPUT_OVERFLOW) with .needed set when `out` is too small."""
    L = _lib.load()
    o, _keep = _opts(format, tile_w, tile_h, planar, device, small_model, devices, chunks_per_device)
    n = C.c_size_t()
    rc = L.llcomp_mi_encode_into(rgb.ctypes.data, width, height, channels, C.byref(o), out.ctypes.data, out.size, C.byref(n))
    if rc != OK:
        e = LlcompError(rc)
        e.needed = n.value
        raise e
    return n.value


def decompress_image_into(data, out, *, device=-1, small_model=False, devices=None, chunks_per_device=0):
    """llcomp_mi_decode_into_flags / _into_devices: `data` / `out` numpy uint8 arrays owned by the caller -> (width, height, channels)."""
    L = _lib.load()
    w, h, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    if devices is not None:
        arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        rc = L.llcomp_mi_decode_into_devices(data.ctypes.data, data.size, arr, len(devices), int(chunks_per_device), 1 if small_model else 0,
                                             out.ctypes.data, out.size, C.byref(w), C.byref(h), C.byref(c))
    else:
        rc = L.llcomp_mi_decode_into_flags(data.ctypes.data, data.size, device, 1 if small_model else 0, out.ctypes.data, out.size, C.byref(w), C.byref(h), C.byref(c))
    if rc != OK:
        e = LlcompError(rc)
        e.shape = (w.value, h.value, c.value)
        raise e
    return w.value, h.value, c.value


StreamJob = namedtuple("StreamJob", "slot kind status tag data")


class Stream:
    """Streaming pipeline (llcomp_mi_stream_*): frames of one shape, host -> GPU -> host, `depth` jobs in flight, a job =
    `frames_per_job` frames.  submit_* return False instead of blocking when every slot is occupied (back-pressure);
    wait() returns the oldest job as StreamJob, valid until release(job).  Its .data is a numpy view of the stream's pinned
    output buffer: frames_per_job == 1: the container (encode) / the frame [h,w,c] (decode); more frames per job: a list of
    containers (encode) / the frames [F,h,w,c] (decode)."""

    def __init__(self, w, h, c, tile_w=0, tile_h=0, planar=True, depth=4, device=-1, frames_per_job=1, devices=None):
        self._L = _lib.load()
        self._h = C.c_void_p()
        if devices is not None:  # one pipeline of `depth` slots per device behind one object, jobs dealt round-robin
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            _check(self._L.llcomp_mi_stream_create_multi(C.byref(self._h), arr, len(devices), w, h, c, tile_w, tile_h, int(bool(planar)), depth, frames_per_job))
        else:
            _check(self._L.llcomp_mi_stream_create_ex(C.byref(self._h), device, w, h, c, tile_w, tile_h, int(bool(planar)), depth, frames_per_job))
        self.n_devices = self._L.llcomp_mi_stream_devices(self._h)
        self.shape = (h, w, c)
        self.frames_per_job = frames_per_job
        self.container_capacity = self._L.llcomp_mi_stream_container_capacity(self._h)

    def close(self):
        if self._h:
            self._L.llcomp_mi_stream_destroy(self._h)
            self._h = None

    __del__ = close

    def _submit(self, rc):
        if rc == BUSY:
            return False
        _check(rc)
        return True

    def submit_encode(self, px, tag=0):
        """px: numpy uint8, frames_per_job frames [h,w,c] back to back (C-contiguous); must stay alive and unchanged until
        the job's result came back."""
        assert px.flags["C_CONTIGUOUS"] and px.dtype == np.uint8 and px.size == self.frames_per_job * self.shape[0] * self.shape[1] * self.shape[2]
        return self._submit(self._L.llcomp_mi_stream_submit_encode(self._h, px.ctypes.data, tag))

    def submit_decode(self, data, tag=0):
        """data: numpy uint8 container (frames_per_job == 1) or a list of frames_per_job containers -- e.g. the .data of an
        encode job that has not been released yet."""
        parts = [data] if self.frames_per_job == 1 and not isinstance(data, (list, tuple)) else list(data)
        assert len(parts) == self.frames_per_job and all(p.flags["C_CONTIGUOUS"] and p.dtype == np.uint8 for p in parts)
        ptrs = (C.c_void_p * len(parts))(*[p.ctypes.data for p in parts])
        lens = (C.c_size_t * len(parts))(*[p.size for p in parts])
        return self._submit(self._L.llcomp_mi_stream_submit_decode_batch(self._h, ptrs, lens, tag))

    def pending(self):
        return self._L.llcomp_mi_stream_pending(self._h)

    def ready(self):
        return self._L.llcomp_mi_stream_poll(self._h) == OK

    def wait(self):
        r = _lib.StreamResult()
        _check(self._L.llcomp_mi_stream_wait(self._h, C.byref(r)))
        data = None
        if r.status == OK:
            whole = np.ctypeslib.as_array(C.cast(r.data, _lib.u8p), shape=(max(int(r.len), 1),))[: int(r.len)]
            if r.kind == JOB_DECODE:
                data = whole.reshape(self.shape) if self.frames_per_job == 1 else whole.reshape((self.frames_per_job,) + self.shape)
            elif self.frames_per_job == 1:
                data = whole
            else:
                data = []
                for f in range(self.frames_per_job):
                    p, n = C.c_void_p(), C.c_uint64()
                    _check(self._L.llcomp_mi_stream_result_part(self._h, r.slot, f, C.byref(p), C.byref(n)))
                    data.append(np.ctypeslib.as_array(C.cast(p, _lib.u8p), shape=(max(int(n.value), 1),))[: int(n.value)])
        return StreamJob(r.slot, r.kind, r.status, r.tag, data)

    def release(self, job):
        _check(self._L.llcomp_mi_stream_release(self._h, job.slot))


def _same_bytes(a, b):
    """bit-exact comparison of two uint8 arrays, eight bytes at a time where the layout allows"""
    a, b = a.reshape(-1), b.reshape(-1)
    if a.size != b.size:
        return False
    if a.size % 8 == 0 and a.ctypes.data % 8 == 0 and b.ctypes.data % 8 == 0 and a.flags["C_CONTIGUOUS"] and b.flags["C_CONTIGUOUS"]:
        return bool(np.array_equal(a.view(np.uint64), b.view(np.uint64)))
    return bool(np.array_equal(a, b))


def pipeline_roundtrip(stream, frames, max_encodes_in_flight=3, on_container=None, verify=True, verify_threads=4, clock_origin=None):
    """Drives BASELINE config 5 through a Stream: every frame host -> GPU -> host (container) -> GPU -> host.  An encode
    result (pinned containers) is handed to submit_decode as it is and released only when that decode has come back;
    submit_* returning False (back-pressure) makes the loop take a finished job first.  frames: list of C-contiguous
    uint8 arrays (pinned for DMA); with frames_per_job > 1 consecutive frames of a job must be adjacent in memory (views
    of one buffer) and len(frames) a multiple of it.  Returns (container lengths, completion time of every frame in
    seconds, number of times back-pressure was hit).  on_container(i, bytes_view) sees every container; verify compares
    every decoded frame with its source bit for bit -- on a few worker threads (numpy releases the GIL), the slot is
    released afterwards.  clock_origin: a time.perf_counter() value the completion times are measured from (several
    pipelines driven from several threads share one; default: this call's start)."""
    import time
    from concurrent.futures import ThreadPoolExecutor

    F = stream.frames_per_job
    n = len(frames)
    assert n % F == 0, "the number of frames must be a multiple of frames_per_job"
    n_jobs = n // F
    raw = frames[0].size
    jobs_px = []
    for j in range(n_jobs):
        if F == 1:
            jobs_px.append(frames[j])
        else:
            base = frames[j * F].ctypes.data
            assert all(frames[j * F + f].ctypes.data == base + f * raw for f in range(F)), "the frames of a job must be adjacent in memory"
            jobs_px.append(np.ctypeslib.as_array(C.cast(base, _lib.u8p), shape=(F * raw,)))
    lens, done_at, busy_seen = [0] * n, [0.0] * n, 0
    enc_held, to_decode, checking = {}, [], []
    next_job = finished = enc_in_flight = 0
    pool = ThreadPoolExecutor(max_workers=verify_threads) if verify and verify_threads > 0 else None

    def reap(block):
        nonlocal finished
        while checking and (block or checking[0][0].done()):
            fut, job = checking.pop(0)
            if not fut.result():
                raise AssertionError(f"job {job.tag} (frames {job.tag * F}..{job.tag * F + F - 1}) is not bit-exact after the round trip")
            stream.release(job)
            finished += 1
            block = False

    t0 = time.perf_counter() if clock_origin is None else clock_origin
    try:
        while finished < n_jobs:
            progressed = False
            reap(False)
            while to_decode:  # containers first: their decode frees two slots
                job = to_decode[0]
                if not stream.submit_decode(job.data, tag=job.tag):
                    busy_seen += 1
                    break
                enc_held[job.tag] = job
                to_decode.pop(0)
                progressed = True
            while next_job < n_jobs and enc_in_flight < max_encodes_in_flight and not to_decode:
                if not stream.submit_encode(jobs_px[next_job], tag=next_job):
                    busy_seen += 1
                    break
                next_job += 1
                enc_in_flight += 1
                progressed = True
            if stream.pending() and (not progressed or stream.ready()):
                job = stream.wait()
                if job.status != OK:
                    raise LlcompError(job.status)
                if job.kind == JOB_ENCODE:
                    enc_in_flight -= 1
                    for f, cont in enumerate([job.data] if F == 1 else job.data):
                        lens[job.tag * F + f] = cont.size
                        if on_container:
                            on_container(job.tag * F + f, cont)
                    to_decode.append(job)
                else:
                    now = time.perf_counter() - t0
                    for f in range(F):
                        done_at[job.tag * F + f] = now
                    stream.release(enc_held.pop(job.tag))
                    if pool:
                        checking.append((pool.submit(_same_bytes, job.data, jobs_px[job.tag]), job))
                    else:
                        if verify and not _same_bytes(job.data, jobs_px[job.tag]):
                            raise AssertionError(f"job {job.tag} is not bit-exact after the round trip")
                        stream.release(job)
                        finished += 1
            elif not progressed:
                reap(True)  # every slot is held by frames that are being compared
    finally:
        if pool:
            pool.shutdown(wait=True)
    return lens, done_at, busy_seen


def probe(data):
    L = _lib.load()
    data = bytes(data)
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    info = Info()
    _check(L.llcomp_mi_probe(C.cast(buf, _lib.u8p), len(data), C.byref(info)))
    return info


def fnv1a64(*pieces):
    """FNV-1a-64 over the concatenation of `pieces` (bytes or contiguous numpy uint8 arrays), as a 16-digit hex string --
    the form tests/golden records container hashes in."""
    L = _lib.load()
    h = 0
    for p in pieces:
        a = np.frombuffer(p, dtype=np.uint8) if isinstance(p, (bytes, bytearray, memoryview)) else np.ascontiguousarray(p, dtype=np.uint8).reshape(-1)
        if a.size:
            h = L.llcomp_mi_fnv1a64(a.ctypes.data, a.size, h)
    return "%016x" % (h or 1469598103934665603)


def suggest_tile_w(frames, w, h, c, planar=True):
    """slice width for one-row slices that keeps the GPU busy when `frames` frames are coded per call (llcomp_mi_suggest_tile_w)"""
    return int(_lib.load().llcomp_mi_suggest_tile_w(frames, w, h, c, int(bool(planar))))


def slice_count(w, h, c, tile_w=0, tile_h=0, planar=False):
    return _lib.load().llcomp_mi_slice_count(w, h, c, tile_w, tile_h, int(bool(planar)))


def merge_bands(bands):
    """Concatenate SLICED containers of consecutive horizontal bands (multi-GPU shards) into one container."""
    L = _lib.load()
    n = len(bands)
    keep = [(C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) or b"\0") for b in bands]
    ptrs = (_lib.u8p * n)(*[C.cast(k, _lib.u8p) for k in keep])
    lens = (C.c_size_t * n)(*[len(b) for b in bands])
    out, m = _lib.u8p(), C.c_size_t()
    _check(L.llcomp_mi_merge_bands(ptrs, lens, n, C.byref(out), C.byref(m)))
    try:
        return C.string_at(out, m.value)
    finally:
        L.llcomp_mi_free(out)


def split_band(data, tile_row0, tile_row1):
    L = _lib.load()
    data = bytes(data)
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    out, m = _lib.u8p(), C.c_size_t()
    _check(L.llcomp_mi_split_band(C.cast(buf, _lib.u8p), len(data), tile_row0, tile_row1, C.byref(out), C.byref(m)))
    try:
        return C.string_at(out, m.value)
    finally:
        L.llcomp_mi_free(out)


class Codec:
    """Device-resident batch codec (llcomp_mi_codec_*): `frames` images of one shape per call, buffers stay in HBM.
    Pointers are raw device addresses (e.g. torch tensor .data_ptr()); `stream` is a hipStream_t handle
    (torch.cuda.current_stream().cuda_stream) or 0."""

    def __init__(self, frames, w, h, c, tile_w=0, tile_h=0, planar=False, device=-1, small_model=False):
        self._L = _lib.load()
        self._h = C.c_void_p()
        _check(self._L.llcomp_mi_codec_create_ex(C.byref(self._h), device, frames, w, h, c, tile_w, tile_h, int(bool(planar)), 1 if small_model else 0))
        self.frames, self.w, self.h, self.c = frames, w, h, c
        self.n_slices = self._L.llcomp_mi_codec_slices(self._h)
        self.max_payload_bytes = self._L.llcomp_mi_codec_max_payload_bytes(self._h)
        self.workspace_bytes = self._L.llcomp_mi_codec_workspace_bytes(self._h)
        fam = self._L.llcomp_mi_codec_kernel_family(self._h)  # (diagnostic: tests make sure they run the family they mean to)
        self.family = {"rows": bool(fam & 1), "lds_table": bool(fam & 2), "snapshot": bool(fam & 16), "bank_cache": bool(fam & 32),
                       "lane_shift": (fam >> 8) & 0xFF, "slices_per_wave": (fam >> 16) & 0xFF}

    def close(self):
        if self._h:
            self._L.llcomp_mi_codec_destroy(self._h)
            self._h = None

    __del__ = close

    def encode(self, d_px, d_payload, payload_cap, d_slice_len, d_total, d_status, stream=0):
        _check(self._L.llcomp_mi_codec_encode(self._h, d_px, d_payload, payload_cap, d_slice_len, d_total, d_status, stream))

    def decode(self, d_payload, payload_bytes, d_slice_len, d_px, d_status, stream=0):
        _check(self._L.llcomp_mi_codec_decode(self._h, d_payload, payload_bytes, d_slice_len, d_px, d_status, stream))

    def model(self, d_px, d_sym, stream=0):
        _check(self._L.llcomp_mi_codec_model(self._h, d_px, d_sym, stream))

    PROFILE_SLOTS = ("clear_states_enc", "k_model_fwd", "k_encode_slices", "scan+pack", "k_scan_lengths_dec", "k_decode_slices", "k_model_inv", "clear_states_dec")

    def set_profiling(self, on=True):
        _check(self._L.llcomp_mi_codec_set_profiling(self._h, int(on)))

    def get_profile(self):
        """-> ({slot: total ms since last call}, n_encode, n_decode); drains the stream."""
        ms = (C.c_double * 8)()
        ne, nd = C.c_uint32(), C.c_uint32()
        _check(self._L.llcomp_mi_codec_get_profile(self._h, ms, C.byref(ne), C.byref(nd)))
        return dict(zip(self.PROFILE_SLOTS, list(ms))), ne.value, nd.value

    def prepare(self, encode=True, decode=True):
        """allocate now what the first encode / decode would allocate inside the call (llcomp_mi_codec_prepare)"""
        _check(self._L.llcomp_mi_codec_prepare(self._h, (1 if encode else 0) | (2 if decode else 0)))

    COUNTERS = ("dec_cached_waves", "dec_bypassed_waves", "cache_lookups", "cache_misses", "cache_writebacks", "dec_replays", "enc_carry_backs",
                "generation_wraps", "dec_launches_cached", "dec_launches_plain")

    def counters(self, reset=False):
        """{name: count} -- what the rare and adaptive paths of this codec's kernels did so far (llcomp_mi_codec_get_counters);
        waits for the codec's last call"""
        v = (C.c_uint64 * 16)()
        _check(self._L.llcomp_mi_codec_get_counters(self._h, v, 16, int(bool(reset))))
        return dict(zip(self.COUNTERS, list(v)))

    def status(self, bits):
        return self._L.llcomp_mi_status_from_bits(int(bits))
