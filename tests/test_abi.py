"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/llcomp_mi.h declares, refuses to compute without a device (no CPU fallback), and the host-only
container tools agree with the oracle's container."""
import ctypes as C
import os
import re

import numpy as np
import pytest
from conftest import ROOT, make_image


@pytest.fixture(scope="module")
def mi():
    import llcomp_amd
    from llcomp_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "llcomp_amd", "csrc")])
    return llcomp_amd


def test_header_symbols_all_exported(mi):
    from llcomp_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "llcomp_mi.h")).read()
    declared = sorted(set(re.findall(r"\b(llcomp_mi_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    assert sorted(_lib.SYMBOLS) == declared, "llcomp_amd/_lib.py SYMBOLS out of sync with include/llcomp_mi.h"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in llcomp_mi.h but not exported by libllcomp_mi.so"
    assert lib.llcomp_mi_abi_version() == _lib.ABI_VERSION == 3


def test_struct_layouts_match_header(mi):
    from llcomp_amd import _lib

    assert C.sizeof(_lib.Opts) == 28
    assert C.sizeof(_lib.Info) == 56
    assert C.sizeof(_lib.StreamResult) == 40


def test_no_cpu_fallback(mi):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert mi.device_count() == 0
    with pytest.raises(mi.LlcompError) as e:
        mi.compress_image(np.zeros(12, np.uint8), 2, 2, 3)
    assert e.value.status == mi.NO_DEVICE
    with pytest.raises(mi.LlcompError) as e:
        mi.decompress_image(bytes([0x79, 3, 2, 0, 2, 0, 1, 2, 3, 4]))
    assert e.value.status == mi.NO_DEVICE
    with pytest.raises(mi.LlcompError):
        mi.Codec(1, 64, 64, 3)


def test_bench_touches_the_oracle_only_in_its_cpu_baseline_leg():
    src = open(os.path.join(ROOT, "bench.py")).read()
    a, b = src.index("def cpu_baseline("), src.index("def main(")
    outside = src[:a] + src[b:]
    for word in ("import orc", "from orc", "liborc", "libllcomp_ref", '"oracle"'):
        assert word not in outside, f"bench.py uses the checker outside cpu_baseline(): {word!r}"
    assert "import orc" in src[a:b]


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under llcomp_amd/, include/ or tools/ may load, link or name it."""
    banned = ("liborc", "llcomp_oracle", "libllcomp_ref", "oracle/", "import orc", "from orc")
    roots = [os.path.join(ROOT, d) for d in ("llcomp_amd", "include", "tools")]
    files = []
    for r in roots:
        if os.path.isdir(r):
            for dp, _, fs in os.walk(r):
                files += [os.path.join(dp, f) for f in fs if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or f == "Makefile"]
        else:
            files.append(r)
    assert files
    for f in files:
        src = open(f).read()
        for word in banned:
            assert word not in src, f"{f} mentions {word!r}"


def test_probe_and_error_strings(mi, orc):
    img = make_image("g1", 19, 13, 3)
    s = orc.compress_sliced(img, 8, 4, True)
    info = mi.probe(s)
    assert (info.format, info.channels, info.width, info.height, info.tile_w, info.tile_h, info.planar) == (1, 3, 19, 13, 8, 4, 1)
    assert info.n_slices == orc.slice_count(19, 13, 3, 8, 4, True) == mi.slice_count(19, 13, 3, 8, 4, True)
    assert info.payload_offset == 24 + 4 * info.n_slices
    leg = orc.compress_image(img)
    info = mi.probe(leg)
    assert (info.format, info.channels, info.width, info.height, info.n_slices, info.payload_offset) == (0, 3, 19, 13, 1, 6)
    for data, code, msg in [(b"", mi.TRUNCATED, None), (bytes([0x79, 1]), mi.TRUNCATED, None),
                            (bytes([0x77]) + bytes(20), mi.BAD_MAGIC, "Invalid magic number"), (bytes([0x9C, 1, 3, 0]), mi.TRUNCATED, None)]:
        with pytest.raises(mi.LlcompError) as e:
            mi.probe(data)
        assert e.value.status == code
        if msg:
            assert str(e.value) == msg
    assert str(mi.LlcompError(mi.BAD_EXPONENT)) == "Invalid exponent"


@pytest.mark.parametrize("planar", [False, True])
@pytest.mark.parametrize("shape", [(40, 33, 3), (17, 64, 1), (33, 20, 4)])
def test_band_concatenator_matches_oracle_container(mi, orc, shape, planar):
    """Host concatenator (multi-GPU stitch): containers of horizontal bands -> container of the whole image,
    equal to the oracle's one-piece container; split_band inverts it."""
    w, h, c = shape
    img = make_image("mid", w, h, c)
    tw, th = 16, 8
    whole = orc.compress_sliced(img, tw, th, planar)
    nty = (h + th - 1) // th
    cuts = [0, nty // 3, 2 * nty // 3 + 1, nty]
    cuts = sorted(set(cuts))
    bands = [orc.compress_sliced(img[a * th:min(h, b * th)], tw, th, planar) for a, b in zip(cuts[:-1], cuts[1:])]
    assert mi.merge_bands(bands) == whole
    for (a, b), band in zip(zip(cuts[:-1], cuts[1:]), bands):
        assert mi.split_band(whole, a, b) == band
    with pytest.raises(mi.LlcompError):
        mi.merge_bands([bands[0], orc.compress_sliced(img[: th], tw + 1, th, planar)])


def test_fnv_helper_matches_the_checker_and_continues_over_pieces(mi, orc):
    """llcomp_mi_fnv1a64 (the checksum of the golden vectors) == the oracle's, also when a container lies in several pieces"""
    from conftest import fnv_hex

    data = bytes((i * 37 + 11) & 0xFF for i in range(5000))
    assert mi.fnv1a64(data) == fnv_hex(orc, data)
    assert mi.fnv1a64(data[:24], data[24:1000], np.frombuffer(data[1000:], dtype=np.uint8)) == fnv_hex(orc, data)
    assert mi.fnv1a64(b"") == "%016x" % 1469598103934665603


def test_suggested_slice_width(mi):
    """llcomp_mi_suggest_tile_w: the widest one-row slice (64..480) that still gives about four wavefronts per SIMD"""
    assert mi.suggest_tile_w(1, 3840, 2160, 3, True) == 80            # one 4K frame: 48 x 6480 slices
    assert mi.suggest_tile_w(32, 3840, 2160, 3, True) == 480          # bench.py's batch: the throughput default
    assert mi.suggest_tile_w(1, 3840, 2160, 3, False) == 64           # interleaved: a third of the slices, the floor
    assert mi.suggest_tile_w(1, 100, 50, 3, True) == 64               # a small image: the floor
    assert mi.suggest_tile_w(1, 40, 50, 3, True) == 40                # never wider than the image
    assert mi.suggest_tile_w(4, 8192, 8192, 3, True) == 480
    assert mi.suggest_tile_w(0, 10, 10, 3, True) == 0
    for f in (1, 2, 3, 4, 8):
        tw = mi.suggest_tile_w(f, 3840, 2160, 3, True)
        assert 64 <= tw <= 480 and 3840 % tw == 0
