"""One image / one frame stream over a LIST of devices inside one process, through the C ABI (llcomp_mi_opts.devices,
llcomp_mi_decode_devices, llcomp_mi_stream_create_multi; csrc/multidev.hip, csrc/stream.hip).  The reference's callers are
in-process C++ (llcompc.cpp:33, llcompd.cpp:26): this is their call with N GPUs behind it.

The test box has ONE GPU, so the lists repeat ordinal 0 ({0,0}, {0,0,0}: two / three lanes on one card) -- every part of the path
runs (the plan, a lane and a thread per part, per-chunk copies straight into the final container / picture, the verdict over all
parts); what does not run here is two distinct GPUs.  Containers must equal the one-device container byte for byte (and, for
BASELINE config 4, the container assembled from the REAL reference's per-slice streams)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mi():
    import llcomp_amd

    assert llcomp_amd.device_count() >= 1
    return llcomp_amd


def _one_piece(mi, img, tw, th, planar, small=False):
    h, w, c = img.shape
    return mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar, device=0, small_model=small)


def test_c4_golden_through_device_lists(mi):
    """BASELINE config 4's image (8192 x 8192 RGB8 noise, planar 512x1) over {0,0} and {0,0,0}: the container assembled from the real
    reference's per-slice streams (tests/golden/c4_bench_slicing.json), decoded back over a device list"""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "c4_bench_slicing.json")))["vectors"][0]
    w, h, c = gold["w"], gold["h"], gold["c"]
    img = make_image(gold["gen"], w, h, c)
    for devs, cpd in (([0, 0], 0), ([0, 0, 0], 3)):
        cont = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=gold["tile_w"], tile_h=gold["tile_h"], planar=gold["planar"],
                                 devices=devs, chunks_per_device=cpd)
        assert len(cont) == gold["container_len"], (devs, len(cont))
        assert mi.fnv1a64(cont) == gold["container_fnv1a64"], f"devices={devs}: container differs from the reference's"
    out = mi.decompress_image(cont, devices=[0, 0, 0])
    assert (out.width, out.height, out.channels) == (w, h, c) and np.array_equal(out.pixels, img)
    mi.trim()


@pytest.mark.parametrize("gen,w,h,c,tw,th,planar,devs,cpd", [
    ("nat", 1024, 1024, 3, 64, 64, True, [0, 0], 0),        # 2-D tiles: snapshot encoder + cached decoder in every part
    ("mid", 1000, 333, 3, 64, 64, False, [0, 0, 0], 2),     # ragged right column and bottom row, channels interleaved
    ("g3", 777, 130, 4, 480, 1, True, [0, 0], 4),           # one-row slices, four channels, ragged width
    ("g2", 256, 200, 1, 128, 48, True, [0, 0, 0], 1),       # 5 tile rows over 3 parts, the last one 8 rows high
    ("mid", 300, 70, 3, 100, 64, True, [0, 0], 1),          # 2 tile rows: the second part owns ONLY the short bottom row (tile_h clamps)
    ("g1", 64, 64, 3, 32, 32, True, [0, 0, 0, 0], 4),       # more devices than tile rows: two parts stay empty
    ("nat", 512, 96, 3, 0, 0, True, [0, 0], 0),             # one slice per plane = one tile row: a one-part plan
    ("mid", 640, 480, 2, 80, 16, False, [0, 0], 8),         # 30 tile rows in 16 chunks
])
def test_device_list_equals_one_device(mi, orc, gen, w, h, c, tw, th, planar, devs, cpd):
    img = make_image(gen, w, h, c)
    want = orc.compress_sliced(img, tw or w, th or h, planar)
    assert _one_piece(mi, img, tw, th, planar) == want
    cont = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar, devices=devs, chunks_per_device=cpd)
    assert cont == want, f"devices={devs}: container differs from the oracle's / the one-device container"
    for dd in (devs, [0, 0], [0]):
        out = mi.decompress_image(cont, devices=dd, chunks_per_device=cpd)
        assert (out.width, out.height, out.channels) == (w, h, c) and np.array_equal(out.pixels, img), dd
    # a container coded in one piece decodes over a list just the same, and the other way round
    assert np.array_equal(mi.decompress_image(want, devices=devs).pixels, img)
    assert np.array_equal(mi.decompress_image(cont, device=0).pixels, img)


def test_device_list_small_model_and_legacy(mi, orc):
    img = make_image("mid", 200, 96, 3)
    one = _one_piece(mi, img, 50, 32, True, small=True)
    many = mi.compress_image(img, 200, 96, 3, format=mi.FORMAT_SLICED, tile_w=50, tile_h=32, planar=True, small_model=True, devices=[0, 0, 0])
    assert many == one and mi.probe(many).small_model == 1
    assert np.array_equal(mi.decompress_image(many, devices=[0, 0]).pixels, img)
    # one serial stream does not shard: devices[0] codes it, and the bytes are the reference's
    legacy = mi.compress_image(img, 200, 96, 3, devices=[0, 0])
    assert legacy == orc.compress_image(img)
    assert np.array_equal(mi.decompress_image(legacy, devices=[0, 0]).pixels, img)


def test_device_list_into_variants_and_capacity(mi):
    """caller-provided buffers: the parts copy straight into them; a buffer that is too small is left alone and told what it takes"""
    img = make_image("nat", 512, 256, 3)
    want = _one_piece(mi, img, 64, 32, True)
    out = np.full(len(want) + 64, 0xAB, np.uint8)
    n = mi.compress_image_into(img.reshape(-1), 512, 256, 3, out, format=mi.FORMAT_SLICED, tile_w=64, tile_h=32, planar=True, devices=[0, 0])
    assert n == len(want) and bytes(out[:n]) == want and (out[n:] == 0xAB).all()
    small = np.full(len(want) - 1, 0xCD, np.uint8)
    with pytest.raises(mi.LlcompError) as e:
        mi.compress_image_into(img.reshape(-1), 512, 256, 3, small, format=mi.FORMAT_SLICED, tile_w=64, tile_h=32, planar=True, devices=[0, 0])
    assert e.value.status == mi.OUTPUT_OVERFLOW and e.value.needed == len(want) and (small == 0xCD).all()
    px = np.full(img.size + 16, 0xEE, np.uint8)
    cont = np.frombuffer(want, np.uint8)
    assert mi.decompress_image_into(cont, px, devices=[0, 0, 0]) == (512, 256, 3)
    assert np.array_equal(px[: img.size].reshape(img.shape), img) and (px[img.size:] == 0xEE).all()
    tiny = np.full(img.size - 1, 0x11, np.uint8)
    with pytest.raises(mi.LlcompError) as e:
        mi.decompress_image_into(cont, tiny, devices=[0, 0])
    assert e.value.status == mi.OUTPUT_OVERFLOW and e.value.shape == (512, 256, 3) and (tiny == 0x11).all()
    # pinned buffers on both sides (plain DMA per chunk)
    pin_in, pin_out = mi.PinnedBuffer(img.size), mi.PinnedBuffer(len(want) + 16)
    pin_in.array[:] = img.reshape(-1)
    n = mi.compress_image_into(pin_in.array, 512, 256, 3, pin_out.array, format=mi.FORMAT_SLICED, tile_w=64, tile_h=32, planar=True, devices=[0, 0])
    assert bytes(pin_out.array[:n]) == want
    pin_in.close()
    pin_out.close()


def test_device_list_failures_publish_nothing(mi):
    img = make_image("mid", 256, 128, 3)
    want = _one_piece(mi, img, 64, 16, True)
    # an ordinal that does not exist: DEVICE_FAILED, and the thread's record says which one and why
    out = np.full(len(want) + 8, 0x5A, np.uint8)
    with pytest.raises(mi.LlcompError) as e:
        mi.compress_image_into(img.reshape(-1), 256, 128, 3, out, format=mi.FORMAT_SLICED, tile_w=64, tile_h=16, planar=True, devices=[0, 99])
    assert e.value.status == mi.DEVICE_FAILED and e.value.device_error == (99, 1, mi.BAD_ARGS), e.value.device_error
    assert (out == 0x5A).all(), "a failed device list wrote into the caller's buffer"
    px = np.full(img.size, 0x5A, np.uint8)
    with pytest.raises(mi.LlcompError) as e:
        mi.decompress_image_into(np.frombuffer(want, np.uint8), px, devices=[99, 0])
    assert e.value.status == mi.DEVICE_FAILED and e.value.device_error == (99, 0, mi.BAD_ARGS) and (px == 0x5A).all()
    assert mi.compress_image(img, 256, 128, 3, format=mi.FORMAT_SLICED, tile_w=64, tile_h=16, planar=True, devices=[0, 0]) == want
    assert mi.last_device_error() is None  # a later success clears the record
    # "current device" has no meaning inside a list; an empty list is no list
    with pytest.raises(mi.LlcompError) as e:
        mi.compress_image(img, 256, 128, 3, format=mi.FORMAT_SLICED, tile_w=64, tile_h=16, planar=True, devices=[0, -1])
    assert e.value.status == mi.BAD_ARGS
    L = mi._lib.load()
    px_p, w, h, c = mi._lib.u8p(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    buf = (C.c_uint8 * len(want)).from_buffer_copy(want)
    assert L.llcomp_mi_decode_devices(C.cast(buf, mi._lib.u8p), len(want), None, 0, 0, 0, C.byref(px_p), C.byref(w), C.byref(h), C.byref(c)) == mi.BAD_ARGS
    # a data verdict comes back as itself from whichever part meets it -- and equals the one-device verdict
    bad = bytearray(want)
    info = mi.probe(want)
    for k in range(info.payload_offset + 40, min(len(bad), info.payload_offset + 1200), 7):
        bad[k] ^= 0xFF
    def verdict(**kw):
        try:
            return ("ok", bytes(mi.decompress_image(bytes(bad), **kw).pixels.reshape(-1)))
        except mi.LlcompError as err:
            return ("err", err.status)
    assert verdict(devices=[0, 0]) == verdict(device=0)


def test_damaged_tables_get_the_one_device_verdict(mi):
    """a slice table that promises more than the payload holds (or a slice longer than any valid stream) is not split over the
    devices: the one-device path forms the verdict, so both calls say the same"""
    img = make_image("g3", 320, 64, 3)
    good = bytearray(_one_piece(mi, img, 80, 8, True))
    info = mi.probe(bytes(good))
    def status(data, **kw):
        try:
            mi.decompress_image(bytes(data), **kw)
            return mi.OK
        except mi.LlcompError as err:
            return err.status
    cut = good[: len(good) - 100]                                  # payload shorter than the table says
    huge = bytearray(good)
    huge[info.table_offset + 4 * 5: info.table_offset + 4 * 5 + 4] = (0x7FFFFFFF).to_bytes(4, "little")  # one absurd length
    for damaged in (cut, huge):
        one, many = status(damaged, device=0), status(damaged, devices=[0, 0, 0])
        assert one == many and one in (mi.TRUNCATED, mi.BAD_EXPONENT), (one, many)


def test_stream_over_a_device_list(mi, orc):
    """llcomp_mi_stream_create_multi: jobs dealt round-robin over two pipelines on the one GPU, results in submission order,
    containers == the oracle's, frames bit-exact after the round trip (BASELINE config 5's path behind a device list)"""
    w, h, c = 192, 80, 3
    frames = [np.ascontiguousarray(np.roll(make_image("mid", w, h, c), 7 * i, axis=1)) for i in range(11)]
    want = [orc.compress_sliced(f, 48, 1, True) for f in frames]
    st = mi.Stream(w, h, c, 48, 1, True, depth=2, devices=[0, 0])
    assert st.n_devices == 2
    seen = {}
    lens, _, _ = mi.pipeline_roundtrip(st, frames, max_encodes_in_flight=3, on_container=lambda i, d: seen.__setitem__(i, d.tobytes()), verify=True, verify_threads=0)
    assert [seen[i] for i in range(len(frames))] == want and lens == [len(x) for x in want]
    # back-pressure: 2 pipelines x 2 slots take four jobs, the fifth is refused until a result is taken and released
    for i in range(4):
        assert st.submit_encode(frames[i], tag=100 + i)
    assert not st.submit_encode(frames[4], tag=104) and st.pending() == 4
    tags = []
    for _ in range(4):
        job = st.wait()
        assert job.status == mi.OK and job.kind == mi.JOB_ENCODE and job.data.tobytes() == want[job.tag - 100]
        tags.append(job.tag)
        st.release(job)
    assert tags == [100, 101, 102, 103] and st.pending() == 0
    st.close()
    # several frames per job through the dealer
    st = mi.Stream(w, h, c, 48, 1, True, depth=2, devices=[0, 0, 0], frames_per_job=2)
    block = np.ascontiguousarray(np.stack(frames[:8]))
    mi.pipeline_roundtrip(st, [block[i] for i in range(8)], max_encodes_in_flight=2, verify=True, verify_threads=0)
    st.close()
    with pytest.raises(mi.LlcompError) as e:
        mi.Stream(w, h, c, 48, 1, True, depth=2, devices=[0, 77])
    assert e.value.status == mi.DEVICE_FAILED and e.value.device_error == (77, 1, mi.BAD_ARGS)


def test_cli_devices_flag(mi, tmp_path):
    """tools/llcompc --devices a,b,... / tools/llcompd --devices: the C++ drop-in header's Options::devices end to end"""
    exe_c, exe_d = os.path.join(ROOT, "tools", "llcompc"), os.path.join(ROOT, "tools", "llcompd")
    if not (os.path.exists(exe_c) and os.path.exists(exe_d)):
        pytest.skip("tools not built")
    img = make_image("mid", 160, 96, 3)
    src = tmp_path / "x.ppm"
    src.write_bytes(b"P6\n160 96\n255\n" + img.tobytes())
    r = subprocess.run([exe_c, str(src), "--sliced", "40x24", "--devices", "0,0,0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cont = (tmp_path / "x.ppm.llcomp").read_bytes()
    assert cont == _one_piece(mi, img, 40, 24, True)
    r = subprocess.run([exe_d, str(tmp_path / "x.ppm.llcomp"), "--devices", "0,0"], capture_output=True, text=True)
    assert r.returncode == 0 and (tmp_path / "x.ppm.llcomp.png").exists(), r.stderr
    r = subprocess.run([exe_c, str(src), "--sliced", "40x24", "--devices", "0,42"], capture_output=True, text=True)
    assert r.returncode == 1 and "device 42" in r.stderr, r.stderr
