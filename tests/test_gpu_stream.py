"""BASELINE config 5 through the product's streaming pipeline (llcomp_mi_stream_*, C ABI): 64 distinct 4K noise frames
host -> GPU -> host (container) -> GPU -> host, several jobs in flight, back-pressure, every frame bit-exact; plus
the caller-provided / pinned-buffer host calls.  Reference analogue: one image in RAM per call, llcompc.cpp:25-41,
llcompd.cpp:17-31."""
import time

import numpy as np
import pytest
from conftest import fnv_hex, load_golden, make_image

pytestmark = pytest.mark.gpu

SLC = load_golden("slice_payloads.json")["vectors"]


@pytest.fixture(scope="module")
def mi():
    import llcomp_amd

    assert llcomp_amd.device_count() >= 1, "GPU tests need a HIP device"
    return llcomp_amd


def golden(gen, w, tw, th, planar):
    return [v for v in SLC if v["gen"] == gen and v["w"] == w and v["tile_w"] == tw and v["tile_h"] == th and v["planar"] == planar][0]


def run_stream(mi, frames, tw, th, planar, depth, max_encodes_in_flight, check=None, frames_per_job=1):
    """Streams every frame through encode and straight back through decode (llcomp_amd.pipeline_roundtrip: the loop
    bench.py's C5 leg uses too); every decoded frame is compared with its source inside."""
    h, w, c = frames[0].shape
    st = mi.Stream(w, h, c, tw, th, planar, depth=depth, frames_per_job=frames_per_job)
    lens, done_at, busy_seen = mi.pipeline_roundtrip(st, frames, max_encodes_in_flight, on_container=check, verify=True)
    assert st.pending() == 0
    st.close()
    return lens, done_at, busy_seen


def test_stream_small_frames_match_oracle(mi, orc):
    """Every container the pipeline returns == the oracle's container of that frame (several slicings, ragged shapes,
    more frames than slots so that slots are reused and back-pressure is exercised)."""
    for (w, h, c, tw, th, planar) in ((200, 37, 3, 50, 1, True), (131, 40, 4, 32, 16, True), (97, 21, 1, 97, 1, False), (64, 64, 3, 64, 64, False)):
        frames = [np.ascontiguousarray(np.roll(make_image(("g3", "mid", "g1", "checker")[i % 4], w, h, c), 3 * i, axis=1)) for i in range(11)]
        want = [orc.compress_sliced(f, tw, th, planar) for f in frames]

        def check(i, data):
            assert data.tobytes() == want[i], f"frame {i}: container differs from the oracle's"

        lens, _, busy = run_stream(mi, frames, tw, th, planar, depth=3, max_encodes_in_flight=2, check=check)
        assert lens == [len(x) for x in want]
        assert busy > 0, "11 frames through 3 slots must have hit back-pressure"


def test_two_pipelines_on_two_threads_match_oracle(mi, orc):
    """bench.py's config-5 leg drives two stream objects from two threads (the library is re-entrant: private HIP streams,
    a mutex around the lane and device-memory caches); every container of both must equal the oracle's."""
    import threading

    rng = np.random.default_rng(21)
    sets = [[rng.integers(0, 256, size=(60, 200, 3), dtype=np.uint8) for _ in range(12)],
            [make_image("mid", 131, 77, 3) + np.uint8(i) for i in range(12)]]
    slicing = [(64, 1, True), (32, 16, False)]
    want = [[orc.compress_sliced(f, *slicing[t]) for f in sets[t]] for t in range(2)]
    errs = []

    def drive(t):
        try:
            def check(i, data):
                assert data.tobytes() == want[t][i], f"pipeline {t}, frame {i}: container differs from the oracle's"

            run_stream(mi, sets[t], *slicing[t], depth=3, max_encodes_in_flight=2, check=check)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=drive, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs


@pytest.mark.parametrize("fpj", [2, 3])
def test_stream_jobs_of_several_frames_match_oracle(mi, orc, fpj):
    """frames_per_job > 1: one launch set and two copies per frame instead of a whole pipeline pass per frame; every
    frame still gets its own container, equal to the oracle's."""
    for (w, h, c, tw, th, planar) in ((200, 37, 3, 50, 1, True), (131, 40, 4, 32, 16, True), (97, 21, 1, 97, 1, False)):
        n = 4 * fpj
        buf = np.empty((n, h, w, c), np.uint8)  # the frames of a job must be adjacent in memory
        for i in range(n):
            buf[i] = np.roll(make_image(("g3", "mid", "g1", "checker")[i % 4], w, h, c), 3 * i, axis=1)
        frames = [buf[i] for i in range(n)]
        want = [orc.compress_sliced(f, tw, th, planar) for f in frames]

        def check(i, data):
            assert data.tobytes() == want[i], f"frame {i}: container differs from the oracle's"

        lens, _, _ = run_stream(mi, frames, tw, th, planar, depth=3, max_encodes_in_flight=2, check=check, frames_per_job=fpj)
        assert lens == [len(x) for x in want]
    st = mi.Stream(64, 16, 3, 16, 1, True, depth=2, frames_per_job=2)
    with pytest.raises(mi.LlcompError):  # the one-container call does not fit jobs of two frames
        mi._lib.load() and mi._check(mi._lib.load().llcomp_mi_stream_submit_decode(st._h, want[0], len(want[0]), 0))
    a = np.zeros((2, 16, 64, 3), np.uint8)
    assert st.submit_encode(a, 5)
    job = st.wait()
    assert job.tag == 5 and len(job.data) == 2 and job.data[0].tobytes() == job.data[1].tobytes()
    cut = job.data[1][: job.data[1].size - 1].copy()  # a container shorter than its table promises
    with pytest.raises(mi.LlcompError) as e:
        st.submit_decode([job.data[0], cut], 6)
    assert e.value.status == mi.TRUNCATED
    st.release(job)
    st.close()


def test_stream_api_errors_and_backpressure(mi):
    w, h, c = 64, 16, 3
    st = mi.Stream(w, h, c, 16, 1, True, depth=2)
    a = np.zeros((h, w, c), np.uint8)
    assert st.submit_encode(a, 1) and st.submit_encode(a, 2)
    assert st.submit_encode(a, 3) is False          # both slots occupied: BUSY, not an exception, nothing queued
    assert st.pending() == 2
    j1 = st.wait()
    assert st.submit_encode(a, 3) is False          # finished but not released: still occupied
    j2 = st.wait()
    assert (j1.tag, j2.tag) == (1, 2) and j1.data.tobytes() == j2.data.tobytes()
    with pytest.raises(mi.LlcompError):              # nothing pending
        st.wait()
    other = mi.compress_image(np.zeros((h, w + 1, c), np.uint8), w + 1, h, c, format=mi.FORMAT_SLICED, tile_w=16, tile_h=1, planar=True)
    with pytest.raises(mi.LlcompError) as e:         # a stream object codes one geometry
        st.submit_decode(np.frombuffer(other, np.uint8))
    assert e.value.status == mi.BAD_ARGS
    st.release(j1)
    with pytest.raises(mi.LlcompError):              # released twice
        st.release(j1)
    bad = j2.data.copy()
    bad[24 + 4 * int.from_bytes(bad[20:24].tobytes(), "little"):] = 0xFF   # payload of all ones: runs of 1 bins -> "Invalid exponent" or garbage, never a fault
    assert st.submit_decode(bad, 9)
    j3 = st.wait()
    assert j3.tag == 9 and j3.status in (mi.OK, mi.BAD_EXPONENT)
    st.release(j2)
    st.release(j3)
    st.close()


def test_host_calls_with_caller_provided_pinned_buffers(mi, orc):
    """llcomp_mi_encode_into / llcomp_mi_decode_into with pinned buffers from llcomp_mi_host_alloc: same bytes as the
    allocating calls; too-small buffers are reported with the size it takes and are not written."""
    img = make_image("mid", 300, 70, 3)
    src = mi.PinnedBuffer(img.size)
    src.array[:] = img.reshape(-1)
    out = mi.PinnedBuffer(2 * img.size + 65536)
    for kw, want in ((dict(), orc.compress_image(img)),
                     (dict(format=mi.FORMAT_SLICED, tile_w=64, tile_h=1, planar=True), orc.compress_sliced(img, 64, 1, True)),
                     (dict(format=mi.FORMAT_SLICED, tile_w=64, tile_h=32, planar=False), orc.compress_sliced(img, 64, 32, False))):
        n = mi.compress_image_into(src.array, 300, 70, 3, out.array, **kw)
        assert out.array[:n].tobytes() == want
        back = mi.PinnedBuffer(img.size)
        back.array[:] = 0xAA
        assert mi.decompress_image_into(out.array[:n], back.array) == (300, 70, 3)
        assert np.array_equal(back.array.reshape(img.shape), img)
        small = np.full(len(want) - 1, 0x55, np.uint8)
        with pytest.raises(mi.LlcompError) as e:
            mi.compress_image_into(src.array, 300, 70, 3, small, **kw)
        assert e.value.status == mi.OUTPUT_OVERFLOW and e.value.needed == len(want) and bool((small == 0x55).all())
        small = np.full(img.size - 1, 0x55, np.uint8)
        with pytest.raises(mi.LlcompError) as e:
            mi.decompress_image_into(out.array[:n], small)
        assert e.value.status == mi.OUTPUT_OVERFLOW and e.value.shape == (300, 70, 3) and bool((small == 0x55).all())
        back.close()
    src.close()
    out.close()


@pytest.mark.parametrize("gen", ["g3", "g2", "mid"])
def test_c3_4k_bench_slicing_golden(mi, orc, gen):
    """The slicing bench.py measures (per-channel planes, 480x1) at full 4K size: container bytes pinned to the real
    reference's per-slice streams (tests/golden/slice_payloads.json), then decoded back."""
    v = golden(gen, 3840, 480, 1, True)
    img = make_image(gen, 3840, 2160, 3)
    s = mi.compress_image(img, 3840, 2160, 3, format=mi.FORMAT_SLICED, tile_w=480, tile_h=1, planar=True)
    assert len(s) == v["container_len"] and fnv_hex(orc, s) == v["container_fnv1a64"]
    assert np.array_equal(mi.decompress_image(s).pixels, img)


@pytest.mark.parametrize("fpj", [1, 4])
def test_c5_stream_64_frames_4k(mi, orc, fpj):
    """BASELINE config 5: 64 distinct 4K RGB8 noise frames (std::mt19937 seeds 1234+i) streamed host -> GPU -> host ->
    GPU -> host through llcomp_mi_stream_*.  Every frame bit-exact; frames 0 and 63 pinned to golden container hashes made
    with the real reference; steady state (first 4 frames excluded) and compression ratio reported."""
    from llcomp_amd import synth

    W, H, C, N = 3840, 2160, 3, 64
    pinned = mi.PinnedBuffer(N * W * H * C)  # source frames in pinned memory: H2D is plain DMA
    frames = []
    for i in range(N):
        f = pinned.array[i * W * H * C:(i + 1) * W * H * C].reshape(H, W, C)
        f[:] = synth.gen_g3(W, H, C, seed=1234 + i)
        frames.append(f)
    pins = {0: golden("g3", 3840, 480, 1, True), 63: golden("g3@1297", 3840, 480, 1, True)}

    def check(i, data):
        if i in pins:
            assert data.size == pins[i]["container_len"] and fnv_hex(orc, data.tobytes()) == pins[i]["container_fnv1a64"], f"frame {i}: golden mismatch"

    lens, done_at, _ = run_stream(mi, frames, 480, 1, True, depth=8 if fpj == 1 else 6, max_encodes_in_flight=3 if fpj == 1 else 2, check=check, frames_per_job=fpj)
    steady = (N - 4) * W * H / 1e6 / (done_at[-1] - done_at[3])  # (jobs of 4 frames: the first job is excluded)
    ratio = N * W * H * C / sum(lens)
    print(f"\nC5 stream ({fpj} frame(s) per job): {N} frames, steady state {steady:.0f} MPix/s end to end over PCIe (first 4 frames excluded), ratio {ratio:.4f} "
          f"(reference whole-image stream of frame 0: 0.8026)")
    assert 0.76 < ratio < 0.79
    assert steady > 500  # one pageable 4K frame through the round-1 host calls ran at 640 MPix/s; a pipeline must not be slower
    pinned.close()


def test_cpp_stream_driver(mi):
    """tools/llcomp_stream: BASELINE config 5's loop written in C++ against the C ABI alone (pinned sources, back-pressure,
    encode results handed to submit_decode_batch, memcmp on worker threads); it exits non-zero on any mismatch."""
    import json
    import os
    import subprocess

    from conftest import ROOT

    exe = os.path.join(ROOT, "tools", "llcomp_stream")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    for args in (["12", "300", "200", "50", "1", "3", "2", "1"], ["12", "300", "200", "64", "16", "3", "2", "3"], ["8", "3840", "2160", "480", "1", "4", "2", "2"],
                 ["24", "300", "200", "64", "1", "3", "2", "2", "2"]):  # (the last: two pipelines on two threads)
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out["verified"] is True and out["frames"] == int(args[0]) and out["frames_per_job"] == int(args[7])
        assert 0.5 < out["compression_ratio"] < 0.85  # xorshift noise


def test_chunked_snapshot_pass_against_the_real_reference_and_at_4k(mi, orc, monkeypatch):
    """Slices above 4096 samples (round 6: the snapshot pass in chunks of 4096, the coder in segments, the pass one chunk ahead on the
    device's second stream).  (1) 1080p in 128x128 planes with the lane-group width forced to 64 -- so that its 405 slices share
    wavefronts instead of getting one each with the table in LDS -- against the golden container made from the REAL reference's
    per-slice streams; (2) at BASELINE's full 4K size, two frames per call through a device-resident codec object: 64x64 tiles with
    the channels interleaved (12288 samples: three chunks) and 128x128 planes (four), every frame's table and payload against the
    oracle's container, decoded back bit-exact; the same with the pass in order (LLCOMP_MI_OVERLAP=0)."""
    import torch

    v = golden("mid", 1920, 128, 128, True)
    img = make_image("mid", 1920, 1080, 3)
    monkeypatch.setenv("LLCOMP_MI_LANE_SHIFT", "6")
    mi.reload_tuning()
    try:
        k = mi.Codec(1, 1920, 1080, 3, 128, 128, True)
        assert k.family["snapshot"] and not k.family["lds_table"], k.family
        k.close()
        s = mi.compress_image(img, 1920, 1080, 3, format=mi.FORMAT_SLICED, tile_w=128, tile_h=128, planar=True)
        assert len(s) == v["container_len"] and fnv_hex(orc, s) == v["container_fnv1a64"], "chunked snapshot container differs from the real reference's"
        assert np.array_equal(mi.decompress_image(s).pixels, img)
    finally:
        monkeypatch.delenv("LLCOMP_MI_LANE_SHIFT")
        mi.reload_tuning()
    frames = np.stack([make_image("nat", 3840, 2160, 3), make_image("g3@77", 3840, 2160, 3)])
    st = torch.cuda.current_stream().cuda_stream
    d_px = torch.from_numpy(frames).cuda()
    for tw, th, planar in ((64, 64, False), (128, 128, True)):
        want = [orc.compress_sliced(frames[f], tw, th, planar) for f in range(2)]
        for overlap in (None, "0"):
            if overlap is not None:
                monkeypatch.setenv("LLCOMP_MI_OVERLAP", overlap)
            mi.reload_tuning()
            codec = mi.Codec(2, 3840, 2160, 3, tw, th, planar)
            assert codec.family["snapshot"] and not codec.family["lds_table"], codec.family
            cap = min(codec.max_payload_bytes, 2 * frames.size + 64 * codec.n_slices + 4096)
            d_pay = torch.empty(cap, dtype=torch.uint8, device="cuda")
            d_len = torch.empty(codec.n_slices, dtype=torch.int32, device="cuda")
            d_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
            d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
            for _ in range(2):  # twice on one object: the carry table's generation moves on, the parked coders are overwritten
                codec.encode(d_px.data_ptr(), d_pay.data_ptr(), cap, d_len.data_ptr(), d_tot.data_ptr(), d_st.data_ptr(), st)
            torch.cuda.synchronize()
            assert int(d_st.item()) == 0
            total, spf = int(d_tot.item()), codec.n_slices // 2
            lens = d_len.cpu().numpy().astype(np.int64)
            pay = d_pay[:total].cpu().numpy().tobytes()
            offs = np.concatenate([[0], np.cumsum(lens)])
            for f in range(2):
                assert lens[f * spf:(f + 1) * spf].astype("<u4").tobytes() == want[f][24:24 + 4 * spf], (tw, th, f, overlap)
                assert pay[offs[f * spf]:offs[(f + 1) * spf]] == want[f][24 + 4 * spf:], (tw, th, f, overlap)
            d_out = torch.zeros_like(d_px)
            codec.decode(d_pay.data_ptr(), total, d_len.data_ptr(), d_out.data_ptr(), d_st.data_ptr(), st)
            torch.cuda.synchronize()
            assert int(d_st.item()) == 0 and torch.equal(d_out, d_px)
            codec.close()
            if overlap is not None:
                monkeypatch.delenv("LLCOMP_MI_OVERLAP")
                mi.reload_tuning()
