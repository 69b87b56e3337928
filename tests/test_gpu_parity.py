"""Parity of the HIP path (through the C ABI of libllcomp_mi.so) against the golden vectors made by the real
reference and against the pinned oracle.  Everything here is integer/byte work: the bar is bit-exact."""
import numpy as np
import pytest
from conftest import fnv_hex, load_golden, make_image

pytestmark = pytest.mark.gpu

KAT = load_golden("kat_streams.json")["vectors"]
SLC = load_golden("slice_payloads.json")["vectors"]
DEC = load_golden("decode_behaviour.json")["vectors"]
SMALL = load_golden("small_model.json")["vectors"]  # the reference compiled with LargeModel = false


def _id(v):
    return "-".join(str(v[k]) for k in ("gen", "w", "h", "c") if k in v) + ("-t%dx%d%s" % (v["tile_w"], v["tile_h"], "p" if v["planar"] else "i") if "tile_w" in v else "")


@pytest.fixture(scope="module")
def mi():
    import llcomp_amd

    assert llcomp_amd.device_count() >= 1, "GPU tests need a HIP device"
    return llcomp_amd


@pytest.fixture
def set_hook(mi, monkeypatch):
    """The library reads its LLCOMP_MI_* test hooks once per process; a test that changes one says so (reload_tuning)."""
    def _set(name, value):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, value)
        mi.reload_tuning()

    yield _set
    monkeypatch.undo()
    mi.reload_tuning()


# ---- P0: legacy whole-image format, one serial lane on the GPU --------------------------------------------------
LEGACY_CASES = [v for v in KAT if v["w"] * v["h"] * v["c"] <= 1920 * 1080 * 3 and not (v["gen"] == "g3" and v["w"] >= 1920)]


@pytest.mark.parametrize("v", LEGACY_CASES, ids=_id)
def test_p0_legacy_stream_equals_reference(mi, orc, v):
    img = make_image(v["gen"], v["w"], v["h"], v["c"])
    s = mi.compress_image(img, v["w"], v["h"], v["c"])
    assert len(s) == v["len"]
    assert fnv_hex(orc, s) == v["fnv1a64"]
    if "hex" in v:
        assert s.hex() == v["hex"]
    out = mi.decompress_image(s)
    assert (out.width, out.height, out.channels) == (v["w"], v["h"], v["c"])
    assert np.array_equal(out.pixels, img)


# ---- P1: sliced container == container assembled from reference payloads of the crops ---------------------------
@pytest.mark.parametrize("v", SLC, ids=_id)
def test_p1_sliced_container_equals_reference_payloads(mi, orc, v):
    img = make_image(v["gen"], v["w"], v["h"], v["c"])
    s = mi.compress_image(img, v["w"], v["h"], v["c"], format=mi.FORMAT_SLICED, tile_w=v["tile_w"], tile_h=v["tile_h"], planar=v["planar"])
    assert len(s) == v["container_len"]
    assert fnv_hex(orc, s) == v["container_fnv1a64"]
    if "container_hex" in v:
        assert s.hex() == v["container_hex"]
    out = mi.decompress_image(s)
    assert np.array_equal(out.pixels, img)


# ---- HIP vs oracle on random shapes / slicings (edge cases: 1-pixel strips, ragged tiles, all channel counts) ---
@pytest.mark.parametrize("seed", range(48))
def test_random_shapes_match_oracle(mi, orc, seed):
    rng = np.random.default_rng(seed)
    w, h, c = int(rng.integers(1, 68)), int(rng.integers(1, 68)), int(rng.integers(1, 5))
    kind = seed % 4
    if kind == 0:
        img = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
    elif kind == 1:
        img = np.full((h, w, c), int(rng.integers(0, 256)), np.uint8)
    elif kind == 2:
        y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
        img = ((x * 3 + y * 5 + k * 11 + rng.integers(-2, 3, size=(h, w, c))) & 0xFF).astype(np.uint8)
    else:
        y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
        img = (((x + y + k) & 1) * 255).astype(np.uint8)  # saturated checkerboard: max residuals, carries
    assert mi.compress_image(img, w, h, c) == orc.compress_image(img)
    tw, th = int(rng.integers(1, w + 3)), int(rng.integers(1, h + 3))
    for planar in (False, True):
        s = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar)
        assert s == orc.compress_sliced(img, tw, th, planar)
        assert np.array_equal(mi.decompress_image(s).pixels, img)
    # the GPU decoder also reads what the oracle wrote in legacy form
    assert np.array_equal(mi.decompress_image(orc.compress_image(img)).pixels, img)


@pytest.mark.parametrize("c", [5, 6, 9, 17])
def test_more_than_four_channels_match_oracle(mi, orc, c):
    """The reference passes channels beyond the third through untransformed for any count its one-byte header holds
    (llcomp.hpp:407-409, 541-543).  Here they run through the generic (any channel count) kernels: legacy stream, sliced
    interleaved and planar, 1-row slices included."""
    rng = np.random.default_rng(700 + c)
    w, h = int(rng.integers(20, 60)), int(rng.integers(6, 30))
    y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
    img = ((x * 3 + y * 5 + k * 29 + rng.integers(-3, 4, size=(h, w, c))) & 0xFF).astype(np.uint8)
    img[:, w // 2:] = rng.integers(0, 256, size=(h, w - w // 2, c), dtype=np.uint8)
    s = mi.compress_image(img, w, h, c)
    assert s == orc.compress_image(img)
    out = mi.decompress_image(s)
    assert out.channels == c and np.array_equal(out.pixels, img)
    for tw, th in ((16, 8), (w, 1), (11, 1)):
        for planar in (False, True):
            t = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar)
            assert t == orc.compress_sliced(img, tw, th, planar), (tw, th, planar)
            assert np.array_equal(mi.decompress_image(t).pixels, img)


@pytest.mark.parametrize("c", [5, 7])
def test_more_than_four_channels_through_the_snapshot_pass(mi, orc, c):
    """Interleaved slices with more than four channels miss the one-row kernels and the specialised 2-D kernels; once there are
    enough of them to share wavefronts (>= 192 slices) they take the snapshot encoder with the generic symbol path -- one-row
    slices and 2-D tiles, the decoder reading its neighbours back from memory.  Containers == the oracle's, round trips lossless."""
    rng = np.random.default_rng(900 + c)
    w, h = 260, 48
    y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
    img = ((x * 3 + y * 5 + k * 29 + rng.integers(-3, 4, size=(h, w, c))) & 0xFF).astype(np.uint8)
    img[:, w // 2:] = rng.integers(0, 256, size=(h, w - w // 2, c), dtype=np.uint8)
    for tw, th in ((52, 1), (20, 12), (9, 5)):  # 240 one-row slices, 52 x 4 = 208 tiles, 29 x 10 = 290 tiles
        t = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=False)
        assert t == orc.compress_sliced(img, tw, th, False), (tw, th)
        assert np.array_equal(mi.decompress_image(t).pixels, img), (tw, th)


@pytest.mark.parametrize("v", SMALL, ids=lambda v: v["kind"] + "-" + _id(v))
def test_small_model_equals_reference_built_with_largemodel_false(mi, orc, v):
    """the bitstream of a reference built with `LargeModel = false` (llcomp.hpp:21, 26-32, 427-429).  Golden
    vectors from the real header compiled with the constant flipped.  The legacy header does not record the variant (the
    caller passes small_model on both sides); the sliced container carries it in its flags byte."""
    img = make_image(v["gen"], v["w"], v["h"], v["c"])
    if v["kind"] == "legacy":
        if v["w"] * v["h"] > 700 * 400:
            pytest.skip("a lone serial stream of this size takes seconds on one GPU lane: test_small_model_large_legacy_streams (slow) runs it")
        s = mi.compress_image(img, v["w"], v["h"], v["c"], small_model=True)
        assert len(s) == v["len"] and fnv_hex(orc, s) == v["fnv1a64"]
        if "hex" in v:
            assert s.hex() == v["hex"]
        assert np.array_equal(mi.decompress_image(s, small_model=True).pixels, img)
        if v["w"] * v["h"] > 64:  # decoded with the wrong model it is a different image (or an error), never a crash
            try:
                assert not np.array_equal(mi.decompress_image(s).pixels, img)
            except mi.LlcompError as e:
                assert e.status == mi.BAD_EXPONENT
    else:
        s = mi.compress_image(img, v["w"], v["h"], v["c"], format=mi.FORMAT_SLICED, tile_w=v["tile_w"], tile_h=v["tile_h"], planar=v["planar"], small_model=True)
        assert len(s) == v["container_len"] and fnv_hex(orc, s) == v["container_fnv1a64"]
        assert mi.probe(s).small_model == 1
        assert np.array_equal(mi.decompress_image(s).pixels, img)  # the container says which model wrote it


@pytest.mark.slow
@pytest.mark.parametrize("v", [v for v in SMALL if v["kind"] == "legacy" and v["w"] * v["h"] > 700 * 400], ids=lambda v: _id(v))
def test_small_model_large_legacy_streams(mi, orc, v):
    """the 1080p whole-image streams of the LargeModel = false reference: one GPU lane each, seconds per stream"""
    img = make_image(v["gen"], v["w"], v["h"], v["c"])
    s = mi.compress_image(img, v["w"], v["h"], v["c"], small_model=True)
    assert len(s) == v["len"] and fnv_hex(orc, s) == v["fnv1a64"]
    assert np.array_equal(mi.decompress_image(s, small_model=True).pixels, img)


# ---- decoder behaviour on damaged streams == the real reference's ------------------------------------------------
@pytest.mark.parametrize("v", DEC, ids=lambda v: v["name"])
def test_decoder_behaviour_equals_reference(mi, orc, v):
    data = bytes.fromhex(v["hex"])
    if v["name"] == "exponent_run_31":
        pytest.skip("e == 31 overflows int32 in the reference (UB)")
    if v["rc"] == 0:
        out = mi.decompress_image(data)
        assert (out.width, out.height, out.channels) == (v["w"], v["h"], v["c"])
        assert fnv_hex(orc, out.pixels.tobytes()) == v["pixels_fnv1a64"]
    else:
        with pytest.raises(mi.LlcompError) as e:
            mi.decompress_image(data)
        if v["rc"] in (1, 2):
            assert e.value.status == v["rc"]
            assert str(e.value) == {1: "Invalid magic number", 2: "Invalid exponent"}[v["rc"]]


def test_error_codes(mi):
    with pytest.raises(mi.LlcompError) as e:
        mi.decompress_image(b"")
    assert e.value.status == mi.TRUNCATED
    with pytest.raises(mi.LlcompError) as e:
        mi.decompress_image(bytes([0x79, 3, 4]))
    assert e.value.status == mi.TRUNCATED
    with pytest.raises(mi.LlcompError) as e:
        mi.compress_image(np.zeros(70000 * 3, np.uint8), 70000, 1, 3)  # u16 header field (D4)
    assert e.value.status == mi.OUT_OF_RANGE
    with pytest.raises(mi.LlcompError) as e:
        mi.compress_image(np.zeros(256 * 4, np.uint8), 2, 2, 256)  # the channel count is one byte in both headers
    assert e.value.status == mi.BAD_ARGS
    with pytest.raises(mi.LlcompError) as e:
        mi.Codec(1, 32768, 32768, 3)  # w*h*c >= 2^31 (llcomp.hpp:359 `int size`)
    assert e.value.status == mi.OUT_OF_RANGE
    # a legacy-sized giant (16384^2 RGB = 805 M samples, beyond the 13 B/sample scratch bound of one 32-bit-addressed slice)
    # is accepted: the codec object can be built (round 1 returned BAD_ARGS here); coding it on one lane would take hours
    big = mi.Codec(1, 16384, 16384, 3)
    assert big.n_slices == 1
    big.close()
    # sliced format has u32 dimensions: a 70000-pixel-wide strip is fine there
    img = (np.arange(70000 * 3) & 0xFF).astype(np.uint8).reshape(1, 70000, 3)
    s = mi.compress_image(img, 70000, 1, 3, format=mi.FORMAT_SLICED, tile_w=1000)
    assert np.array_equal(mi.decompress_image(s).pixels, img)


@pytest.mark.parametrize("shape", [(65535, 1, 1), (1, 65535, 3), (65535, 2, 2)])
def test_legacy_header_maximum_dimensions(mi, orc, shape):
    """the largest width / height the reference's u16 header fields hold (llcomp.hpp:377-378); one more is OUT_OF_RANGE"""
    w, h, c = shape
    rng = np.random.default_rng(w + h)
    img = (np.cumsum(rng.integers(-3, 4, size=(h, w, c)), axis=1 if w > 1 else 0) & 0xFF).astype(np.uint8)
    s = mi.compress_image(img, w, h, c)
    assert s == orc.compress_image(img)
    out = mi.decompress_image(s)
    assert (out.width, out.height, out.channels) == (w, h, c) and np.array_equal(out.pixels, img)
    with pytest.raises(mi.LlcompError) as e:
        mi.compress_image(np.zeros((h + (w == 1), w + (w > 1), c), np.uint8), w + (w > 1), h + (w == 1), c)
    assert e.value.status == mi.OUT_OF_RANGE


def test_empty_and_mismatched_inputs_are_rejected(mi):
    for w, h, c in ((0, 4, 3), (4, 0, 3), (4, 4, 0)):
        with pytest.raises(mi.LlcompError) as e:
            mi.compress_image(np.zeros(0, np.uint8), w, h, c)
        assert e.value.status == mi.BAD_ARGS
    with pytest.raises(mi.LlcompError) as e:   # the reference only asserts size == w*h*c (llcomp.hpp:361)
        mi.compress_image(np.zeros(47, np.uint8), 4, 4, 3)
    assert e.value.status == mi.BAD_ARGS
    with pytest.raises(mi.LlcompError) as e:   # degenerate header: zero width
        mi.decompress_image(bytes([0x79, 3, 0, 0, 4, 0, 1, 2, 3]))
    assert e.value.status == mi.BAD_ARGS


def test_opts_struct_of_another_abi_is_refused(mi, orc):
    """One struct layout per ABI version (llcomp_mi.h): llcomp_mi_opts is checked through struct_size and a caller built
    against another header -- ABI 1 had no small_model field and passed 24, ABI 2 / 3 had no device list and passed 28 -- is refused
    instead of being half-read; the binding checks llcomp_mi_abi_version() when it loads the library."""
    import ctypes as C

    from llcomp_amd import _lib

    class Opts1(C.Structure):
        _fields_ = [("struct_size", C.c_uint32), ("format", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
                    ("planar", C.c_uint32), ("device", C.c_int32), ("pad", C.c_uint32)]

    img = make_image("mid", 50, 20, 3)
    L = _lib.load()
    assert L.llcomp_mi_abi_version() == _lib.ABI_VERSION
    out, n = _lib.u8p(), C.c_size_t()
    for size in (24, 20, 32, 28):
        o = Opts1(size, mi.FORMAT_SLICED, 16, 1, 1, -1, 0)
        assert L.llcomp_mi_encode(img.ctypes.data_as(_lib.u8p), 50, 20, 3, C.cast(C.byref(o), C.POINTER(_lib.Opts)), C.byref(out), C.byref(n)) == mi.BAD_ARGS
    o = _lib.Opts(C.sizeof(_lib.Opts), mi.FORMAT_SLICED, 16, 1, 1, -1, 0, 0, None, 0, 0)
    assert L.llcomp_mi_encode(img.ctypes.data_as(_lib.u8p), 50, 20, 3, C.cast(C.byref(o), C.POINTER(_lib.Opts)), C.byref(out), C.byref(n)) == mi.OK
    try:
        assert C.string_at(out, n.value) == orc.compress_sliced(img, 16, 1, True)
    finally:
        L.llcomp_mi_free(out)


# ---- stage A kernel alone vs the oracle's intermediate dump ------------------------------------------------------
@pytest.mark.parametrize("shape", [(300, 70, 3), (257, 33, 1), (64, 64, 4), (5, 3, 2), (1030, 40, 3)])
def test_model_kernel_matches_oracle_symbols(mi, orc, shape):
    import torch

    w, h, c = shape
    img = np.random.default_rng(w * h + c).integers(0, 256, size=(h, w, c), dtype=np.uint8)
    img[h // 2:, :, :] = (img[h // 2:, :, :] >> 5) + 100  # low-entropy half
    codec = mi.Codec(1, w, h, c)
    d_px = torch.from_numpy(img).cuda()
    d_sym = torch.zeros(h * w * c, dtype=torch.int32, device="cuda")
    codec.model(d_px.data_ptr(), d_sym.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sym = d_sym.cpu().numpy().view(np.uint32).reshape(h, w, c)
    ctx, res = orc.model_samples(orc.forward_rct(img))
    assert np.array_equal(sym & 0xFFFF, ctx.astype(np.uint32))
    assert np.array_equal((sym >> 16).astype(np.uint16).view(np.int16), res)
    codec.close()


# ---- device-resident batch codec --------------------------------------------------------------------------------
def _batch_roundtrip(mi, orc, frames, w, h, c, tw, th, planar, gens, small_model=False, expect=None):
    """`expect`: kernel-family properties the codec object must report (Codec.family), so that a test runs the family it means to"""
    import torch

    imgs = np.stack([make_image(gens[i % len(gens)], w, h, c) for i in range(frames)])
    for i in range(frames):
        imgs[i] = np.roll(imgs[i], i * 7, axis=1)
    codec = mi.Codec(frames, w, h, c, tw, th, planar, small_model=small_model) if small_model else mi.Codec(frames, w, h, c, tw, th, planar)
    for key, val in (expect or {}).items():
        assert codec.family[key] == val, (key, codec.family)
    st = torch.cuda.current_stream().cuda_stream
    d_px = torch.from_numpy(imgs).cuda()
    cap = min(codec.max_payload_bytes, 2 * imgs.size + 64 * codec.n_slices + 4096)
    d_pay = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_len = torch.empty(codec.n_slices, dtype=torch.int32, device="cuda")
    d_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
    codec.encode(d_px.data_ptr(), d_pay.data_ptr(), cap, d_len.data_ptr(), d_tot.data_ptr(), d_st.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(d_st.item()) == 0
    total = int(d_tot.item())
    lens = d_len.cpu().numpy().astype(np.int64)
    assert lens.sum() == total
    pay = d_pay[:total].cpu().numpy().tobytes()
    spf = codec.n_slices // frames
    offs = np.concatenate([[0], np.cumsum(lens)])
    for f in range(frames):  # every frame's slices == the oracle's container of that frame
        ref = orc.compress_sliced(imgs[f], tw, th, planar)
        ref_lens = np.frombuffer(ref[24:24 + 4 * spf], dtype="<u4").astype(np.int64)
        assert np.array_equal(lens[f * spf:(f + 1) * spf], ref_lens)
        assert pay[offs[f * spf]:offs[(f + 1) * spf]] == ref[24 + 4 * spf:]
    d_out = torch.zeros_like(d_px)
    codec.decode(d_pay.data_ptr(), total, d_len.data_ptr(), d_out.data_ptr(), d_st.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(d_st.item()) == 0
    assert torch.equal(d_out, d_px)
    codec.close()
    return total


def test_batch_codec_small(mi, orc):
    _batch_roundtrip(mi, orc, 5, 100, 37, 3, 32, 16, True, ["g1", "g3", "mid", "checker"])
    _batch_roundtrip(mi, orc, 3, 100, 37, 4, 32, 16, False, ["g1", "g3", "mid"])
    _batch_roundtrip(mi, orc, 2, 61, 50, 1, 61, 1, False, ["g3", "g2"])


@pytest.mark.parametrize("nocache", ["0", "1"])
def test_state_tables_survive_generation_wrap(mi, orc, set_hook, nocache):
    """State tables in HBM are not cleared per call: every bank carries the 8-bit generation of the call that wrote it
    (slice_kernels.hip, bank_fresh) and the table is cleared for real once per 255 calls.  600 calls on one codec object,
    three different batches in turn, across two wraps of the generation: every payload equals the oracle's, every decode
    is lossless -- stale states of an earlier call must never leak into a later one."""
    import torch

    set_hook("LLCOMP_MI_NOCACHE", nocache)  # (the decoder's bank cache in LDS in front of the tables, or every bank straight from them)
    frames, w, h, c, tw, th = 2, 96, 80, 3, 8, 8   # 720 slices per call: four slices per wavefront -> tables in HBM (fewer than 192
    codec = mi.Codec(frames, w, h, c, tw, th, True)  # slices would get one wavefront each and their tables in LDS)
    assert not codec.family["lds_table"] and not codec.family["rows"] and codec.family["bank_cache"] == (nocache == "0"), codec.family
    st = torch.cuda.current_stream().cuda_stream
    batches, wants = [], []
    for k, gens in enumerate((["g3", "mid"], ["checker", "g1"], ["mid", "g3"])):
        imgs = np.stack([np.roll(make_image(g, w, h, c), 3 * k + i, axis=1) for i, g in enumerate(gens)])
        batches.append(torch.from_numpy(imgs).cuda())
        ref = [orc.compress_sliced(imgs[f], tw, th, True) for f in range(frames)]
        spf = codec.n_slices // frames
        wants.append(b"".join(r[24 + 4 * spf:] for r in ref))
    cap = max(len(x) for x in wants) + 4096
    d_pay = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_len = torch.empty(codec.n_slices, dtype=torch.int32, device="cuda")
    d_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    d_st = torch.zeros(2, dtype=torch.int32, device="cuda")
    d_out = torch.empty_like(batches[0])
    for i in range(300):  # 600 calls: encode + decode each use a generation
        b = i % 3
        codec.encode(batches[b].data_ptr(), d_pay.data_ptr(), cap, d_len.data_ptr(), d_tot.data_ptr(), d_st.data_ptr(), st)
        codec.decode(d_pay.data_ptr(), cap, d_len.data_ptr(), d_out.data_ptr(), d_st[1:].data_ptr(), st)
        if i < 6 or i % 17 == 0 or 120 <= i <= 135 or i >= 290:  # dense around the wraps (call 255 = i 127, call 510 = i 254/255)
            torch.cuda.synchronize()
            assert d_st.tolist() == [0, 0]
            total = int(d_tot.item())
            assert d_pay[:total].cpu().numpy().tobytes() == wants[b], f"call {2 * i}: payload differs from the oracle's"
            assert torch.equal(d_out, batches[b]), f"call {2 * i + 1}: round trip"
    for i in range(250, 262):
        b = i % 3
        codec.encode(batches[b].data_ptr(), d_pay.data_ptr(), cap, d_len.data_ptr(), d_tot.data_ptr(), d_st.data_ptr(), st)
        codec.decode(d_pay.data_ptr(), cap, d_len.data_ptr(), d_out.data_ptr(), d_st[1:].data_ptr(), st)
        torch.cuda.synchronize()
        assert d_st.tolist() == [0, 0] and d_pay[:int(d_tot.item())].cpu().numpy().tobytes() == wants[b] and torch.equal(d_out, batches[b])
    codec.close()


def test_device_range_sums_and_pool_limit(mi):
    """two small entry points of the multi-GPU path / the device-memory cache"""
    import ctypes as C

    import torch

    from llcomp_amd import _lib

    L = _lib.load()
    rng = np.random.default_rng(5)
    vals = rng.integers(0, 5000, size=100_000, dtype=np.uint32)
    vals[77] = 0xFFFFFFF0  # a damaged length: clamped to the cap, never negative
    start = np.array([0, 10, 70, 5000, 99_990, 100_000], dtype=np.int64)
    count = np.array([10, 100, 20, 60_000, 10, 0], dtype=np.int64)
    cap = 4000
    d_vals, d_start, d_count = (torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda() for a in (vals, start, count))
    d_out = torch.full((len(start),), -1, dtype=torch.int64, device="cuda")
    assert L.llcomp_mi_device_range_sums(d_vals.data_ptr(), d_start.data_ptr(), d_count.data_ptr(), d_out.data_ptr(), len(start), cap,
                                         torch.cuda.current_stream().cuda_stream) == mi.OK
    want = [int(np.minimum(vals[s:s + n], cap).astype(np.int64).sum()) for s, n in zip(start, count)]
    assert d_out.cpu().tolist() == want
    # pool limit: with a budget of 0 nothing stays parked when a codec object goes
    before = int(L.llcomp_mi_pool_limit())
    try:
        mi.trim()
        mi.set_pool_limit(1 << 30)
        c1 = mi.Codec(2, 640, 360, 3, 64, 1, True)
        c1.close()
        parked = mi.pool_idle_bytes()
        assert 0 < parked <= 1 << 30
        mi.set_pool_limit(0)
        assert mi.pool_idle_bytes() == 0
        c2 = mi.Codec(2, 640, 360, 3, 64, 1, True)
        c2.close()
        assert mi.pool_idle_bytes() == 0
    finally:
        mi.set_pool_limit(before)


def test_payload_capacity_overflow_is_reported(mi):
    import torch

    w, h, c = 64, 64, 3
    img = np.random.default_rng(3).integers(0, 256, size=(h, w, c), dtype=np.uint8)
    codec = mi.Codec(1, w, h, c, 16, 16, True)
    st = torch.cuda.current_stream().cuda_stream
    d_px = torch.from_numpy(img).cuda()
    cap = 1000  # far too small for noise
    d_pay = torch.full((cap + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    d_len = torch.empty(codec.n_slices, dtype=torch.int32, device="cuda")
    d_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
    codec.encode(d_px.data_ptr(), d_pay.data_ptr(), cap, d_len.data_ptr(), d_tot.data_ptr(), d_st.data_ptr(), st)
    torch.cuda.synchronize()
    assert codec.status(int(d_st.item())) == mi.OUTPUT_OVERFLOW
    assert int(d_tot.item()) > cap
    assert bool((d_pay[cap:] == 0xAB).all()), "nothing may be written past the caller's capacity"
    codec.close()


# ---- full BASELINE sizes: golden hashes + size-independent properties -------------------------------------------
@pytest.mark.parametrize("gen", ["g2", "g3", "mid", "nat"])
def test_c3_4k_planar_tiles_golden_and_roundtrip(mi, orc, gen):
    v = [x for x in SLC if x["w"] == 3840 and x["gen"] == gen and x["planar"] and x["tile_w"] == 64][0]
    img = make_image(gen, 3840, 2160, 3)
    s = mi.compress_image(img, 3840, 2160, 3, format=mi.FORMAT_SLICED, tile_w=64, tile_h=64, planar=True)
    assert len(s) == v["container_len"] and fnv_hex(orc, s) == v["container_fnv1a64"]
    assert np.array_equal(mi.decompress_image(s).pixels, img)


def test_c2_1080p_noise_one_slice_per_row(mi, orc):
    v = [x for x in SLC if x["w"] == 1920 and x["gen"] == "g3" and x["tile_h"] == 1][0]
    img = make_image("g3", 1920, 1080, 3)
    s = mi.compress_image(img, 1920, 1080, 3, format=mi.FORMAT_SLICED, tile_w=1920, tile_h=1, planar=False)
    assert len(s) == v["container_len"] and fnv_hex(orc, s) == v["container_fnv1a64"]
    assert np.array_equal(mi.decompress_image(s).pixels, img)


def test_carries_through_ff_runs_match_oracle(mi, orc):
    """The encoder propagates carries eagerly into bytes it has already written (LDS ring, or its units in HBM behind
    the last 16-byte flush) where the reference resolves a held run lazily (llcomp.hpp:40-57).  Noise makes both happen
    thousands of times: the oracle counts its carries through 0xFF runs for this very input, the containers must agree
    byte for byte, and a run of 2+ bytes plus the number of events make a flush-boundary crossing certain."""
    img = make_image("g3", 1920, 1080, 3)
    # (64x64 and 32x32 tiles: the same through the 2-D encoder, whose hand-written sample is the snapshot form of the block)
    for tw, th, planar in ((480, 1, True), (1920, 1, False), (64, 64, True), (32, 32, False)):
        orc.carry_stats(reset=True)
        want = orc.compress_sliced(img, tile_w=tw, tile_h=th, planar=planar)
        runs, longest = orc.carry_stats()
        assert runs > 1000 and longest >= 2, (runs, longest)
        got = mi.compress_image(img, 1920, 1080, 3, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar)
        assert got == want
        assert np.array_equal(mi.decompress_image(got).pixels, img)


def test_c4_8k_roundtrip_and_band_merge_property(mi, orc):
    """8192x8192 RGB8: encode two halves as separate bands (what two GPUs would do), merge with the host
    concatenator, compare with the one-piece encode, decode, compare with the input."""
    w = h = 8192
    img = make_image("mid", w, h, 3)
    kw = dict(format=mi.FORMAT_SLICED, tile_w=128, tile_h=128, planar=True)
    whole = mi.compress_image(img, w, h, 3, **kw)
    top = mi.compress_image(img[: h // 2], w, h // 2, 3, **kw)
    bot = mi.compress_image(img[h // 2:], w, h // 2, 3, **kw)
    assert mi.merge_bands([top, bot]) == whole
    assert mi.split_band(whole, 0, 32) == top and mi.split_band(whole, 32, 64) == bot
    assert np.array_equal(mi.decompress_image(whole).pixels, img)


@pytest.mark.parametrize("lpw", ["64", "7", "1"])
def test_lanes_per_wave_does_not_change_bytes(mi, orc, lpw, set_hook):
    """The launcher spreads slices over wavefronts (1..64 slices per wave); the bytes must not depend on it."""
    set_hook("LLCOMP_MI_LPW", lpw)
    img = make_image("g3", 200, 150, 3)
    img[:, 100:] = make_image("mid", 100, 150, 3)
    for planar in (False, True):
        s = mi.compress_image(img, 200, 150, 3, format=mi.FORMAT_SLICED, tile_w=16, tile_h=16, planar=planar)
        assert s == orc.compress_sliced(img, 16, 16, planar)
        assert np.array_equal(mi.decompress_image(s).pixels, img)


# ---- the reference-compatible CLIs (tools/llcompc, tools/llcompd) --------------------------------------------------
def test_cli_roundtrip_matches_reference_behaviour(mi, orc, tmp_path):
    import os
    import struct
    import subprocess
    import zlib

    from conftest import ROOT

    exe_c, exe_d = os.path.join(ROOT, "tools", "llcompc"), os.path.join(ROOT, "tools", "llcompd")
    if not (os.path.exists(exe_c) and os.path.exists(exe_d)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    img = make_image("mid", 61, 47, 3)
    ppm = tmp_path / "a.ppm"
    ppm.write_bytes(b"P6\n# comment\n61 47\n255\n" + img.tobytes())
    assert subprocess.run([exe_c]).returncode == 1                      # usage
    big = tmp_path / "big.pgm"                                          # > 1 MPix in the reference's format: a note on stderr
    big.write_bytes(b"P5\n1100 1000\n255\n" + bytes(1100 * 1000))
    r = subprocess.run([exe_c, str(big)], capture_output=True, text=True)
    assert r.returncode == 0 and "single-stream format" in r.stderr
    r = subprocess.run([exe_c, str(big), "--legacy"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == "" and (tmp_path / "big.pgm.llcomp").read_bytes()[:6] == bytes([0x79, 1, 0x4C, 0x04, 0xE8, 0x03])
    assert subprocess.run([exe_c, str(tmp_path / "missing.ppm")]).returncode == 1
    assert subprocess.run([exe_c, str(ppm)]).returncode == 0
    stream = (tmp_path / "a.ppm.llcomp").read_bytes()                   # llcompc.cpp:34: <path> + ".llcomp"
    assert stream == orc.compress_image(img)                            # == reference stream
    assert subprocess.run([exe_d, str(tmp_path / "a.ppm.llcomp")]).returncode == 0
    png = (tmp_path / "a.ppm.llcomp.png").read_bytes()                  # llcompd.cpp:28: <path> + ".png"
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(png):
        n, typ = struct.unpack(">I4s", png[pos:pos + 8])
        body = png[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", png[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        if typ == b"IDAT":
            idat += body
        pos += 12 + n
    assert ihdr == (61, 47, 8, 2, 0, 0, 0)
    from test_cli_image_io import decode_png_py

    assert np.array_equal(decode_png_py(png)[0], img)  # zlib inflates it, any of the five row filters undone
    # the PNG llcompd wrote goes back in through llcompc's own PNG reader: same stream as from the PPM
    (tmp_path / "b.png").write_bytes(png)
    assert subprocess.run([exe_c, str(tmp_path / "b.png")]).returncode == 0
    assert (tmp_path / "b.png.llcomp").read_bytes() == stream
    # a PNG as an encoder would write it (dynamic Huffman blocks, Paeth rows), RGBA
    rgba = make_image("mid", 40, 33, 4)
    from test_cli_image_io import make_png

    (tmp_path / "c.png").write_bytes(make_png(rgba, 6, filters=(4, 1, 3), level=9))
    assert subprocess.run([exe_c, str(tmp_path / "c.png")]).returncode == 0
    assert (tmp_path / "c.png.llcomp").read_bytes() == orc.compress_image(rgba)
    # sliced container through the CLI, damaged input -> exit code 1 with the reference's message
    assert subprocess.run([exe_c, str(ppm), "--sliced", "16x1"]).returncode == 0
    assert (tmp_path / "a.ppm.llcomp").read_bytes() == orc.compress_sliced(img, 16, 1, True)
    # --sliced auto: one-row slices of the width the library suggests for one image per call
    h_, w_, c_ = img.shape
    assert subprocess.run([exe_c, str(ppm), "--sliced", "auto"]).returncode == 0
    assert (tmp_path / "a.ppm.llcomp").read_bytes() == orc.compress_sliced(img, mi.suggest_tile_w(1, w_, h_, c_, True), 1, True)
    # the LargeModel = false variant through the CLIs: the sliced container carries the flag, the reference format needs it said
    orc.set_small_model(True)
    try:
        assert subprocess.run([exe_c, str(ppm), "--small-model"]).returncode == 0
        assert (tmp_path / "a.ppm.llcomp").read_bytes() == orc.compress_image(img)
    finally:
        orc.set_small_model(False)
    assert subprocess.run([exe_d, str(tmp_path / "a.ppm.llcomp"), "--small-model"]).returncode == 0
    assert (tmp_path / "a.ppm.llcomp.png").read_bytes() == png
    bad = tmp_path / "bad.llcomp"
    bad.write_bytes(bytes([0x42]) + stream[1:])
    r = subprocess.run([exe_d, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid magic number" in r.stderr


def test_cli_config1_gradient_pgm_equals_reference_stream(mi, orc, tmp_path):
    """BASELINE config 1 proper, through the tool: the 256x256 single-channel G2 gradient as a PGM file -> tools/llcompc ->
    <file>.llcomp must be, byte for byte, what the REAL reference's compressImage gives for the same pixels (llcompc.cpp:25-41;
    kat_streams.json g2-256-256-1: 1277 bytes, hex recorded) -- and tools/llcompd takes it back to the same pixels (the
    reference's own decoder cannot: SURVEY D2, c < 3)."""
    import os
    import subprocess

    from conftest import ROOT

    exe_c, exe_d = os.path.join(ROOT, "tools", "llcompc"), os.path.join(ROOT, "tools", "llcompd")
    if not (os.path.exists(exe_c) and os.path.exists(exe_d)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    v = [x for x in KAT if (x["gen"], x["w"], x["h"], x["c"]) == ("g2", 256, 256, 1)][0]
    img = make_image("g2", 256, 256, 1)
    pgm = tmp_path / "c1.pgm"
    pgm.write_bytes(b"P5\n256 256\n255\n" + img.tobytes())
    assert subprocess.run([exe_c, str(pgm)]).returncode == 0
    stream = (tmp_path / "c1.pgm.llcomp").read_bytes()  # llcompc.cpp:34: <path> + ".llcomp"
    assert len(stream) == v["len"] == 1277 and fnv_hex(orc, stream) == v["fnv1a64"]
    if "hex" in v:
        assert stream.hex() == v["hex"]
    assert subprocess.run([exe_d, str(tmp_path / "c1.pgm.llcomp")]).returncode == 0
    from test_cli_image_io import decode_png_py

    assert np.array_equal(decode_png_py((tmp_path / "c1.pgm.llcomp.png").read_bytes())[0], img)


def test_one_row_slices_through_both_kernel_families(mi, orc, set_hook):
    """tile_h == 1 normally runs the register-resident kernels (+ the fused 16-bit-symbol stage A when planar);
    LLCOMP_MI_NOROWS=1 forces the same slicing through the general table-in-HBM kernels.  Same bytes either way."""
    img = make_image("g3", 300, 9, 3)
    img[:, 150:] = make_image("mid", 150, 9, 3)
    for planar in (False, True):
        want = orc.compress_sliced(img, 70, 1, planar)
        for norows in ("0", "1"):
            set_hook("LLCOMP_MI_NOROWS", norows)
            s = mi.compress_image(img, 300, 9, 3, format=mi.FORMAT_SLICED, tile_w=70, tile_h=1, planar=planar)
            assert s == want
            assert np.array_equal(mi.decompress_image(s).pixels, img)


def test_2d_encoder_snapshot_pass_or_state_tables_same_bytes(mi, orc, set_hook):
    """Slices of several rows are encoded through the state snapshot pass (context sort, walk, unpermute: snapshot_kernels.hip) --
    up to 4096 samples in one go, above that and up to 16384 (128x128 planes, 64x64 interleaved RGB: 4 and 3 chunks) chunk after
    chunk with the contexts' states carried through the slice's table in HBM (round 6); LLCOMP_MI_NOSNAP=1 keeps the table encoder
    (a read-modify-write of the table per sample), which slices above 16384 samples (256x256 planes) always take.  Same bytes either way: ragged tiles (slices with fewer chunks than their
    neighbours, a last chunk of a few samples), every channel count, interleaved and planar, both model sizes, a slice of exactly
    4096 samples, slices narrower than a lane group."""
    cases = [(200, 150, 3, 64, 64, True), (200, 150, 3, 32, 32, False), (130, 67, 1, 64, 64, True), (97, 41, 2, 50, 21, False),
             (300, 20, 4, 128, 8, True), (64, 64, 3, 64, 64, True), (37, 29, 4, 16, 16, False), (500, 9, 3, 480, 2, True),
             # slices above 4096 samples: the chunked snapshot pass against the table encoder (16384 = 4 chunks, 12288 = 3, 65536 = 16;
             # 65x64 = 4160: a second chunk of 64 samples; 37x37x3 = 4107: of 11; 100x90x2 interleaved: 18000 = 4 full chunks + 1616)
             (300, 280, 3, 128, 128, True), (200, 150, 3, 64, 64, False), (520, 300, 1, 256, 256, True), (150, 140, 3, 65, 64, True),
             (90, 80, 3, 37, 37, False), (230, 200, 2, 100, 90, False)]
    for i, (w, h, c, tw, th, planar) in enumerate(cases):
        img = make_image("g3", w, h, c)
        img[:, w // 2:] = make_image("nat" if i & 1 else "mid", w - w // 2, h, c)
        for small in (False, True):
            orc.set_small_model(small)
            try:
                want = orc.compress_sliced(img, tw, th, planar)
            finally:
                orc.set_small_model(False)
            # (images this small have so few slices that each would get a wavefront of its own and its table in LDS -- neither 2-D
            # encoder: the lane-group width is forced; the last round runs the geometry as the library picks it)
            # (OVERLAP: slices above 4096 samples -- the pass of chunk c + 1 beside the coding of chunk c on the device's shared second
            # stream (2, the default), on one of the codec's own (1), or behind it on the caller's (0))
            for nosnap, shift, noov in (("0", "6", None), ("1", "6", None), ("0", "3", None), ("0", None, None), ("0", "6", "0"), ("0", "6", "1")):
                if noov is not None and tw * th * (1 if planar else c) <= 4096:
                    continue
                set_hook("LLCOMP_MI_NOSNAP", nosnap)
                set_hook("LLCOMP_MI_OVERLAP", noov)
                set_hook("LLCOMP_MI_LANE_SHIFT", shift)  # (None: unset)
                s = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar, small_model=small)
                assert s == want, (w, h, c, tw, th, planar, small, nosnap, shift, noov)
                assert np.array_equal(mi.decompress_image(s, small_model=small).pixels, img)
    set_hook("LLCOMP_MI_OVERLAP", None)
    set_hook("LLCOMP_MI_LANE_SHIFT", "6")
    for nosnap in ("0", "1"):
        set_hook("LLCOMP_MI_NOSNAP", nosnap)
        k = mi.Codec(1, 200, 150, 3, 64, 64, True)
        assert k.family["snapshot"] == (nosnap == "0") and not k.family["lds_table"], k.family
        k.close()


def test_2d_decoder_bank_cache_or_plain_same_pixels(mi, orc, set_hook):
    """The 2-D decoder keeps a per-lane write-back cache of 32 state banks in LDS (slice_kernels.hip, CACHE) and gives it up in
    the middle of a slice when fewer than one access in eight hits; LLCOMP_MI_NOCACHE=1 fetches and writes every bank in HBM (the
    path before round 5).  Same pixels either way, and the containers equal the oracle's: contents that stay cached (noise,
    photo-like, flat), content that trips the bypass (dithered gradient: the slices are 40 rows high, the decision falls at rows
    8, 12, ...), mixtures inside one wavefront, every channel count, interleaved and planar, both model sizes, slices above
    4096 samples (the decoder is the same kernel there), batches of three frames on one codec object called twice (entries of the
    first call must not leak into the second: tables are per call)."""
    cases = [(640, 80, 3, 64, 40, True, "mid"), (640, 80, 3, 64, 40, True, "nat"), (640, 80, 1, 64, 40, True, "g3"), (520, 90, 3, 32, 45, False, "mid"),
             (300, 150, 4, 40, 25, False, "g2"), (400, 100, 2, 50, 50, False, "mid"), (700, 140, 3, 128, 70, True, "mid"), (512, 64, 3, 64, 64, True, "checker")]
    for i, (w, h, c, tw, th, planar, gen) in enumerate(cases):
        img = make_image(gen, w, h, c)
        if i & 1:
            img[:, w // 2:] = make_image("g3", w - w // 2, h, c)  # half of the lanes of a wavefront keep hitting, the others do not
        for small in ((False, True) if i < 4 else (False,)):
            orc.set_small_model(small)
            try:
                want = orc.compress_sliced(img, tw, th, planar)
            finally:
                orc.set_small_model(False)
            # (so few slices would get one wavefront each and their tables in LDS: the lane-group width is forced, 64 and 8 lanes)
            for nocache, shift in (("0", "6"), ("0", "3"), ("1", "6")):
                set_hook("LLCOMP_MI_NOCACHE", nocache)
                set_hook("LLCOMP_MI_LANE_SHIFT", shift)
                s = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar, small_model=small)
                assert s == want, (w, h, c, tw, th, planar, gen, small, nocache, shift)
                assert np.array_equal(mi.decompress_image(s, small_model=small).pixels, img), (w, h, c, tw, th, planar, gen, small, nocache, shift)
    set_hook("LLCOMP_MI_LANE_SHIFT", "6")
    for nocache in ("0", "1"):
        set_hook("LLCOMP_MI_NOCACHE", nocache)
        for gens in (["mid", "g3", "nat"], ["nat", "mid", "mid"]):
            assert _batch_roundtrip(mi, orc, 3, 260, 90, 3, 64, 45, True, gens, expect={"bank_cache": nocache == "0", "lds_table": False, "snapshot": True}) > 0


@pytest.mark.parametrize("tile", [(32, 32, True), (33, 31, True), (32, 33, True), (64, 32, True), (64, 33, True), (65, 63, True), (64, 64, True),
                                  (1365, 3, True), (16, 21, False), (26, 26, False), (37, 37, False), (4, 2, True),
                                  (64, 65, True), (128, 64, True), (64, 64, False), (91, 45, False), (128, 129, True)],
                         ids=lambda t: "%dx%d%s" % (t[0], t[1], "p" if t[2] else "i"))
def test_snapshot_pass_capacity_classes(mi, orc, tile, set_hook):
    """The snapshot pass sorts a slice's samples in one of three capacity classes (1024 / 2048 / 4096 keys) and moves its arrays
    in pieces of 8 / 16 samples: slices of exactly, one below and one above every boundary (planar: tile_w x tile_h samples;
    interleaved RGB: x 3), a batch of three frames so that the slice count is no multiple of a lane group, ragged last tiles.
    Above 4096 samples the pass runs in chunks of 4096 (one above: a second chunk of 64 or 11 samples; 8192, 12288, 12285); 128x129 =
    16512 samples is one above what the pass takes (four chunks): the table encoder."""
    tw, th, planar = tile
    w, h = min(2 * tw + 5, 1400), 2 * th + 3
    total = 0
    set_hook("LLCOMP_MI_LANE_SHIFT", "6")  # (so few slices would get one wavefront each and their tables in LDS: no snapshot pass)
    for small in (False, True):
        orc.set_small_model(small)
        try:
            total += _batch_roundtrip(mi, orc, 3, w, h, 3, tw, th, planar, ["g3", "nat", "mid"], small_model=small,
                                      expect={"snapshot": tw * th * (1 if planar else 3) <= 16384, "lds_table": False, "bank_cache": True})
        finally:
            orc.set_small_model(False)
    assert total > 0


def test_state_table_in_lds_or_hbm_same_bytes(mi, orc, set_hook):
    """With one slice per wavefront (a lone legacy stream, a few big tiles) the 63 KB state table of the slice lives in
    LDS; LLCOMP_MI_NOLDSTAB=1 keeps it in HBM like every multi-lane launch.  Same bytes either way, all channel counts."""
    for c in (1, 2, 3, 4):
        img = make_image("g3", 97, 41, c)
        img[:, 40:] = make_image("mid", 57, 41, c)
        legacy = orc.compress_image(img)
        tiled = orc.compress_sliced(img, 50, 21, False)
        for off in ("0", "1"):
            set_hook("LLCOMP_MI_NOLDSTAB", off)
            s = mi.compress_image(img, 97, 41, c)
            assert s == legacy
            assert np.array_equal(mi.decompress_image(s).pixels, img)
            t = mi.compress_image(img, 97, 41, c, format=mi.FORMAT_SLICED, tile_w=50, tile_h=21, planar=False)
            assert t == tiled
            assert np.array_equal(mi.decompress_image(t).pixels, img)


def test_host_calls_are_reentrant(mi, orc):
    """the host-buffer calls are re-entrant.  Four threads encode and decode different images at the
    same time (ctypes drops the GIL inside the library; idle codec objects are shared through a locked cache)."""
    import threading

    jobs = []
    for i in range(8):
        rng = np.random.default_rng(100 + i)
        w, h, c = int(rng.integers(40, 200)), int(rng.integers(8, 60)), int(rng.integers(1, 5))
        img = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
        tw = int(rng.integers(16, w + 1))
        jobs.append((img, w, h, c, tw, orc.compress_sliced(img, tw, 1, True), orc.compress_image(img)))
    errors = []

    def work(k):
        try:
            for rep in range(3):
                for img, w, h, c, tw, want_sliced, want_legacy in jobs[k::4]:
                    s = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=1, planar=True)
                    assert s == want_sliced
                    assert np.array_equal(mi.decompress_image(s).pixels, img)
                    assert mi.compress_image(img, w, h, c) == want_legacy
        except Exception as e:  # noqa: BLE001 -- reported below, in the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_decoder_rollback_and_checked_replay(mi, orc, set_hook):
    """The decoder's fast path never checks its input window; a sample that outruns the window is rolled back and
    replayed with per-step refills.  That almost never happens on real data, so LLCOMP_MI_FORCE_REPLAY=1 sends EVERY
    sample through rollback + replay: the pixels must not change.  Also a hostile-statistics image (long constant runs,
    then maximal spikes) through the normal path."""
    y, x, k = np.meshgrid(np.arange(24), np.arange(400), np.arange(3), indexing="ij")
    spikes = np.where((x % 97 == 96) & (k != 1), 255, np.where((x % 2 == 0) & (k == 0), 128, 0)).astype(np.uint8)
    for img in (make_image("g3", 160, 12, 3), make_image("mid", 160, 12, 3), spikes):
        h, w, c = img.shape
        for tw, th, planar in ((w, 1, True), (50, 1, False), (64, 8, True)):
            want = orc.compress_sliced(img, tw, th, planar)
            for force in ("0", "1"):
                set_hook("LLCOMP_MI_FORCE_REPLAY", force)
                assert mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar) == want
                assert np.array_equal(mi.decompress_image(want).pixels, img)


@pytest.mark.parametrize("seed", range(12))
def test_damaged_sliced_containers_never_crash(mi, orc, seed):
    """Valid header, damaged slice table and/or payload: the call must either decode (to whatever the bits say) or
    raise LlcompError -- never fault.  Truncation and over-long table entries must be reported."""
    rng = np.random.default_rng(4000 + seed)
    w, h, c = int(rng.integers(8, 90)), int(rng.integers(1, 30)), int(rng.integers(1, 5))
    tw, th = int(rng.integers(1, w + 1)), int(rng.integers(1, min(h, 9) + 1))
    planar = bool(rng.integers(0, 2))
    img = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
    good = bytearray(orc.compress_sliced(img, tw, th, planar))
    n = int.from_bytes(good[20:24], "little")
    kind = seed % 4
    bad = bytearray(good)
    if kind == 0:      # random payload, table intact
        bad[24 + 4 * n:] = rng.integers(0, 256, size=len(bad) - 24 - 4 * n, dtype=np.uint8).tobytes()
    elif kind == 1:    # one table entry claims far too much
        k = int(rng.integers(0, n))
        bad[24 + 4 * k:28 + 4 * k] = (0x7FFFFFF0).to_bytes(4, "little")
    elif kind == 2:    # payload cut short
        bad = bad[: 24 + 4 * n + max(0, (len(bad) - 24 - 4 * n) // 2)]
    else:              # random table
        bad[24:24 + 4 * n] = rng.integers(0, 2000, size=n, dtype=np.uint32).astype("<u4").tobytes()
    try:
        out = mi.decompress_image(bytes(bad))
        assert out.pixels.shape == (h, w, c)
        assert kind == 0 or kind == 3, "a truncated container must not decode silently"
    except mi.LlcompError as e:
        assert e.status in (mi.TRUNCATED, mi.BAD_EXPONENT, mi.BAD_ARGS)
    # and the good one still decodes afterwards (codec cache state is clean)
    assert np.array_equal(mi.decompress_image(bytes(good)).pixels, img)


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_batch_codec_fused_rows_all_channel_counts(mi, orc, c):
    """Planar 1-row slices go through the fused stage-A kernels and 16-bit symbols; tiles straddle lane-group
    boundaries differently for every channel count, widths are ragged, several frames per call."""
    _batch_roundtrip(mi, orc, 3, 250, 11, c, 70, 1, True, ["g3", "mid", "g1"])
    _batch_roundtrip(mi, orc, 2, 131, 40, c, 131, 1, True, ["mid", "checker"])
    _batch_roundtrip(mi, orc, 2, 97, 5, c, 16, 1, False, ["g3", "g2"])  # interleaved rows: register kernels, u32 symbols


@pytest.mark.parametrize("shift", ["0", "2", "5"])
def test_lane_group_width_does_not_change_bytes(mi, orc, shift, set_hook):
    """Lane groups narrower than 64 (few slices, or forced here) change every HBM layout but not a single byte."""
    set_hook("LLCOMP_MI_LANE_SHIFT", shift)
    img = make_image("g3", 190, 21, 3)
    img[:, 90:] = make_image("mid", 100, 21, 3)
    for tw, th, planar in ((45, 1, True), (45, 1, False), (32, 8, True), (190, 21, False)):
        s = mi.compress_image(img, 190, 21, 3, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar)
        assert s == orc.compress_sliced(img, tw, th, planar)
        assert np.array_equal(mi.decompress_image(s).pixels, img)


# ---- device-memory cache (devmem.hip): nothing goes back to the driver until trim() --------------------------------
def test_lane_churn_reuses_parked_blocks_and_trim_returns_them(mi, orc):
    """Shapes come and go (the host API keeps 4 idle lanes): the buffers of dropped lanes are parked and reused, not
    hipFree'd (a buffer taken right after a hipFree showed zeroed cache lines on this stack, see devmem.hip); trim()
    hands everything back.  Containers must equal the oracle's throughout."""
    import torch

    mi.trim()
    free0 = torch.cuda.mem_get_info(0)[0]
    rng = np.random.default_rng(77)
    for i in range(12):  # 12 distinct shapes: lanes are dropped from the 5th on
        w, h = 200 + 37 * i, 120 + 11 * i
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        s = mi.compress_image(img, w, h, 3, format=mi.FORMAT_SLICED, tile_w=64, tile_h=32, planar=bool(i & 1))
        assert s == orc.compress_sliced(img, 64, 32, bool(i & 1))
        assert np.array_equal(mi.decompress_image(s).pixels, img)
    held = free0 - torch.cuda.mem_get_info(0)[0]
    assert held > (8 << 20), "lanes and parked blocks should hold device memory here"
    mi.trim()
    assert torch.cuda.mem_get_info(0)[0] >= free0 - (64 << 20), "trim() must return lanes and parked blocks to the driver"
