"""Event counters of a codec object (llcomp_mi_codec_get_counters): proof that the rare and the adaptive branches of the kernels RAN.
Parity passes whether or not the 2-D decoder's bank cache hit, gave itself up, wrote victims back, whether the decoder's checked
replay or the encoder's carry into flushed units ever executed -- these tests pin the counts: against the oracle's context traces
(the cache's look-ups, misses and write-backs must be EXACTLY what a direct-mapped cache of 32 entries does on the slices' context
sequences, llcomp.hpp:424-436), against the oracle's own carry statistics, and against the forced-replay hook.  Also the feedback
that takes the cache away per launch from content whose wavefronts all give it up, and gives it back (csrc/codec.hip)."""
import numpy as np
import pytest

from conftest import make_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mi():
    import llcomp_amd

    assert llcomp_amd.device_count() >= 1
    return llcomp_amd


@pytest.fixture
def set_hook(mi, monkeypatch):
    def _set(name, value):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, value)
        mi.reload_tuning()

    yield _set
    monkeypatch.undo()
    mi.reload_tuning()


class Batch:
    """frames [F,h,w,c] on the GPU behind one device-resident codec object; encode() / decode() check status and losslessness"""

    def __init__(self, mi, imgs, tw, th, planar):
        import torch

        self.torch, self.imgs = torch, imgs
        F, h, w, c = imgs.shape
        self.codec = mi.Codec(F, w, h, c, tw, th, planar)
        self.st = torch.cuda.current_stream().cuda_stream
        self.d_px = torch.from_numpy(imgs).cuda()
        self.cap = min(self.codec.max_payload_bytes, 2 * imgs.size + 64 * self.codec.n_slices + 4096)
        self.d_pay = torch.empty(self.cap, dtype=torch.uint8, device="cuda")
        self.d_len = torch.empty(self.codec.n_slices, dtype=torch.int32, device="cuda")
        self.d_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
        self.d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.d_out = torch.zeros_like(self.d_px)
        self.total = None

    def load(self, imgs):
        self.d_px.copy_(self.torch.from_numpy(imgs))

    def encode(self):
        self.codec.encode(self.d_px.data_ptr(), self.d_pay.data_ptr(), self.cap, self.d_len.data_ptr(), self.d_tot.data_ptr(), self.d_st.data_ptr(), self.st)
        self.torch.cuda.synchronize()
        assert int(self.d_st.item()) == 0
        self.total = int(self.d_tot.item())

    def decode(self):
        self.d_out.zero_()
        self.codec.decode(self.d_pay.data_ptr(), self.total, self.d_len.data_ptr(), self.d_out.data_ptr(), self.d_st.data_ptr(), self.st)
        self.torch.cuda.synchronize()
        assert int(self.d_st.item()) == 0 and self.torch.equal(self.d_out, self.d_px), "round trip is not lossless"

    def close(self):
        self.codec.close()


def _cache_trace(orc, img, tw, th, entries=32):
    """what a per-slice direct-mapped cache of `entries` state banks (index = low context bits) does on the planar slices of `img`:
    (look-ups, misses, write-backs of valid victims) from the oracle's context sequence of every slice"""
    h, w, c = img.shape
    rct = orc.forward_rct(img)
    lookups = misses = wbs = 0
    for y0 in range(0, h, th):
        for x0 in range(0, w, tw):
            for ch in range(c):
                ctx, _ = orc.model_samples(rct[y0:y0 + th, x0:x0 + tw, ch:ch + 1])
                tags = [-1] * entries
                for v in ctx.reshape(-1).tolist():
                    e = v & (entries - 1)
                    lookups += 1
                    if tags[e] != v:
                        misses += 1
                        wbs += tags[e] >= 0
                        tags[e] = v
    return lookups, misses, wbs


def test_bank_cache_counts_equal_the_context_trace(mi, orc, set_hook):
    """Noise and photo-like content keep the cache: every sample is one look-up, and misses / write-backs are exactly those of a
    direct-mapped 32-entry cache on the oracle's context sequences.  (Slices are 32 rows high here and four frames make 384 slices;
    the lane-group width is forced to 64 so that they share wavefronts and their tables live in HBM.)"""
    set_hook("LLCOMP_MI_LANE_SHIFT", "6")
    for gen in ("g3", "nat"):
        imgs = np.stack([np.roll(make_image(gen, 256, 128, 3), 9 * i, axis=1) for i in range(4)])
        b = Batch(mi, imgs, 32, 32, True)
        assert b.codec.family["bank_cache"] and not b.codec.family["lds_table"], b.codec.family
        b.encode()
        b.decode()
        got = b.codec.counters()
        want = [sum(x) for x in zip(*(_cache_trace(orc, imgs[i], 32, 32) for i in range(4)))]
        assert got["dec_launches_cached"] == 1 and got["dec_launches_plain"] == 0
        assert got["dec_cached_waves"] == (b.codec.n_slices + 63) // 64 and got["dec_bypassed_waves"] == 0, got
        assert [got["cache_lookups"], got["cache_misses"], got["cache_writebacks"]] == want, (gen, got, want)
        hit = 1 - got["cache_misses"] / got["cache_lookups"]
        assert 0.15 < hit < 0.75, (gen, hit)  # (the cache is worth having on these: profiles/r05_bank_cache_ab.txt)
        # reset: everything back to zero, and the next call counts from there
        assert b.codec.counters(reset=True)["cache_lookups"] == want[0]
        assert not any(b.codec.counters().values())
        b.decode()
        assert b.codec.counters()["cache_lookups"] == want[0]
        b.close()


def test_dithered_content_gives_the_cache_up_and_the_codec_stops_asking_for_it(mi, orc, set_hook):
    """The dithered gradient's contexts do not come back inside 32 entries: its wavefronts give the cache up at row 8 (counter), and
    because ALL of them did, the codec's following decode calls run the plain kernel (no LDS held for a cache nobody uses) until the
    16th call probes again.  Noise frames loaded into the same codec object flip it back to the cache at that probe, dithered frames
    after that flip it away again: same pixels every time."""
    set_hook("LLCOMP_MI_LANE_SHIFT", "6")
    mid = np.stack([np.roll(make_image("mid", 512, 128, 3), 5 * i, axis=1) for i in range(2)])
    noise = np.stack([make_image("g3@%d" % (77 + i), 512, 128, 3) for i in range(2)])
    b = Batch(mi, mid, 64, 64, True)
    waves = (b.codec.n_slices + 63) // 64
    b.encode()
    b.decode()
    c = b.codec.counters()
    assert c["dec_cached_waves"] == waves and c["dec_bypassed_waves"] == waves, c
    assert mid.size // 8 <= c["cache_lookups"] <= mid.size // 4 and c["cache_misses"] * 4 > c["cache_lookups"] * 3, c  # gave up at row 8 (or 12)
    for _ in range(15):  # the run of plain launches
        b.decode()
    c = b.codec.counters()
    assert (c["dec_launches_cached"], c["dec_launches_plain"], c["dec_cached_waves"]) == (1, 15, waves), c
    b.decode()  # the probe: with the cache, gives it up again
    c = b.codec.counters()
    assert (c["dec_launches_cached"], c["dec_launches_plain"], c["dec_bypassed_waves"]) == (2, 15, 2 * waves), c
    # now noise through the same object: the plain run that the probe started goes on, then the probe finds hits and the cache stays
    b.load(noise)
    b.encode()
    for _ in range(15):
        b.decode()
    c = b.codec.counters()
    assert (c["dec_launches_cached"], c["dec_launches_plain"]) == (2, 30), c
    for _ in range(3):
        b.decode()
    c = b.codec.counters()
    assert (c["dec_launches_cached"], c["dec_launches_plain"], c["dec_bypassed_waves"]) == (5, 30, 2 * waves), c
    # ... and back: the first dithered call runs with the cache, gives up, the next ones are plain
    b.load(mid)
    b.encode()
    b.decode()
    b.decode()
    c = b.codec.counters()
    assert (c["dec_launches_cached"], c["dec_launches_plain"], c["dec_bypassed_waves"]) == (6, 31, 3 * waves), c
    b.close()


def test_mixed_wavefronts_keep_the_cache(mi, orc, set_hook):
    """a launch where only SOME wavefronts give the cache up keeps it: the plain kernel is for content that makes (nearly) all give up"""
    set_hook("LLCOMP_MI_LANE_SHIFT", "6")
    img = np.concatenate([make_image("mid", 1024, 128, 3), make_image("g3", 1024, 384, 3)], axis=0)  # 96 + 288 slices of 64x64: 1.5 + 4.5 wavefronts
    b = Batch(mi, img[None], 64, 64, True)
    b.encode()
    for _ in range(3):
        b.decode()
    c = b.codec.counters()
    assert c["dec_launches_cached"] == 3 and c["dec_launches_plain"] == 0 and 0 < c["dec_bypassed_waves"] < c["dec_cached_waves"], c
    b.close()


def test_checked_replays_are_counted(mi, orc, set_hook):
    """LLCOMP_MI_FORCE_REPLAY=1 sends every sample through rollback + checked replay: the counter then equals the sample count, in the
    one-row kernel and in the 2-D kernel; without the hook the fast path takes (nearly) everything, but noise at 1.3 bytes per sample
    does outrun the window now and then -- the replay is live code on ordinary input, not only under the hook"""
    img = make_image("g3", 480, 96, 3)
    for tw, th in ((480, 1), (32, 32)):
        set_hook("LLCOMP_MI_LANE_SHIFT", "6")
        set_hook("LLCOMP_MI_FORCE_REPLAY", "1")
        b = Batch(mi, img[None], tw, th, True)
        b.encode()
        b.decode()
        assert b.codec.counters()["dec_replays"] == img.size, (tw, th, b.codec.counters())
        b.close()
        set_hook("LLCOMP_MI_FORCE_REPLAY", "0")
        b = Batch(mi, img[None], tw, th, True)
        b.encode()
        b.decode()
        n = b.codec.counters()["dec_replays"]
        assert n < img.size // 20, (tw, th, n)
        b.close()
    # ... and it is live code on ordinary input, not only under the hook: on noise the fast path takes everything (1080p: not one
    # replay), but long constant runs saturate the models and a maximal spike behind them costs 6-11 bytes in one sample -- more than
    # the four the window is guaranteed to hold
    set_hook("LLCOMP_MI_LANE_SHIFT", None)
    y, x, k = np.meshgrid(np.arange(48), np.arange(1600), np.arange(3), indexing="ij")
    spikes = np.ascontiguousarray(np.where((x % 97 == 96) & (k != 1), 255, np.where((x % 2 == 0) & (k == 0), 128, 0)).astype(np.uint8))
    b = Batch(mi, spikes[None], 400, 1, True)
    b.encode()
    b.decode()
    n = b.codec.counters()["dec_replays"]
    assert 0 < n < spikes.size // 50, n
    b.close()


def test_carries_into_flushed_units_are_counted(mi, orc):
    """Noise makes the encoder's carry into a held 0xFF happen thousands of times (the oracle counts the reference's outstanding runs,
    llcomp.hpp:40-57, for this very input); some of those carries must go on into bytes that have already left for HBM -- the rare
    path behind the hand-written block -- in the one-row kernel and in the 2-D snapshot encoder"""
    img = make_image("g3", 1920, 1080, 3)
    for tw, th in ((480, 1), (64, 64)):
        orc.carry_stats(reset=True)
        orc.compress_sliced(img, tile_w=tw, tile_h=th, planar=True)
        runs, _ = orc.carry_stats()
        b = Batch(mi, img[None], tw, th, True)
        b.encode()
        n = b.codec.counters()["enc_carry_backs"]
        assert 0 < n <= runs, (tw, th, n, runs)
        b.decode()
        b.close()


def test_generation_wraps_are_counted(mi, orc, set_hook):
    set_hook("LLCOMP_MI_LANE_SHIFT", "6")
    b = Batch(mi, make_image("nat", 128, 64, 1)[None], 16, 16, True)
    b.encode()
    for _ in range(300):
        b.codec.decode(b.d_pay.data_ptr(), b.total, b.d_len.data_ptr(), b.d_out.data_ptr(), b.d_st.data_ptr(), b.st)
    b.decode()
    assert b.codec.counters()["generation_wraps"] == 1
    b.close()


def test_prepare_allocates_ahead_of_the_first_call(mi, orc, set_hook):
    """llcomp_mi_codec_prepare: the state tables (decode) and the snapshot arrays (encode) exist before the first call asks for them;
    idempotent; same bytes and pixels as without it"""
    set_hook("LLCOMP_MI_LANE_SHIFT", "6")
    img = make_image("nat", 256, 128, 3)
    b = Batch(mi, img[None], 32, 32, True)
    before = mi.pool_idle_bytes()
    b.codec.prepare()
    b.codec.prepare(encode=True, decode=False)
    assert mi.pool_idle_bytes() <= before
    b.encode()
    b.decode()
    want = orc.compress_sliced(img, 32, 32, True)
    n = b.codec.n_slices
    assert b.d_len.cpu().numpy().astype("<u4").tobytes() == want[24:24 + 4 * n] and b.d_pay[: b.total].cpu().numpy().tobytes() == want[24 + 4 * n:]
    L = mi._lib.load()
    assert L.llcomp_mi_codec_prepare(b.codec._h, 4) == mi.BAD_ARGS and L.llcomp_mi_codec_prepare(None, 1) == mi.BAD_ARGS
    b.close()
