"""N > 1 path on CPU: gloo ranks shard images by interleaved chunks of tile rows (llcomp_amd/sharding.py), exchange the
slice tables (all_gather / broadcast) and one payload message per rank, and the gathering rank's concatenator interleaves
the pieces into image order.  Every container must equal the one-piece container byte for byte, and decode must give
every rank its rows back.  The local coder plugged in here is the oracle (no GPU in this test; the HIP coder under the
same code runs in tests/test_gpu_sharding.py) -- what is under test is the distributed logic: chunk plan, slice
permutation, segment tables, message sizes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def oracle_band_factory(orc):
    import torch

    class OracleBand:
        """CPU stand-in for the device codec object: same contract (packed payload + slice lengths, frame-major)."""

        def __init__(self, images, w, h, c, tile_w, tile_h, planar, device):
            self.geo = (images, w, h, c, tile_w, min(tile_h, h), planar)
            self.n_slices = images * orc.slice_count(w, h, c, tile_w, min(tile_h, h), planar)

        def encode(self, px):
            images, w, h, c, tw, th, planar = self.geo
            pays, lens = [], []
            for b in range(images):
                s = orc.compress_sliced(px[b].numpy(), tw, th, planar)
                n = int.from_bytes(s[20:24], "little")
                lens.append(np.frombuffer(s[24:24 + 4 * n], dtype="<u4"))
                pays.append(s[24 + 4 * n:])
            payload = torch.frombuffer(bytearray(b"".join(pays) + bytes(16)), dtype=torch.uint8)
            lens = torch.from_numpy(np.concatenate(lens).astype(np.int32))
            return payload, lens, torch.tensor([int(lens.sum())]), torch.zeros(1, dtype=torch.int32)

        def decode(self, payload, payload_bytes, lens, out):
            images, w, h, c, tw, th, planar = self.geo
            per = self.n_slices // images
            lens = lens.numpy().astype(np.int64)
            data = payload.numpy().tobytes()
            pos = 0
            for b in range(images):
                ln = lens[b * per:(b + 1) * per]
                head = bytes([0x9C, 1, c, 1 if planar else 0]) + b"".join(int(v).to_bytes(4, "little") for v in (w, h, tw, th, per))
                body = data[pos:pos + int(ln.sum())]
                pos += int(ln.sum())
                rc, px = orc.decompress(head + ln.astype("<u4").tobytes() + body)
                assert rc == 0
                out[b] = torch.from_numpy(px.copy())
            return torch.zeros(1, dtype=torch.int32)

        def check(self, status):
            assert int(status.item()) == 0

    return OracleBand


def _worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist

    import orc as orc_mod
    from llcomp_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = orc_mod.Orc()
        (w, h, c), (tw, th), planar, images, cpr, root = case
        full = np.stack([np.roll(orc_mod.gen_mid(w, h, c) if b % 2 == 0 else orc_mod.gen_g3(w, h, c, seed=5 + b), 3 * b, axis=1) for b in range(images)])
        sc = sharding.ShardedCodec(w, h, c, tw, th, planar, images=images, chunks_per_rank=cpr, root=root, device=torch.device("cpu"),
                                   band_factory=oracle_band_factory(orc))
        band = sc.take_local(full)
        rows = [y for y0, y1 in sc.rows for y in range(y0, y1)]
        assert band.shape == (images, len(rows), w, c)
        conts = sc.encode(band)
        want_mine = [b for b in range(images) if (b % world if root is None else root) == rank]
        assert sorted(conts) == want_mine
        for b in want_mine:
            assert bytes(conts[b].numpy()) == orc.compress_sliced(full[b], tw, th, planar), f"image {b}: sharded container differs from the one-piece container"
        out = sc.decode(conts)
        assert np.array_equal(out.numpy(), full[sc.frame_images][:, rows]), "decoded rows differ"
        px = sc.gather_pixels(out)
        if rank == 0:
            assert np.array_equal(px.numpy(), full)
        # the same through the chunked exchange (messages cut into rounds of at most MAX_MESSAGE bytes per peer)
        sc3 = sharding.ShardedCodec(w, h, c, tw, th, planar, images=images, chunks_per_rank=cpr, root=root, device=torch.device("cpu"),
                                    band_factory=oracle_band_factory(orc))
        sc3.MAX_MESSAGE = 97
        c3 = sc3.encode(band)
        for b in want_mine:
            assert bytes(c3[b].numpy()) == bytes(conts[b].numpy())
        assert np.array_equal(sc3.decode(c3).numpy(), full[sc3.frame_images][:, rows])
        # two part batches in bench.py's software-pipelined order (begin / finish halves interleaved): same bytes, same rows
        sc2 = sharding.ShardedCodec(w, h, c, tw, th, planar, images=images, chunks_per_rank=cpr, root=root, device=torch.device("cpu"),
                                    band_factory=oracle_band_factory(orc))
        full2 = np.ascontiguousarray(full[:, ::-1])  # the second part: the images upside down
        band2 = sc2.take_local(full2)
        sc.encode_begin(band)
        sc2.encode_begin(band2)
        c1 = sc.encode_finish()
        sc.decode_begin(c1)
        c2 = sc2.encode_finish()
        sc2.decode_begin(c2)
        o1 = sc.decode_finish()
        o2 = sc2.decode_finish()
        for b in want_mine:
            assert bytes(c1[b].numpy()) == bytes(conts[b].numpy())
            assert bytes(c2[b].numpy()) == orc.compress_sliced(full2[b], tw, th, planar)
        assert np.array_equal(o1.numpy(), full[sc.frame_images][:, rows]) and np.array_equal(o2.numpy(), full2[sc2.frame_images][:, rows])
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


CASES = [  # shape, tile, planar, images, chunks per rank, root (None = containers spread round-robin over the ranks)
    (2, ((70, 50, 3), (32, 8), True, 1, 4, 0)),       # ragged last tile row, planar, everything gathered on rank 0
    (2, ((33, 9, 1), (16, 4), False, 2, 1, None)),    # contiguous bands (one chunk per rank), two images, one per rank
    (3, ((40, 20, 4), (40, 1), True, 2, 2, None)),    # one-row slices, three ranks, rank 2 gathers nothing
    (2, ((20, 5, 3), (8, 8), True, 1, 4, 1)),         # a single tile row: rank 1 has nothing to code but gathers the image
    (3, ((64, 37, 3), (16, 5), False, 5, 4, None)),   # 8 tile rows over 3 ranks x 4 chunks: uneven chunks, ragged tail, 5 images over 3 roots
    (3, ((64, 37, 3), (16, 5), True, 3, 4, 2)),       # funnel to a rank that is not 0
]


@pytest.mark.parametrize("world,case", CASES)
def test_sharded_containers_equal_one_piece(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(r, "ok") for r in range(world)], res
    assert all(p.exitcode == 0 for p in procs)


# ---- world 8: the size the 8-GPU node runs, rehearsed on the CPU ----------------------------------------------------------
def _worker_world8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist

    import orc as orc_mod
    from llcomp_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = orc_mod.Orc()
        # BASELINE config 4's plan scaled down: one-row slices narrower than the image (bench.py: 512 x 1 of 8192), planar, four
        # chunks of tile rows per rank; 67 rows do not divide by the 32 chunks (sizes 2 and 3).  Image counts below, at and above
        # the world size, none of them but 8 a multiple of it; containers spread round-robin (root None) or funnelled to rank 5.
        w, h, c, tw, th, cpr = 96, 67, 3, 40, 1, 4
        done = []
        for images in (1, 3, 8, 11):
            full = np.stack([np.roll(orc_mod.gen_mid(w, h, c) if b % 2 == 0 else orc_mod.gen_g3(w, h, c, seed=5 + b), 3 * b, axis=1) for b in range(images)])
            want = [orc.compress_sliced(full[b], tw, th, True) for b in range(images)] if rank in (0, 5) else None
            for root in (None, 5):
                sc = sharding.ShardedCodec(w, h, c, tw, th, True, images=images, chunks_per_rank=cpr, root=root, device=torch.device("cpu"),
                                           band_factory=oracle_band_factory(orc))
                sc.MAX_MESSAGE = 4096 if images == 8 else 1 << 30  # (one count also through the rounds of the chunked exchange)
                band = sc.take_local(full)
                rows = [y for y0, y1 in sc.rows for y in range(y0, y1)]
                assert band.shape == (images, len(rows), w, c) and len(sc.rows) == cpr
                conts = sc.encode(band)
                mine = [b for b in range(images) if (b % world if root is None else root) == rank]
                assert sorted(conts) == mine, (images, root, sorted(conts))
                for b in mine:
                    one_piece = want[b] if want is not None else orc.compress_sliced(full[b], tw, th, True)
                    assert bytes(conts[b].numpy()) == one_piece, f"images {images} root {root}: container {b} differs from the one-piece container"
                out = sc.decode(conts)
                assert np.array_equal(out.numpy(), full[sc.frame_images][:, rows]), (images, root)
                px = sc.gather_pixels(out)
                if rank == 0:
                    assert np.array_equal(px.numpy(), full)
                done.append((images, root, len(mine)))
        q.put((rank, "ok", done))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc(), None))
        raise
    finally:
        dist.destroy_process_group()


def test_world8_plan_shapes_of_config4():
    """Largest world the N > 1 path is meant for (the 8-GPU node), on the CPU with the oracle as local coder: per_root / perm index
    arithmetic with image counts that do not divide by the world, both gathering modes, the exchange in rounds."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_world8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(400)
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert [(r, m) for r, m, _ in res] == [(r, "ok") for r in range(world)], [m for _, m, _ in res if m != "ok"][:1]
    assert all(p.exitcode == 0 for p in procs)
    # every container was gathered exactly once: 1 + 3 + 8 + 11 images, twice (two gathering modes)
    assert sum(n for _, _, d in res for _, _, n in d) == 2 * (1 + 3 + 8 + 11)
    assert [n for _, _, n in res[5][2] if True][1::2] == [1, 3, 8, 11]  # root 5 holds all of them in the funnel mode


def test_chunk_plan_covers_image():
    from llcomp_amd import sharding

    for h, th, world, cpr in [(2160, 64, 8, 4), (8192, 128, 8, 4), (8192, 1, 8, 4), (5, 8, 2, 4), (10, 1, 4, 1), (7, 3, 3, 2), (1, 1, 8, 4)]:
        chunks = sharding.plan_chunks(h, th, world, cpr)
        nty = (h + min(th, h) - 1) // min(th, h)
        assert chunks[0][0] == 0 and chunks[-1][1] == nty
        assert all(a[1] == b[0] for a, b in zip(chunks, chunks[1:])) and all(t1 > t0 for t0, t1, _ in chunks)
        assert [o for _, _, o in chunks] == [i % world for i in range(len(chunks))]
        sizes = [t1 - t0 for t0, t1, _ in chunks]
        assert max(sizes) - min(sizes) <= 1
        flat = sorted(y for r in range(world) for y0, y1 in sharding.local_rows(h, th, world, r, cpr) for y in range(y0, y1))
        assert flat == list(range(h))


# ---- failure agreement and untrusted tables (a lone raise before a collective hangs the others) -------------
def _worker_faults(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import datetime

    import torch
    import torch.distributed as dist

    import orc as orc_mod
    from llcomp_amd import LlcompError, sharding

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        orc = orc_mod.Orc()
        w, h, c, tw, th = 48, 24, 3, 16, 4
        Band = oracle_band_factory(orc)

        class FailingBand(Band):  # rank 1's coder reports an overflow
            def encode(self, px):
                payload, lens, total, status = super().encode(px)
                if rank == 1:
                    status = torch.tensor([1], dtype=torch.int32)
                return payload, lens, total, status

        full = np.stack([orc_mod.gen_mid(w, h, c), orc_mod.gen_g3(w, h, c, seed=9)])
        mk = lambda factory, **kw: sharding.ShardedCodec(w, h, c, tw, th, True, images=2, device=torch.device("cpu"), band_factory=factory, **kw)  # noqa: E731
        # 1. one rank's local encode fails: EVERY rank raises, before the payload collective
        sc = mk(FailingBand)
        try:
            sc.encode(sc.take_local(full))
            raise AssertionError("no error raised")
        except LlcompError as e:
            assert "rank 1" in str(e) and sc.exchanges == 0
        dist.barrier()  # nobody is stuck in a collective
        # 2. a container that does not belong to this geometry on ONE rank: every rank raises ValueError together
        sc = mk(Band)
        conts = sc.encode(sc.take_local(full))
        bad = dict(conts)
        if rank == 1:
            for b in bad:
                bad[b] = bad[b].clone()
                bad[b][8] ^= 1  # width
        try:
            sc.decode(bad)
            raise AssertionError("no error raised")
        except ValueError as e:
            assert "[1]" in str(e)
        dist.barrier()
        # 3. a damaged slice table (a length of 0xFFFFFFF0 = negative as int32, and one far beyond the container): sizes stay
        #    bounded by the slice capacity, nothing is copied from outside the container, the decoder gets zero-filled bytes
        dmg = dict(conts)
        for b in dmg:
            t = dmg[b].clone()
            t[24:28] = torch.tensor([0xF0, 0xFF, 0xFF, 0xFF], dtype=torch.uint8)
            t[32:36] = torch.tensor([0x00, 0x00, 0x00, 0x7F], dtype=torch.uint8)
            dmg[b] = t
        lens_seen = []
        orig = sc.band.decode

        def spy(payload, payload_bytes, lens, out):
            lens_seen.append(lens.clone())
            assert int(lens.min()) >= 0 and int(lens.max()) <= sc.slice_cap and payload_bytes <= sc.slice_cap * lens.numel()
            out.zero_()
            return torch.zeros(1, dtype=torch.int32)

        sc.band.decode = spy
        sc.decode(dmg)
        sc.band.decode = orig
        assert lens_seen and int(torch.stack([x.max() for x in lens_seen]).max()) <= sc.slice_cap
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_failures_are_agreed_before_collectives_and_tables_are_clamped():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_faults, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, "ok"), (1, "ok")], res


def _worker_forced(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist

    import orc as orc_mod
    from llcomp_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = orc_mod.Orc()
        w, h, c, tw, th = 50, 21, 3, 16, 4
        full = np.stack([orc_mod.gen_mid(w, h, c), orc_mod.gen_g3(w, h, c, seed=3)])
        sc = sharding.ShardedCodec(w, h, c, tw, th, True, images=2, device=torch.device("cpu"), band_factory=oracle_band_factory(orc), force_exchange=True)
        conts = sc.encode(sc.take_local(full))
        assert sc.exchanges == 1  # the all_to_all ran although it is the identity at world 1
        for b in range(2):
            assert bytes(conts[b].numpy()) == orc.compress_sliced(full[b], tw, th, True)
        out = sc.decode(conts)
        assert sc.exchanges == 2 and np.array_equal(out.numpy(), full[sc.frame_images])
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_world1_forced_exchange_runs_the_payload_collective():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_forced, args=(0, 1, _free_port(), q))
    p.start()
    p.join(120)
    assert q.get(timeout=5) == (0, "ok")
