"""N > 1 path on CPU: two gloo ranks shard one image by tile rows, gather the per-rank containers to rank 0 and
stitch them with the product's host concatenator; the result must equal the one-piece container.  The encoder /
decoder plugged in here is the oracle (there is no GPU in this test) -- what is under test is the distributed
plumbing (band split, variable-length gather/scatter) and llcomp_mi_merge_bands / llcomp_mi_split_band."""
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


def _worker(rank, world, port, shape, tile, planar, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist

    import orc as orc_mod
    from llcomp_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = orc_mod.Orc()
        w, h, c = shape
        tw, th = tile
        img = orc_mod.gen_mid(w, h, c)
        y0, y1 = sharding.band_rows(h, th, world)[rank]

        def enc(b, bw, bh, bc, tile_w, tile_h, pl):
            return orc.compress_sliced(np.ascontiguousarray(b).reshape(bh, bw, bc), tile_w, tile_h, pl)

        def dec(data):
            rc, px = orc.decompress(data)
            assert rc == 0
            return px

        whole = sharding.encode_image_sharded(img[y0:y1], w, y1 - y0, c, tile_w=tw, tile_h=th, planar=planar, encode_fn=enc)
        if rank == 0:
            assert whole == orc.compress_sliced(img, tw, th, planar), "stitched container differs from one-piece container"
        px = sharding.decode_image_sharded(whole, decode_fn=dec)
        if rank == 0:
            assert np.array_equal(px, img)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,tile,planar", [(2, (70, 50, 3), (32, 8), True), (2, (33, 9, 1), (16, 4), False), (3, (40, 20, 4), (40, 1), True), (2, (20, 5, 3), (8, 8), True)])
def test_two_rank_shard_gather_stitch(world, shape, tile, planar):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, tile, planar, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(r, "ok") for r in range(world)], res
    assert all(p.exitcode == 0 for p in procs)


def test_band_rows_cover_image():
    from llcomp_amd import sharding

    for h, th, world in [(2160, 64, 8), (8192, 128, 8), (5, 8, 2), (10, 1, 4), (7, 3, 3), (1, 1, 8)]:
        bands = sharding.band_rows(h, th, world)
        assert bands[0][0] == 0 and bands[-1][1] == h or any(b[1] == h for b in bands)
        flat = [y for y0, y1 in bands for y in range(y0, y1)]
        assert flat == list(range(h))
        for (y0, y1) in bands[:-1]:
            assert y0 % min(th, h) == 0
