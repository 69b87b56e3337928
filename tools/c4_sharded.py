#!/usr/bin/env python3
"""BASELINE config 4: ONE 8192x8192 RGB8 image, tiles sharded over the ranks of a torch.distributed job (one process per
GPU, RCCL over xGMI), per-rank containers gathered to rank 0 and stitched by the host concatenator; decode mirrors it.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/c4_sharded.py [--content mid]

Prints one JSON line on rank 0: end-to-end MPix/s through the HOST-buffer API (PCIe and the gather are inside), so this
is a functional / scaling check of the sharded path, not bench.py's device-resident metric."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--tile-w", type=int, default=512)
    ap.add_argument("--tile-h", type=int, default=1)
    ap.add_argument("--content", default="mid", choices=["mid", "g2", "g3"])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import llcomp_amd as mi
    from llcomp_amd import synth as orc_mod
    from llcomp_amd import sharding

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    w = h = args.size
    y0, y1 = sharding.band_rows(h, args.tile_h, world)[rank]
    # every rank builds only ITS band (the generators are position based, so bands of a 2048-wide pattern are tiled)
    base = orc_mod.GENERATORS[args.content](min(w, 2048), min(h, 2048), 3)
    band = np.ascontiguousarray(np.tile(base, (h // base.shape[0] + 1, w // base.shape[1] + 1, 1))[y0:y1, :w])
    band += (np.arange(y0, y1, dtype=np.uint32)[:, None, None] >> 6).astype(np.uint8)  # make the bands differ
    times = []
    for _ in range(args.reps):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        whole = sharding.encode_image_sharded(band, w, y1 - y0, 3, tile_w=args.tile_w, tile_h=args.tile_h, planar=True)
        px = sharding.decode_image_sharded(whole)
        dist.barrier()
        times.append(time.perf_counter() - t0)
    ok = True
    if rank == 0:
        ok = bool(np.array_equal(px[y0:y1], band))  # rank 0 checks its own rows of the reassembled image
    rows = sharding.gather_bytes(band.tobytes(), dst=0)
    if rank == 0:
        ok = ok and np.array_equal(px, np.frombuffer(b"".join(rows), np.uint8).reshape(h, w, 3))
        print(json.dumps({"config": f"C4 {w}x{h} RGB8 {args.content}, planar {args.tile_w}x{args.tile_h} slices, {world} GPU(s), gather to rank 0",
                          "bit_exact": ok, "container_bytes": len(whole), "ratio": round(w * h * 3 / len(whole), 4),
                          "end_to_end_MPix_s": round(w * h / min(times) / 1e6, 1), "seconds": [round(t, 4) for t in times]}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
