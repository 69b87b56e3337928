#!/usr/bin/env python3
"""What the sharded code path costs beside the coding itself, on ONE GPU: the per-rank share of BASELINE config 4 at N ranks
(24 images, every rank codes 8192/N rows of each) through ShardedCodec with the payload collective forced (RCCL, world 1), against
the same pixels through the plain device-resident codec (bench.measure).  The difference is host logic + collectives + the
concatenator: what strong scaling has to amortise.     python tools/c4_overhead.py [N ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LLCOMP_MI_FORCE_EXCHANGE"] = "1"
import socket

import torch
import torch.distributed as dist

import bench

sk = socket.socket()
sk.bind(("127.0.0.1", 0))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
sk.close()
torch.cuda.set_device(0)
fd = os.dup(1)
os.dup2(2, 1)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.all_reduce(torch.zeros(1, device="cuda"))
torch.cuda.synchronize()
os.dup2(fd, 1)
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    h = 8192 // n
    frames = bench.make_frames("g3", 24, 0, w=8192, h=h, c=3, distinct=2)
    m = bench.measure(frames, 512, 1, True, 3, 6, 2, 0)
    plain_ms = m["dt"] / m["steps"] * 1e3
    del frames
    for order in ("deep", "one-ahead"):  # bench.py's step order (all parts' coding queued first) and round 2's
        os.environ["LLCOMP_BENCH_C4_ORDER"] = order
        dt, pay = bench.c4_run(24, 8192, 512, 1, 6, 3, 0, 1, 0, check_one_piece=False, height=h)
        sharded_ms = dt / 6 * 1e3
        print(json.dumps({"ranks_modelled": n, "rows_per_rank": h, "order": order, "sharded_path_ms_per_step": round(sharded_ms, 2), "plain_codec_ms_per_step": round(plain_ms, 2),
                          "overhead_ms": round(sharded_ms - plain_ms, 2), "efficiency": round(plain_ms / sharded_ms, 3), **bench.c4_run.last_detail}), flush=True)
dist.destroy_process_group()
