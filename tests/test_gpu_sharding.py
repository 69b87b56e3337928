"""The multi-GPU path (llcomp_amd/sharding.py) with the PRODUCT's local coder -- the device-resident HIP codec object --
under torch.distributed:
  * world size 1 on the nccl (= RCCL) backend: every collective of the path runs on device tensors -- including, with the
    force_exchange hook, the variable-size all_to_all of the payloads that a single rank would otherwise skip;
  * world size 2 on the nccl backend, one rank per GPU (skipped on a one-GPU box): the path as bench.py --gpus 2 runs it;
  * world size 2 on the gloo backend, both ranks on the one GPU of the test box: the full exchange logic (slice-table
    all_gather / broadcast, one payload message per rank, device concatenator) around real HIP encodes / decodes.
Containers must equal the one-piece container of the host call byte for byte (BASELINE config 4 sizes included)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _golden_container(gen, w, h, c, tw, th, planar):
    import json

    for v in json.load(open(os.path.join(ROOT, "tests", "golden", "c4_bench_slicing.json")))["vectors"]:
        if (v["gen"], v["w"], v["h"], v["c"], v["tile_w"], v["tile_h"], v["planar"]) == (gen, w, h, c, tw, th, planar):
            return v
    return None


def _oracle_container(img, tw, th, planar):
    """the checker's container of a small image (the big cases fall back to the one-piece HIP container, which the parity
    tests pin elsewhere)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc as orc_mod

    return orc_mod.Orc().compress_sliced(img, tw, th, planar)


def _worker(rank, world, backend, port, cases, q, own_gpu=False, force=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import llcomp_amd as mi
    from llcomp_amd import sharding, synth

    ordinal = rank if own_gpu else 0
    torch.cuda.set_device(ordinal)
    dev = torch.device("cuda", ordinal)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for (gen, w, h, c), (tw, th), planar, images, cpr, *rest in cases:
            root = rest[0] if rest else None
            full = np.stack([np.roll(synth.GENERATORS[gen](w, h, c), 5 * b, axis=1) for b in range(images)])
            sc = sharding.ShardedCodec(w, h, c, tw, th, planar, images=images, chunks_per_rank=cpr, root=root, device=dev, force_exchange=force)
            assert sc.band is None or type(sc.band).__name__ == "_HipBand"
            band = sc.take_local(full)
            conts = sc.encode(band)
            assert sorted(conts) == [b for b in range(images) if (b % world if root is None else root) == rank]   # spread round-robin, or funnelled
            for b in conts:
                assert conts[b].is_cuda
                gold = _golden_container(gen, w, h, c, tw, th, planar) if b == 0 else None
                if gold is not None:
                    # BASELINE config 4 at the benchmarked slicing: against the container assembled from the REAL reference's
                    # per-slice streams (tests/golden/c4_bench_slicing.json, oracle/gen_golden.py c4) -- not against another HIP result
                    host = conts[b].cpu().numpy()
                    assert host.size == gold["container_len"], f"{gen} {w}x{h}: {host.size} bytes, the reference's container has {gold['container_len']}"
                    assert mi.fnv1a64(host) == gold["container_fnv1a64"], f"{gen} {w}x{h}: sharded container differs from the reference's"
                    continue
                want = _oracle_container(full[b], tw, th, planar) if w * h * c <= 4_000_000 else \
                    mi.compress_image(full[b], w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar, device=ordinal)
                assert bytes(conts[b].cpu().numpy()) == want, f"{gen} {w}x{h}: image {b} differs from the one-piece container"
            out = sc.decode(conts)
            assert out.is_cuda and torch.equal(out, band), "decoded rows differ from the source rows"
            px = sc.gather_pixels(out)
            if rank == 0:
                assert np.array_equal(px.cpu().numpy(), full)
            if force or world > 1:  # the payload collective ran on device tensors, once per direction
                assert sc.exchanges == 2 and (backend != "nccl" or sc.comm_device.type == "cuda")
            del sc
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def _run(world, backend, cases, own_gpu=False, force=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, backend, port, cases, q, own_gpu, force)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(r, "ok") for r in range(world)], res
    assert all(p.exitcode == 0 for p in procs)


def test_sharded_path_world1_rccl_device_tensors():
    _run(1, "nccl", [(("mid", 2048, 1024, 3), (128, 128), True, 2, 4),
                     (("g3", 1000, 333, 4), (480, 1), True, 1, 4),
                     (("mid", 8192, 8192, 3), (128, 128), True, 1, 4)])   # BASELINE config 4 size


def test_sharded_path_world2_hip_coder_gloo_exchange():
    _run(2, "gloo", [(("mid", 2048, 1024, 3), (128, 128), True, 2, 4),
                     (("g3", 1000, 333, 4), (480, 1), True, 2, 3),
                     (("nat", 777, 130, 3), (64, 16), False, 1, 1),
                     (("mid", 640, 360, 3), (64, 64), True, 3, 2, 1),                 # every container funnelled to rank 1
                     (("g3", 8192, 2048, 3), (480, 1), True, 1, 4),       # a quarter of config 4, noise
                     (("g3", 8192, 8192, 3), (512, 1), True, 1, 4)])      # config 4 itself at the benchmarked slicing: golden from the real reference


def test_sharded_path_world1_rccl_forced_alltoallv():
    """the RCCL all_to_all_single with uneven uint8 splits on device tensors, executed (not short-circuited) on one GPU"""
    _run(1, "nccl", [(("mid", 2048, 1024, 3), (128, 128), True, 2, 4),
                     (("g3", 1000, 333, 4), (480, 1), True, 3, 4),
                     (("nat", 777, 130, 3), (64, 16), False, 1, 1),
                     (("g3", 8192, 2048, 3), (512, 1), True, 2, 4),
                     (("g3", 8192, 8192, 3), (512, 1), True, 1, 4)], force=True)   # BASELINE config 4, benchmarked slicing: golden from the real reference


def _gpus():
    import torch

    return torch.cuda.device_count()  # (counting devices does not initialise the GPU in this process)


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs: one RCCL rank per device")
def test_sharded_path_world2_rccl():
    _run(2, "nccl", [(("mid", 2048, 1024, 3), (128, 128), True, 2, 4),
                     (("g3", 1000, 333, 4), (480, 1), True, 2, 3),
                     (("nat", 777, 130, 3), (64, 16), False, 1, 1),
                     (("mid", 640, 360, 3), (64, 64), True, 3, 2, 1),
                     (("g3", 8192, 2048, 3), (512, 1), True, 2, 4),
                     (("g3", 8192, 8192, 3), (512, 1), True, 1, 4)], own_gpu=True)


def test_bench_line_from_four_ranks_rehearsed_on_one_gpu():
    """bench.py's N > 1 path end to end, as the driver starts it (bare `--gpus N`: bench.py spawns torch.distributed.run itself):
    four ranks share cuda:0 and exchange over gloo (`--rehearse-one-gpu`; a one-GPU box allows at most six processes on the card,
    so 8 ranks cannot be rehearsed here -- the world-8 index arithmetic runs on the CPU in tests/test_sharding_gloo.py).  Exactly
    one JSON line must come out, carrying the contract's keys, all four ranks in every leg, and a sharded config-4 leg whose
    containers were verified against the one-piece container inside bench.py.  The numbers themselves mean nothing."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearse-one-gpu", "--steps", "2", "--warmup", "1",
                        "--frames", "4", "--streams", "2", "--c4-images", "4", "--no-cpu-baseline", "--legs-timeout", "400"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 4 and d["ranks_seen"] == 4 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert len(d["per_rank_one_gpu_value"]) == 4 and all(v > 0 for v in d["per_rank_one_gpu_value"])
    assert d["collective_backend"] == "gloo"  # (the rehearsal; the driver's run says nccl)
    c4 = d["c4_sharded"]
    assert c4.get("ranks_seen") == 4 and c4["value"] > 0 and c4["scaling"] == "strong" and c4["images_per_step"] == 4, c4
    c5 = d["c5_replica_pcie"]
    assert c5["ranks_ok"] == 4 and c5["value"] > 0, c5
    # the device-list leg: rank 0 alone through llcomp_mi_opts.devices (here {0,0,0}: three lanes on the one GPU) while three ranks wait on the store
    ip = d["c4_inprocess_devices"]
    assert "failed" not in ip and ip["value"] > 0 and ip["devices"] == [0, 0, 0] and ip["ranks_waiting"] == 3 and ip["golden_pin"], ip
    assert list(d)[-1] == "c4_inprocess_devices"
