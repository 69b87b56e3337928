"""bench.py's N > 1 bookkeeping at WORLD SIZE 8 on the CPU.  The first contact of this file with eight ranks must not be the graded
run: the launcher path (`python bench.py --gpus 8` -> torch.distributed.run -> 8 ranks), the reductions behind `value` /
`per_rank_one_gpu_value`, the config-4 strong-scaling leg (the REAL llcomp_amd.sharding code with 3 images over 8 ranks of a 67-row
image: neither divides), the config-5 replica leg, the in-process device-list leg's store hand-shake and the watchdog all run here over
gloo, with the GPU coders replaced by the oracle through LLCOMP_BENCH_STANDIN=bench_standin (tests/bench_standin.py).  The numbers mean
nothing and no curve is claimed: what is checked is that ONE well-formed line comes out and every rank took part in every leg."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_line_at_world_8_over_gloo():
    env = dict(os.environ, LLCOMP_BENCH_STANDIN="bench_standin", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), env.get("PYTHONPATH", "")])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--frames", "2", "--c4-images", "3",
           "--c4-size", "67", "--c4-tile-w", "16", "--c4-tile-h", "1", "--legs-timeout", "400", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["scaling"] == "weak" and d["steps"] == 2 and d["value"] > 0
    assert len(d["per_rank_one_gpu_value"]) == 8 and all(v > 0 for v in d["per_rank_one_gpu_value"])
    assert abs(d["per_gpu_value"] * 8 - d["value"]) < 0.1 * 8
    assert d["collective_backend"] == "gloo" and "lost_legs" not in d
    c4 = d["c4_sharded"]
    assert c4["ranks_seen"] == 8 and c4["images_per_step"] == 3 and c4["scaling"] == "strong" and c4["value"] > 0 and c4["one_gpu_value"] > 0, c4
    assert c4["payload_collectives_per_step"] == 2 and c4["parts"] == 1, c4
    c5 = d["c5_replica_pcie"]
    assert c5["ranks_ok"] == 8 and c5["value"] > 0 and c5["frames_per_rank"] == 64, c5
    inproc = d["c4_inprocess_devices"]
    assert inproc["devices"] == list(range(8)) and inproc["ranks_waiting"] == 7, inproc
    assert list(d)[-1] == "c4_inprocess_devices"
