"""CPU stand-ins for bench.py's GPU coders -- TEST INFRASTRUCTURE, loaded by bench.py only when LLCOMP_BENCH_STANDIN names this module
(tests/test_bench_world8.py).  Purpose: let the N > 1 BOOKKEEPING of bench.py (rank layout, the reductions behind `value`,
`per_rank_one_gpu_value`, the config-4 strong-scaling leg over the REAL llcomp_amd.sharding code, the config-5 replica leg, the
in-process leg's store hand-shake, the watchdog) run at world size 8 on a machine without a GPU, over gloo.  The local coder is the
oracle (as in tests/test_sharding_gloo.py); the numbers such a run prints mean nothing.  Nothing of this file is in the product path."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc as orc_mod  # noqa: E402  (the checker: this module lives under tests/)
from test_sharding_gloo import oracle_band_factory  # noqa: E402

_orc = orc_mod.Orc()


def _measure(frames_np, tile_w, tile_h, planar, streams, steps, warmup, local_rank, barrier=None, isolated=False, per_step=False):
    """the dict bench.measure returns, from a 96x16 crop of every frame through the oracle (same slicing)"""
    F, h, w, c = frames_np.shape
    crop = np.ascontiguousarray(frames_np[:, :16, :96])
    tw, th = min(tile_w or 96, 96), min(tile_h or 16, 16)
    S = max(1, min(streams, F))
    conts = [_orc.compress_sliced(crop[i], tw, th, planar) for i in range(F)]
    for i in range(F):
        rc, px = _orc.decompress(conts[i])
        assert rc == 0 and np.array_equal(px, crop[i])
    if barrier:
        barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        for i in range(F):
            _orc.decompress(_orc.compress_sliced(crop[i], tw, th, planar))
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    n_slices = F * _orc.slice_count(96, 16, c, tw, th, planar)
    container_bytes = sum(len(x) for x in conts)
    prof = {k: 1.0 for k in ("clear_states_enc", "k_model_fwd", "k_encode_slices", "scan+pack", "k_scan_lengths_dec", "k_decode_slices", "k_model_inv", "clear_states_dec")}
    return dict(dt=dt, steps=steps, F=F, S=S, w=w, h=h, c=c, n_slices=n_slices, payload=container_bytes, container_bytes=container_bytes, raw_bytes=int(crop.size),
                prof=prof, counters={}, step_ms=[], frame0_container=len(conts[0]), frame0_fnv="", n_enc=steps * S, n_dec=steps * S, iso={}, iso_enc=0, iso_dec=0,
                mpix=F * w * h * steps / dt / 1e6, ratio=crop.size / container_bytes)


def _c5_stream(frames_np, tile_w, tile_h, planar, **kw):
    F, h, w, c = frames_np.shape
    t0 = time.perf_counter()
    for i in range(F):
        crop = np.ascontiguousarray(frames_np[i, :8, :64])
        rc, px = _orc.decompress(_orc.compress_sliced(crop, 64, 1, planar))
        assert rc == 0 and np.array_equal(px, crop)
    dt = time.perf_counter() - t0
    return {"value": round(F * w * h / 1e6 / dt, 1), "unit": "MPix/s", "frames": F, "seconds_first_submit_to_last_result": dt, "stand_in": True}


def _one_piece(img, w, h, tile_w, tile_h):
    return _orc.compress_sliced(img, tile_w, tile_h, True)


def _inprocess(devices, images=2, size=64, tile_w=16, tile_h=1, **kw):
    img = orc_mod.gen_mid(size, size, 3)
    for _ in range(images):
        rc, px = _orc.decompress(_orc.compress_sliced(img, tile_w, tile_h, True))
        assert rc == 0 and np.array_equal(px, img)
    return {"value": 1.0, "unit": "MPix/s", "devices": list(devices), "distinct_gpus": len(set(devices)), "images_per_step": images, "stand_in": True}


def install(hooks):
    hooks.device, hooks.backend = "cpu", "gloo"
    hooks.measure, hooks.c5_stream, hooks.one_piece, hooks.inprocess = _measure, _c5_stream, _one_piece, _inprocess
    hooks.band_factory = oracle_band_factory(_orc)
