#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: encode+decode MPix/s on 4K RGB8, bit-exact, with achieved HBM GB/s vs peak.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--content g3|g2|mid|nat] [--tile-w TW --tile-h TH] [--interleaved]

N = 1 (default): BASELINE config 3.  A "step" = one pass of the hot path over one batch: F frames of 3840x2160 RGB8 already
resident in HBM are encoded into the sliced container payload (stage A -> k_encode_slices -> scan + pack) and decoded back
(stage streams -> k_decode_slices -> stage A inverse); the round trip is verified bit-exact outside the timed region.  The
frames of a step are split over --streams independent pipelines (codec object + HIP stream each) so that the memory-bound
kernels of one overlap the issue-bound slice kernels of another.  value = pixels coded / wall time, i.e.
w*h / (t_enc + t_dec) per frame.  `also` = the other workloads of DESIGN.md section 7, a few steps each, measured in the
same run: other contents, 2-D tiles, one-frame latency, batched legacy streams, the PCIe-inclusive C5 stream, and the C4
workload on one GPU (the N = 1 point of the strong-scaling curve).

N > 1: one rank per GPU over RCCL.  Started bare (`python bench.py --gpus N`, no WORLD_SIZE in the environment) this
process touches no GPU: it starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a
child, relays rank 0's JSON line and exits with the child's code; started by torch.distributed.run it is one of the ranks.
`value` is the SAME metric and workload as at N = 1 -- config 3, F frames per GPU per step, frames dealt to the ranks
(independent objects: no data-path collective, weak scaling), barrier + synchronize on both sides, max over ranks -- so
value(N) / value(1) is a scaling point.  `c4_sharded` = BASELINE config 4, the path that has a real exchange step, STRONG
scaling: a fixed batch of 8192x8192 RGB8 images, every image sharded over all ranks by interleaved chunks of tile rows
(llcomp_amd/sharding.py): local encode, slice-table all_gather, one variable-size all-to-all of the payloads (RCCL, image b
is gathered on rank b % N), device concatenator -> complete containers spread over the ranks; decode mirrors it; the
exchange is inside the timed region and container 0 is compared with the one-piece container.  Its one-GPU point is
measured in the same run (rank 0 codes the same batch through a one-rank subgroup first): `one_gpu_value`,
`scaling_vs_one_gpu`, `ranks_seen`.  `c5_replica_pcie` = config 5 in replica mode.  `c4_inprocess_devices` = config 4 from ONE
process through the C ABI's device list (llcomp_mi_opts.devices / llcomp_mi_decode_into_devices, host buffers in, host buffers out):
rank 0 alone drives all N GPUs while the other ranks wait on the store (no collective, N PCIe links).  The secondary legs sit behind a
watchdog: if one of them fails or hangs, the line still goes out with the headline and says which leg was lost.

Key order of the line: the full-model figures (`full_model`, then `value_full_model`) are the LAST keys, behind `also` and the CPU
legs, so that a record which keeps only the tail of the line still shows the real-model number.

LLCOMP_BENCH_STANDIN=<module> (tests/test_bench_world8.py only): the named module -- it lives under tests/, nothing of it is in the
product path -- replaces the GPU coders by CPU stand-ins and the backend by gloo, so that the N > 1 BOOKKEEPING of this file (rank
layout, reductions, the legs' keys) can run at world size 8 where there is no GPU.  The numbers of such a run mean nothing.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W4K, H4K, C4K = 3840, 2160, 3
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK = 1024 * 2.4e9 / 2  # wave-instructions/s: 256 CUs x 4 SIMDs, one VALU instruction per 2 cycles at 2.4 GHz
RMW_UBENCH = 24.06e9       # uniformly random dependent 8-byte read-modify-writes/s into 12.4 GB of per-lane tables at the wavefront count
                           # of the 2-D tile legs (3060): tools/ubench/rand_table.hip, profiles/r03_rand_table.txt.  NOT a ceiling for
                           # the kernels: their contexts are not uniformly random (lanes share lines, the decoder skips unchanged banks)
# HBM bytes per sample of the 2-D tile legs' kernels, (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction) / samples, from the
# committed PMC passes of 16 frames / one pipeline (tools/prof_2d.sh).  encode_all_kernels = stage A 5.4 + k_snap_sort 8.0 + k_snap_walk
# 12.0 + k_snap_unperm 22.2 + k_encode_slices + k_pack_payload; decode_all_kernels = k_stage_streams + k_decode_slices + k_from_lane_order
# 4.0 + inverse stage A 3.0.  Round 3's encoder (state tables in HBM) moved 131 (g3) / 147 (nat) in k_encode_slices alone; round 4's
# decoder (two memory round trips per sample, every bank fetched and written in HBM) 156 / 168 in k_decode_slices.
TILE_HBM_SOURCE = "profiles/r05_tiles64_f16_{g3,nat}_pmc_summary.txt (KiB counters x 1024)"
TILE_HBM_BYTES_PER_SAMPLE = {"g3": {"k_encode_slices": 12.9, "encode_all_kernels": 64.1, "k_decode_slices": 98.7, "decode_all_kernels": 108.9},
                             "nat": {"k_encode_slices": 10.7, "encode_all_kernels": 59.1, "k_decode_slices": 120.6, "decode_all_kernels": 128.4}}


class Hooks:
    """what the N > 1 branch calls; a stand-in module named by LLCOMP_BENCH_STANDIN (tests only) replaces the members"""
    standin = None           # the module, when there is one
    device = "cuda"          # where the reduction tensors live
    backend = "nccl"
    measure = None           # filled in below (the functions of this file)
    c5_stream = None
    band_factory = None      # sharding.ShardedCodec's local coder: None = the HIP codec object
    one_piece = None         # (img, w, h, tile_w, tile_h) -> container bytes, for c4_run's check
    inprocess = None         # c4_inprocess

    @staticmethod
    def sync():
        if Hooks.device == "cuda":
            import torch

            torch.cuda.synchronize()


def make_frames(content, frames, rank, w=W4K, h=H4K, c=C4K, distinct=None):
    """frames x [h,w,c] uint8.  `distinct` < frames: only that many frames are generated, the rest are horizontal rotations
    of them (cheap, and different bytes for the coder)."""
    import numpy as np
    from llcomp_amd import synth

    out = np.empty((frames, h, w, c), dtype=np.uint8)
    distinct = frames if distinct is None else max(1, min(distinct, frames))
    for i in range(frames):
        seed = 1234 + rank * frames + i
        if i >= distinct:
            out[i] = np.roll(out[i % distinct], 11 * (i // distinct), axis=1)
        elif content == "g3":
            out[i] = synth.gen_g3(w, h, c, seed=seed)
        elif content == "g2":
            out[i] = np.roll(synth.gen_g2(w, h, c), (seed - 1234) * 5, axis=1)
        elif content == "nat":
            out[i] = synth.gen_nat(w, h, c, seed=seed)
        else:
            out[i] = synth.gen_mid(w, h, c, seed=seed)
    return out


def host_cpu():
    """model name and logical CPU count of the host the CPU legs run on"""
    name = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return f"{name} ({os.cpu_count()} logical CPUs, {usable} usable by this process)"


def cpu_baseline(img, label, tile_w, tile_h, planar, same_slicing=False):
    """Time the CPU path on ONE frame, single thread.  kind 'reference' = the real llcomp.hpp compiled in place
    (oracle/_ref, whole-image stream: O2 encode + unmodified decompressImage); kind 'port' = the plain-C restatement
    (same sliced container as the GPU produces)."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))  # the checker: imported by this leg of the bench only
    import orc as orc_mod

    h, w, _ = img.shape
    if orc_mod.Ref.available():
        ref = orc_mod.Ref()
        t0 = time.perf_counter()
        s = ref.o2_compress_image(img)
        t1 = time.perf_counter()
        rc, px = ref.o1_decompress_image(s)
        t2 = time.perf_counter()
        assert rc == 0 and np.array_equal(px, img)
        kind, sample = "reference", f"1 frame {label}, whole-image stream (ratio {img.size / len(s):.4f}), llcomp.hpp -O2 -DNDEBUG (enc {t1 - t0:.2f}s + dec {t2 - t1:.2f}s)"
    else:
        orc = orc_mod.Orc()
        t0 = time.perf_counter()
        s = orc.compress_sliced(img, tile_w, tile_h, planar)
        t1 = time.perf_counter()
        rc, px = orc.decompress(s)
        t2 = time.perf_counter()
        assert rc == 0 and np.array_equal(px, img)
        kind, sample = "port", f"1 frame {label}, same slicing, plain-C oracle (enc {t1 - t0:.2f}s + dec {t2 - t1:.2f}s)"
    res = {"value": round(w * h / 1e6 / (t2 - t0), 4), "unit": "MPix/s", "cores": 1, "kind": kind, "sample": sample, "cpu": host_cpu()}
    if kind == "reference" and same_slicing:
        # like for like: the plain-C port on the SAME slicing the GPU codes (same container bytes, same ratio), one thread
        orc = orc_mod.Orc()
        t0 = time.perf_counter()
        s2 = orc.compress_sliced(img, tile_w, tile_h, planar)
        t1 = time.perf_counter()
        rc, px = orc.decompress(s2)
        t2 = time.perf_counter()
        assert rc == 0 and np.array_equal(px, img)
        res["same_slicing_port"] = {"value": round(w * h / 1e6 / (t2 - t0), 4), "unit": "MPix/s", "cores": 1, "kind": "port",
                                    "sample": f"the same frame, {tile_w}x{tile_h} {'planar' if planar else 'interleaved'} slices (ratio {img.size / len(s2):.4f}), plain-C oracle (enc {t1 - t0:.2f}s + dec {t2 - t1:.2f}s)"}
        # ... and the same on every host core this process may use (slices are independent: one frame per thread; ctypes
        # releases the GIL).  The reference itself has no threads; this is the generous CPU figure.
        from concurrent.futures import ThreadPoolExecutor

        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = max(1, min(cores, 32))

        def one(_):
            o = orc_mod.Orc()
            rc2, px2 = o.decompress(o.compress_sliced(img, tile_w, tile_h, planar))
            return rc2 == 0 and bool(np.array_equal(px2, img))

        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            ok = list(ex.map(one, range(cores)))
        t1 = time.perf_counter()
        assert all(ok)
        res["same_slicing_port_all_cores"] = {"value": round(cores * w * h / 1e6 / (t1 - t0), 3), "unit": "MPix/s", "cores": cores, "kind": "port",
                                              "sample": f"{cores} threads, one frame each, same slicing ({t1 - t0:.2f}s)"}
    return res


def measure(frames_np, tile_w, tile_h, planar, streams, steps, warmup, local_rank, barrier=None, isolated=False, per_step=False):
    """The device-resident round trip of bench.py: frames_np [F,h,w,c] resident in HBM, split over `streams` pipelines.
    Returns a dict with the wall time of `steps` timed steps and the per-kernel event timings.  per_step: every step is
    bracketed by a device synchronisation of its own and timed alone (`step_ms`, for legs that report median and range); the
    wall time is then the sum of the steps."""
    import torch

    import llcomp_amd as mi

    F, h, w, c = frames_np.shape
    S = max(1, min(streams, F))
    d_px = torch.from_numpy(frames_np).cuda()
    d_out = torch.empty_like(d_px)
    parts, lo = [], 0
    for i in range(S):
        n = F // S + (1 if i < F % S else 0)
        codec = mi.Codec(n, w, h, c, tile_w, tile_h, planar, device=local_rank)
        cap = min(codec.max_payload_bytes, 2 * n * w * h * c + 64 * codec.n_slices + 4096)
        parts.append(dict(codec=codec, lo=lo, n=n, cap=cap, stream=torch.cuda.Stream() if S > 1 else torch.cuda.current_stream(),
                          pay=torch.empty(cap, dtype=torch.uint8, device="cuda"), len=torch.empty(codec.n_slices, dtype=torch.int32, device="cuda"),
                          tot=torch.zeros(1, dtype=torch.int64, device="cuda"), st=torch.zeros(2, dtype=torch.int32, device="cuda"), total=None))
        lo += n
    n_slices = sum(p["codec"].n_slices for p in parts)

    def run(p):
        cd, st = p["codec"], p["stream"].cuda_stream
        px, out = d_px[p["lo"]:p["lo"] + p["n"]], d_out[p["lo"]:p["lo"] + p["n"]]
        cd.encode(px.data_ptr(), p["pay"].data_ptr(), p["cap"], p["len"].data_ptr(), p["tot"].data_ptr(), p["st"].data_ptr(), st)
        # decode needs the payload size on the host only as an upper bound for its bounds checks
        cd.decode(p["pay"].data_ptr(), p["total"] if p["total"] is not None else p["cap"], p["len"].data_ptr(), out.data_ptr(), p["st"][1:].data_ptr(), st)

    def step():
        for p in parts:
            run(p)

    def check():
        torch.cuda.synchronize()
        if os.environ.get("LLCOMP_BENCH_NOCHECK"):  # tools/attic/exp_time.py with a diagnostic build whose bytes are wrong on purpose
            return
        for p in parts:
            assert int(p["st"][0].item()) == 0 and int(p["st"][1].item()) == 0, f"status {p['st'].tolist()}"
        assert torch.equal(d_out, d_px), "round trip is not lossless"

    # first pass: learn the payload sizes, check status and the bit-exact round trip (outside the timed region)
    torch.cuda.synchronize()
    step()
    check()
    for p in parts:
        p["total"] = int(p["tot"].item())
    total = sum(p["total"] for p in parts)
    spf0 = parts[0]["codec"].n_slices // parts[0]["n"]
    frame0_payload = int(parts[0]["len"][:spf0].to(torch.int64).sum().item())
    frame0_container = 24 + 4 * spf0 + frame0_payload  # bytes of frame 0's container
    # ... and its FNV-1a-64, as the golden vectors record it: [24-byte header][u32 slice table][payload]
    head0 = bytes([0x9C, 1, c, 1 if planar else 0]) + b"".join(int(v).to_bytes(4, "little") for v in (w, h, min(tile_w or w, w), min(tile_h or h, h), spf0))
    frame0_fnv = mi.fnv1a64(head0, parts[0]["len"][:spf0].cpu().numpy().astype("<u4").view("uint8"), parts[0]["pay"][:frame0_payload].cpu().numpy())
    for _ in range(max(0, warmup - 1)):
        step()
    torch.cuda.synchronize()
    for p in parts:
        p["codec"].set_profiling(True)
        p["codec"].get_profile()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    step_ms = []
    t0 = time.perf_counter()
    if per_step:
        for _ in range(steps):
            t1 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            step_ms.append((time.perf_counter() - t1) * 1e3)
    else:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    prof, n_enc, n_dec, counters = {}, 0, 0, {}
    for p in parts:
        for k, v in p["codec"].counters().items():  # what the adaptive / rare paths did since the codec was made (llcomp_mi_codec_get_counters)
            counters[k] = counters.get(k, 0) + v
        pr, ne, nd = p["codec"].get_profile()
        p["codec"].set_profiling(False)
        for k, v in pr.items():
            prof[k] = prof.get(k, 0.0) + v
        n_enc, n_dec = n_enc + ne, n_dec + nd
    check()

    # The same launches once more with every pipeline ALONE on the GPU (outside the timed region): per-launch durations
    # without the other streams' kernels beside them, reported next to the live ones (roofline.isolated).
    iso, iso_enc, iso_dec = {}, 0, 0
    if S > 1 and isolated:
        for p in parts:
            cd = p["codec"]
            cd.set_profiling(True)
            cd.get_profile()
            torch.cuda.synchronize()
            for _ in range(2):
                run(p)
                torch.cuda.synchronize()
            pr, ne, nd = cd.get_profile()
            cd.set_profiling(False)
            for k, v in pr.items():
                iso[k] = iso.get(k, 0.0) + v
            iso_enc, iso_dec = iso_enc + ne, iso_dec + nd
    for p in parts:
        p["codec"].close()
    del d_px, d_out, parts
    torch.cuda.empty_cache()
    container_bytes = total + 24 * F + 4 * n_slices  # per-frame container headers + slice tables
    return dict(dt=dt, steps=steps, F=F, S=S, w=w, h=h, c=c, n_slices=n_slices, payload=total, container_bytes=container_bytes,
                raw_bytes=int(frames_np.size), prof=prof, counters=counters, step_ms=step_ms, frame0_container=frame0_container, frame0_fnv=frame0_fnv, n_enc=n_enc, n_dec=n_dec, iso=iso, iso_enc=iso_enc, iso_dec=iso_dec,
                mpix=F * w * h * steps / dt / 1e6, ratio=frames_np.size / container_bytes)


def brief(m, **extra):
    d = {"value": round(m["mpix"], 1), "unit": "MPix/s", "ms_per_step": round(m["dt"] / m["steps"] * 1e3, 3), "steps": m["steps"],
         "frames": m["F"], "streams": m["S"], "compression_ratio": round(m["ratio"], 4)}
    if m.get("step_ms"):  # steps timed one by one: the value is the MEDIAN step's rate, the range says how far single steps fall from it
        ms = sorted(m["step_ms"])
        med = ms[len(ms) // 2] if len(ms) & 1 else (ms[len(ms) // 2 - 1] + ms[len(ms) // 2]) / 2
        pix = m["F"] * m["w"] * m["h"] / 1e6
        d.update({"value": round(pix / (med * 1e-3), 1), "value_is": "median step", "ms_per_step": round(med, 3),
                  "ms_per_step_min_max": [round(ms[0], 3), round(ms[-1], 3)], "value_min_max": [round(pix / (ms[-1] * 1e-3), 1), round(pix / (ms[0] * 1e-3), 1)]})
    d.update(extra)
    return d


def cache_counters(m):
    """the 2-D decoder's bank cache over all calls of the leg (warm-up included), from the codec objects' event counters"""
    c = m.get("counters") or {}
    if not c.get("dec_launches_cached") and not c.get("dec_launches_plain"):
        return {}
    out = {"decode_launches_with_cache": c["dec_launches_cached"], "decode_launches_plain_by_feedback": c["dec_launches_plain"],
           "wavefronts_that_gave_the_cache_up": round(c["dec_bypassed_waves"] / max(1, c["dec_cached_waves"]), 4)}
    if c.get("cache_lookups"):
        out["hit_rate_while_cached"] = round(1 - c["cache_misses"] / c["cache_lookups"], 4)
    return {"bank_cache": out}


def tile_sides(m):
    """per launch (one pipeline's frames), summed over the kernels of a direction, from the library's own events"""
    p, ne, nd = m["prof"], max(1, m["n_enc"]), max(1, m["n_dec"])
    enc = {"stage_a": p["k_model_fwd"] / ne, "state_snapshot_pass": p["clear_states_enc"] / ne, "k_encode_slices": p["k_encode_slices"] / ne, "scan+pack": p["scan+pack"] / ne}
    dec = {"scan+stage": p["k_scan_lengths_dec"] / nd, "k_decode_slices": p["k_decode_slices"] / nd, "stage_a_inverse": (p["k_model_inv"] + p["clear_states_dec"]) / nd}
    return {"encode_ms_per_launch": {k_: round(v_, 3) for k_, v_ in enc.items()}, "encode_ms_per_launch_sum": round(sum(enc.values()), 3),
            "decode_ms_per_launch": {k_: round(v_, 3) for k_, v_ in dec.items()}, "decode_ms_per_launch_sum": round(sum(dec.values()), 3)}


def profile_numbers(F, tile_w, tile_h, planar, content, S, dom):
    """HBM-side bytes and VALU instructions per launch of kernel `dom` from the committed rocprofv3 PMC passes of THIS
    configuration (profiles/*_traffic.json); (None, None, None) when no committed profile matches."""
    for name in ("r06_default_traffic.json", "r06_full_model_traffic.json", "r06_single_stream_traffic.json", "r05_default_traffic.json", "r05_single_stream_traffic.json", "r04_default_traffic.json", "r04_single_stream_traffic.json", "r03_default_traffic.json", "r03_single_stream_traffic.json",
                 "r02_default_traffic.json", "r02_single_stream_traffic.json", "r01_default_traffic.json", "r01_single_stream_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
            same = all(tj["config"].get(k) == v for k, v in (("frames_per_step_per_gpu", F), ("tile_w", tile_w), ("tile_h", tile_h),
                                                            ("planar", planar), ("content", content), ("streams", S)))
            if same:
                k = tj["per_launch"][dom]
                # (the commit the profiled tree was at: the passes have to be re-run whenever the dominant kernel's source changes)
                return k.get("hbm_bytes_corrected"), k.get("valu_insts"), "profiles/" + name + (" @ " + tj["profiled_at_commit"] if tj.get("profiled_at_commit") else "")
        except (OSError, KeyError, ValueError):
            pass
    return None, None, None


def gpu_local_cpus(device=0):
    """CPUs of the NUMA node the GPU hangs off (sysfs local_cpulist of its PCI function) that this process may run on; None
    when the topology is not visible (containers often hide it)"""
    try:
        import torch

        p = torch.cuda.get_device_properties(device)
        path = f"/sys/bus/pci/devices/{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0/local_cpulist"
        cpus = set()
        for part in open(path).read().strip().split(","):
            if part:
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        return cpus or None
    except Exception:  # noqa: BLE001
        return None


def link_rate(n=256 << 20, reps=6, streams=None):
    """what the host link gives on this box right now: copies between hipHostMalloc'ed (pinned, GPU-local) memory and HBM, H2D
    alone, D2H alone and both directions at once on two HIP streams, 256 MiB each, in GB/s -- what the PCIe-inclusive config-5
    leg is to be read against.  (Plain hipMemcpyAsync through ctypes: the same calls and the same kind of buffers as the
    streaming pipeline uses.)"""
    import ctypes as C

    import torch

    import llcomp_amd as mi

    mi.device_count()  # (loads the library and with it the process's one HIP runtime, globally)
    hip = C.CDLL(None)
    hip.hipMemcpyAsync.restype = C.c_int
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    h_a, h_b = mi.PinnedBuffer(n), mi.PinnedBuffer(n)
    d_a, d_b = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")
    # FRESH HIP streams of this function's own (not torch's pooled ones): which copy engine a stream's copies run on is decided
    # inside the HIP runtime when the stream first copies, and a pooled stream that has already been used elsewhere can sit on
    # a shared engine (profiles/r03_c5_repeat.txt, run 6)
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    own = []
    if streams is None:
        for _ in range(2):
            h = C.c_void_p()
            assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0
            own.append(h)
        s1h, s2h = own[0].value, own[1].value
    else:
        s1h, s2h = streams[0].cuda_stream, streams[1].cuda_stream

    def run(h2d, d2h, k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            if h2d:
                assert hip.hipMemcpyAsync(d_a.data_ptr(), h_a.ptr, n, 1, s1h) == 0
            if d2h:
                assert hip.hipMemcpyAsync(h_b.ptr, d_b.data_ptr(), n, 2, s2h) == 0
        torch.cuda.synchronize()
        return (h2d + d2h) * k * n / (time.perf_counter() - t0) / 1e9

    run(True, True, 2)
    out = {"h2d_alone_GBps": round(run(True, False, reps), 1), "d2h_alone_GBps": round(run(False, True, reps), 1), "both_directions_GBps": round(run(True, True, reps), 1)}
    torch.cuda.synchronize()
    for h in own:
        hip.hipStreamDestroy(h)
    h_a.close()
    h_b.close()
    del d_a, d_b
    return out


def c5_stream(frames_np, tile_w, tile_h, planar, depth=6, frames_per_job=8, pipelines=2, encodes_in_flight=2, passes=4, pin=True, verify=True, link=None, devices=None):
    """BASELINE config 5, PCIe inclusive: the frames stream host -> GPU -> host (container) -> GPU -> host through the
    product's pipeline (llcomp_mi_stream_*, jobs of `frames_per_job` frames), `passes` times over the batch; every frame verified
    bit-exact.  `pipelines` stream objects, each driven by its own thread over its own share of the frames: one pipeline
    hands its results back in submission order and leaves the two DMA directions idle now and then.  Defaults = the shape
    that was both fastest and steadiest in tools/attic/c5_repeat.py (profiles/r03_c5_repeat.txt): two pipelines, jobs of 8 frames
    (200 MB copies), 6 slots, 2 encodes in flight each -- 5.65-6.07 GPix/s over five repetitions against 5.3-6.4 with jobs of
    4 frames in 8 slots; the link gives 97 GB/s with both directions busy = 7.1 GPix/s at 114 MB per frame."""
    import threading

    import llcomp_amd as mi

    F, h, w, c = frames_np.shape
    per = (F // pipelines) // frames_per_job * frames_per_job  # frames per pipeline
    assert per >= 4, "too few frames for this many pipelines"
    F = per * pipelines
    pinned = mi.PinnedBuffer(F * h * w * c)
    pinned.array[:] = frames_np[:F].reshape(-1)
    views = [pinned.array[i * h * w * c:(i + 1) * h * w * c].reshape(h, w, c) for i in range(F)]
    # devices: every pipeline object is a dealer over that device list (llcomp_mi_stream_create_multi: one pipeline of `depth` slots per
    # device behind one object, jobs dealt round-robin, results in submission order)
    sts = [mi.Stream(w, h, c, tile_w, tile_h, planar, depth=depth, frames_per_job=frames_per_job, devices=devices) for _ in range(pipelines)]
    res, errs = [None] * pipelines, []
    origin = time.perf_counter()

    local = gpu_local_cpus() if pin else None

    def drive(t):
        try:
            if local:  # the driving thread (and the verification workers it starts) stay on the GPU's NUMA node, next to the pinned buffers
                os.sched_setaffinity(0, local)
            mine = views[t * per:(t + 1) * per]  # (the frames of a job must be adjacent in memory)
            res[t] = mi.pipeline_roundtrip(sts[t], mine * passes, max_encodes_in_flight=encodes_in_flight, verify=verify,
                                           verify_threads=max(2, 6 // pipelines), clock_origin=origin)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=drive, args=(t,)) for t in range(pipelines)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for st in sts:
        st.close()
    pinned.close()
    if errs:
        raise errs[0]
    n = passes * F
    skip = max(4, frames_per_job)  # every pipeline's first 4 frames = its first job are left out
    begin = max(r[1][skip - 1] for r in res)   # the moment the last pipeline has its first job back
    end = max(r[1][-1] for r in res)
    counted = sum(sum(1 for t in r[1] if t > begin) for r in res)
    steady = counted * w * h / 1e6 / (end - begin)
    total_len = sum(sum(r[0]) for r in res)
    per_frame = int(2 * (h * w * c + total_len / n))
    extra = {}
    if link:  # bytes over the link per second against what the link gave in this process a moment ago
        extra = {"link": link, "pcie_GBps": round(per_frame * steady * 1e6 / (w * h) / 1e9, 1),
                 "pcie_frac": round(per_frame * steady * 1e6 / (w * h) / 1e9 / link["both_directions_GBps"], 3)}
    return {"value": round(steady, 1), "unit": "MPix/s", "frames": n, "seconds_first_submit_to_last_result": round(end, 4),
            "whole_run_MPix_s": round(n * w * h / 1e6 / end, 1), "frames_per_job": frames_per_job, "depth": depth, "pipelines": pipelines,
            "compression_ratio": round(n * h * w * c / total_len, 4), "threads_pinned_to_gpu_numa_node": bool(local), **extra,
            "pcie_bytes_per_frame": per_frame, "backpressure_hits": sum(r[2] for r in res),
            "note": "end to end over PCIe from/to pinned host memory, steady state (every pipeline's first 4 frames excluded), every frame bit-exact; never part of `value`"}


def c4_run(images, size, tile_w, tile_h, steps, warmup, local_rank, world, rank, check_one_piece=True, parts=0, group=None, height=None):
    """(see below; device-agnostic where it matters: under LLCOMP_BENCH_STANDIN the tensors are CPU tensors and there are no HIP streams)"""
    return _c4_run(images, size, tile_w, tile_h, steps, warmup, local_rank, world, rank, check_one_piece, parts, group, height)


def _c4_run(images, size, tile_w, tile_h, steps, warmup, local_rank, world, rank, check_one_piece, parts, group, height):
    """BASELINE config 4: `images` noise images of size x size RGB8, each sharded over all ranks; encode (+ exchange of the
    bitstream) and decode (+ exchange back) per step.  The batch is coded as up to three part batches on as many HIP streams
    (a ShardedCodec each) so that the exchange of one part overlaps the coding of the others.  Returns the max-over-ranks wall time."""
    import torch
    import torch.distributed as dist

    import contextlib

    import llcomp_amd as mi
    from llcomp_amd import sharding

    cuda = Hooks.device == "cuda"
    dev = torch.device("cuda", local_rank) if cuda else torch.device("cpu")
    sync = Hooks.sync
    height = height or size  # (tools/attic/c4_overhead.py codes bands of the config-4 image: one rank's share at N > 1, on one GPU)
    # 3, 2 or 1 part batches (measured on one GPU: 2, 3 and 6 parts all take 82 ms per step), each spreading its containers evenly over the ranks
    halves = next(p for p in ((parts,) if parts else ()) + (3, 2, 1) if images % p == 0 and (images // p) % world == 0 or p == 1)
    per = images // halves
    scs = [sharding.ShardedCodec(size, height, 3, tile_w, tile_h, True, images=per, device=dev, group=group, band_factory=Hooks.band_factory) for _ in range(halves)]
    streams = [torch.cuda.Stream(device=dev) if cuda else None for _ in range(halves)]
    # uniform byte noise (torch Philox, seed 1234 + image): every rank draws the full image on its GPU and keeps its rows
    bands, first = [], None
    for k, sc in enumerate(scs):
        rows = {}
        for j in range(per):
            b = k * per + j
            g = torch.Generator(device=dev)
            g.manual_seed(1234 + b)
            full = torch.randint(0, 256, (1, height, size, 3), dtype=torch.uint8, device=dev, generator=g)
            rows[j] = torch.cat([full[:, y0:y1] for y0, y1 in sc.rows], dim=1) if sc.rows else full[:, :0]
            if b == 0 and rank == sc.root_of[0] and check_one_piece:
                first = full[0].cpu().numpy()
            del full
        bands.append(torch.cat([rows[j] for j in sc.frame_images], dim=0).contiguous())  # frames in the codec's order
        del rows
    sync()

    def step():
        """software pipeline over the parts: while the host waits for one part's sizes (the exchange needs them) the GPU
        already holds the next part's coding, and a part's decode is queued as soon as its containers exist -- so encode
        and decode kernels of different parts, memory-bound and issue-bound ones, run beside each other."""
        conts, outs = [None] * halves, [None] * halves

        def on(k, fn, *a, **kw):
            with (torch.cuda.stream(streams[k]) if cuda else contextlib.nullcontext()):
                return fn(*a, **kw)

        if os.environ.get("LLCOMP_BENCH_C4_ORDER", "one-ahead") == "deep":
            # every part's coding queued before the host waits for the first part's sizes.  Measured (tools/attic/c4_overhead.py,
            # profiles/r03_c4_overhead.jsonl): 2-3 % SLOWER than one part ahead at every modelled rank count -- the path's cost
            # beside the coding is not host latency
            for k in range(halves):
                on(k, scs[k].encode_begin, bands[k])
            for k in range(halves):
                conts[k] = on(k, scs[k].encode_finish)
                on(k, scs[k].decode_begin, conts[k], validate=False)  # straight from encode: no header round trip
            for k in range(halves):
                outs[k] = on(k, scs[k].decode_finish)
        else:  # round 2's order: one part of coding ahead of the host
            for k in range(halves + 2):
                if k < halves:
                    on(k, scs[k].encode_begin, bands[k])
                if 1 <= k <= halves:
                    conts[k - 1] = on(k - 1, scs[k - 1].encode_finish)
                    on(k - 1, scs[k - 1].decode_begin, conts[k - 1], validate=False)
                if k >= 2:
                    outs[k - 2] = on(k - 2, scs[k - 2].decode_finish)
        # every part has drained its own stream by now (decode_finish reads its status); the device-wide wait costs
        # microseconds and keeps steps from interleaving in the runtime's queues (without it some runs of this leg took
        # twice as long per step, with identical kernels and allocator statistics)
        sync()
        return conts, outs

    conts, outs = step()
    sync()
    for k in range(halves):
        assert torch.equal(outs[k], bands[k]), "sharded round trip is not lossless"
    pb = torch.tensor([sum(int(c_.numel()) for cs in conts for c_ in cs.values())], dtype=torch.int64, device=dev)
    dist.all_reduce(pb, op=dist.ReduceOp.SUM, group=group)
    payload_bytes = int(pb.item())
    if first is not None:
        one = Hooks.one_piece(first, size, height, tile_w, tile_h) if Hooks.one_piece else \
            mi.compress_image(first, size, height, 3, format=mi.FORMAT_SLICED, tile_w=tile_w, tile_h=tile_h, planar=True, device=local_rank)
        assert bytes(conts[0][0].cpu().numpy()) == one, "sharded container differs from the one-piece container"
    # At least three more untimed passes, bound exactly like the timed ones (the results of pass i stay alive while pass
    # i + 1 runs): that is when torch's allocator takes its last segments from the driver.  With the results dropped at
    # once, the first such hipMalloc fell into the SECOND timed step instead -- 260 ms on a freshly provisioned box, one
    # step of four at 340 ms beside three at 80 ms.
    for _ in range(max(3, warmup - 1)):
        conts, outs = step()
    sync()
    # N = 1 only (an `also` leg there; at N > 1 this IS the timed headline and carries no instrumentation)
    codecs = [sc.band.codec for sc in scs if getattr(sc.band, "codec", None) is not None] if world == 1 else []
    for cd in codecs:  # hipEvent spans of the local coding kernels, so that a slow step can be told from a slow exchange
        cd.set_profiling(True)
        cd.get_profile()
    dist.barrier(group=group)
    sync()
    t0 = time.perf_counter()
    marks = []
    for _ in range(steps):
        conts, outs = step()
        marks.append(time.perf_counter())
    sync()
    dist.barrier(group=group)
    dt = time.perf_counter() - t0
    spans = {}
    for cd in codecs:
        pr, _, _ = cd.get_profile()
        cd.set_profiling(False)
        for k_, v_ in pr.items():
            spans[k_] = spans.get(k_, 0.0) + v_ / steps
    c4_run.last_detail = {"kernel_ms_per_step": {k_: round(v_, 3) for k_, v_ in spans.items()},
                          "step_ms": [round(1e3 * (b - a), 1) for a, b in zip([t0] + marks[:-1], marks)], "parts": halves,
                          "payload_collectives_per_step": sum(sc.exchanges for sc in scs) // max(1, steps + max(3, warmup - 1) + 1)}
    if os.environ.get("LLCOMP_BENCH_ALLOC") and rank == 0:
        st_ = torch.cuda.memory_stats()
        print("c4 allocator:", {k: st_.get(k) for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.peak", "allocated_bytes.all.peak")},
              "ms/step", round(1e3 * dt / steps, 1), file=sys.stderr, flush=True)
    if os.environ.get("LLCOMP_BENCH_DEBUG") and rank == 0:
        print("c4 step times (ms):", [round(1e3 * (b - a), 1) for a, b in zip([t0] + marks[:-1], marks)], "reserved GB", round(torch.cuda.memory_reserved() / 1e9, 1),
              "free GB", round(torch.cuda.mem_get_info()[0] / 1e9, 1), file=sys.stderr, flush=True)
    for k in range(halves):
        assert torch.equal(outs[k], bands[k])
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    del scs, bands, outs, conts
    if cuda:
        torch.cuda.empty_cache()
    return float(t.item()), payload_bytes


def c4_inprocess(devices, images=4, size=8192, tile_w=512, tile_h=1, steps=2, callers=2, compare_one_device=True):
    """BASELINE config 4 from ONE process through the C ABI's device list (csrc/multidev.hip): every image goes host buffer ->
    llcomp_mi_encode_into(opts.devices) -> container in host memory -> llcomp_mi_decode_into_devices -> host buffer.  Each device gets
    only its tile rows over its own PCIe link and copies its payload / its rows straight to their place: no collective, no gather.
    `callers` threads call concurrently (the C ABI is thread-safe; a second caller's copies overlap the first one's kernels), each
    with its own pinned buffers.  PCIe inclusive by nature -- never part of `value`.  Image 0 is the golden vector's image
    (std::mt19937(1234)): its container must be the one assembled from the REAL reference's per-slice streams
    (tests/golden/c4_bench_slicing.json); the other images are rotations of it."""
    import threading

    import numpy as np

    import llcomp_amd as mi
    from llcomp_amd import synth

    base = synth.gen_g3(size, size, 3)
    raw = base.size
    kw = dict(format=mi.FORMAT_SLICED, tile_w=tile_w, tile_h=tile_h, planar=True)
    try:
        gold = [v for v in json.load(open(os.path.join(ROOT, "tests", "golden", "c4_bench_slicing.json")))["vectors"]
                if (v["gen"], v["w"], v["h"], v["tile_w"], v["tile_h"], v["planar"]) == ("g3", size, size, tile_w, tile_h, True)]
    except OSError:
        gold = []
    callers = max(1, min(callers, images))
    srcs = [mi.PinnedBuffer(raw) for _ in range(images)]          # every image's pixels, pinned (plain DMA per chunk of tile rows)
    for b_, sb in enumerate(srcs):
        sb.array[:] = np.roll(base, 11 * b_, axis=1).reshape(-1)
    del base
    outs = [(mi.PinnedBuffer(2 * raw), mi.PinnedBuffer(raw)) for _ in range(callers)]  # per caller: container, decoded pixels
    bufs = [(x,) for x in srcs] + [tuple(o) for o in outs]

    def run(devs, n_steps, check):
        """every caller codes images t, t + callers, ... n_steps times; returns (seconds, container lengths)"""
        errs, lens = [], {}

        def work(t):
            try:
                cont, back = outs[t]
                for _ in range(n_steps):
                    for b in range(t, images, callers):
                        src = srcs[b]
                        n = mi.compress_image_into(src.array, size, size, 3, cont.array, devices=devs, **kw)
                        lens[b] = n
                        if check and b == 0 and gold:
                            assert n == gold[0]["container_len"] and mi.fnv1a64(cont.array[:n]) == gold[0]["container_fnv1a64"], \
                                "the device-list container of image 0 differs from the reference's"
                        mi.decompress_image_into(cont.array[:n], back.array, devices=devs)
                        if check:
                            assert mi._same_bytes(back.array, src.array), "device-list round trip is not lossless"
            except BaseException as e:  # noqa: BLE001
                errs.append(e)

        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(t,)) for t in range(callers)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        if errs:
            raise errs[0]
        return time.perf_counter() - t0, lens

    try:
        run(devices, 1, True)            # untimed: lanes, code objects, first touch of the pinned buffers; golden pin + lossless check
        dt, lens = run(devices, steps, False)
        out = {"value": round(images * steps * size * size / dt / 1e6, 1), "unit": "MPix/s", "devices": list(devices), "distinct_gpus": len(set(devices)),
               "images_per_step": images, "steps": steps, "concurrent_callers": callers, "ms_per_image_encode_plus_decode": round(dt / (images * steps) * callers * 1e3, 2),
               "compression_ratio": round(images * raw / sum(lens.values()), 4), "golden_pin": bool(gold),
               "workload": f"C4 {images} x {size}x{size} RGB8 std::mt19937 noise, {tile_w}x{tile_h} planar, host buffer -> llcomp_mi_encode_into(devices) -> "
                           f"host container -> llcomp_mi_decode_into_devices -> host buffer; {callers} concurrent callers with pinned buffers; "
                           "PCIe inclusive (every byte crosses the link twice per direction of the round trip)"}
        if compare_one_device:
            one = [devices[0]]
            run(one, 1, False)
            dt1, _ = run(one, steps, False)
            out["one_device_value"] = round(images * steps * size * size / dt1 / 1e6, 1)
            out["vs_one_device"] = round(dt1 / dt, 3)
        if len(set(devices)) < 2:
            out["note"] = "the list repeats one ordinal: two lanes on ONE GPU -- the path's overhead on one card, not a multi-GPU number"
        return out
    finally:
        for tr in bufs:
            for b in tr:
                b.close()
        mi.trim()


def spawn_ranks(n):
    """`python bench.py --gpus N` started bare: this process must not touch the GPU (a process that has initialised HIP may
    neither fork ranks nor be replaced by another program on this pool), so it only starts the launcher as a child, relays
    rank 0's JSON line and hands the child's exit code on."""
    import socket
    import subprocess

    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host driver
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in child.stdout:  # the ranks' stdout: exactly one JSON line is expected (anything else goes to stderr)
        t = out.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    raise SystemExit(rc)


def headline(args, m, world, planar):
    """the JSON line's common part: BASELINE's metric on config 3 from measure()'s numbers (at N > 1: rank 0's kernels, the
    whole job's pixels over the max-over-ranks time)"""
    F = args.frames
    S, prof = m["S"], m["prof"]
    # roofline of the dominant kernel (algorithmic bytes of one coding direction = raw + stream), per launch
    # (a launch covers F/S frames), duration measured live with hipEvents on the launching stream
    k_enc = prof["k_encode_slices"] / max(1, m["n_enc"])
    k_dec = prof["k_decode_slices"] / max(1, m["n_dec"])
    dom, dom_ms = ("k_decode_slices", k_dec) if k_dec >= k_enc else ("k_encode_slices", k_enc)
    algo = (m["raw_bytes"] + m["container_bytes"]) // S
    achieved = algo / (dom_ms * 1e-3) / 1e9
    traffic, valu, source = profile_numbers(F, args.tile_w, args.tile_h, planar, args.content, S, dom)
    clock = clock_numbers()
    isolated = None
    if m["iso"]:
        iso_ms = m["iso"][dom] / max(1, m["iso_enc"] if dom == "k_encode_slices" else m["iso_dec"])
        isolated = {"avg_launch_ms": round(iso_ms, 4), "achieved": round(algo / (iso_ms * 1e-3) / 1e9, 3),
                    "frac": round(algo / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                    "note": "the same launches with one pipeline at a time on the GPU, outside the timed region"}
        if valu:
            isolated["valu_issue_frac"] = round(valu / (iso_ms * 1e-3) / VALU_PEAK, 4)
    valu_issue = None
    if valu:
        valu_issue = {"valu_wave_insts_per_launch": valu, "source": source, "achieved": round(valu / (dom_ms * 1e-3) / 1e9, 2), "peak": VALU_PEAK / 1e9,
                      "unit": "G wave-instructions/s", "frac": round(valu / (dom_ms * 1e-3) / VALU_PEAK, 4),
                      "note": "VALU wave-instructions of one launch (committed rocprofv3 SQ_INSTS_VALU pass of this configuration) / live launch duration, "
                              "against 1024 SIMDs x 2.4 GHz / 2; live launches share the SIMDs with the other pipelines' kernels, isolated.valu_issue_frac is the kernel alone"}
        if clock:  # the same against the clock the kernel was MEASURED to run at (s_memtime / s_memrealtime in a diagnostic build)
            ghz = clock.get(dom, {}).get("clock_ghz")
            if ghz:
                true_peak = 1024 * ghz * 1e9 / 2
                valu_issue.update({"clock_ghz": ghz, "clock_source": clock.get("source"), "peak_at_clock": round(true_peak / 1e9, 1),
                                   "frac_at_clock": round(valu / (dom_ms * 1e-3) / true_peak, 4)})
                if isolated and "valu_issue_frac" in isolated:
                    isolated["valu_issue_frac_at_clock"] = round(isolated["valu_issue_frac"] * 2.4 / ghz, 4)
    return {
        "metric": "encode+decode MPix/s on 4K RGB8, bit-exact",
        "value": round(m["mpix"], 2),
        "unit": "MPix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(m["dt"] / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {
            "workload": f"C3 3840x2160 RGB8 {args.content} ({'std::mt19937 noise' if args.content == 'g3' else args.content}), "
                        f"{F} frames/step/GPU resident in HBM, sliced container: {args.tile_w}x{args.tile_h} tiles, "
                        f"{'per-channel planes' if planar else 'channels interleaved'}, {m['n_slices'] // F} slices/frame, {S} stream(s)"
                        + ("" if world == 1 else f"; {world} GPUs, {F} frames each per step (frames are independent objects: dealt to the ranks, no data-path collective)"),
            "frames_per_step_per_gpu": F, "tile_w": args.tile_w, "tile_h": args.tile_h, "planar": planar, "content": args.content,
            "slices_per_frame": m["n_slices"] // F, "streams": S,
            "compression_ratio": round(m["ratio"], 4),
            "parallelism": "one GPU" if world == 1 else f"dp{world}: one process per GPU, frames sharded over the ranks; config 4 (tiles of one image sharded, RCCL exchange) is the c4_sharded key",
        },
        "roofline": {
            "bound": "valu_issue", "hbm_bound_as_contract_asks": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": source,
            "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(dom_ms, 4), "isolated": isolated,
            "limiter": "valu_issue",
            "valu_issue": valu_issue,
            "note": "achieved/peak/frac are the contract's HBM figures (algorithmic bytes of one coding direction per launch / live launch duration vs 8 TB/s); "
                    "`bound` names what limits the kernel: VALU issue along its serial dependency chain (one lane per slice), DESIGN.md 4 -- its own roofline is "
                    "valu_issue (frac_at_clock = against the measured in-kernel clock); launch durations are measured while the other stream(s)' pipelines run beside them",
        },
        "kernel_ms_per_step": {k: round(v / max(1, args.steps), 4) for k, v in prof.items()},
    }


def clock_numbers():
    """in-kernel shader clock of the slice kernels, measured with a diagnostic build (tools/clock_probe.py: delta s_memtime /
    delta s_memrealtime x 100 MHz, median over wavefronts) and committed under profiles/; None when there is no such file"""
    for name in ("r03_inkernel_clock.json",):
        try:
            cj = json.load(open(os.path.join(ROOT, "profiles", name)))
            cj["source"] = "profiles/" + name
            return cj
        except (OSError, ValueError):
            pass
    return None


class quiet_stdout:
    """RCCL prints a version banner on STDOUT when a communicator comes up; stdout carries exactly one JSON line, so the file
    descriptor points at stderr while communicators are created (init, new_group, their first collective)"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


class Watchdog:
    """The secondary legs of the N > 1 line run behind this: if they have not finished `seconds` after arm(), rank 0 prints
    the line with what it has (the headline is complete by then; `lost_legs` names what is missing) and every rank leaves with
    EXIT CODE 3 -- a leg that hangs in a collective, or a rank that died in one, costs that leg, and a harness that looks at the
    exit code alone does not take the run for a clean one."""

    def __init__(self, rank, res):
        import threading

        self.rank, self.res, self.lock, self.done = rank, res, threading.Lock(), False
        self.timer = None

    def bail(self, what, why):
        """give up on the remaining legs NOW: rank 0 prints the line as it stands, the process leaves with exit code 3"""
        with self.lock:
            if self.done:
                return
            self.done = True
            print(f"bench: rank {self.rank}: giving up on '{what}': {why}", file=sys.stderr, flush=True)
            if self.rank == 0 and self.res is not None:
                self.res.setdefault("lost_legs", []).append({"leg": what, "why": why[:300]})
                print(json.dumps(self.res), flush=True)
        sys.stdout.flush()
        time.sleep(0 if self.rank == 0 else 3.0)
        os._exit(3)

    def arm(self, seconds, what):
        import threading

        self.timer = threading.Timer(seconds, self.bail, args=(what, f"not finished after {seconds:.0f} s (a rank is stuck or gone)"))
        self.timer.daemon = True
        self.timer.start()

    def finish(self):
        """True when the caller may print (the watchdog has not already done so)"""
        with self.lock:
            if self.done:
                return False
            self.done = True
        if self.timer:
            self.timer.cancel()
        return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=32, help="4K frames per step per GPU")
    ap.add_argument("--content", default="g3", choices=["g3", "g2", "mid", "nat"])
    ap.add_argument("--tile-w", type=int, default=480)
    ap.add_argument("--tile-h", type=int, default=1)
    ap.add_argument("--interleaved", action="store_true", help="channels interleaved in one slice instead of per-channel planes")
    ap.add_argument("--streams", type=int, default=3, help="split the frames of a step over this many HIP streams (codec objects)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated", action="store_true", help="skip the extra one-pipeline-at-a-time launches behind the timed region (profiler runs)")
    ap.add_argument("--no-also", action="store_true", help="only the headline workload (profiler runs, sweeps)")
    ap.add_argument("--also-only", default="", help="comma-separated subset of the `also` legs: full,contents,tiles,latency,legacy,c5,c4,inproc,c2 (profiling)")
    ap.add_argument("--c4-parts", type=int, default=0, help="part batches (ShardedCodec objects on their own HIP streams) of the config-4 step; 0 = three where the image count allows")
    ap.add_argument("--c4-images", type=int, default=24, help="8192x8192 images per step of the sharded (config 4) workload (fixed total: strong scaling)")
    ap.add_argument("--c4-size", type=int, default=8192, help="side of the square config-4 images (8192 = BASELINE; smaller only for rehearsals)")
    ap.add_argument("--c4-tile-w", type=int, default=512)
    ap.add_argument("--c4-tile-h", type=int, default=1)
    ap.add_argument("--legs-timeout", type=float, default=300.0, help="N > 1: seconds the secondary legs (config 4, config 5) may take before the line goes out without them")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="rehearsal of the N > 1 code path on a one-GPU box: all ranks use cuda:0 and exchange over gloo (RCCL refuses two ranks on one device); the numbers mean nothing")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)  # (does not return)

    import numpy as np
    import torch
    import torch.distributed as dist

    import llcomp_amd as mi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    Hooks.measure, Hooks.c5_stream, Hooks.inprocess = measure, c5_stream, c4_inprocess
    if os.environ.get("LLCOMP_BENCH_STANDIN"):  # tests/test_bench_world8.py: CPU stand-ins for the coders, gloo for RCCL (docstring)
        import importlib

        Hooks.standin = importlib.import_module(os.environ["LLCOMP_BENCH_STANDIN"])
        Hooks.standin.install(Hooks)
        if world < 2:
            raise SystemExit("LLCOMP_BENCH_STANDIN rehearses the N > 1 bookkeeping only")
    elif not torch.cuda.is_available() or mi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: llcomp_amd has no CPU path")
    if Hooks.device == "cuda":
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local_rank)
    DEV = Hooks.device
    if world == 1:  # the sharded workload runs under torch.distributed at every N, so the N = 1 point is the same code
        import socket

        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        sk.close()
    with quiet_stdout():
        if args.rehearse_one_gpu or Hooks.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        warm = torch.zeros(1, device=DEV)
        dist.all_reduce(warm)
        Hooks.sync()
    barrier = dist.barrier if world > 1 else None

    planar = not args.interleaved
    F = args.frames

    if world > 1:
        # ---- headline: BASELINE's metric on config 3, frames dealt to the ranks (same workload per GPU as at N = 1) ----
        m = Hooks.measure(make_frames(args.content, F, rank, distinct=min(F, 8)), args.tile_w, args.tile_h, planar, args.streams, args.steps, args.warmup,
                          local_rank, barrier=barrier)
        t = torch.tensor([m["dt"]], dtype=torch.float64, device=DEV)
        every = torch.zeros(world, dtype=torch.float64, device=DEV)
        every[rank] = m["dt"]
        dist.all_reduce(every)  # every rank's own time for the same K steps (the line's value uses the slowest)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_all = float(t.item())
        res = None
        if rank == 0:
            res = headline(args, dict(m, dt=dt_all, mpix=world * m["F"] * W4K * H4K * m["steps"] / dt_all / 1e6), world, planar)
            res["ranks_seen"] = dist.get_world_size()
            res["per_gpu_value"] = round(res["value"] / world, 2)
            # what each rank did on its own GPU in the same timed region (MPix/s): the spread says whether one GPU held the others up
            res["per_rank_one_gpu_value"] = [round(m["F"] * W4K * H4K * m["steps"] / float(e) / 1e6, 1) for e in every.tolist()]
            try:
                res["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # noqa: BLE001
                res["rccl_version"] = f"unknown ({type(e).__name__})"
            res["collective_backend"] = dist.get_backend()
        if dist.get_world_size() != args.gpus:  # (cannot happen behind the check above; a line from fewer ranks must not look green)
            raise SystemExit(4)
        dog = Watchdog(rank, res)
        dog.arm(args.legs_timeout, "c4_sharded + c5_replica_pcie + c4_inprocess_devices")
        t_legs = time.perf_counter()
        empty_cache = torch.cuda.empty_cache if DEV == "cuda" else (lambda: None)
        # ---- BASELINE config 4: strong scaling of sharded 8192^2 images, gather + scatter inside the timed region -------
        size, B = args.c4_size, args.c4_images
        n4 = max(3, args.steps // 2)
        c4 = {}
        try:
            # rank 0 alone = the one-GPU point of the config-4 curve.  The subgroup is made HERE, behind the watchdog and
            # after the headline is complete: a communicator split that the runtime refuses costs this leg, not the line
            with quiet_stdout():
                solo = dist.new_group(ranks=[0])
                if rank == 0:
                    dist.all_reduce(warm, group=solo)
                    Hooks.sync()
            one = None
            if rank == 0:  # the one-GPU point of the same batch, same code path, on a one-rank subgroup
                dt1, pay1 = c4_run(B, size, args.c4_tile_w, args.c4_tile_h, n4, 3, local_rank, 1, 0, parts=args.c4_parts, group=solo)
                mi.trim()
                empty_cache()
                one = B * size * size * n4 / dt1 / 1e6
            dist.barrier()
            dtn, payload = c4_run(B, size, args.c4_tile_w, args.c4_tile_h, n4, 3, local_rank, world, rank, parts=args.c4_parts)
            mi.trim()
            empty_cache()
            if rank == 0:
                raw = B * size * size * 3
                vn = B * size * size * n4 / dtn / 1e6
                c4 = {"value": round(vn, 1), "unit": "MPix/s", "scaling": "strong", "ms_per_step": round(dtn / n4 * 1e3, 3), "steps": n4,
                      "one_gpu_value": round(one, 1), "one_gpu_ms_per_step": round(dt1 / n4 * 1e3, 3), "scaling_vs_one_gpu": round(vn / one, 3),
                      "ranks_seen": dist.get_world_size(), "images_per_step": B, "compression_ratio": round(raw / payload, 4),
                      "hbm_roofline_frac": round(2 * (raw + payload) * n4 / dtn / 1e9 / (HBM_PEAK_GBS * world), 6),
                      **getattr(c4_run, "last_detail", {}),
                      "workload": f"C4 {B} x {size}x{size} RGB8 uniform noise per step (fixed total), every image sharded over {world} GPUs by interleaved chunks of "
                                  f"tile rows; sliced container {args.c4_tile_w}x{args.c4_tile_h} tiles, per-channel planes; per step: local encode, slice-table "
                                  f"all_gather, variable-size all-to-all of the packed payloads over RCCL (image b is gathered on rank b % {world}), device "
                                  f"concatenator -> {B} complete containers spread over the ranks; then table all_gather, all-to-all back, local decode (decoded "
                                  f"rows stay on their ranks); one_gpu_value = the same batch through the same code on rank 0 alone, same run"}
        except Exception as e:  # noqa: BLE001  (the other ranks are very likely stuck in a collective now: their watchdogs end them)
            dog.bail("c4_sharded", f"rank {rank}: {type(e).__name__}: {e}")
        if rank == 0:
            res["c4_sharded"] = c4
        # ---- BASELINE config 5 in replica mode: every rank streams its own share of the 64 frames
        # host -> GPU -> host -> GPU -> host through llcomp_mi_stream_* over its own PCIe link, all ranks at the same time.
        # A rank whose leg fails still takes part in the reductions (nobody is left waiting) and the key says so.
        c5 = {"value": 0.0, "frames": 0, "seconds_first_submit_to_last_result": 0.0}
        ok = 1.0
        try:
            dist.barrier()
            c5 = Hooks.c5_stream(make_frames(args.content, max(16, 64 // world), rank, distinct=8), args.tile_w, args.tile_h, planar)
        except Exception as e:  # noqa: BLE001
            ok = 0.0
            print(f"bench: rank {rank}: config-5 replica leg failed: {e!r}", file=sys.stderr, flush=True)
        v5 = torch.tensor([c5["value"], ok, float(c5["frames"])], dtype=torch.float64, device=DEV)
        dist.all_reduce(v5, op=dist.ReduceOp.SUM)
        t5 = torch.tensor([c5["seconds_first_submit_to_last_result"]], dtype=torch.float64, device=DEV)
        dist.all_reduce(t5, op=dist.ReduceOp.MAX)
        if rank == 0:
            whole = float(v5[2].item()) * W4K * H4K / 1e6 / float(t5.item()) if float(t5.item()) > 0 else 0.0
            res["c5_replica_pcie"] = {"value": round(whole, 1), "unit": "MPix/s", "ranks_ok": int(round(float(v5[1].item()))), "scaling": "weak",
                                      "value_is": "all ranks' frames / the slowest rank's time from its first submit to its last result (ramp-up included; the ranks start together behind a barrier)",
                                      "sum_of_steady_state_rates": round(float(v5[0].item()), 1),
                                      "frames_per_rank": 4 * max(16, 64 // world), "rank0": c5,
                                      "workload": f"C5: every rank streams its own {max(16, 64 // world)} 4K frames (four passes) host -> GPU -> host -> GPU -> host "
                                                  f"through llcomp_mi_stream_*, all {world} ranks at once; sum of the ranks' steady-state rates, PCIe inclusive"}
        # ---- BASELINE config 4 from ONE process: rank 0 drives all N GPUs through the C ABI's device list (llcomp_mi_opts.devices) while
        # the other ranks wait on the STORE (a collective would park a spinning RCCL kernel on the very GPUs rank 0 is about to use).
        # Their GPUs are idle by now (lanes and torch's cache released); rank 0 opens a HIP context on each of them.
        inproc, hung = None, False
        try:
            import datetime
            import threading

            mi.trim()
            empty_cache()
            dist.barrier()
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                # on a thread with a deadline of its own: this leg is the newest code on the newest ground (one process on N GPUs
                # that other processes hold contexts on) -- if it fails or hangs it costs ITS key, not the line and not the exit code
                box = {}

                def work():
                    try:
                        # (a launcher that narrows every rank's view to its own GPU leaves rank 0 fewer ordinals than ranks: it then drives what it sees)
                        seen = mi.device_count() if Hooks.device == "cuda" else world
                        # (the one-GPU rehearsal repeats its one ordinal, so that the list path -- a lane and a thread per part -- runs beside the waiting ranks)
                        devs = [local_rank] * min(world, 3) if args.rehearse_one_gpu else list(range(max(1, min(world, seen))))
                        box["out"] = Hooks.inprocess(devs, images=max(2, min(4, args.c4_images)), size=size, tile_w=args.c4_tile_w, tile_h=args.c4_tile_h)
                    except BaseException as e:  # noqa: BLE001
                        box["err"] = f"{type(e).__name__}: {e}"[:300]

                # (its deadline also stays clear of the watchdog's: what is left of --legs-timeout minus 30 s)
                budget = min(150.0, args.legs_timeout - (time.perf_counter() - t_legs) - 30.0)
                th = threading.Thread(target=work, daemon=True)
                if budget >= 20.0:
                    th.start()
                    th.join(budget)
                else:
                    box["err"] = "skipped: the legs before it left less than 50 s of --legs-timeout"
                if th.is_alive():
                    inproc, hung = {"failed": "not finished after its own deadline; the line goes out without it"}, True
                elif "err" in box:
                    inproc = {"failed": box["err"]}
                    print(f"bench: in-process device-list leg failed: {box['err']}", file=sys.stderr, flush=True)
                else:
                    inproc = dict(box["out"], ranks_waiting=world - 1)
                store.set("llcomp_bench_inprocess_done", "hung" if hung else "ok")
            else:
                store.wait(["llcomp_bench_inprocess_done"], datetime.timedelta(seconds=args.legs_timeout))
                hung = store.get("llcomp_bench_inprocess_done") == b"hung"  # (every rank skips the barrier then, and leaves the same way)
            if not hung:
                dist.barrier()
        except Exception as e:  # noqa: BLE001
            dog.bail("c4_inprocess_devices", f"rank {rank}: {type(e).__name__}: {e}")
        if rank == 0:
            res["c4_inprocess_devices"] = inproc
        if dog.finish() and rank == 0:
            print(json.dumps(res), flush=True)
        if hung:  # a HIP call of the abandoned leg may never return: leave without waiting for its thread (the line is out, exit code 0)
            sys.stdout.flush()
            os._exit(0)
        dist.destroy_process_group()
        return

    # ---- N = 1: BASELINE config 3 (headline) ------------------------------------------------------------------------
    # What the host link gives, measured FIRST: the rate a stream's copies get depends on what the process has done before
    # (the HIP runtime's copy-engine assignment: 57 / 57 / 97 GB/s in a fresh process, 57 / 30 / 57-80 on streams made after the
    # headline leg -- profiles/r03_c5_repeat.txt run 6), so the reference for the config-5 leg is taken while the process is young
    link0 = link_rate() if not args.no_also and (not args.also_only or "c5" in args.also_only.split(",")) else None
    frames_np = make_frames(args.content, F, rank)
    m = measure(frames_np, args.tile_w, args.tile_h, planar, args.streams, args.steps, args.warmup, local_rank, isolated=not args.no_isolated)
    res = headline(args, m, world, planar)
    # byte-level pin of the measured workload: frame 0 of the default batch is the golden vector's image (std::mt19937(1234)),
    # its container length must be the one the real reference's per-slice streams add up to (tests/golden, also hashed in
    # tests/test_gpu_stream.py::test_c3_4k_bench_slicing_golden)
    try:
        gold = [v for v in json.load(open(os.path.join(ROOT, "tests", "golden", "slice_payloads.json")))["vectors"]
                if (v["gen"], v["w"], v["tile_w"], v["tile_h"], v["planar"]) == (args.content, W4K, args.tile_w, args.tile_h, planar)]
        if gold and args.content in ("g3", "g2"):  # frame 0 of these two IS the golden vector's image (mid / nat use other seeds here)
            res["golden_pin"] = {"frame0_container_bytes": m["frame0_container"], "reference_bytes": gold[0]["container_len"],
                                 "frame0_container_fnv1a64": m["frame0_fnv"], "reference_fnv1a64": gold[0]["container_fnv1a64"],
                                 "match": m["frame0_container"] == gold[0]["container_len"] and m["frame0_fnv"] == gold[0]["container_fnv1a64"],
                                 "source": "tests/golden/slice_payloads.json (container assembled from the real reference's per-slice streams)"}
            assert res["golden_pin"]["match"], "frame 0's container differs from the reference's (length or FNV-1a-64)"
    except OSError:
        pass
    full_box = {}
    if not args.no_also:
        also = {}
        sub = max(3, args.steps // 3)
        t_also = time.perf_counter()
        only = set(x for x in args.also_only.split(",") if x)
        want = lambda leg: not only or leg in only  # noqa: E731
        legacy_box = {}

        def leg_full():
            # The headline's slicing (one-row slices) leaves llcomp's vertical context -- quant11 over three gradients, the two-row window, the
            # median of left / top / gradient -- out of the timed region: with no row above, the context collapses to 605 * quant5(L - l).
            # The same batch and pipelines through 64x64 planar tiles run the FULL model (and keep the reference's ratio): reported next to
            # `value`, as the LAST keys of the line.
            if args.tile_h != 1:
                return
            # Timed like `value`: K steps queued back to back between two device synchronisations (the kernels of step i + 1 start while
            # the tail of step i drains).  The second pass brackets every step with a synchronisation of its own and reports median and
            # range: what one isolated step costs, and how far single steps fall from each other.
            mf = measure(frames_np, 64, 64, True, args.streams, max(12, args.steps), 2, local_rank)
            bf = brief(mf)
            ms_ = measure(frames_np, 64, 64, True, args.streams, max(12, args.steps), 2, local_rank, per_step=True)
            bs = brief(ms_)
            bf["steps_synchronised_one_by_one"] = {k: bs[k] for k in ("value", "value_is", "ms_per_step", "ms_per_step_min_max", "value_min_max")}
            # the same contract figures for the full model's dominant kernel: algorithmic bytes of one direction per launch / its live duration
            sides_f = tile_sides(mf)
            kd, ke = sides_f["decode_ms_per_launch"]["k_decode_slices"], sides_f["encode_ms_per_launch"]["k_encode_slices"]
            dom_f, dom_ms_f = ("k_decode_slices", kd) if kd >= ke else ("k_encode_slices", ke)
            algo_f = (mf["raw_bytes"] + mf["container_bytes"]) // mf["S"]
            roof_f = {"bound": "hbm", "limiter": "random state-bank transactions (decode) / one wavefront's dependent chain per slice (few frames in flight)",
                      "kernel": dom_f, "achieved": round(algo_f / (dom_ms_f * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(algo_f / (dom_ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": algo_f, "avg_launch_ms": round(dom_ms_f, 4)}
            # HBM-side bytes of that kernel per launch: the committed PMC passes of THIS configuration (frames, pipelines, 64x64 planar,
            # content), looked up like the headline's (profiles/r06_full_model_traffic.json names the commit it was taken at)
            traffic_f, _valu_f, source_f = profile_numbers(F, 64, 64, True, args.content, mf["S"], dom_f)
            roof_f["traffic"], roof_f["traffic_source"] = traffic_f, source_f
            full_box["full_model"] = dict(bf, **sides_f, **cache_counters(mf), roofline=roof_f,
                                          workload=f"the headline's batch ({F} frames 4K {args.content}, {mf['S']} pipelines) in 64x64 planar tiles: "
                                                   f"{mf['n_slices'] // F} slices per frame, all five context terms and the median predictor live",
                                          vs_value=round(bf["value"] / res["value"], 4))
            full_box["value_full_model"] = bf["value"]

        def leg_inproc():  # BASELINE config 4 through the C ABI's device list, from this one process: {0,0} = two lanes on the one GPU
            also["c4_inprocess_devices"] = c4_inprocess([local_rank, local_rank], images=4, tile_w=args.c4_tile_w, tile_h=args.c4_tile_h)

        def leg_c5():  # three repetitions: the pipeline's steady state is sensitive to how the copies of the jobs fall over each other (tools/attic/c5_repeat.py)
            link = dict(link0, measured="at the start of the process")
            c5_stream(frames_np, args.tile_w, args.tile_h, planar, passes=2)  # untimed: first touch of the pinned buffers, lanes, code objects
            runs = sorted((c5_stream(frames_np, args.tile_w, args.tile_h, planar, link=link) for _ in range(3)), key=lambda r: r["value"])
            leg = dict(runs[1])  # the median run
            leg["runs_MPix_s"] = [r["value"] for r in runs]
            leg["spread"] = round((runs[-1]["value"] - runs[0]["value"]) / runs[-1]["value"], 3)
            leg["note"] = "median of three repetitions in this process; " + leg["note"]
            also["c5_stream_pcie"] = leg
            # the same frames through ONE pipeline object that deals its jobs over a device list ({0,0}: two pipelines of six slots on
            # the one GPU behind one object -- the same twelve slots as above --, ONE driving thread) -- config 5's "round-robin over
            # the GPUs" from one process
            dl = c5_stream(frames_np, args.tile_w, args.tile_h, planar, depth=6, pipelines=1, encodes_in_flight=4, passes=4, link=link, devices=[local_rank, local_rank])
            dl["devices"] = [local_rank, local_rank]
            dl["note"] = "one llcomp_mi_stream_create_multi object over {0,0}, one driving thread; " + dl["note"]
            also["c5_stream_device_list"] = dl

        def leg_c4():
            n4 = max(4, sub // 2)
            dt4, pay4 = c4_run(args.c4_images, 8192, args.c4_tile_w, args.c4_tile_h, n4, 3, local_rank, world, rank, parts=args.c4_parts)
            mi.trim()  # the one-piece check went through a host-buffer call: give its cached lane (10 GB) back
            also["c4_sharded_one_gpu"] = {"value": round(args.c4_images * 8192 * 8192 * n4 / dt4 / 1e6, 1), "unit": "MPix/s",
                                          "ms_per_step": round(dt4 / n4 * 1e3, 3), "steps": n4, "images_per_step": args.c4_images,
                                          "compression_ratio": round(args.c4_images * 8192 * 8192 * 3 / pay4, 4),
                                          "workload": f"C4 {args.c4_images} x 8192x8192 RGB8 uniform noise, {args.c4_tile_w}x{args.c4_tile_h} planar, the sharded code path on 1 GPU",
                                          **getattr(c4_run, "last_detail", {})}

        def leg_contents():  # other contents at the default slicing (4 distinct frames, the rest rotations)
            for content in ("g2", "mid", "nat"):  # clean gradient, dithered gradient, photo-like (synth.py)
                if content != args.content:
                    also[f"{content}_default_slicing"] = brief(measure(make_frames(content, F, 0, distinct=4), args.tile_w, args.tile_h, planar, args.streams, sub, 1, local_rank),
                                                               workload=f"{F} frames 4K {content}, {args.tile_w}x{args.tile_h} planar")

        def leg_tiles():  # 2-D tiles keep vertical prediction (and the reference's ratio): the mode that runs llcomp's full context model
            for content in ("nat", "mid", "g3"):
                for frames, streams in ((16, 2), (48, 3)):  # round 3's leg (8 frames per pipeline: latency-bound) and the throughput configuration
                    fr = frames_np[:frames] if content == args.content and frames <= len(frames_np) else make_frames(content, frames, 0, distinct=4)
                    m2 = measure(fr, 64, 64, True, streams, max(12, sub), 2, local_rank, per_step=True)
                    samples = 2 * frames * W4K * H4K * C4K * m2["steps"]
                    extra = {}
                    if content in TILE_HBM_BYTES_PER_SAMPLE and frames == 16:  # HBM-side traffic from the committed PMC passes of 16 frames (one pipeline); the
                        # 48-frame leg's traffic is NOT extrapolated from it -- its configuration at the headline's batch size has its own PMC passes (full_model)
                        t = TILE_HBM_BYTES_PER_SAMPLE[content]
                        gbs = (t["encode_all_kernels"] + t["decode_all_kernels"]) / 2 * samples / m2["dt"] / 1e9
                        extra = {"hbm_bytes_per_sample": t, "hbm_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 3), "hbm_source": TILE_HBM_SOURCE}
                    also[f"{content}_tiles64x64_{frames}frames"] = brief(
                        m2, workload=f"{frames} frames 4K {content}, 64x64 planar tiles, {streams} pipelines; encoder: state snapshot pass + sequential coder (no state "
                                     f"table), decoder: per-slice state tables in HBM (generation-tagged)",
                        samples_per_s=round(samples / m2["dt"] / 1e9, 2), **tile_sides(m2), **cache_counters(m2), **extra,
                        note="the decoder is bound by random state-bank transactions (the next context needs the sample just decoded), cut by a per-lane bank "
                             "cache in LDS; the encoder knows every context in advance and streams its states")

        def leg_latency():  # latency of ONE frame: at the throughput slicing, and at the width the library suggests for one frame per call
            m1 = measure(frames_np[:1], args.tile_w, args.tile_h, planar, 1, 20, 2, local_rank)
            also["one_frame_latency"] = brief(m1, workload=f"1 frame 4K {args.content}, {args.tile_w}x{args.tile_h} planar", ms_enc_plus_dec=round(m1["dt"] / m1["steps"] * 1e3, 3))
            tw1 = mi.suggest_tile_w(1, W4K, H4K, C4K, planar)
            m1s = measure(frames_np[:1], tw1, 1, planar, 1, 20, 2, local_rank)
            also["one_frame_latency_suggested_tile_w"] = brief(m1s, workload=f"1 frame 4K {args.content}, {tw1}x1 planar (llcomp_mi_suggest_tile_w for one frame per call)",
                                                               tile_w=tw1, ms_enc_plus_dec=round(m1s["dt"] / m1s["steps"] * 1e3, 3))

        def leg_c2():  # BASELINE config 2: 1920x1080 RGB8 noise, ONE SLICE PER ROW, one frame and a batch of 32
            c2 = make_frames("g3", 32, 0, w=1920, h=1080, c=3, distinct=8)
            out = {}
            for label, pl in (("interleaved", False), ("planar", True)):
                one = measure(c2[:1], 1920, 1, pl, 1, 20, 2, local_rank)
                many = measure(c2, 1920, 1, pl, 2, sub, 1, local_rank)
                out[label] = {"one_frame": brief(one, ms_enc_plus_dec=round(one["dt"] / one["steps"] * 1e3, 3), slices=one["n_slices"]),
                              "batch_32_frames": brief(many, slices=many["n_slices"])}
            out["workload"] = ("C2 1920x1080 RGB8 std::mt19937 noise, one slice per row (1920x1): 1080 slices per frame with the channels interleaved "
                               "(payload of a slice == reference stream of that row), 3240 as per-channel planes; one frame = 17 / 51 wavefronts on 1024 SIMDs")
            also["c2_rows_1080p"] = out

        def leg_legacy():  # the reference's own format in bulk: 512 whole-image streams (one lane each) of 256x256 RGB8
            leg = make_frames("mid", 512, 0, w=256, h=256, c=3, distinct=16)
            ml = measure(leg, 256, 256, False, 1, 1, 1, local_rank)
            also["legacy_streams_batched"] = brief(ml, workload="512 frames 256x256 RGB8 mid, one whole-image stream each (payload == reference stream), one GPU lane per stream")
            legacy_box["frame"] = leg[0].copy()

        # The two big legs go first (BASELINE config 5 through the streaming pipeline, PCIe inclusive -- 3.9 instead of
        # 4.8-5.2 GPix/s when its pinned buffers are allocated behind config 4's 150 GB; then BASELINE config 4 on one GPU =
        # the N = 1 point of the strong-scaling curve), before the allocate / free cycles of the others fragment HBM.  A secondary leg that fails is
        # reported as such; it never costs the headline line.
        for name, fn in (("c5", leg_c5), ("full", leg_full), ("c4", leg_c4), ("inproc", leg_inproc), ("contents", leg_contents), ("tiles", leg_tiles), ("latency", leg_latency), ("c2", leg_c2), ("legacy", leg_legacy)):
            if not want(name):
                continue
            try:
                fn()
            except Exception as e:  # noqa: BLE001
                also[f"{name}_failed"] = f"{type(e).__name__}: {e}"[:300]
                print(f"bench: also leg {name} failed: {e!r}", file=sys.stderr, flush=True)
                try:
                    torch.cuda.synchronize()
                    mi.trim()
                    torch.cuda.empty_cache()
                except Exception:  # noqa: BLE001
                    pass
        also["seconds"] = round(time.perf_counter() - t_also, 1)
        res["also"] = also
    # The CPU legs run LAST: seconds of single-thread coding churn gigabytes of host memory, and pinned buffers allocated
    # after that are slower to DMA (measured: the PCIe-inclusive C5 leg drops from 4.7 to 3.8 GPix/s when it runs behind them).
    if not args.no_cpu_baseline:  # the CPU reference is timed at N=1 only
        res["cpu_baseline"] = cpu_baseline(frames_np[0], f"3840x2160 RGB8 {args.content}", args.tile_w, args.tile_h, planar, same_slicing=True)
        res["speedup_vs_cpu_baseline"] = round(m["mpix"] / res["cpu_baseline"]["value"], 1)
        if "same_slicing_port" in res["cpu_baseline"]:
            res["speedup_vs_cpu_same_slicing"] = round(m["mpix"] / res["cpu_baseline"]["same_slicing_port"]["value"], 1)
            res["speedup_vs_cpu_same_slicing_all_cores"] = round(m["mpix"] / res["cpu_baseline"]["same_slicing_port_all_cores"]["value"], 1)
        res["cpu_baseline"]["note"] = ("the reference codes one whole-image stream; the GPU figure is on independent slices (ratio in config.compression_ratio vs "
                                       "the whole-image ratio in `sample`), so speedup_vs_cpu_baseline is throughput at unequal compression; same_slicing_port / "
                                       "speedup_vs_cpu_same_slicing is the like-for-like figure (identical container bytes)")

        if not args.no_also and "legacy_streams_batched" in res.get("also", {}):
            res["also"]["legacy_streams_batched"]["cpu_reference"] = cpu_baseline(legacy_box["frame"], "256x256 RGB8 mid", 256, 256, False)
    if not args.no_also:  # the real-model figures go LAST: a record that keeps the tail of the line keeps them
        for k in ("full_model", "value_full_model"):
            if k in full_box:
                res[k] = full_box[k]
    print(json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
