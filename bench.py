#!/usr/bin/env python3
"""Benchmark of the PIV cross-correlation hot path (driver contract).

    python bench.py --gpus N --steps K --warmup W [--config 1|2] [--precision exact|f64|fast|reference]

Metric (BASELINE.json): image-pairs/sec at 4 MP, wind=64 ov=32, 2-pass CWS x2.0.

--config 1 (default, the headline; BASELINE.json configs[1]): synthetic 2048x2048 pairs, batch = 256
  pairs resident in HBM per GPU.  A step = one pass of the whole hot path (pass 1 + predictor + CWS
  pass 2: every kernel of tpiv_plan_run) over that batch.  N > 1: every rank owns its own batch
  (weak scaling, no data-path collective) and ONE RCCL gather of the (u, v) fields of the last step
  onto rank 0 closes the timed region.
--config 2 (BASELINE.json configs[2]): a 4000-pair DWS stream in 500-pair shards, STRONG scaling: the
  4000 pairs are split over the ranks (PIVbackend.py:744-812 per pair, no state between pairs), every
  rank runs its pairs shard by shard, and the single gather of all (u, v) fields onto rank 0 is part
  of EVERY step.  A step = the whole stream once.

N > 1 without a launcher: `python bench.py --gpus N` starts the N ranks itself (a parent that never
touches the GPU runs `python -m torch.distributed.run --nproc-per-node N bench.py ...`); under
torchrun (WORLD_SIZE set) it is a rank.  The line is refused unless n_gpus == --gpus.

Precision.  `value` is measured at --precision exact (the default of the library): every transform of the shifted
passes in the reference's types (float32 with the float64 epilogue, B:249-257, B:382), and pass 1 -- float64 in the
reference, B:513-514 -- from EXACT integer correlation sums at the map cells that reach the result (a float32 FFT pass only
locates them inside an error band; undecided windows run the float64 transform; csrc/xcorr_exact.hip).  The fields are
within 1e-14 px of the float64 pass 1 (tests/test_gpu_exact.py) -- not a narrower type.  At N = 1 the same process then
times the same workload with pass 1 through the float64 FFT kernel (`f64_transform`, the headline of rounds 2-3) and
with pass 1 in float32 (`fast`), reported BESIDE the headline, never as `value`; then `end_to_end` (the generator through
post-validation and yield, tools/e2e_generator.py) and `cpu_baseline`.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel: HIP-event durations on the
launch stream during the timed steps, bound = "valu" (the path sits ~10x above the
HBM ridge), frac = max(B_alg / 8 TB/s, F_alg / 157.3 TFLOP/s) / t, the HBM view beside it, and --
at N = 1 -- PMC counters of THIS build collected live by rocprofv3 child runs of this script
(FETCH_SIZE, WRITE_SIZE, SQ_* in separate passes; --pmc file reads profiles/r05/pmc_counters.json
instead, --pmc off skips).  `cpu_baseline` (N = 1): the CPU oracle on the host cores.
"""
import argparse
import csv
import glob
import json
import math
import os
import shutil
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_VALU_PEAK_TFLOPS = 157.3   # vector fp32 peak, same guide
FP64_VALU_PEAK_TFLOPS = 78.6    # vector fp64 peak (half rate)
ISSUE_PEAK_G = 256 * 4 * 2.4 / 2.0      # wave-instructions/s: one per SIMD every 2 cycles at 2.4 GHz
PMC_FILE = os.path.join(ROOT, "profiles", "r05", "pmc_counters.json")


def alg_bytes(H, W, n_windows, first_pass):
    """both uint8 frames once per pass + 9 B/window out (+ 8 B/window predictor in)."""
    return 2 * H * W + n_windows * 9 + (0 if first_pass else n_windows * 8)


def alg_flops(ws, n_windows, cws):
    """three real 2-D FFTs + cross-spectrum per window (+ 14 flop/px/frame bilinear)."""
    f = n_windows * (7.5 * ws * ws * math.log2(ws * ws) + 6 * ws * (ws / 2 + 1))
    if cws:
        f += 2 * n_windows * ws * ws * 14
    return f


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 for config 1, 10 for config 2)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=1, choices=(1, 2))
    ap.add_argument("--precision", default="exact",
                    help="exact (default): pass 1 from exact integer correlation sums (64x64 windows), shifted passes float32; "
                         "f64: pass 1 through a float64 FFT like the reference (PIVbackend.py:513-514); "
                         "reference: the same with the reference's operation order in the CWS sampling; fast: pass 1 in float32 too")
    ap.add_argument("--no-gpu-sampling", action="store_true", help="do not sample the GPU clock / power from sysfs during the timed steps")
    ap.add_argument("--no-fast", action="store_true", help="N = 1: skip the float64-FFT and all-float32 runs reported beside the headline")
    ap.add_argument("--no-e2e", action="store_true", help="N = 1: skip the end_to_end block (generator rates)")
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU per launch (config 1: 256; config 2: shard 500)")
    ap.add_argument("--stream", type=int, default=4000, help="config 2: pairs in the stream (all ranks together)")
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--ws", type=int, default=64)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--mode", default=None, help="CWS (config 1) / DWS (config 2)")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic pairs to render (0 = a whole batch)")
    ap.add_argument("--pmc", default="live", choices=("live", "file", "off"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e", action="store_true",
                    help="end-to-end generator mode instead of the kernel-path bench: ResidentPIV / OfflinePIV.batched incl. "
                         "device post-validation, counted host fallbacks, BMP ingest (one JSON line, N = 1)")
    ap.add_argument("--e2e-pairs", type=int, default=256)
    ap.add_argument("--e2e-block-pairs", type=int, default=128, help="pairs per case of the end_to_end block of the default run")
    ap.add_argument("--fill-workers", type=int, default=8)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if not all(p_ in ("exact", "f64", "fast", "reference") for p_ in a.precision.split(",")):
        ap.error("--precision must be exact, f64, fast or reference")
    if a.mode is None:
        a.mode = "CWS" if a.config == 1 else "DWS"
    if a.batch is None:
        a.batch = 256 if a.config == 1 else 500
    if a.steps is None:
        a.steps = 200 if a.config == 1 else 10
    if a.warmup is None:
        a.warmup = 5 if a.config == 1 else 1
    return a


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process stays off the GPU and starts the
    N ranks as children through torch.distributed.run (never re-exec a process that touched HIP)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle (a port of the reference's algorithm) on the host cores
# ---------------------------------------------------------------------------------------------
def cpu_baseline(H, W, ws, ov, n_pass, mode, n_kernels=8, n_e2e=4):
    """Bounded sample of the SAME workload through oracle/piv_oracle.py (numpy + torch-CPU FFT: pass 1 in float64,
    shifted passes float32 like the reference): `n_kernels` pairs through the passes alone (the scope of `value`),
    `n_e2e` pairs end to end like OfflinePIV.__call__ incl. post-validation -- about 25 s on the box's host cores."""
    import torch
    from oracle import piv_oracle as O
    from torchpiv_amd import synth
    threads = torch.get_num_threads()
    pairs = [synth.make_pair(H, W, 900 + i, noise=2.0) for i in range(4)]
    pairs = [(a.numpy().copy(), b.numpy().copy()) for a, b in pairs]
    for a, b in pairs:                     # a few dead windows: the pair has invalid vectors, so the hole fill runs
        a[300:340, 500:540] = 0            # (a clean pair is DROPPED by the reference's quirk before any fill, B:300-304)
        b[300:340, 500:540] = 0
        a[1200:1240, 900:940] = 0
        b[1200:1240, 900:940] = 0

    def kernels_only(a, b):
        u, v, x, y, val = O.pass1(a, b, ws, ov, validate=True)
        w, o = ws, ov
        for _ in range(n_pass - 1):
            w, o = w // 2, o // 2
            u, v, x, y, val = O.ITER[mode](a.shape, w, o)(a, b, x, y, u, v, val)
        return u, v, val

    kernels_only(*pairs[0])                      # warm-up (MKL plans, page faults)
    t0 = time.perf_counter()
    for k in range(n_kernels):
        kernels_only(*pairs[k % len(pairs)])
    dt = time.perf_counter() - t0
    # end to end like OfflinePIV.__call__ (B:873-901): the passes plus NaN-out, border interpolation,
    # Delaunay hole fill, flip and scaling (frames already decoded, as for the GPU figure)
    t1 = time.perf_counter()
    ref_out = list(O.offline_piv([pairs[k % len(pairs)] for k in range(n_e2e)], ws, ov, multipass=n_pass, mode=mode))
    dt2 = time.perf_counter() - t1
    n_yield = len(ref_out)
    # ... and the same pairs through the drop-in's generator on the GPU, compared tuple by tuple: a live parity datum at
    # the BASELINE size in the driver's own record (reported, never gating; the gates are the -m gpu tests)
    check = None
    try:
        import numpy as np
        import torchpiv_amd as T
        A = torch.stack([torch.from_numpy(pairs[k % len(pairs)][0]) for k in range(n_e2e)]).cuda()
        B = torch.stack([torch.from_numpy(pairs[k % len(pairs)][1]) for k in range(n_e2e)]).cuda()
        piv = T.ResidentPIV(A, B, ws, ov, multipass=n_pass, multipass_mode=mode)
        got = list(piv())
        piv.close()
        check = {"gpu_yielded": len(got), "oracle_yielded": n_yield}
        if len(got) == n_yield and n_yield:
            # (the tuples carry u * scale / dt * 1000 with scale = dt = 1: back to pixels)
            d_uv = max(float(np.nanmax(np.abs(g_[k_] - r_[k_]))) for g_, r_ in zip(got, ref_out) for k_ in (2, 3)) / 1000.0
            d_xy = max(float(np.abs(g_[k_] - r_[k_]).max()) for g_, r_ in zip(got, ref_out) for k_ in (0, 1))
            cells = sum(g_[2].size for g_ in got)
            off = sum(int(((np.abs(g_[2] - r_[2]) > 1.0) | (np.abs(g_[3] - r_[3]) > 1.0)).sum()) for g_, r_ in zip(got, ref_out))
            check.update({"max_abs_diff_uv_px": d_uv, "max_abs_diff_xy": d_xy, "cells": cells, "cells_beyond_1e-3_px": off,
                          "nan_pattern_equal": all(np.array_equal(np.isnan(g_[2]), np.isnan(r_[2])) for g_, r_ in zip(got, ref_out))})
    except Exception as exc:            # noqa: BLE001 -- a report, not a gate
        check = {"error": repr(exc)}
    return {"value": n_kernels / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "end_to_end": {"value": n_e2e / dt2, "unit": "pairs/s", "pairs": n_e2e, "yielded": n_yield,
                           "checked_against_the_gpu_generator": check,
                           "what": "oracle offline_piv on resident frames: passes + NaN-out + border interpolation + "
                                   "Delaunay hole fill + flip/scale (PIVbackend.py:873-901)"},
            "sample": f"{n_kernels} pairs (kernels only) + {n_e2e} pairs (end to end) of the same {H}x{W} {n_pass}-pass {mode} "
                      f"workload (noise 2, two dead spots per frame), oracle/piv_oracle.py (numpy + torch-CPU FFT, pass 1 in "
                      f"float64), no file I/O; survey-container figure for the reference itself: 0.24 pairs/s on 8 threads "
                      f"(BASELINE.md)"}


# ---------------------------------------------------------------------------------------------
# PMC counters of the dominant kernels: rocprofv3 child runs of this script (N = 1 only)
# ---------------------------------------------------------------------------------------------
PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"]]


def collect_pmc_live(args, precisions, timeout_s=170):
    """{kernel name: {counter: mean per launch}} or None.  One rocprofv3 child per counter group (FETCH_SIZE and
    WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots'); the program itself follows `--`
    (no env/bash hop: the profiler's library initialises the GPU).  The child runs 2 plain steps at every
    precision in `precisions`, so one set of passes covers the kernels of the headline AND of the fast run."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    out = {}
    t_end = time.time() + timeout_s
    tmp = tempfile.mkdtemp(prefix="tpiv_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, ctrs in enumerate(PMC_PASSES):
            d = os.path.join(tmp, f"p{i}")
            cmd = [exe, "--pmc", *ctrs, "--kernel-include-regex", "xcorr", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--pmc", "off", "--no-cpu-baseline",
                   "--config", str(args.config), "--precision", ",".join(precisions), "--steps", "2", "--warmup", "1",
                   "--batch", str(args.batch), "--size", str(args.size), "--ws", str(args.ws),
                   "--passes", str(args.passes), "--mode", args.mode, "--distinct", "8", "--stream", str(args.batch)]
            left = t_end - time.time()
            if left < 20:
                return (out or None), "time budget of the PMC passes exhausted"
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                               timeout=left)
            if r.returncode != 0:
                return (out or None), f"rocprofv3 pass {ctrs} failed: {r.stderr.decode(errors='replace')[-300:]}"
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                agg, disp = {}, {}
                for row in csv.DictReader(open(f)):
                    k = row["Kernel_Name"]
                    agg.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
                    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    disp.setdefault(k, set()).add(row["Dispatch_Id"])
                for k, v in agg.items():
                    for c, val in v.items():
                        out.setdefault(k, {})[c] = val / len(disp[k])
    except subprocess.TimeoutExpired:
        return (out or None), "rocprofv3 child timed out"
    except Exception as exc:                              # never let the counters break the bench line
        return (out or None), f"{type(exc).__name__}: {exc}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, "live: rocprofv3 --pmc child runs of this bench.py (2 timed steps per precision each)"


def pmc_for(pmc, kernel_name):
    if not pmc:
        return None
    for k, v in pmc.items():
        if kernel_name in k:
            return v
    return None


def e2e_cases(n, workers, files=True, budget_s=150.0):
    """Generator rates end to end (tools/e2e_generator.py) at the default precision of the drop-in ("exact")."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import contextlib
    import io as _io
    import e2e_generator
    log = _io.StringIO()
    with contextlib.redirect_stdout(log):
        r = e2e_generator.main(n=n, workers=workers, files=files, budget_s=budget_s)
    return r, log.getvalue().strip().splitlines()


def e2e_block(r, log, n, workers):
    return {"unit": "pairs/s", "pairs_per_case": r.get("pairs_streamed", n), "pairs_per_resident_case": r.get("resident_pairs_streamed", n), "distinct_pairs": n, "batch": {"resident": r.get("resident_batch", 32), "files": r.get("files_batch", 32)}, "fill_workers": workers, "precision": r.get("precision", "exact"),
            "what": "generator end to end at 4 MP, wind=64 ov=32 2-pass CWS: passes + device post-validation + counted host "
                    "fallbacks + flip/scale + yield (ResidentPIV / OfflinePIV.batched)",
            "resident_clean_all_dropped": r.get("clean"), "resident_straight_runs": r.get("runs"),
            "resident_isolated_spots": r.get("spots"), "bmp_files_isolated_spots": r.get("files"),
            "bmp_files_generator_call": r.get("files_call"), "trials": r.get("trials"),
            "host_cpu_s_per_pair": r.get("host_cpu_s_per_pair"),
            "host": host_block(workers),
            "note": "every figure is the median of three runs of pairs_per_case (files) / pairs_per_resident_case pairs behind one untimed run (the distinct pairs "
                    "streamed repeatedly); host_cpu_s_per_pair = user + system CPU seconds of the process, its threads and "
                    "its fill-worker processes per pair over the timed runs",
            "post_validation": r.get("stats"), "log": log}


def host_block(workers=None):
    """What the host side of a rank may count on, and what 8 ranks at the single-GPU generator rate would need."""
    from torchpiv_amd import hostcfg
    b = hostcfg.host_budget(1)
    b8 = hostcfg.host_budget(8)
    return {"cores_available": b["cores"], "cgroup_cpu_quota": hostcfg.cgroup_cpu_quota(),
            "affinity_mask": len(os.sched_getaffinity(0)), "fill_workers_used": workers,
            "budget_1_rank": {k: b[k] for k in ("per_rank", "read_threads", "fill_workers")},
            "budget_8_ranks_on_this_node": {k: b8[k] for k in ("per_rank", "read_threads", "fill_workers")}}


def e2e_sharded(args):
    """`--e2e --gpus N`: the GENERATOR path sharded like dist.run_sharded -- every rank streams its own resident pairs
    (isolated dead spots: nearly every pair needs the host triangulation) through ResidentPIV.batched with reader / worker
    counts taken from its share of the node's cores, the finished fields stay on the device, ONE gather of (u, v) onto
    rank 0 closes the timed region.  One JSON line from rank 0; `value` = pairs of all ranks / wall time (max over ranks)."""
    from torchpiv_amd import hostcfg
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None:
        raise SystemExit(spawn_ranks(args))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    pinned = hostcfg.pin_rank(local, int(world_env))        # before any GPU call and before the workers are started
    import torch
    import torch.distributed as dist
    import torchpiv_amd as T
    from torchpiv_amd import dist as pdist
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_generator
    rank, world, local = pdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: {world} rank(s) running, --gpus {args.gpus} asked for")
    if os.environ.get("TPIV_DIST_BACKEND") == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    n, reps, batch = args.e2e_pairs, 4, 32
    A, B = e2e_generator.make_frames(n, args.size, args.size, "spots")
    piv = T.ResidentPIV(A, B, args.ws, args.ws // 2, multipass=args.passes, multipass_mode=args.mode, precision=args.precision)
    budget = piv.auto_host_config(world)
    order = list(range(n)) * reps
    ids_base = rank * n * reps

    def one_run():
        piv.device_out = True
        ids, uv = [], []
        for i, x, y, u, v in piv.batched(batch, indices=order):
            ids.append(ids_base + len(ids))
            uv.append(torch.stack([u, v]))
        f = torch.stack(uv) if uv else torch.zeros((0, 2, 0, 0), dtype=torch.float64, device=dev)
        out = pdist.gather_fields(torch.tensor(ids, dtype=torch.int64, device=dev), f)
        return len(ids), out

    one_run()                                   # untimed: plan creation, worker start, communicator
    piv.reset_stats()
    dist.barrier()
    torch.cuda.synchronize()
    c0 = hostcfg.tree_cpu_seconds()
    t0 = time.perf_counter()
    yielded, (ids_all, f_all) = one_run()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    cpu = hostcfg.tree_cpu_seconds() - c0
    backend = dist.get_backend()
    t = torch.tensor([elapsed, cpu, float(yielded)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    me = {"rank": rank, "local_rank": local, "device": torch.cuda.get_device_name(dev), "device_index": dev.index,
          "cores_pinned": len(pinned) if pinned else None, "post_validation": dict(piv.stats)}
    gathered = [None] * world
    dist.all_gather_object(gathered, me)
    if rank == 0:
        total = n * reps * world
        rec = {"metric": "generator pairs/s end to end at 4 MP, wind=64 ov=32 2-pass CWS (passes + post-validation + flip/scale "
                         "+ yield), pairs sharded over the ranks, one gather of the fields at the end",
               "value": total / float(tmax[0]), "unit": "pairs/s", "n_gpus": world, "higher_is_better": True, "scaling": "weak",
               "data": "synthetic", "dtype": DTYPE[args.precision],
               "config": {"workload": f"{n * reps} resident {args.size}x{args.size} pairs per rank ({n} distinct, isolated dead "
                                      f"spots), batch {batch}, wind={args.ws} overlap={args.ws // 2} {args.passes}-pass {args.mode}",
                          "pairs_total": total, "yielded_total": int(t[2]), "gathered_on_rank0": int(ids_all.numel())},
               "host": {"budget_per_rank": budget, "host_cpu_s_per_pair": float(t[1]) / total,
                        "cores_needed_at_this_rate": float(t[1]) / float(tmax[0])},
               "distributed": {"world_size": world, "backend": backend, "collectives_per_gather": 2, "ranks": gathered}}
        devices = {(g_["device_index"]) for g_ in gathered}
        if len(devices) < world:
            rec["rehearsal"] = (f"{world} ranks on {len(devices)} GPU(s) (TPIV_DIST_BACKEND=gloo): the ranks time-slice one device and the "
                                "payload of the gather goes through the host -- a check of the launch, the per-rank host budget and "
                                "the gather from device-resident fields, NOT a rate")
        print(json.dumps(rec), flush=True)
    piv.close()
    dist.barrier()
    dist.destroy_process_group()


def e2e_mode(args):
    """`--e2e`: the generator rates as the line's own metric (not the BASELINE.json metric) -- `value` is the
    resident-frame rate on frames that all hold isolated invalid vectors, the other cases ride along."""
    r, log = e2e_cases(args.e2e_pairs, args.fill_workers)
    rec = {"metric": "generator pairs/s end to end at 4 MP, wind=64 ov=32 2-pass CWS (passes + post-validation + flip/scale + yield)",
           "value": r["spots"], "unit": "pairs/s", "n_gpus": 1, "higher_is_better": True, "data": "synthetic",
           "config": {"workload": f"{args.e2e_pairs} synthetic 2048x2048 pairs, batch 32, {args.fill_workers} Qhull worker processes",
                      "cases": e2e_block(r, [], args.e2e_pairs, args.fill_workers)},
           "log": log}
    print(json.dumps(rec), flush=True)


DTYPE = {"exact": "u32 exact sums + f64 / f32", "fast": "f32", "f64": "f64/f32", "reference": "f64/f32"}
PREC_NOTE = {
    "exact": "exact (pass 1: the map cells behind the result -- arg-max, neighbours, second peak, minimum -- as exact uint8 "
             "correlation sums (u32), located by a float32 FFT pass inside an error band, float64 epilogue; undecided windows "
             "through the float64 FFT; within 1e-14 px of the reference's float64 pass 1, B:513-518; passes >= 2 float32 + float64 "
             "epilogue as in the reference, B:249-257 / B:382)",
    "fast": "fast (pass 1 float32; passes >= 2 float32 + float64 epilogue as in the reference)",
    "f64": "f64 (pass 1 float64 as in the reference, B:513-514; passes >= 2 float32 + float64 epilogue as in the reference, "
           "B:249-257 / B:382, CWS sample formed as row lerps + column lerp)",
    "reference": "reference (pass 1 float64, B:513-514; passes >= 2 float32 + float64 epilogue with the reference's "
                 "operation order in the CWS sampling, B:187-193: bit-identical staged windows)",
}


class GpuSampler:
    """Shader clock and board power of one GPU, sampled from sysfs (amdgpu hwmon: freq1_input in Hz, power1_average /
    power1_input in microwatts) by a thread while the timed steps run.  The roofline peaks assume the 2.4 GHz boost clock;
    the in-kernel stamps of round 4 read 2.08-2.13 GHz under these kernels.  Everything is optional: a box that does not show
    the files yields {"available": False, ...}."""

    def __init__(self, torch_dev_index, period_s=0.02):
        import threading
        self.period = period_s
        self.clk, self.pwr = [], []
        self.files = {}
        self.note = None
        self._stop = threading.Event()
        self._thread = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(torch_dev_index)
            bdf = None
            if hasattr(pr, "pci_bus_id"):
                bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}.0"
            cands = []
            if bdf and os.path.isdir(f"/sys/bus/pci/devices/{bdf}"):
                cands = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if not cands:
                cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
                if len(cards) == 1:
                    cands = cards
                else:
                    self.note = f"{len(cards)} amdgpu hwmon nodes, none matched PCI {bdf}"
            for h in cands[:1]:
                for key, names in (("clk", ("freq1_input",)), ("pwr", ("power1_average", "power1_input"))):
                    for n_ in names:
                        f_ = os.path.join(h, n_)
                        if os.path.exists(f_):
                            self.files[key] = f_
                            break
        except Exception as exc:
            self.note = f"{type(exc).__name__}: {exc}"

    def _read(self, key):
        try:
            with open(self.files[key]) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            c = self._read("clk") if "clk" in self.files else None
            w = self._read("pwr") if "pwr" in self.files else None
            if c:
                self.clk.append(c / 1e6)
            if w:
                self.pwr.append(w / 1e6)
            self._stop.wait(self.period)

    def start(self):
        import threading
        if self.files:
            self._stop.clear()
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None

    def summary(self):
        def st(v):
            if not v:
                return None
            v = sorted(v)
            return {"median": v[len(v) // 2], "min": v[0], "max": v[-1], "samples": len(v)}
        out = {"available": bool(self.clk or self.pwr), "gpu_clock_mhz": st(self.clk), "power_w": st(self.pwr),
               "source": self.files or None,
               "note": "sampled every 20 ms from the amdgpu hwmon files while the timed steps ran; the vector peaks of `roofline` "
                       "assume 2400 MHz"}
        if self.note:
            out["problem"] = self.note
        return out


def main():
    args = parse_args()
    if args.e2e:
        if args.gpus != 1 or os.environ.get("WORLD_SIZE") is not None:
            return e2e_sharded(args)
        return e2e_mode(args)
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    if world_env is not None and int(world_env) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}")

    import torch
    import torch.distributed as dist
    from torchpiv_amd import dist as pdist
    from torchpiv_amd import engine, synth

    rank, world, local = pdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: {world} rank(s) running, --gpus {args.gpus} asked for")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the hot path has no CPU fallback")
    if os.environ.get("TPIV_DIST_BACKEND") == "gloo":
        local = local % torch.cuda.device_count()      # rehearsal of the N-rank path on fewer GPUs (gloo only)
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local}, this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    H = W = args.size
    ws, ov = args.ws, args.ws // 2
    batch = args.batch

    # ---- synthetic frames rendered on the device (seed = 1234 + global pair index), resident in HBM
    distinct = batch if args.distinct <= 0 else min(args.distinct, batch)
    A0, B0 = synth.make_batch(distinct, H, W, first_index=rank * batch, device=dev)
    reps = (batch + distinct - 1) // distinct
    A = A0.repeat(reps, 1, 1)[:batch].contiguous()
    B = B0.repeat(reps, 1, 1)[:batch].contiguous()
    del A0, B0

    if args.config == 1:
        n_local = batch                                  # pairs this rank processes per step
        shards = [(0, batch)]
        pair_ids = torch.arange(rank * batch, (rank + 1) * batch, device=dev)
        total_per_step = batch * world
        scaling = "weak"
    else:
        mine = pdist.shard_indices(args.stream, rank, world, "block")
        n_local = len(mine)
        shards = [(s, min(batch, n_local - s)) for s in range(0, n_local, batch)]
        pair_ids = torch.tensor(mine, dtype=torch.int64, device=dev)
        total_per_step = args.stream
        scaling = "strong"
    gather_every_step = args.config == 2

    def measure(precision, steps, warmup, timed=True):
        """W warm-up steps, then exactly `steps` timed steps between barrier + synchronize pairs; returns the
        elapsed wall time (max over ranks), the per-step event times, the per-kernel event times and the plan's
        geometry / kernel names."""
        plan = engine.Plan(H, W, ws, ov, n_pass=args.passes, mode=args.mode, max_batch=batch, device=dev,
                           precision=precision)
        nr, nc = plan.out_shape
        u_all = torch.empty(max(n_local, 1), nr, nc, dtype=torch.float64, device=dev)
        v_all = torch.empty_like(u_all)
        i_all = torch.empty(max(n_local, 1), nr, nc, dtype=torch.uint8, device=dev)

        gather_ev, gather_stats = [], {}

        def one_step(gather, timed_gather=False):
            for s_, n_ in shards:
                # (every shard reads the same resident synthetic frames; its fields land in its own slice)
                plan.run(A[:n_], B[:n_], out=(u_all[s_:s_ + n_], v_all[s_:s_ + n_], i_all[s_:s_ + n_]))
            if gather and world > 1:
                # the single collective: (u, v) of every rank's pairs onto rank 0, float64 as yielded
                if timed_gather:        # events on the launch stream around the two collectives (the stream waits for them)
                    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    g0.record()
                uv = torch.stack([u_all[:n_local], v_all[:n_local]], dim=1)
                pdist.gather_fields(pair_ids, uv, stats=gather_stats)
                if timed_gather:
                    g1.record()
                    gather_ev.append((g0, g1))

        for _ in range(warmup):
            one_step(True)       # (also rehearses the gather: communicator / buffer setup of the first call)
        torch.cuda.synchronize()
        if not timed:            # profiled child: a couple of plain steps are all the counters need
            for _ in range(steps):
                one_step(False)
            torch.cuda.synchronize()
            plan.close()
            return None
        plan.set_timing(True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        sampler = GpuSampler(dev.index) if (rank == 0 and not args.no_gpu_sampling) else None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if sampler:
            sampler.start()
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(steps):
            one_step(gather_every_step or k == steps - 1, timed_gather=True)
            ev[k + 1].record()
        torch.cuda.synchronize()
        elapsed_local = time.perf_counter() - t0           # this rank's own clock, before it waits for the others
        if sampler:
            sampler.stop()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        step_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]
        timing, n_runs = plan.get_timing()
        plan.set_timing(False)
        res = {"precision": precision, "elapsed": elapsed, "step_ms": step_ms, "timing": timing, "n_runs": n_runs,
               "geometry": list(plan.geometry), "names": [plan.kernel_name(p_) for p_ in range(plan.n_pass)],
               "n_pass": plan.n_pass, "elapsed_local": elapsed_local,
               "gather_ms": [g0.elapsed_time(g1) for g0, g1 in gather_ev], "gather_stats": dict(gather_stats),
               "gpu": sampler.summary() if sampler else None}
        if precision == "exact" and plan.exact_capable():
            # slot pass1_xcorr taken apart (events behind each kernel), and how many windows the float64 transform decided
            res["exact_timing"] = plan.exact_timing()
            res["exact_fallbacks"] = {"windows": plan.exact_fallbacks(), "of": shards[-1][1] * plan.geometry[0][2] * plan.geometry[0][3],
                                      "note": "first-pass windows of the last launch that took the float64 transform"}
        plan.close()
        del u_all, v_all, i_all
        return res

    def exact_against_f64():
        """The first-pass fields of the bench batch at precision "exact" against the same batch through the float64 FFT
        kernel (what the exact sums replace): every window of the workload the headline is measured on."""
        out = {}
        fields = {}
        for prec in ("exact", "f64"):
            plan = engine.Plan(H, W, ws, ov, n_pass=args.passes, mode=args.mode, max_batch=shards[0][1], device=dev, precision=prec)
            plan.run(A[:shards[0][1]], B[:shards[0][1]])
            fields[prec] = plan.pass_fields(0, shards[0][1]) if plan.n_pass > 1 else None
            if prec == "exact":
                out["undecided_windows"] = plan.exact_fallbacks()
            plan.close()
        if fields["exact"] is None:
            return None
        (ue, ve, ie), (uf, vf, i_f) = fields["exact"], fields["f64"]
        d = torch.maximum((ue - uf).abs(), (ve - vf).abs())
        out.update({"windows": int(d.numel()), "max_abs_diff_px": float(d.max()), "bit_identical_windows": int((d == 0).sum()),
                    "validity_flags_differing": int((ie != i_f).sum()),
                    "note": "first-pass u, v, invalid of one launch of the bench batch: precision \"exact\" against the float64 FFT of every window"})
        return out

    if args.pmc_child:
        for prec in args.precision.split(","):
            measure(prec, args.steps, args.warmup, timed=False)
        return
    if "," in args.precision:
        raise SystemExit("bench.py: one --precision (the comma form is the profiled child's)")

    head = measure(args.precision, args.steps, args.warmup)
    also_fast = world == 1 and args.config == 1 and not args.no_fast and args.precision != "fast"
    also_f64 = also_fast and args.precision == "exact"
    f64_run = measure("f64", args.steps, args.warmup) if also_f64 else None
    fast = measure("fast", args.steps, args.warmup) if also_fast else None
    parity = None
    if also_f64 and "exact_timing" in head:
        try:
            parity = exact_against_f64()
        except Exception as exc:                          # never let the side check break the bench line
            parity = {"error": f"{type(exc).__name__}: {exc}"}

    # who ran: backend and devices as torch.distributed saw them (every rank reports, rank 0 prints)
    # ... and what a scaling curve needs to be read: every rank's own step times (HIP events on its launch stream), its own
    # wall clock over the timed steps (before it waits for the others), its gather times and the bytes it put into the gather
    st_me = sorted(head["step_ms"])
    me = {"rank": rank, "local_rank": local, "device": torch.cuda.get_device_name(dev), "device_index": dev.index,
          "step_ms_median": statistics.median(st_me), "step_ms_min": st_me[0], "step_ms_max": st_me[-1],
          "ms_per_step_own_clock": head["elapsed_local"] / args.steps * 1e3,
          "gather_ms": head["gather_ms"], "gather_payload_bytes": head["gather_stats"].get("payload_bytes", 0),
          "gather_useful_bytes": head["gather_stats"].get("useful_bytes", 0)}
    ranks_info = [me]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        ranks_info = gathered

    if rank == 0:
        # ---- PMC counters (N = 1): one set of child passes covers both precisions
        pmc, pmc_src = None, "off"
        if world == 1 and args.pmc == "live":
            pmc, pmc_src = collect_pmc_live(args, [args.precision] + (["f64"] if also_f64 else []) + (["fast"] if also_fast else []))
        if pmc is None and args.pmc in ("live", "file") and os.path.exists(PMC_FILE):
            try:
                with open(PMC_FILE) as f:
                    blob = json.load(f)
                if (args.batch, args.size, args.ws, args.passes, args.config) == (blob.get("batch"), 2048, 64, 2, 1):
                    pmc = blob["kernels"]
                    pmc_src = f"file: {os.path.relpath(PMC_FILE, ROOT)} ({blob.get('source', '')}); live pass: {pmc_src}"
            except Exception as exc:
                pmc_src = f"{pmc_src}; file unreadable: {exc}"

        def analyse(m):
            """Per-kernel roofline entries of one measured run; dominant = the xcorr slot with the largest mean duration."""
            kernels = {}
            for p_idx in range(m["n_pass"]):
                slot = "pass1_xcorr" if p_idx == 0 else f"pass{p_idx + 1}_xcorr"
                g_ws, g_ov, g_nr, g_nc = m["geometry"][p_idx]
                n_win = g_nr * g_nc
                launch_pairs = shards[0][1]                   # pairs per launch (full shards)
                exact = p_idx == 0 and "exact_timing" in m
                f64 = p_idx == 0 and m["precision"] not in ("fast",) and not exact
                b_launch = alg_bytes(H, W, n_win, p_idx == 0) * launch_pairs
                f_launch = alg_flops(g_ws, n_win, args.mode == "CWS" and p_idx > 0) * launch_pairs
                # (exact: this entry is the float32 locating kernel alone; the other kernels of the slot follow below)
                slot_ms = m["exact_timing"]["locate_f32"] if exact else m["timing"][slot]
                t_k = slot_ms * 1e-3
                peak_tf = FP64_VALU_PEAK_TFLOPS if f64 else FP32_VALU_PEAK_TFLOPS
                name = m["names"][p_idx]
                ctr = pmc_for(pmc, name)
                ent = {
                    "kernel": name, "arith": "f64" if f64 else "f32", "launch_ms": slot_ms,
                    "launches_timed": m["n_runs"], "pairs_per_launch": launch_pairs,
                    "alg_flops_per_launch": f_launch, "alg_bytes_per_launch": b_launch,
                    "valu": {"achieved": f_launch / t_k / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                             "frac": f_launch / t_k / 1e12 / peak_tf},
                    "hbm": {"achieved": b_launch / t_k / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": b_launch / t_k / 1e9 / HBM_PEAK_GBS, "traffic": None},
                }
                if ctr:
                    scale = launch_pairs / args.batch if args.config == 1 else 1.0      # child ran the same launch size
                    if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
                        # KiB units; on gfx950 FETCH_SIZE tallies 64 B per 128-B request of wide streaming reads
                        ent["hbm"]["traffic"] = (2.0 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024.0 * scale
                        ent["hbm"]["traffic_over_alg"] = ent["hbm"]["traffic"] / b_launch
                    if "SQ_INSTS_VALU" in ctr:
                        insts = ctr["SQ_INSTS_VALU"] * scale
                        # float64 VALU instructions issue at half rate: the wave-instruction peak is halved for them
                        ipk = ISSUE_PEAK_G
                        ent["valu_issue"] = {"insts_per_launch": insts, "achieved": insts / t_k / 1e9, "peak": ipk,
                                             "unit": "G wave-instr/s", "frac": insts / t_k / 1e9 / ipk,
                                             "alg_flops_per_wave_instr": f_launch / insts / 64.0}
                    if "SQ_LDS_BANK_CONFLICT" in ctr and ctr.get("SQ_ACTIVE_INST_LDS"):
                        ent["lds_conflict_share"] = ctr["SQ_LDS_BANK_CONFLICT"] / ctr["SQ_ACTIVE_INST_LDS"]
                    ent["counters_per_launch"] = {k: v * scale for k, v in ctr.items()}
                kernels["pass1_locate" if exact else slot] = ent
                if exact:
                    et = m["exact_timing"]
                    # the refinement: ~8 cells per window (arg-max + 4 neighbours + second peak + minimum ...), 4096 u8
                    # multiply-adds each; it re-reads both windows (L2 hits next to the locating pass: the images once)
                    kernels["pass1_refine"] = {
                        "kernel": "xcorr_exact_refine_kernel", "arith": "u8 dot4 -> u32", "launch_ms": et["refine_exact"],
                        "launches_timed": m["n_runs"], "pairs_per_launch": launch_pairs,
                        "alg_int_ops_per_launch": 2.0 * 8 * g_ws * g_ws * n_win * launch_pairs,
                        "alg_bytes_per_launch": (2 * H * W + 64 * n_win) * launch_pairs,
                        "counters_per_launch": pmc_for(pmc, "xcorr_exact_refine_kernel")}
                    kernels["pass1_undecided_f64"] = {
                        "kernel": "xcorr_f64_list_kernel<64>", "arith": "f64", "launch_ms": et["undecided_f64"],
                        "launches_timed": m["n_runs"], "windows": m["exact_fallbacks"]}
                    kernels["pass1_finalize"] = {"kernel": "finalize_kernel<true>", "launch_ms": et["finalize"]}
            dom = max((k for k in kernels if "valu" in kernels[k]), key=lambda k: kernels[k]["launch_ms"])
            d = kernels[dom]
            roof = {
                "bound": "valu",
                "kernel": d["kernel"] + f" ({dom})",
                "achieved": d["valu"]["achieved"],
                "peak": d["valu"]["peak"],
                "unit": "TFLOP/s",
                "frac": max(d["valu"]["frac"], d["hbm"]["frac"]),
                "traffic": d["hbm"]["traffic"],
                "hbm": d["hbm"],
                "valu_issue": d.get("valu_issue"),
                "lds_conflict_share": d.get("lds_conflict_share"),
                "alg_flops_per_launch": d["alg_flops_per_launch"],
                "alg_bytes_per_launch": d["alg_bytes_per_launch"],
                "launch_ms": d["launch_ms"],
                "launches_timed": d["launches_timed"],
                "counters": pmc_src,
                "note": "frac = max(B_alg / 8 TB/s, F_alg / vector peak of the kernel's arithmetic type: 78.6 TFLOP/s "
                        "float64, 157.3 float32) / t (SURVEY.md 8d): the path is VALU-bound (~190 flop/B against a ridge "
                        "of ~20), `hbm` is the same launch seen from the HBM side; traffic = (2 FETCH_SIZE + WRITE_SIZE) "
                        "KiB per launch from the PMC passes",
            }
            return kernels, roof

        def spread(v):
            v = sorted(float(t) for t in v)
            return {"min": v[0], "median": statistics.median(v), "max": v[-1]} if v else None

        def stats(m):
            st = sorted(m["step_ms"])
            return {"median": statistics.median(st), "min": st[0], "max": st[-1],
                    "p05": st[int(0.05 * (len(st) - 1))], "p95": st[int(math.ceil(0.95 * (len(st) - 1)))],
                    "pairs_per_s_at_median": total_per_step / (statistics.median(st) * 1e-3),
                    "note": "per-step HIP-event times on rank 0's launch stream; `value` uses the wall clock "
                            "around all steps (max over ranks)"}

        kernels, roof = analyse(head)
        value = total_per_step * args.steps / head["elapsed"]
        geo = head["geometry"]
        b_pair = sum(alg_bytes(H, W, g[2] * g[3], i == 0) for i, g in enumerate(geo))
        f_pair = sum(alg_flops(g[0], g[2] * g[3], args.mode == "CWS" and i > 0) for i, g in enumerate(geo))
        rec = {
            "metric": "image-pairs/sec at 4 MP, wind=64 ov=32 2-pass CWS; % HBM roofline",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": DTYPE[args.precision],
            "data": "synthetic",
            "config": {
                "workload": (f"synthetic {H}x{W} pair batch={batch} per GPU, wind={ws} overlap={ov}, {args.passes}-pass "
                             f"{args.mode} x2.0 (BASELINE.json configs[1])") if args.config == 1 else
                            (f"synthetic {H}x{W} pair stream={args.stream} in {batch}-pair shards, wind={ws} overlap={ov}, "
                             f"{args.passes}-pass {args.mode}, sharded over {world} GPU(s) (BASELINE.json configs[2])"),
                "precision": PREC_NOTE[args.precision],
                "batch_per_gpu": batch, "distinct_pairs": distinct, "pairs_per_step": total_per_step,
                "parallelism": (f"pair-sharded x{world}, one RCCL gather of (u,v) onto rank 0 "
                                + ("at the end of the timed region" if args.config == 1 else "inside every step")),
            },
            "step_ms": stats(head),
            "roofline": roof,
            "kernels": kernels,
            "kernel_ms": head["timing"],
            "whole_path": {"alg_bytes_per_pair": b_pair, "hbm_frac": b_pair * value / world / 1e9 / HBM_PEAK_GBS,
                           "alg_flops_per_pair": f_pair,
                           "valu_frac_of_f32_peak": f_pair * value / world / 1e12 / FP32_VALU_PEAK_TFLOPS},
            "distributed": {"world_size": dist.get_world_size() if world > 1 else 1,
                            "backend": dist.get_backend() if world > 1 else None,
                            "collectives_per_gather": 2 if world > 1 else 0,
                            "per_rank_ms_per_step": spread([r_["ms_per_step_own_clock"] for r_ in ranks_info]),
                            "per_rank_step_ms_median": spread([r_["step_ms_median"] for r_ in ranks_info]),
                            "gather_ms": spread([g_ for r_ in ranks_info for g_ in r_["gather_ms"]]),
                            "gathers_timed_per_rank": len(ranks_info[0]["gather_ms"]),
                            "gather_payload_bytes_per_rank": spread([r_["gather_payload_bytes"] for r_ in ranks_info]),
                            "gather_useful_bytes_total": sum(r_["gather_useful_bytes"] for r_ in ranks_info),
                            "note": "per_rank_ms_per_step: every rank's own wall clock over the timed steps (min / median / max "
                                    "over the ranks; `ms_per_step` of the line is the max plus the closing barrier); gather_ms: "
                                    "HIP events around the two collectives of a gather on every rank's launch stream",
                            "ranks": ranks_info},
            "gpu": head["gpu"],
            "published_ref": {"value": 6.7, "unit": "pairs/s",
                              "note": "'>6.7 pairs/s' incl. file I/O, GPU unnamed (GTX 1660 Ti era), "
                                      "reference README.md:58; not this exact metric, so vs_baseline is null"},
        }
        if "exact_timing" in head:
            rec["exact"] = {"pass1_ms": head["exact_timing"], "float64_path": head["exact_fallbacks"],
                            "band": {"per_unit_of_E_plus": 2.0 * (3 * 12 * 6.66 + 7) * (1 + 1 / 16) * 2.0 ** -24,
                                     "gamma_bound_u": 3 * 12 * 6.66 + 7,
                                     "note": "decision band of the locating pass = 2 Gamma (1 + 1/16) E+, Gamma = 247 u the proven bound "
                                             "on the float32 map's cell error (DESIGN.md 3.4b), E+ from exact integer window sums"},
                            "against_float64_fft": parity,
                            "note": "pass1_xcorr of kernel_ms = locate_f32 + refine_exact + undecided_f64 + finalize (HIP events "
                                    "behind each kernel); parity gates: tests/test_gpu_exact.py (<= 1e-11 px against the float64 "
                                    "kernel and the oracle, identical masks), error bound: DESIGN.md 3.4b, measured: tools/research/exact_band.py, "
                                    "tools/research/exact_adversarial.py"}
        if f64_run is not None:
            k64, roof64 = analyse(f64_run)
            rec["f64_transform"] = {"dtype": DTYPE["f64"], "precision": PREC_NOTE["f64"],
                                    "value": total_per_step * args.steps / f64_run["elapsed"], "unit": "pairs/s",
                                    "steps": args.steps, "warmup": args.warmup,
                                    "ms_per_step": f64_run["elapsed"] / args.steps * 1e3, "step_ms": stats(f64_run),
                                    "kernel_ms": f64_run["timing"], "roofline": roof64, "kernels": k64,
                                    "note": "the same workload, same process, pass 1 through the float64 FFT kernel "
                                            "(--precision f64: the headline of rounds 2-3, and the path the exact mode's undecided "
                                            "windows take)"}
        if fast is not None:
            fk, froof = analyse(fast)
            rec["fast"] = {"dtype": DTYPE["fast"], "precision": PREC_NOTE["fast"],
                           "value": total_per_step * args.steps / fast["elapsed"], "unit": "pairs/s",
                           "steps": args.steps, "warmup": args.warmup,
                           "ms_per_step": fast["elapsed"] / args.steps * 1e3, "step_ms": stats(fast),
                           "kernel_ms": fast["timing"], "roofline": froof, "kernels": fk,
                           "note": "the same workload, same process, timed right after the headline with pass 1 in float32 "
                                   "(narrower than the reference's float64 pass 1: reported beside `value`, never as it)"}
        if world == 1 and args.config == 1 and not args.no_e2e:
            del A, B
            torch.cuda.empty_cache()
            try:
                r, log = e2e_cases(args.e2e_block_pairs, args.fill_workers, files=True, budget_s=100.0)
                rec["end_to_end"] = e2e_block(r, log, args.e2e_block_pairs, args.fill_workers)
            except Exception as exc:                      # never let the side block break the bench line
                rec["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(H, W, ws, ov, args.passes, args.mode)
        assert rec["n_gpus"] == args.gpus
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
