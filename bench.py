#!/usr/bin/env python3
"""Benchmark of the PIV cross-correlation hot path (driver contract).

    python bench.py --gpus N --steps K --warmup W [--config 1|2] [--precision fast|reference]

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

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel: HIP-event durations on the
launch stream during the timed steps, bound = "valu" (the path sits ~10x above the
HBM ridge), frac = max(B_alg / 8 TB/s, F_alg / 157.3 TFLOP/s) / t, the HBM view beside it, and --
at N = 1 -- PMC counters of THIS build collected live by rocprofv3 child runs of this script
(FETCH_SIZE, WRITE_SIZE, SQ_* in separate passes; --pmc file reads profiles/r02/pmc_counters.json
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
PMC_FILE = os.path.join(ROOT, "profiles", "r02", "pmc_counters.json")


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
    ap.add_argument("--precision", default="fast", choices=("fast", "reference"),
                    help="arithmetic of pass 1: float32 (fast) or float64 like the reference (PIVbackend.py:513-514)")
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
    ap.add_argument("--fill-workers", type=int, default=8)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
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
def cpu_baseline(H, W, ws, ov, n_pass, mode, budget_s=20.0):
    import torch
    from oracle import piv_oracle as O
    from torchpiv_amd import synth
    threads = torch.get_num_threads()
    pairs = [synth.make_pair(H, W, 900 + i, noise=2.0) for i in range(2)]
    pairs = [(a.numpy(), b.numpy()) for a, b in pairs]

    def kernels_only(a, b):
        u, v, x, y, val = O.pass1(a, b, ws, ov, validate=True)
        w, o = ws, ov
        for _ in range(n_pass - 1):
            w, o = w // 2, o // 2
            u, v, x, y, val = O.ITER[mode](a.shape, w, o)(a, b, x, y, u, v, val)
        return u, v, val

    kernels_only(*pairs[0])                      # warm-up (MKL plans, page faults)
    n, t0 = 0, time.perf_counter()
    while True:
        kernels_only(*pairs[n % len(pairs)])
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s * 0.7 or n >= 8:
            break
    # end to end like OfflinePIV.__call__ (B:873-901): the passes plus NaN-out, border interpolation,
    # Delaunay hole fill, flip and scaling (frames already decoded, as for the GPU figure)
    t1 = time.perf_counter()
    n2 = 0
    for _ in O.offline_piv(pairs[:1], ws, ov, multipass=n_pass, mode=mode):
        pass
    n2 += 1
    dt2 = time.perf_counter() - t1
    return {"value": n / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "end_to_end": {"value": n2 / dt2, "unit": "pairs/s",
                           "what": "oracle offline_piv on resident frames: passes + NaN-out + border interpolation + "
                                   "Delaunay hole fill + flip/scale (PIVbackend.py:873-901), 1 pair"},
            "sample": f"{n} pairs of the same {H}x{W} {n_pass}-pass {mode} workload (noise 2), oracle/piv_oracle.py "
                      f"(numpy + torch-CPU FFT), no file I/O; survey-container figure for the reference itself: "
                      f"0.24 pairs/s on 8 threads (BASELINE.md)"}


# ---------------------------------------------------------------------------------------------
# PMC counters of the dominant kernels: rocprofv3 child runs of this script (N = 1 only)
# ---------------------------------------------------------------------------------------------
PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"]]


def collect_pmc_live(args, timeout_s=170):
    """{kernel name substring: {counter: mean per launch}} or None.  One rocprofv3 child per counter
    group (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots');
    the program itself follows `--` (no env/bash hop: the profiler's library initialises the GPU)."""
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
                   "--config", str(args.config), "--precision", args.precision, "--steps", "2", "--warmup", "1",
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
    return out, "live: rocprofv3 --pmc child runs of this bench.py (2 timed steps each)"


def pmc_for(pmc, kernel_name):
    if not pmc:
        return None
    for k, v in pmc.items():
        if kernel_name in k:
            return v
    return None


def e2e_mode(args):
    """Generator rates end to end (tools/e2e_generator.py): not the BASELINE.json metric -- `value` is the
    resident-frame rate on frames that all need the host triangulation, the other cases ride along."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import contextlib
    import io as _io
    import e2e_generator
    log = _io.StringIO()
    with contextlib.redirect_stdout(log):
        r = e2e_generator.main(n=args.e2e_pairs, workers=args.fill_workers)
    rec = {"metric": "generator pairs/s end to end at 4 MP, wind=64 ov=32 2-pass CWS (passes + post-validation + flip/scale + yield)",
           "value": r["spots"], "unit": "pairs/s", "n_gpus": 1, "higher_is_better": True, "data": "synthetic",
           "config": {"workload": f"{args.e2e_pairs} synthetic 2048x2048 pairs, batch 32, {args.fill_workers} Qhull worker processes",
                      "cases": {"resident_clean_all_dropped": r["clean"], "resident_straight_runs": r["runs"],
                                "resident_isolated_spots": r["spots"], "bmp_files_isolated_spots": r.get("files"),
                                "bmp_files_generator_call": r.get("files_call")}},
           "log": log.getvalue().strip().splitlines()}
    print(json.dumps(rec), flush=True)


def main():
    args = parse_args()
    if args.e2e:
        if args.gpus != 1:
            raise SystemExit("bench.py --e2e is a single-GPU mode")
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

    plan = engine.Plan(H, W, ws, ov, n_pass=args.passes, mode=args.mode, max_batch=batch, device=dev,
                       precision=args.precision)
    nr, nc = plan.out_shape

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
    u_all = torch.empty(max(n_local, 1), nr, nc, dtype=torch.float64, device=dev)
    v_all = torch.empty_like(u_all)
    i_all = torch.empty(max(n_local, 1), nr, nc, dtype=torch.uint8, device=dev)

    def one_step(gather):
        for s, n in shards:
            # (every shard reads the same resident synthetic frames; its fields land in its own slice)
            plan.run(A[:n], B[:n], out=(u_all[s:s + n], v_all[s:s + n], i_all[s:s + n]))
        if gather and world > 1:
            # the single collective: (u, v) of every rank's pairs onto rank 0, float64 as yielded
            uv = torch.stack([u_all[:n_local], v_all[:n_local]], dim=1)
            pdist.gather_fields(pair_ids, uv)

    gather_every_step = args.config == 2
    for _ in range(args.warmup):
        one_step(True)       # (also rehearses the gather: communicator / buffer setup of the first call)
    torch.cuda.synchronize()
    if args.pmc_child:       # profiled child: a couple of plain steps are all the counters need
        for _ in range(args.steps):
            one_step(False)
        torch.cuda.synchronize()
        plan.close()
        return
    plan.set_timing(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(args.steps):
        one_step(gather_every_step or k == args.steps - 1)
        ev[k + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    timing, n_runs = plan.get_timing()
    plan.set_timing(False)

    if rank == 0:
        value = total_per_step * args.steps / elapsed
        # ---- per-kernel roofline entries, dominant = the xcorr slot with the largest mean duration
        pmc, pmc_src = None, "off"
        if world == 1 and args.pmc == "live":
            pmc, pmc_src = collect_pmc_live(args)
        if pmc is None and args.pmc in ("live", "file") and os.path.exists(PMC_FILE):
            try:
                with open(PMC_FILE) as f:
                    blob = json.load(f)
                key = f"config{args.config}_{args.precision}"
                if key in blob and (args.batch, args.size, args.ws, args.passes) == (blob[key]["batch"], 2048, 64, 2):
                    pmc = blob[key]["kernels"]
                    pmc_src = f"file: profiles/r02/pmc_counters.json ({blob[key].get('source', '')}); live pass: {pmc_src}"
            except Exception as exc:
                pmc_src = f"{pmc_src}; file unreadable: {exc}"
        kernels = {}
        for p_idx in range(plan.n_pass):
            slot = "pass1_xcorr" if p_idx == 0 else f"pass{p_idx + 1}_xcorr"
            g_ws, g_ov, g_nr, g_nc = plan.geometry[p_idx]
            n_win = g_nr * g_nc
            launch_pairs = shards[0][1]                   # pairs per launch (full shards)
            f64 = p_idx == 0 and args.precision == "reference"
            b_launch = alg_bytes(H, W, n_win, p_idx == 0) * launch_pairs
            f_launch = alg_flops(g_ws, n_win, args.mode == "CWS" and p_idx > 0) * launch_pairs
            t_k = timing[slot] * 1e-3
            peak_tf = FP64_VALU_PEAK_TFLOPS if f64 else FP32_VALU_PEAK_TFLOPS
            name = plan.kernel_name(p_idx)
            ctr = pmc_for(pmc, name)
            ent = {
                "kernel": name, "launch_ms": timing[slot], "launches_timed": n_runs, "pairs_per_launch": launch_pairs,
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
                    ent["valu_issue"] = {"insts_per_launch": insts, "achieved": insts / t_k / 1e9, "peak": ISSUE_PEAK_G,
                                         "unit": "G wave-instr/s", "frac": insts / t_k / 1e9 / ISSUE_PEAK_G,
                                         "alg_flops_per_wave_instr": f_launch / insts / 64.0}
                ent["counters_per_launch"] = {k: v * scale for k, v in ctr.items()}
            kernels[slot] = ent
        dom = max(kernels, key=lambda k: kernels[k]["launch_ms"])
        d = kernels[dom]
        frac = max(d["valu"]["frac"], d["hbm"]["frac"])
        b_pair = sum(alg_bytes(H, W, g[2] * g[3], i == 0) for i, g in enumerate(plan.geometry))
        f_pair = sum(alg_flops(g[0], g[2] * g[3], args.mode == "CWS" and i > 0) for i, g in enumerate(plan.geometry))
        srt = sorted(step_ms)
        rec = {
            "metric": "image-pairs/sec at 4 MP, wind=64 ov=32 2-pass CWS; % HBM roofline",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fast" else "f64/f32",
            "data": "synthetic",
            "config": {
                "workload": (f"synthetic {H}x{W} pair batch={batch} per GPU, wind={ws} overlap={ov}, {args.passes}-pass "
                             f"{args.mode} x2.0 (BASELINE.json configs[1])") if args.config == 1 else
                            (f"synthetic {H}x{W} pair stream={args.stream} in {batch}-pair shards, wind={ws} overlap={ov}, "
                             f"{args.passes}-pass {args.mode}, sharded over {world} GPU(s) (BASELINE.json configs[2])"),
                "precision": args.precision + (" (pass 1 float32; passes >= 2 float32 + float64 epilogue as in the reference)"
                                               if args.precision == "fast" else
                                               " (pass 1 float64 as in the reference, B:513-514; passes >= 2 float32 + float64 epilogue)"),
                "batch_per_gpu": batch, "distinct_pairs": distinct, "pairs_per_step": total_per_step,
                "parallelism": (f"pair-sharded x{world}, one RCCL gather of (u,v) onto rank 0 "
                                + ("at the end of the timed region" if args.config == 1 else "inside every step")),
            },
            "step_ms": {"median": statistics.median(step_ms), "min": srt[0], "max": srt[-1],
                        "p05": srt[int(0.05 * (len(srt) - 1))], "p95": srt[int(math.ceil(0.95 * (len(srt) - 1)))],
                        "pairs_per_s_at_median": total_per_step / world / (statistics.median(step_ms) * 1e-3) * world,
                        "note": "per-step HIP-event times on rank 0's launch stream; `value` uses the wall clock "
                                "around all steps (max over ranks)"},
            "roofline": {
                "bound": "valu",
                "kernel": d["kernel"] + f" ({dom})",
                "achieved": d["valu"]["achieved"],
                "peak": d["valu"]["peak"],
                "unit": "TFLOP/s",
                "frac": frac,
                "traffic": d["hbm"]["traffic"],
                "hbm": d["hbm"],
                "valu_issue": d.get("valu_issue"),
                "alg_flops_per_launch": d["alg_flops_per_launch"],
                "alg_bytes_per_launch": d["alg_bytes_per_launch"],
                "launch_ms": d["launch_ms"],
                "launches_timed": d["launches_timed"],
                "counters": pmc_src,
                "note": "frac = max(B_alg / 8 TB/s, F_alg / vector peak) / t (SURVEY.md 8d): the path is VALU-bound "
                        "(~190 flop/B against a ridge of ~20), `hbm` is the same launch seen from the HBM side; "
                        "traffic = (2 FETCH_SIZE + WRITE_SIZE) KiB per launch from the PMC passes",
            },
            "kernels": kernels,
            "kernel_ms": timing,
            "whole_path": {"alg_bytes_per_pair": b_pair, "hbm_frac": b_pair * value / world / 1e9 / HBM_PEAK_GBS,
                           "alg_flops_per_pair": f_pair,
                           "valu_frac": f_pair * value / world / 1e12 / FP32_VALU_PEAK_TFLOPS},
            "published_ref": {"value": 6.7, "unit": "pairs/s",
                              "note": "'>6.7 pairs/s' incl. file I/O, GPU unnamed (GTX 1660 Ti era), "
                                      "reference README.md:58; not this exact metric, so vs_baseline is null"},
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(H, W, ws, ov, args.passes, args.mode)
        assert rec["n_gpus"] == args.gpus
        print(json.dumps(rec), flush=True)
    plan.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
