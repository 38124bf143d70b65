"""Copy a tools/profile_round.sh output directory (gpurun_out/<tag>) into profiles/<round>/ and rebuild
profiles/<round>/pmc_counters.json (the per-kernel counter means `bench.py --pmc file` reads) from the bench line.
    python tools/collect_profile.py gpurun_out/r03_prof profiles/r03
"""
import glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.join(dst, "other_configs"), exist_ok=True)
for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "*.csv")) + glob.glob(os.path.join(src, "*.txt")):
    shutil.copy(f, dst)
for f in glob.glob(os.path.join(src, "other_configs", "*.txt")) + glob.glob(os.path.join(src, "other_configs", "*.csv")):
    shutil.copy(f, os.path.join(dst, "other_configs"))
# the kernel statistics of the headline command list every kernel of the process (the synthetic frames are rendered by tens of
# thousands of small torch kernels before the timed region): a companion file with this library's kernels only
import csv
for stats in glob.glob(os.path.join(dst, "rocprofv3_kernel_stats.csv")) + glob.glob(os.path.join(dst, "other_configs", "*_kernel_stats.csv")):
    rows = list(csv.DictReader(open(stats)))
    if rows and not stats.endswith("_tpiv.csv"):
        keep = [r for r in rows if "tpiv::" in r["Name"]]
        tot = sum(float(r["TotalDurationNs"]) for r in keep) or 1.0
        with open(stats[:-4] + "_tpiv.csv", "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()) + ["PercentageOfTpivKernels"])
            w.writeheader()
            for r in keep:
                w.writerow(dict(r, PercentageOfTpivKernels=f"{100.0 * float(r['TotalDurationNs']) / tot:.2f}"))
path = os.path.join(dst, "bench_n1.json")
r = json.loads(open(path).read().strip().splitlines()[-1])
ks = {}
for block in (r, r.get("fast") or {}):
    for e in (block.get("kernels") or {}).values():
        if "counters_per_launch" in e:
            ks[e["kernel"]] = e["counters_per_launch"]
out = {"batch": r["config"]["batch_per_gpu"], "kernels": ks,
       "source": f"{dst}/bench_n1.json (live rocprofv3 --pmc child passes of bench.py, both precisions)"}
print(round(r["value"]), "pairs/s", r["dtype"], "fast", round((r.get("fast") or {}).get("value", 0)), list(ks))
json.dump(out, open(os.path.join(dst, "pmc_counters.json"), "w"), indent=1)
