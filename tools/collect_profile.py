"""Copy a tools/profile_round.sh output directory (gpurun_out/<tag>) into profiles/<round>/ and rebuild
profiles/<round>/pmc_counters.json (the per-kernel counter means `bench.py --pmc file` reads) from the bench lines.
    python tools/collect_profile.py gpurun_out/r02_prof3 profiles/r02
"""
import glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.join(dst, "other_configs"), exist_ok=True)
for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "*.csv")):
    shutil.copy(f, dst)
for f in glob.glob(os.path.join(src, "other_configs", "*.txt")) + glob.glob(os.path.join(src, "other_configs", "*.csv")):
    shutil.copy(f, os.path.join(dst, "other_configs"))
out = {}
for key, name in (("config1_fast", "bench_n1.json"), ("config1_reference", "bench_n1_reference.json"),
                  ("config2_fast", "bench_n1_config2.json")):
    path = os.path.join(dst, name)
    if not os.path.exists(path):
        continue
    r = json.loads(open(path).read().strip().splitlines()[-1])
    ks = {e["kernel"]: e["counters_per_launch"] for e in r["kernels"].values() if "counters_per_launch" in e}
    out[key] = {"batch": r["config"]["batch_per_gpu"],
                "source": f"{dst}/{name} (live rocprofv3 --pmc child passes of bench.py)", "kernels": ks}
    print(key, round(r["value"]), "pairs/s", list(ks))
json.dump(out, open(os.path.join(dst, "pmc_counters.json"), "w"), indent=1)
