"""Summarise rocprofv3 --pmc counter_collection CSVs (sum over dispatches per kernel)."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        meta = {}
        for r in rows:
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size"])
        for k, v in agg.items():
            print(f"{k}  dispatches={len(disp[k])} vgpr/agpr/sgpr/lds/scratch/grid={meta[k]}")
            for c, val in sorted(v.items()):
                print(f"    {c:28s} {val:.5g}   per dispatch {val/len(disp[k]):.5g}")
