"""All kernels around the k-th 32x32 CWS launch of a rocprofv3 kernel trace (development aid)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in rows), key=lambda t: t[0])
marks = [i for i, e in enumerate(ks) if "xcorr_tile_kernel<32, 2" in e[2]]
s0, e0 = ks[marks[k]][0], ks[marks[k]][1]
prev = ks[marks[k - 1]][1]
for s, e, n, q, wg, grid in ks:
    if e >= prev and s <= e0 + 200000:
        print(f"{(s - s0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  queue {q:>3s} grid {grid:>9s} wg {wg:>4s}  {n[:80]}")
