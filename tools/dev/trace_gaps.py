"""Gaps on the GPU timeline of a rocprofv3 --kernel-trace run (development aid): python tools/dev/trace_gaps.py <kernel_trace.csv> [skip]
prints, for the kernels after the first `skip` xcorr launches, the busy share and the idle time in front of each kernel name."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
# start at the `skip`-th 32x32 CWS launch, end at the last one
marks = [i for i, k in enumerate(ks) if "xcorr_tile_kernel<32, 2" in k[2]]
lo, hi = marks[skip], marks[-1]
sel = ks[lo:hi]
span = sel[-1][1] - sel[0][0]
busy, end = 0, sel[0][0]
gap_by = collections.Counter()
for s, e, nme in sel:
    if s > end:
        gap_by[nme[:70]] += s - end
    busy += max(0, e - max(s, end))
    end = max(end, e)
n_b = len(marks) - 1 - skip
print(f"{n_b} launches: period {span / n_b / 1e6:.3f} ms, busy {busy / n_b / 1e6:.3f} ms, idle {(span - busy) / n_b / 1e6:.3f} ms per launch")
for nme, g in gap_by.most_common(12):
    print(f"   idle in front of {nme:70s} {g / n_b / 1e3:8.1f} us per launch")
tot = collections.Counter()
cnt = collections.Counter()
for s, e, nme in sel:
    tot[nme[:80]] += e - s
    cnt[nme[:80]] += 1
print("   kernel time per launch:")
for nme, t in tot.most_common(14):
    print(f"   {nme:80s} {t / n_b / 1e3:9.1f} us  ({cnt[nme] / n_b:.1f} calls)")
