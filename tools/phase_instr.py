"""Static instruction counts per phase of a tile kernel: the stamped diagnostic build (-DTPIV_STAMPS) puts an
s_memtime between the phases, so counting the instructions between consecutive s_memtime in the device assembly
gives VALU / LDS / VMEM / SALU instructions per phase and item (the hot path is straight-line code; the rare
per-pixel paths sit in branches that are counted too, see the note printed).
    hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=fast-honor-pragmas -fno-slp-vectorize -DTPIV_STAMPS \\
          --cuda-device-only -S torchpiv_amd/csrc/xcorr_ws32.hip -o /tmp/ws32.s
    python tools/phase_instr.py /tmp/ws32.s 32 2        # WS, MODE (0 pass 1, 1 DWS, 2 CWS)
"""
import re, sys
path, ws, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
names = ["kernel prologue", "(start)", "loop head", "convert (incl. slow paths)", "mean / issue of next rows", "fwd row FFT", "transpose 1",
         "fwd col FFT", "cross-spectrum", "inv col FFT", "transpose 2", "inv row FFT", "prefetch issue", "peak + record", "loop tail", "epilogue"]
txt = open(path).read()
occ = {8: 2, 16: 4, 32: 3, 64: 3 if mode != 2 else 2}[ws]
kern = f"_ZN4tpiv17xcorr_tile_kernelILi{ws}ELi{mode}ELi{occ}ELb1EEEvNS_10PassParamsE"
body = re.search(re.escape(kern) + r":(.*?)s_endpgm", txt, re.S).group(1).split("\n")
segs, cur = [], dict(valu=0, lds=0, vmem=0, salu=0)
for line in body:
    t = line.strip().split()
    if not t:
        continue
    op = t[0]
    if op == "s_memtime":
        segs.append(cur); cur = dict(valu=0, lds=0, vmem=0, salu=0); continue
    if op.startswith("v_"): cur["valu"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    elif op.startswith(("global_", "scratch_", "buffer_")): cur["vmem"] += 1
    elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")): cur["salu"] += 1
segs.append(cur)
print(f"{kern}: {len(segs)} segments")
for i, sg in enumerate(segs):
    print(f"  {names[i] if i < len(names) else i:32s} VALU {sg['valu']:5d}  LDS {sg['lds']:4d}  VMEM {sg['vmem']:3d}  SALU {sg['salu']:3d}")
print("  total VALU", sum(s["valu"] for s in segs))
