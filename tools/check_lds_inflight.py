"""Static check of the hand-issued LDS reads of the float64 kernels (xcorr_f64_split.hpp, xcorr_f64.hip).

Those kernels issue `ds_read_b64 / ds_read_b128` through inline asm and wait for them with a separate counted
`s_waitcnt lgkmcnt(N)` asm statement, so that the next batch of reads is in flight while the current one is consumed.  The
compiler's own wait-count insertion does not see inline asm: nothing but the data dependency through the wait statement
tells it that the destination registers are not valid yet.  If it ever copied, spilled or otherwise READ such a register
between the read and the wait that covers it, the kernel would silently use stale data.  This script compiles
the translation unit to assembly and walks every kernel: LDS operations are kept in issue order (the LDS returns in order),
`s_waitcnt lgkmcnt(N)` retires all but the N youngest, and any instruction that names a register of a read still in flight
is reported.

    python tools/check_lds_inflight.py [xcorr_f64]      -> exit code 1 and the offending lines if the constraint is broken
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=fast-honor-pragmas", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-Wno-unused-result", "-S", "--cuda-device-only"]


def regs(tok):
    m = re.match(r"-?\|?v\[(\d+):(\d+)\]\|?$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?\|?v(\d+)\|?$", tok)
    return {int(m.group(1))} if m else set()


def check(asm_text):
    problems, n_reads = [], 0
    kernel, fifo = None, []                      # fifo: [registers written by the op (empty for stores / others)]
    in_asm = False                               # between ;;#ASMSTART and ;;#ASMEND: a hand-issued instruction
    for ln, line in enumerate(asm_text.splitlines(), 1):
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        s = line.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, fifo = m.group(1), []
            continue
        if s.endswith(":") or s.startswith("."):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        toks = [t.strip().split(" ")[0] for t in parts[1].split(",")] if len(parts) > 1 else []
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                keep = int(m.group(1))
                fifo = fifo[len(fifo) - keep:] if keep else []
            continue
        if op in ("s_barrier", "s_endpgm") or op.startswith("s_cbranch") or op == "s_branch":
            # (control flow: the kernels wait for lgkmcnt(0) in front of every barrier; across branches the walk is linear,
            #  which is conservative enough for straight-line read batches)
            continue
        used = set().union(*[regs(t) for t in toks]) if toks else set()
        inflight = set().union(*fifo) if fifo else set()
        if op.startswith("ds_read") or op.startswith("ds_load"):
            dst = regs(toks[0]) if toks else set()
            if (used - dst) & inflight or dst & inflight:
                problems.append((kernel, ln, line.strip()))
            # only the hand-issued reads are at risk: the compiler waits for its own reads before it touches their registers
            fifo.append(dst if in_asm else set())
            n_reads += in_asm
            continue
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            if used & inflight:
                problems.append((kernel, ln, line.strip()))
            fifo.append(set())
            continue
        if used & inflight:
            problems.append((kernel, ln, line.strip()))
    return problems, n_reads


def main(unit="xcorr_f64"):
    src = os.path.join(ROOT, "torchpiv_amd", "csrc", unit + ".hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, unit + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        problems, n_reads = check(open(out).read())
    print(f"{unit}: {n_reads} hand-issued LDS reads walked, {len(problems)} use(s) of a register whose read is still in flight")
    for k, ln, text in problems[:20]:
        print(f"  {k} line {ln}: {text}")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:]))
