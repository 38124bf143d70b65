"""Rough VGPR liveness at a line of a kernel's device assembly (development aid): follows the fall-through path from the
line to the next s_barrier, skipping blocks whose label is not reached by fall-through is NOT attempted -- pass a range
that is straight-line.   python tools/dev/asm_live.py file.s <abs line> <abs end line>"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
a, b = int(sys.argv[2]), int(sys.argv[3])
def expand(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'-?\|?v(\d+)\|?$', tok)
    if m: return [int(m.group(1))]
    m = re.match(r'-?v\[(\d+):(\d+)\]', tok)
    if m: return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []
first = {}
for l in lines[a:b]:
    s = l.strip()
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'): continue
    s = s.split(';')[0].strip()
    parts = s.split(None, 1)
    op = parts[0]
    ops = [t.strip() for t in parts[1].split(',')] if len(parts) > 1 else []
    ops = [t.split(' ')[0] for t in ops]
    regs = [expand(t) for t in ops]
    store = op.startswith(('ds_write', 'scratch_store', 'global_store', 'buffer_store', 'flat_store'))
    rw = op.startswith(('v_fmac', 'v_mac', 'v_writelane')) or 'dpp' in op
    defs = [] if store else (regs[0] if regs else [])
    uses = [r for i, rs in enumerate(regs) for r in rs if store or i > 0 or rw]
    for r in uses:
        first.setdefault(r, 'R')
    for r in defs:
        first.setdefault(r, 'W')
live = sorted(r for r, k in first.items() if k == 'R')
print(len(live), 'VGPRs live-in (read before written) on the path', a, '..', b)
print(live)
if len(sys.argv) > 4:
    # where (line offset from the start) each live-in register is first read
    mid = int(sys.argv[4])
    firstread = {}
    for i, l in enumerate(lines[a:b]):
        s = l.strip()
        if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'): continue
        s = s.split(';')[0].strip()
        parts = s.split(None, 1)
        ops = [t.strip().split(' ')[0] for t in parts[1].split(',')] if len(parts) > 1 else []
        for t in ops:
            for r in expand(t):
                if r in live and r not in firstread: firstread[r] = i
    late = sorted(r for r in live if firstread.get(r, 0) >= mid - a)
    print(len(late), 'of them are first touched at or after line', mid, ':', late)
    import collections
    print(sorted(collections.Counter((firstread[r] // 100) for r in late).items()))
