"""Where the scratch traffic, barriers and LDS instructions of a kernel sit in its device assembly (development aid):
   python tools/dev/asm_scan.py /tmp/x.s <first line> <last line>"""
import sys, collections
lines = open(sys.argv[1]).read().splitlines()[int(sys.argv[2]):int(sys.argv[3])]
cnt = collections.Counter()
marks = []
for i, l in enumerate(lines):
    t = l.strip().split(' ')[0] if l.strip() else ''
    cnt[t] += 1
    if t.startswith('scratch_') or t == 's_barrier':
        marks.append((i, t))
print(len(lines), 'lines')
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:45]:
    print(f'  {k:28s} {v}')
run = []
for i, t in marks:
    if t == 's_barrier':
        print('   barrier @', i)
    else:
        print('     ', t, '@', i, lines[i].strip()[:70])
