#!/usr/bin/env python3
"""isa_lines.py <kernel.s> <lo> <hi> [file-substring] -- instruction histogram of the fat window kernel (vio_window_kernel<true,
true, 256>) attributed to source lines lo..hi of solver_core.h (or another header), from a -gline-tables-only --save-temps build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I../../include -ffp-contract=fast -gline-tables-only \
          -c vio_backend.hip -o /tmp/isa_dbg/x.o --save-temps=obj
With --dump the instructions themselves are printed in order."""
import collections, re, sys
path, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sub = next((a for a in sys.argv[4:] if not a.startswith('--')), 'solver_core.h')
dump = '--dump' in sys.argv
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN6vio_wk17vio_window_kernelILb1ELb1ELi256ELb0E'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]+)"', l)
    if m: files[m.group(1)] = m.group(2)
ids = {k for k, v in files.items() if sub in v}
cur, sel, total = None, [], 0
for l in lines[start:end]:
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (m.group(1), int(m.group(2)))
        continue
    if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        total += 1
        if cur and cur[0] in ids and lo <= cur[1] <= hi: sel.append((cur[1], l.strip()))
print('%d of %d instructions attributed to %s:%d-%d' % (len(sel), total, sub, lo, hi))
print(collections.Counter(i.split()[0] for _, i in sel).most_common(30))
print(sorted(collections.Counter(c for c, _ in sel).items()))
if dump:
    for c, i in sel: print(c, i)
