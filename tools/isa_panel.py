#!/usr/bin/env python3
"""isa_panel.py <kernel.s> [lo hi] -- instruction categories of the code attributed to panel_step_static (all six instantiations of the
W = 10 kernel together; -gline-tables-only --save-temps build, see isa_lines.py): the per-category table of the round-6 review."""
import re, collections, sys
lines = open(sys.argv[1]).read().split('\n')
src = open(__file__.replace('tools/isa_panel.py', 'vins-mobile_amd/csrc/solver_core.h')).read().split('\n')
lo = next(i for i, l in enumerate(src) if 'VIO_DEV void panel_step_static(' in l) + 1
hi = next(i for i in range(lo, len(src)) if src[i].startswith('}')) + 1
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]+)"', l)
    if m: files[m.group(1)] = m.group(2).split('/')[-1]
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN6vio_wk17vio_window_kernel'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
cur, seq = None, []
for l in lines[start:end]:
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(m.group(1)), int(m.group(2)))
        continue
    if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'): seq.append((cur, l.strip()))
ps = [i for i, (c, ins) in enumerate(seq) if c and c[0] == 'solver_core.h' and lo <= c[1] <= hi]
a, b = ps[0], ps[-1]
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_read'): return 'lds_read'
    if op.startswith(('ds_write', 'ds_add')): return 'lds_write'
    if op.startswith(('global_', 'flat_', 'buffer_')): return 'vmem'
    if op.startswith('scratch_'): return 'scratch'
    if op in ('v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32'): return 'lane'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_nop'): return 's_nop'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('v_cndmask', 'v_cmp')): return 'select/cmp'
    if op.startswith(('v_mov', 'v_accvgpr')): return 'v_mov'
    if 'f64' in op: return 'valu_f64'
    return 'valu_int'
c = collections.Counter(cls(ins.split()[0]) for _, ins in seq[a:b + 1])
n = b - a + 1
NI = 18
print('panel_step_static (solver_core.h:%d-%d), %d instantiations (3 waves x (5 first-tile values + the prior block)): %d instructions, %d matrix instructions -> %.0f per instantiation' % (lo, hi, NI, n, c["mfma"], n / float(NI)))
for k, v in sorted(c.items(), key=lambda kv: -kv[1]): print('  %-12s %5d  (%.0f per instantiation)' % (k, v, v / float(NI)))
