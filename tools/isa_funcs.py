#!/usr/bin/env python3
"""isa_funcs.py <kernel.s> [kernel-symbol-prefix] -- static instruction counts of one kernel attributed to the FUNCTIONS of
solver_core.h / marg_core.h / batch.h (innermost inlined location of a -gline-tables-only --save-temps build, see isa_lines.py),
with the classes the instruction diet of round 6 tracks: matrix, LDS, global, scalar, v_readlane / v_writelane (scalar
spills), waitcnt, the rest of the vector unit."""
import collections, os, re, sys
path = sys.argv[1]
sym = sys.argv[2] if len(sys.argv) > 2 else '_ZN6vio_wk17vio_window_kernelILb1ELb1ELi256ELb0E'
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vins-mobile_amd', 'csrc')
def func_ranges(fn):
    out, src = [], open(os.path.join(here, fn)).read().split('\n')
    for i, l in enumerate(src):
        m = re.match(r'^(?:VIO_DEV|VIO_HD|__device__ __forceinline__|template.*\n)?\s*(?:VIO_DEV|VIO_HD|__device__ __forceinline__|inline)\s+[\w:<>\*& ]+?\s+(\w+)\s*\(', l)
        if m and not l.startswith(' '): out.append((i + 1, m.group(1)))
    return out
ranges = {fn: func_ranges(fn) for fn in ('solver_core.h', 'marg_core.h', 'batch.h', 'vio_math.h', 'vio_window_kernel.inc')}
def func_of(fn, line):
    best = '?'
    for l0, name in ranges.get(fn, []):
        if l0 <= line: best = name
        else: break
    return best
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(sym))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]+)"', l)
    if m: files[m.group(1)] = os.path.basename(m.group(2))
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'flat_', 'buffer_')): return 'vmem'
    if op.startswith('scratch_'): return 'scratch'
    if op in ('v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32'): return 'lane'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('v_cndmask', 'v_cmp')): return 'sel'
    if op.endswith('_f64') or 'f64' in op: return 'vf64'
    return 'valu'
tab = collections.defaultdict(collections.Counter)
cur = None
for l in lines[start:end]:
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(m.group(1), '?'), int(m.group(2)))
        continue
    if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        op = l.strip().split()[0]
        key = '%s:%s' % (cur[0], func_of(cur[0], cur[1])) if cur else '?'
        tab[key][cls(op)] += 1
cols = ['mfma', 'vf64', 'valu', 'sel', 'lane', 'salu', 'lds', 'vmem', 'scratch', 'wait']
tot = collections.Counter()
print('%-46s %7s ' % ('function', 'total') + ' '.join('%7s' % c for c in cols))
for k, c in sorted(tab.items(), key=lambda kv: -sum(kv[1].values())):
    n = sum(c.values())
    tot.update(c)
    if n >= 150: print('%-46s %7d ' % (k[:46], n) + ' '.join('%7d' % c[x] for x in cols))
print('%-46s %7d ' % ('TOTAL', sum(tot.values())) + ' '.join('%7d' % tot[x] for x in cols))
