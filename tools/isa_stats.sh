#!/bin/bash
# isa_stats.sh <file.hip> [extra hipcc flags] — compiles one HIP source for gfx950 with --save-temps into /tmp/isa_<name>
# and prints, per kernel, the register / spill / scratch metadata and the instruction-class counts of its ISA.
set -e
SRC=$1; shift
NAME=$(basename $SRC .hip)
OUT=/tmp/isa_$NAME
mkdir -p $OUT
DIR=$(cd $(dirname $SRC) && pwd)
(cd $DIR && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$DIR/../../include "$@" -c $(basename $SRC) -o $OUT/x.o --save-temps=obj 2>/dev/null)
S=$(ls $OUT/*gfx950.s)
python3 - "$S" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
# split into functions by label lines "name:" followed by code until .Lfunc_end
funcs = re.findall(r'\n(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end\d+:', txt, re.S)
meta = {}
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
    d = dict(re.findall(r'\.(\w+):\s+(\d+)', m.group(2)))
    meta[m.group(1)] = d
pats = ['scratch_load', 'scratch_store', 'flat_load', 'flat_store', 'global_load', 'global_store', 'global_atomic', 'ds_read', 'ds_write', 'ds_add', 'v_mfma',
        'v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_readlane', 'v_writelane', 's_barrier', 's_waitcnt', 's_swappc']
for name, body in funcs:
    lines = [l for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    if len(lines) < 50: continue
    print('%s: %d instructions' % (name[:90], len(lines)))
    if name in meta:
        d = meta[name]
        print('   vgpr %s agpr %s sgpr %s vgpr_spill %s sgpr_spill %s scratch %s B' % (d.get('vgpr_count'), d.get('agpr_count'), d.get('sgpr_count'), d.get('vgpr_spill_count'), d.get('sgpr_spill_count'), d.get('private_segment_fixed_size')))
    print('   ' + ', '.join('%s %d' % (p, sum(1 for l in lines if l.lstrip().startswith(p))) for p in pats))
PY
