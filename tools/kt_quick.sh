# Quick kernel-trace summary of one bench run: gpurun -- 'bash tools/kt_quick.sh [bench args]'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ktq
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --quick --no-cpu-baseline --steps 20 --warmup 3 "$@" > $O/bench.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) $O/kernel_trace.txt > /dev/null
head -14 $O/kernel_trace.txt
tail -1 $O/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'], d['config']['kernel_ms'], 'frac', d['roofline']['frac'])"
rm -rf $O/kt
