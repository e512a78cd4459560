# gpurun -- 'bash tools/fe_gpu.sh': front-end parity tests + kernel trace of the bench's front-end half
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fe
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_track_update_gpu.py tests/test_shim_gpu.py tests/test_dbow.py tests/test_pipeline_gpu.py -x -q -m gpu > $O/tests.log 2>&1
grep -v "marginaliz\|release\|parallax" $O/tests.log | tail -6
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --quick --no-cpu-baseline --steps 20 --warmup 3 > $O/bench.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) $O/kernel_trace.txt > /dev/null
head -14 $O/kernel_trace.txt
tail -1 $O/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'], d['config']['kernel_ms'], 'frac', d['roofline']['frac'])"
rm -rf $O/kt
