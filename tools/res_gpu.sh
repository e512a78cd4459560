export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_estimator.py tests/test_threads_gpu.py tests/test_shim_gpu.py -x -q -m gpu 2>&1 | tail -4 ) | tee gpurun_out/res_gpu.log
( time python bench.py > gpurun_out/bench_try.json 2> gpurun_out/bench_try.err ) 2>&1 | tail -3 | tee -a gpurun_out/res_gpu.log
tail -c 600 gpurun_out/bench_try.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/bench_try.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('end_to_end','end_to_end_full'):
    print(k, json.dumps(d[k])[:1500])
P
