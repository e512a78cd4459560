export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_estimator.py tests/test_closed_loop.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3 ) | tee gpurun_out/res_gpu.log
for n in 256 512 1024; do
python tools/time_estimator.py $n 40 2>&1 | tail -2 | tee -a gpurun_out/res_gpu.log
done
