# gpurun -- 'bash tools/res_gpu.sh': resident tests, then the whole pipeline timed with both paths
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_estimator.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -4 ) | tee gpurun_out/res_gpu.log
for n in 256 512; do
  for r in 0 1; do
    echo "== pipeline: sequences $n resident $r" | tee -a gpurun_out/res_gpu.log
    VIO_AMD_RESIDENT=$r timeout 900 python tools/time_pipeline.py $n 22 1 1 2>&1 | tail -4 | tee -a gpurun_out/res_gpu.log
  done
done
