# gpurun -- 'bash tools/res_gpu.sh': the whole pipeline timed: resident estimator, synchronous / asynchronous front-end submit
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
: > gpurun_out/res_gpu.log
for n in 256 512; do
  for o in 1 2; do
    echo "== pipeline: sequences $n overlap $o" | tee -a gpurun_out/res_gpu.log
    timeout 900 python tools/time_pipeline.py $n 22 $o 1 2>&1 | tail -1 | tee -a gpurun_out/res_gpu.log
  done
done
echo "== cadence 3, 256, async" | tee -a gpurun_out/res_gpu.log
timeout 900 python tools/time_pipeline.py 256 22 2 3 2>&1 | tail -1 | tee -a gpurun_out/res_gpu.log
