# gpurun -- 'bash tools/res_gpu.sh': whole GPU suite (resident stores are the default), then timing of both paths
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) | tee gpurun_out/res_gpu.log
for n in 256 512 1024; do
  for r in 0 1; do
    echo "== sequences $n resident $r" | tee -a gpurun_out/res_gpu.log
    VIO_AMD_RESIDENT=$r timeout 600 python tools/time_estimator.py $n 40 2>&1 | tail -2 | tee -a gpurun_out/res_gpu.log
  done
done
