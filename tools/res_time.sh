# gpurun -- 'bash tools/res_time.sh': end-to-end estimator timing, resident path, by number of groups
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
: > gpurun_out/res_time.log
for n in 256 512 1024; do
  for g in 1 2; do
    echo "== sequences $n resident 1 groups $g" | tee -a gpurun_out/res_time.log
    VIO_AMD_EST_GROUPS=$g VIO_AMD_RESIDENT=1 timeout 600 python tools/time_estimator.py $n 40 2>&1 | tail -2 | tee -a gpurun_out/res_time.log
  done
done
