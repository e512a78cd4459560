export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
: > gpurun_out/res_time.log
for rep in 1 2 3; do
for n in 256 512; do
for g in 1 2; do
  echo "== rep $rep: $n sequences, groups $g" | tee -a gpurun_out/res_time.log
  VIO_AMD_EST_GROUPS=$g timeout 600 python tools/time_estimator.py $n 40 2>&1 | tail -2 | head -1 | tee -a gpurun_out/res_time.log
done
done
done
