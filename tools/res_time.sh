export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
: > gpurun_out/res_time.log
for rep in 1 2; do
for t in 0 1; do
  echo "== rep $rep: pipeline 256, window LDS tight $t" | tee -a gpurun_out/res_time.log
  VIO_AMD_WINDOW_LDS_TIGHT=$t timeout 600 python tools/time_pipeline.py 256 30 2 1 2>&1 | tail -1 | cut -c1-420 | tee -a gpurun_out/res_time.log
  echo "== rep $rep: estimator 256, window LDS tight $t" | tee -a gpurun_out/res_time.log
  VIO_AMD_WINDOW_LDS_TIGHT=$t timeout 600 python tools/time_estimator.py 256 40 2>&1 | tail -2 | head -1 | tee -a gpurun_out/res_time.log
done
done
