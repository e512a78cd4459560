export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
: > gpurun_out/res_time.log
for rep in 1 2; do
for pe in 1 2; do
  echo "== rep $rep: pipeline 256, peers $pe" | tee -a gpurun_out/res_time.log
  VIO_AMD_EST_PEERS=$pe timeout 600 python tools/time_pipeline.py 256 30 2 1 2>&1 | tail -1 | cut -c1-420 | tee -a gpurun_out/res_time.log
done
done
