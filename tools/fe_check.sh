# gpurun -- 'bash tools/fe_check.sh': front-end bit-exactness tests + per-kernel times of the front-end step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_frontend_gpu.py tests/test_properties_gpu.py tests/test_track_update_gpu.py -x -q -m gpu 2>&1 | tail -3 ) | tee gpurun_out/fe_check.log
O=$R/gpurun_out/fe_check
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/t -- python $R/bench.py --quick --no-cpu-baseline --only frontend --steps 20 --warmup 3 > $O/b.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/t -name "*.db" | head -1) 2>&1 | head -8 | cut -c1-110 | tee -a gpurun_out/fe_check.log
rm -rf $O/t
