# gpurun -- 'bash tools/fe_iters.sh': lk_track time with the iteration cap at 30 (default) and at 1 -> what the iterations cost
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for it in 30 1; do
  O=$R/gpurun_out/fe_iters_$it; rm -rf $O; mkdir -p $O
  VIO_BENCH_LK_ITERS=$it rocprofv3 --kernel-trace --stats -d $O/t -- python $R/bench.py --quick --no-cpu-baseline --only frontend --steps 12 --warmup 3 > $O/b.log 2>&1
  echo "lk_max_iters=$it"; python $R/tools/rocpd_summary.py $(find $O/t -name "*.db" | head -1) 2>&1 | grep "lk_track\|detect_k" | cut -c1-100
  rm -rf $O/t
done
