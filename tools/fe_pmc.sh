# gpurun -- 'bash tools/fe_pmc.sh': SQ counters of the front-end kernels (one pass per counter group)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fe_pmc
rm -rf $O; mkdir -p $O
BENCH="python $R/bench.py --quick --no-cpu-baseline --only frontend --steps 6 --warmup 2"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/a -- $BENCH > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/b -- $BENCH > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $O/c -- $BENCH > $O/c.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/d -- $BENCH > $O/d.log 2>&1
cd $R
for k in a b c d; do python tools/rocpd_pmc_summary.py $(find $O/$k -name "*.db" | head -1) 2>&1 | grep "lk_track\|detect\|track_update" >> $O/sq.txt; done
cat $O/sq.txt | cut -c1-150
rm -rf $O/a $O/b $O/c $O/d
