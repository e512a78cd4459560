# gpurun -- 'bash tools/r05_probe.sh': round-5 first look -- instruction-cache behaviour of straight-line code (microbench + the
# window kernel's SQC counters), baseline timings of the window kernel on this box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_probe
rm -rf $O; mkdir -p $O
cd $R
for kb in 8 32 48 64 96 192 384; do tools/microbench/bin/icache_probe_$kb; done > $O/icache_probe.txt 2>&1
python tools/time_backend.py --path=single 1 128 256 512 2>&1 | grep "path=\|stage" > $O/time_single.txt
BENCH="python $R/bench.py --quick --no-cpu-baseline --only backend --steps 6 --warmup 2"
cd /tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $O/ic1 -- $BENCH > $O/ic1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/ic2 -- $BENCH > $O/ic2.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQ_INSTS_SALU SQ_INSTS_VALU -d $O/ic3 -- $BENCH > $O/ic3.log 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
for k in 1 2 3; do python tools/rocpd_pmc_summary.py $(db ic$k) 2>&1 | grep "vio_window" >> $O/pmc_icache.txt; done
rm -rf $O/ic1 $O/ic2 $O/ic3
cat $O/icache_probe.txt $O/time_single.txt $O/pmc_icache.txt | cut -c1-700
