# Round profiling on the GPU box (gpurun): kernel trace, HBM counters (+ calibration), SQ counters, stage cycles.
# usage: gpurun -- 'bash tools/profile_round.sh r03_a'     -> gpurun_out/<tag>/..., summaries to copy into profiles/
set -x
export TMPDIR=/tmp
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
BENCH="python $R/bench.py --quick --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- $BENCH --steps 20 --warmup 3 > $O/bench_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- $BENCH --steps 6 --warmup 2 > $O/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- $BENCH --steps 6 --warmup 2 > $O/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib_fetch -- $R/tools/microbench/bin/pmc_calib > $O/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calib_write -- $R/tools/microbench/bin/pmc_calib > $O/calib_write.log 2>&1
# SQ passes on the window kernel alone (counters that fit one pass each)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq1 -- $BENCH --only backend --steps 6 --warmup 2 > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/sq2 -- $BENCH --only backend --steps 6 --warmup 2 > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -d $O/sq3 -- $BENCH --only backend --steps 6 --warmup 2 > $O/sq3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM -d $O/sq4 -- $BENCH --only backend --steps 6 --warmup 2 > $O/sq4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT -d $O/sq5 -- $BENCH --only backend --steps 6 --warmup 2 > $O/sq5.log 2>&1
# dynamic instruction mix per trust-region iteration: the same counters with 2 instead of 10 iterations (difference / 8)
VIO_BENCH_MAX_ITER=2 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -d $O/sq3b -- $BENCH --only backend --steps 6 --warmup 2 > $O/sq3b.log 2>&1
VIO_BENCH_MAX_ITER=2 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $O/sq4b -- $BENCH --only backend --steps 6 --warmup 2 > $O/sq4b.log 2>&1
# L2 hit rate of the window kernel's scratch traffic (requests that hit / miss in the XCD's L2)
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/tcc -- $BENCH --only backend --steps 6 --warmup 2 > $O/tcc.log 2>&1
# BASELINE configs[2] / configs[4] end to end (front-end at 1280x720 / 1920x1080 + the cooperative W = 20 / 30 window solve)
for leg in configs2 configs4; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$leg -- python $R/bench.py --leg $leg --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_$leg.log 2>&1
done
# instruction cache of the window kernel (one body of ~70 k instructions against 64 KB per CU pair)
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d $O/ic -- $BENCH --only backend --steps 6 --warmup 2 > $O/ic.log 2>&1
# the resident estimator path (landmark stores + window assembly on the device): per-kernel trace and timeline of one process
rocprofv3 --kernel-trace --stats -d $O/kt_res -- python $R/tools/time_estimator.py 512 24 > $O/estimator_kt.log 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_summary.py $(db kt) $O/kernel_trace.txt > /dev/null
python tools/rocpd_pmc_summary.py $(db fetch) $(db write) > $O/pmc_hbm.txt 2>&1
python tools/rocpd_pmc_summary.py $(db calib_fetch) $(db calib_write) > $O/pmc_calib.txt 2>&1
for k in 1 2 3 4 5; do python tools/rocpd_pmc_summary.py $(db sq$k) 2>&1 | grep vio_window >> $O/pmc_sq.txt; done
python tools/rocpd_pmc_summary.py $(db tcc) 2>&1 | grep vio_window >> $O/pmc_sq.txt
for k in 3b 4b; do python tools/rocpd_pmc_summary.py $(db sq$k) 2>&1 | grep vio_window | sed 's/^/max_iter=2: /' >> $O/pmc_sq.txt; done
for leg in configs2 configs4; do python tools/rocpd_summary.py $(db kt_$leg) $O/kernel_trace_$leg.txt > /dev/null; grep "^{\"workload\"" $O/bench_$leg.log | tail -1 > $O/bench_$leg.json; done
python tools/rocpd_pmc_summary.py $(db ic) 2>&1 | grep vio_window >> $O/pmc_sq.txt
for kb in 8 32 48 64 96 192 384; do $R/tools/microbench/bin/icache_probe_$kb; done > $O/icache_probe.txt 2>&1
python tools/rocpd_pmc_summary.py --json $O/pmc.json --workload "configs[1] x 512 sequences, prior 75" \
  --calib $(db calib_fetch) $(db calib_write) --fetch $(db fetch) --write $(db write) > /dev/null 2> $O/pmc_json.err
python tools/rocpd_summary.py $(db kt_res) $O/kernel_trace_resident_estimator.txt > /dev/null
python tools/rocpd_timeline.py $(db kt_res) 24 >> $O/kernel_trace_resident_estimator.txt 2>&1
( for n in 256 512 1024; do for r in 1 0; do echo "== time_estimator.py $n sequences, VIO_AMD_RESIDENT=$r"; VIO_AMD_RESIDENT=$r python tools/time_estimator.py $n 40 2>&1 | tail -2; done; done
  VIO_AMD_STORE_PROF=1 python tools/time_estimator.py 512 16 2>&1 | grep -A1 "store cycles" | tail -2
  g++ -O2 -std=c++17 -Iinclude tools/estimator_throughput.cpp -Lvins-mobile_amd/csrc -lvio_amd -Wl,-rpath,$R/vins-mobile_amd/csrc -lpthread -o /tmp/estimator_throughput
  python tools/estimator_dataset.py /tmp/est.bin 8 60 > /dev/null 2>&1
  for cfg in "256 1" "512 1" "256 2" "512 2"; do set -- $cfg; echo "== estimator_throughput (C++ driver): $1 sequences x $2 estimator objects"; /tmp/estimator_throughput /tmp/est.bin $1 $2 2>&1 | tail -1; done
  for n in 256 512; do echo "== time_pipeline.py $n sequences, asynchronous submit"; python tools/time_pipeline.py $n 30 2 1 2>&1 | tail -1; done ) > $O/estimator_paths.txt 2>&1
python tools/time_backend.py --path=single 1 256 512 1024 > $O/stage_cycles.txt 2>&1
VIO_AMD_PROF_TID=64 python tools/time_backend.py 1 2>&1 | grep "stage cycles" | sed "s/^/clock on a panel wave: /" >> $O/stage_cycles.txt
$R/tools/microbench/bin/band_bench > $O/microbench.txt 2>&1
$R/tools/microbench/bin/mfma_share >> $O/microbench.txt 2>&1
python tools/time_large.py > $O/large_windows.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
rm -rf $O/sq3b $O/sq4b $O/kt_configs2 $O/kt_configs4 $O/ic $O/kt $O/fetch $O/write $O/calib_fetch $O/calib_write $O/sq1 $O/sq2 $O/sq3 $O/sq4 $O/sq5 $O/tcc $O/kt_res
ls -la $O
