# gpurun -- 'bash tools/profile_pmc.sh <tag>': the HBM counter passes of tools/profile_round.sh alone (FETCH_SIZE / WRITE_SIZE of the bench and of
# the calibration streams -> pmc_hbm.txt, pmc_calib.txt, pmc.json) + the instruction-cache probe and the microbenchmarks
export TMPDIR=/tmp
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
BENCH="python $R/bench.py --quick --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- $BENCH --steps 6 --warmup 2 > $O/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- $BENCH --steps 6 --warmup 2 > $O/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib_fetch -- $R/tools/microbench/bin/pmc_calib > $O/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calib_write -- $R/tools/microbench/bin/pmc_calib > $O/calib_write.log 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_pmc_summary.py $(db fetch) $(db write) > $O/pmc_hbm.txt 2>&1
python tools/rocpd_pmc_summary.py $(db calib_fetch) $(db calib_write) > $O/pmc_calib.txt 2>&1
python tools/rocpd_pmc_summary.py --json $O/pmc.json --workload "configs[1] x 512 sequences, prior 75" \
  --calib $(db calib_fetch) $(db calib_write) --fetch $(db fetch) --write $(db write) > /dev/null 2> $O/pmc_json.err
for kb in 8 32 48 64 96 192 384; do $R/tools/microbench/bin/icache_probe_$kb; done > $O/icache_probe.txt 2>&1
$R/tools/microbench/bin/band_bench > $O/microbench.txt 2>&1
$R/tools/microbench/bin/mfma_share >> $O/microbench.txt 2>&1
rm -rf $O/fetch $O/write $O/calib_fetch $O/calib_write
cat $O/pmc_json.err; head -c 600 $O/pmc.json
