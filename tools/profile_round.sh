set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_i
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_i/kt -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_i/bench_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_i/fetch -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_i/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_i/write -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_i/bench_write.log 2>&1
cd $R
find gpurun_out/prof_i -name "*.db" | head
python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/prof_i/bench.json 2> gpurun_out/prof_i/bench.err
python tools/time_backend.py 1 256 > gpurun_out/prof_i/stage_cycles.txt 2>&1
python tools/time_preprocess.py > gpurun_out/prof_i/preprocess.txt 2>&1
python tools/time_estimator.py 256 30 > gpurun_out/prof_i/estimator.txt 2>&1
