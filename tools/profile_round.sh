set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_f
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f/kt -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_f/bench_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_f/fetch -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_f/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_f/write -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_f/bench_write.log 2>&1
cd $R
find gpurun_out/prof_f -name "*.db" | head
python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/prof_f/bench.json 2> gpurun_out/prof_f/bench.err
python tools/time_backend.py 1 256 > gpurun_out/prof_f/stage_cycles.txt 2>&1
python tools/time_preprocess.py > gpurun_out/prof_f/preprocess.txt 2>&1
python tools/time_estimator.py 256 30 > gpurun_out/prof_f/estimator.txt 2>&1
