export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
: > gpurun_out/res_time.log
g++ -O2 -std=c++17 -Iinclude tools/estimator_throughput.cpp -Lvins-mobile_amd/csrc -lvio_amd -Wl,-rpath,$R/vins-mobile_amd/csrc -lpthread -o /tmp/estimator_throughput 2>&1 | tail -3
python tools/estimator_dataset.py /tmp/est.bin 8 60 2>&1 | tail -2
for cfg in "256 1" "128 2" "256 2" "512 1" "128 4" "256 4" "512 2"; do
  set -- $cfg
  echo "== $1 sequences x $2 estimator objects" | tee -a gpurun_out/res_time.log
  timeout 600 /tmp/estimator_throughput /tmp/est.bin $1 $2 2>&1 | tail -2 | tee -a gpurun_out/res_time.log
done
