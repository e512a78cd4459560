# Look at the phase path on the device: gpurun -- 'bash tools/phase_probe.sh [tests]'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/phase_probe
rm -rf $O; mkdir -p $O
cd $R
if [ "$1" = "tests" ]; then
  timeout 900 python -m pytest tests/test_backend_gpu.py -x -q -m gpu -k "not poisoned" > $O/tests.log 2>&1
  grep -v "marginaliz\|release" $O/tests.log | tail -5
fi
python tools/time_backend.py --path=single 1 512 2>&1 | grep "path=" > $O/time_single.txt
python tools/time_backend.py --path=phase 1 8 256 512 768 1024 2>&1 | grep "path=" > $O/time_phase.txt
cat $O/time_single.txt $O/time_phase.txt
python tools/phase_stages.py phase 1 > $O/stages.txt 2>&1
python tools/phase_stages.py phase 512 >> $O/stages.txt 2>&1
VIO_AMD_PROF_TID=192 python tools/phase_stages.py phase 512 >> $O/stages.txt 2>&1
python tools/phase_stages.py single 512 >> $O/stages.txt 2>&1
grep "path=" $O/stages.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/time_backend.py --path=phase 512 > $O/kt.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) $O/kernel_trace.txt > /dev/null
head -12 $O/kernel_trace.txt
rm -rf $O/kt
