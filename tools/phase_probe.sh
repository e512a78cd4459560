# First look at the phase path on the device: gpurun -- 'bash tools/phase_probe.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/phase_probe
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_backend_gpu.py -x -q -m gpu -k "not poisoned" > $O/tests.log 2>&1
tail -5 $O/tests.log
python tools/time_backend.py --path=single 1 8 64 256 512 1024 > $O/time_single.txt 2>&1
python tools/time_backend.py --path=phase 1 8 64 256 512 768 1024 > $O/time_phase.txt 2>&1
grep "path=" $O/time_single.txt $O/time_phase.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/time_backend.py --path=phase 512 > $O/kt.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) $O/kernel_trace.txt > /dev/null
head -12 $O/kernel_trace.txt
rm -rf $O/kt
