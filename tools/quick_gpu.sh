# gpurun -- 'bash tools/quick_gpu.sh [tests] [large]': backend GPU tests (optional) + window-kernel timings
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quick
rm -rf $O; mkdir -p $O
cd $R
for a in "$@"; do
  if [ "$a" = "tests" ]; then
    timeout 900 python -m pytest tests/test_backend_gpu.py tests/test_closed_loop.py -x -q -m gpu -k "not poisoned" > $O/tests.log 2>&1
    grep -v "marginaliz\|release" $O/tests.log | tail -4
  fi
  if [ "$a" = "large" ]; then python tools/time_large.py 2>&1 | grep "kernel" ; fi
done
python tools/time_backend.py --path=single 1 8 256 512 1024 2>&1 | grep "path=\|stage" > $O/time_single.txt
cat $O/time_single.txt | cut -c1-1200
