export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/extra
( VIO_AMD_POISON=1 timeout 900 python -m pytest tests/test_backend_gpu.py -x -q -m gpu 2>&1 | grep -v "parallax\|marginaliz\|release\|initial succ" | tail -4 ) | tee gpurun_out/extra/poison_tests.log
