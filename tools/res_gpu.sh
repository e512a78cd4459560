# gpurun -- 'bash tools/res_gpu.sh [pytest -k expr]': the resident-path tests on the GPU box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_estimator.py -x -q -m gpu -k "${1:-resident}" 2>&1 | tail -40 | tee gpurun_out/res_gpu.log
