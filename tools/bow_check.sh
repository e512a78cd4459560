# The bag-of-words database on the GPU box: parity tests + the bench leg. usage: gpurun -- 'bash tools/bow_check.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_dbow.py -x -q -m gpu 2>&1 | tail -5
python - <<'P' 2>&1 | tail -5
import importlib, json, sys
sys.path.insert(0, ".")
import bench
pkg = importlib.import_module("vins-mobile_amd")
print(json.dumps(bench.loop_closure(pkg)["bow_query"], indent=1))
P
