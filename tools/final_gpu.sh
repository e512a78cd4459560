# gpurun -- 'bash tools/final_gpu.sh': whole GPU suite, smoke, torchrun world-1 bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/final
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) | tee gpurun_out/final/pytest_gpu.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) | tee gpurun_out/final/smoke.log
( timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 2> gpurun_out/final/torchrun.err | tail -1 ) > gpurun_out/final/bench_torchrun_world1.json
python - <<'P'
import json
d=json.loads(open('gpurun_out/final/bench_torchrun_world1.json').read().strip().splitlines()[-1])
print(d['value'], d['n_gpus'], json.dumps(d.get('multi_gpu',{}).get('replicas'))[:600])
P
