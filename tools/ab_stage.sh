# gpurun -- 'bash tools/ab_stage.sh a.so b.so ...': one-window stage cycles (p_gram, p_fact, total) + 512-window time of each library variant
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  cp vins-mobile_amd/csrc/$v vins-mobile_amd/csrc/libvio_amd.so
  echo "== $v: $(python tools/time_backend.py --path=single 1 512 2>&1 | grep "stage\|path=" | tr ',' '\n' | grep "p_gram\|p_fact\|total=\|kernel" | cut -c1-60 | tr '\n' ' ')"
done
