# gpurun -- 'bash tools/ab_large.sh a.so b.so ...': the large-window kernel times (tools/time_large.py) with each library variant (files under csrc/)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp vins-mobile_amd/csrc/libvio_amd.so /tmp/lib_keep.so
for rep in 1 2; do
for v in "$@"; do
  cp vins-mobile_amd/csrc/$v vins-mobile_amd/csrc/libvio_amd.so
  echo "== $v: $(python tools/time_large.py ${BATCH:-64} 2>&1 | grep kernel | sed 's/; iters.*//' | tr '\n' ' ')"
done
done
cp /tmp/lib_keep.so vins-mobile_amd/csrc/libvio_amd.so
