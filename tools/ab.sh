# gpurun -- 'bash tools/ab.sh a.so b.so ...': the window kernel timed with each library variant on the same box
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  cp vins-mobile_amd/csrc/$v vins-mobile_amd/csrc/libvio_amd.so
  echo "== $v: $(python tools/time_backend.py 256 512 2>&1 | tail -2 | sed 's/path=auto //; s/(wall.*//' | tr '\n' ' ')"
done
done
