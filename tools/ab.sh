# gpurun -- 'bash tools/ab.sh a.so b.so ...': the window kernel timed with each library variant (files under csrc/) on the same box
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp vins-mobile_amd/csrc/libvio_amd.so /tmp/lib_keep.so
for rep in 1 2; do
for v in "$@"; do
  cp vins-mobile_amd/csrc/$v vins-mobile_amd/csrc/libvio_amd.so
  echo "== $v: $(python tools/time_backend.py 1 256 512 2>&1 | grep 'path=' | sed 's/path=single //; s/(wall.*//' | tr '\n' ' ')"
done
done
cp /tmp/lib_keep.so vins-mobile_amd/csrc/libvio_amd.so
