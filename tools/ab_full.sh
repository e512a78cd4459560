# gpurun -- 'bash tools/ab_full.sh a.so b.so ...': all non-zero stage cycles of one window + 1/256/512-window times of each library variant (files under csrc/)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  cp vins-mobile_amd/csrc/$v vins-mobile_amd/csrc/libvio_amd.so
  echo "== $v"
  python tools/time_backend.py --prof-batch=1 1 256 512 2>&1 | grep "stage\|path=" | tr ',' '\n' | grep -v "=0(" | tr '\n' ',' | sed 's/path=single/\n/g; s/(wall[^>]*>//g; s/us\/solve[^;]*;//g'
  echo
done
