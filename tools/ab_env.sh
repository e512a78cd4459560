# gpurun -- 'bash tools/ab_env.sh "VAR=a" "VAR=b" ...': the window kernel timed under each environment setting on the same box
# (two rounds; boxes of the pool differ by up to 1.5x, only numbers from one call compare)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  echo "== $v: $(env $v python tools/time_backend.py 1 256 512 2>&1 | grep 'path=' | sed 's/path=single //; s/(wall.*//' | tr '\n' ' ')"
done
done
