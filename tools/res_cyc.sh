export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
VIO_AMD_STORE_PROF=1 python tools/time_estimator.py ${1:-512} 16 2>&1 | grep -A1 "store cycles" | tail -6
