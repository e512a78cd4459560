# gpurun -- 'bash tools/tu_prof.sh': stage cycles of track_update_kernel (sequence 0's workgroup) during the bench's front-end steps
cd $GRAFT_REPO_ROOT
VIO_AMD_TU_PROF=1 python bench.py --quick --no-cpu-baseline --only frontend --steps 4 --warmup 2 2>&1 | grep -A2 "track_update cycles" | tail -3
