# gpurun -- 'bash tools/res_prof.sh [sequences]': kernel timeline of the resident estimator path
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-512}
O=$R/gpurun_out/res_prof
rm -rf $O; mkdir -p $O
cd /tmp
VIO_AMD_RESIDENT=${2:-1} rocprofv3 --kernel-trace --stats -d $O/t -- python $R/tools/time_estimator.py $N 24 > $O/trace.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/t -name "*.db" | head -1) 2>&1 | head -12 | tee $O/kernels.txt
python tools/rocpd_timeline.py $(find $O/t -name "*.db" | head -1) 40 2>&1 | tee $O/timeline.txt
rm -rf $O/t
