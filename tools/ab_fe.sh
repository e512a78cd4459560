# gpurun -- 'bash tools/ab_fe.sh a.so b.so ...': front-end kernel times (rocprofv3 kernel trace of the bench's front-end half) with each library variant (files under csrc/)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
cp vins-mobile_amd/csrc/libvio_amd.so /tmp/lib_keep.so
for v in "$@"; do
  cp vins-mobile_amd/csrc/$v vins-mobile_amd/csrc/libvio_amd.so
  O=/tmp/abfe_$v; rm -rf $O; mkdir -p $O
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --quick --no-cpu-baseline --only frontend --steps 30 --warmup 3 > $O/bench.log 2>&1)
  python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) $O/kernel_trace.txt > /dev/null
  echo "== $v: $(grep -v "rocclr\|^pct" $O/kernel_trace.txt | awk '{n=$0; sub(/^[^a-z(]*/, "", n); split(n, a, "("); k=a[1]; if (k=="") k=a[2]; printf "%s x%s %s us | ", substr(k, length(k)-24), $2, $4}')"
done
cp /tmp/lib_keep.so vins-mobile_amd/csrc/libvio_amd.so
