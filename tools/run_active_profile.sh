# Per-kernel times of one active-learning iteration (rocprofv3 kernel stats of tools/prof_active_train.py).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/active
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/active -- python3 tools/prof_active_train.py > gpurun_out/active.log 2>&1
tail -3 gpurun_out/active.log
find gpurun_out/active -name "*kernel_stats.csv" | while read f; do head -30 "$f" | cut -c1-70,'-'  | awk -F'",' '{n=split($1,a,"("); printf "%-60s %s\n", substr(a[1],2,58), $2}'; done
find gpurun_out/active -name "*kernel_trace.csv" -delete
