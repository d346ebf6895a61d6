# Per-kernel times and the kernel timeline of the N = 10000 factorisation (panel path, look-ahead): rocprofv3 kernel trace of
# tools/prof_fit.py C5.  Run on the GPU box through gpurun; results under gpurun_out/chol10k/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chol10k -- python3 tools/prof_fit.py C5 3 > gpurun_out/chol10k.log 2>&1
f=$(find gpurun_out/chol10k -name "*kernel_stats.csv" | tail -1)
test -n "$f" && cut -c1-200 "$f" | sed -n 1,14p
t=$(find gpurun_out/chol10k -name "*kernel_trace.csv" | tail -1)
test -n "$t" && python3 tools/chol_timeline.py "$t" > gpurun_out/chol10k_timeline.txt && tail -60 gpurun_out/chol10k_timeline.txt
