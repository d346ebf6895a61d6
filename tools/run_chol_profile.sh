# Cholesky per-kernel times: rocprofv3 kernel stats of tools/prof_fit.py (run on the GPU box through gpurun).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/prof_fit.py C3 20 2>&1 | tail -1 &&
timeout -k 10 200 python tools/prof_fit.py C5 3 2>&1 | tail -1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chol -- python3 tools/prof_fit.py C3 10 > gpurun_out/chol.log 2>&1
find gpurun_out/chol -name "*kernel_stats.csv" | while read f; do grep -i "potrf\|trsm\|syrk\|chol" "$f" | cut -c1-160; done
