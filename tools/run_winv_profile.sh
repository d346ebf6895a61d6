# Per-kernel times of the L^-1 build (rocprofv3 kernel stats of tools/prof_winv.py C3).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/winv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/winv -- python3 tools/prof_winv.py ${1:-C3} > gpurun_out/winv.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/winv/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("inv_", "ident_seed", "retile", "predict_var_ws", "predict_var_small")):
        print("%-70s calls %4s avg %9.1f us min %9.1f max %9.1f" % (r["Name"].split("(")[0][-68:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
find gpurun_out/winv -name "*kernel_trace.csv" -delete
