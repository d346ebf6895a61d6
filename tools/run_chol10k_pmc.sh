# PMC passes over the N = 10000 factorisation (tools/prof_fit.py C5 1): HBM/fabric traffic and matrix-core occupancy of
# syrk_panel_kernel.  Separate --pmc runs, nothing else traced.  Results: gpurun_out/chol10k_pmc.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/chol10k_pmc; mkdir -p $O
P="python3 tools/prof_fit.py C5 1"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $P > /dev/null 2> $O/fetch.err &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $P > /dev/null 2> $O/write.err &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- $P > /dev/null 2> $O/mfma.err &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $O/wave -- $P > /dev/null 2> $O/wave.err &&
python tools/summarize_pmc.py gpurun_out/chol10k_pmc.json $O/fetch $O/write $O/mfma $O/wave
rm -rf $O
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/chol10k_pmc.json"))
for k, v in d.items():
    if "syrk_panel" in k or "trsm_panel" in k or "syrk_update" in k:
        print(k[:60])
        for c, s in v.items():
            if isinstance(s, dict) and "mean" in s: print(f"   {c:32s} calls {s['calls']:5d} mean {s['mean']:.4g} max {s['max']:.4g} total {s['total']:.4g}")
PY
