# PMC passes over the C4-sized half-step kernel (tools/prof_c4_half_step.py): instruction and wait counters of ens_half_multi_kernel.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c4_pmc; mkdir -p $O
P="python3 tools/prof_c4_half_step.py"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/a -- $P > /dev/null 2> $O/a.err &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/b -- $P > /dev/null 2> $O/b.err &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/c -- $P > /dev/null 2> $O/c.err
python tools/summarize_pmc.py gpurun_out/c4_pmc.json $O/a $O/b $O/c
rm -rf $O
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/c4_pmc.json"))
for k, v in d.items():
    if "ens_half_multi" in k:
        print(k[:70])
        for c, s in v.items():
            if isinstance(s, dict) and "mean" in s: print(f"   {c:28s} calls {s['calls']:5d} mean {s['mean']:.4g}")
PY
