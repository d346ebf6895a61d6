# Round-end evidence: the bench line, rocprofv3 kernel stats of the same command, and PMC passes (separate runs, --pmc only:
# FETCH_SIZE, WRITE_SIZE, matrix-core counters, wave / issue counters).  Run on the GPU box from the repo root:
#   bash tools/collect_profiles.sh        -> gpurun_out/final/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
P="python3 bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $P > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $P > /dev/null 2> $O/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- $P > /dev/null 2> $O/pmc_mfma.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_WAVES --output-format csv -d $O/pmc_wave -- $P > /dev/null 2> $O/pmc_wave.err
python tools/summarize_pmc.py $O/pmc_by_kernel.json $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_wave
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_wave
find $O/stats -name "*kernel_trace.csv" -delete
head -c 600 $O/bench.json; echo; head -12 $O/kernel_stats.csv
