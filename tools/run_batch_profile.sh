# rocprofv3 kernel stats + PMC passes of the batched CV fit (tools/prof_batch_cv.py, B = 500 jobs of N = 1600, d = 10).
# Run on the GPU box from the repo root: bash tools/run_batch_profile.sh  -> gpurun_out/batch_prof/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/batch_prof; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/prof_batch_cv.py 500 > $O/run.log 2> $O/rocprof.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/stats -name "*kernel_trace.csv" -delete
P="python3 tools/prof_batch_cv.py 500"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $P > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $P > /dev/null 2> $O/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- $P > /dev/null 2> $O/pmc_mfma.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_WAVES --output-format csv -d $O/pmc_wave -- $P > /dev/null 2> $O/pmc_wave.err
python tools/summarize_pmc.py $O/pmc_by_kernel.json $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_wave
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_wave $O/stats
head -14 $O/kernel_stats.csv; tail -6 $O/run.log
