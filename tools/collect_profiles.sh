# Round-end evidence: bench line, rocprofv3 kernel stats of the same command, FETCH/WRITE PMC passes (separate runs).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write.err
python tools/summarize_pmc.py $O/pmc_by_kernel.json $O/pmc_fetch $O/pmc_write
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/pmc_fetch $O/pmc_write
find $O/stats -name "*kernel_trace.csv" -delete
head -c 600 $O/bench.json; echo; head -8 $O/kernel_stats.csv
