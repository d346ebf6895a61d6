#!/bin/bash
# Group ensemble kernel: tests, timings, rocprofv3 kernel stats of the C4 run, then a -DALABI_GROUP_PROF build for the phase stamps.
# Usage (on the GPU box): bash tools/run_group_profile.sh <tag>
cd "$(dirname "$0")/.."
tag=${1:-x}
out=gpurun_out/r3_group_$tag
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_ensemble_group.py -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
timeout -k 10 200 python tools/prof_group_kernel.py --paths group > $out/timing.log 2>&1; grep "half step" $out/timing.log
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o c4 -- python3 tools/prof_group_kernel.py --paths group --configs C4 > $out/rocprof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-220
# the instrumented object must not outlive this script: the Makefile does not track EXTRA, so every later `make` (and the incremental
# build in _lib.build()) would otherwise keep the kernel with its s_memrealtime stamps -- restore the normal build on ANY exit
restore() { (cd alabi_amd/csrc && rm -f ens_group.o && make -j16 > /dev/null 2>&1); }
trap restore EXIT
(cd alabi_amd/csrc && rm -f ens_group.o && make EXTRA=-DALABI_GROUP_PROF -j16 > /dev/null 2>&1)
(timeout -k 10 120 python tools/prof_group_phases.py C4; timeout -k 10 120 python tools/prof_group_phases.py C5) > $out/phases.log 2>&1; grep -v amdgpu.ids $out/phases.log
