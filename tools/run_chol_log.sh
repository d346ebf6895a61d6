# Event log of the CHAIN tasks of the task-queue Cholesky (ALABI_CHOL_LOG build: time stamps by plain stores, no counters on the chain).
# Run on the GPU box through gpurun; the product build is restored on exit.   usage: bash tools/run_chol_log.sh [N ...]
cd $GRAFT_REPO_ROOT
restore() { (cd alabi_amd/csrc && rm -f gp_cholesky.o && make -j16 > /dev/null 2>&1); }
trap restore EXIT
(cd alabi_amd/csrc && rm -f gp_cholesky.o && make EXTRA=-DALABI_CHOL_LOG -j16 > /dev/null 2>&1) || exit 1
ALABI_CHOL_LOG_PRINT=1 ALABI_CHOL_TASKS=1 timeout -k 10 300 python tools/prof_chol_sizes.py ${@:-2000} 2>&1 | grep -v amdgpu.ids | tail -22
