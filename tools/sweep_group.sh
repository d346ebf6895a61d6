#!/bin/bash
# Sweep of the group ensemble kernel's blocking switches (one line per setting); output to stdout.
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python tools/prof_group_kernel.py --configs ${CFG:-C4} --paths group ${EXTRA} 2>&1 | grep "us per half"; }
CFG=C4 EXTRA="--N 512" run ALABI_ENS_GROUP_THREADS=256
CFG=C4 EXTRA="--N 512" run ALABI_ENS_GROUP_THREADS=512
CFG=C4 EXTRA="--N 2000" run ALABI_ENS_GROUP_THREADS=512
CFG=C4 EXTRA="" run ALABI_ENS_GROUP_THREADS=256
CFG=C4 EXTRA="" run ALABI_ENS_GROUP_THREADS=512
CFG=C4 EXTRA="" run ALABI_ENS_GROUP_THREADS=512 ALABI_ENS_GROUP_XCD=0
CFG=C4 EXTRA="" run ALABI_ENS_GROUP_Q=2
CFG=C4 EXTRA="" run ALABI_ENS_GROUP_G=4
CFG=C5 EXTRA="--N 1024" run ALABI_ENS_GROUP_THREADS=512
CFG=C5 EXTRA="" run ALABI_ENS_GROUP_THREADS=256
CFG=C5 EXTRA="" run ALABI_ENS_GROUP_Q=8
CFG=C5 EXTRA="" run ALABI_ENS_GROUP_XCD=0
CFG=C3 EXTRA="" run ALABI_ENS_GROUP_THREADS=256
CFG=C3 EXTRA="" run ALABI_ENS_GROUP_THREADS=512
