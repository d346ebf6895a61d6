#!/bin/bash
# Sweep of the group ensemble kernel's switches (one line per setting); output to stdout.
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python tools/prof_group_kernel.py --configs ${CFG:-C4} --paths group ${EXTRA} 2>&1 | grep "us per half"; }
for dly in 0 2 4 6 8 10 12 16 20; do CFG=C4 run ALABI_ENS_GROUP_POLL_DELAY=$dly; done
for dly in 0 4 8 16 24; do CFG=C5 run ALABI_ENS_GROUP_POLL_DELAY=$dly; done
