#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4g
mkdir -p $O
timeout -k 10 400 orb-slam-free-space-carving_amd/lib/ubench_big_grid > $O/big_grid.txt 2>&1 || echo "big_grid rc $?"
grep "K1-like" $O/big_grid.txt
# the {4,3} scan plan: parity first, then the A/B against the round-3 scan
bash tools/run_ab.sh default r3scan
