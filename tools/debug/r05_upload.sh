#!/bin/bash
cd $GRAFT_REPO_ROOT
nproc
for t in 4 8 12; do echo "stager threads $t"; SDM_STAGER_THREADS=$t timeout -k 10 200 python3 tools/debug/upload_probe.py 2>&1 | grep -v amdgpu; done
