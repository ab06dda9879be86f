#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/probe5
rm -rf $OUT && mkdir -p $OUT
export SDM_LIB_PATH=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_noxyz.so
AHEAD=2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 tools/debug/streaming_probe.py > $OUT/probe.txt 2>&1
python3 tools/debug/streaming_account.py $OUT/tr 10 | tee -a $OUT/account.txt
rm -rf $OUT/tr
