#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/probe7
rm -rf $OUT && mkdir -p $OUT
for v in "" nt; do
if [ -n "$v" ]; then export SDM_LIB_PATH=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_$v.so; fi
AHEAD=2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$v -- python3 tools/debug/streaming_probe.py > $OUT/probe$v.txt 2>&1
echo "variant [$v]" | tee -a $OUT/account.txt
python3 tools/debug/streaming_account.py $OUT/tr$v 10 | tee -a $OUT/account.txt
rm -rf $OUT/tr$v
done
