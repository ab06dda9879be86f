#!/bin/bash
# kernel trace of the streaming probe + per-block accounting (on the GPU box, from the repo root)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/stracc
rm -rf $OUT && mkdir -p $OUT
AHEAD=2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 tools/debug/streaming_probe.py > $OUT/probe.txt 2>&1
cat $OUT/probe.txt | tail -3
python3 tools/debug/streaming_account.py $OUT/tr 10 | tee $OUT/account.txt
rm -rf $OUT/tr
