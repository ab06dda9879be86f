#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/probe2
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 240 python3 tools/debug/cumask_probe.py > $OUT/cumask.txt 2>&1
cat $OUT/cumask.txt
AHEAD=2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 tools/debug/streaming_probe.py > $OUT/probe.txt 2>&1
python3 tools/debug/streaming_block_dump.py $OUT/tr > $OUT/block.txt 2>&1
rm -rf $OUT/tr
