#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/probe3
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_ingest.py tests/test_gpu_longscan.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_statefuzz.py -x -q -m gpu > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -3 $OUT/tests.txt
AHEAD=2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 tools/debug/streaming_probe.py > $OUT/probe.txt 2>&1
tail -2 $OUT/probe.txt
python3 tools/debug/streaming_account.py $OUT/tr 10 | tee $OUT/account.txt
python3 tools/debug/streaming_block_dump.py $OUT/tr > $OUT/block.txt 2>&1
rm -rf $OUT/tr
AHEAD=2 timeout -k 10 120 python3 tools/debug/streaming_probe.py 2>&1 | tail -2 | tee $OUT/probe_noprof.txt
