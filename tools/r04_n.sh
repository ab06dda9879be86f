#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4n
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_golden.py tests/test_gpu_statefuzz.py -m gpu -x -q > $O/tests.log 2>&1 || { echo FAILED; tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 100 python tools/latency.py 7 2>&1 | grep -v amdgpu.ids
timeout -k 10 100 python tools/latency.py 20 2>&1 | grep -v amdgpu.ids
