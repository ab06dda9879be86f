#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/deep5 gpurun_out/last
SDM_FUZZ_INGEST=300 timeout -k 10 400 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q -k "random_sizes" > gpurun_out/deep5/ingest_sizes.log 2>&1; echo "ingest sizes rc=$?"; tail -2 gpurun_out/deep5/ingest_sizes.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/last/tests.txt 2>&1; echo "suite rc=$?"; tail -2 gpurun_out/last/tests.txt
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/last/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/last/smoke.txt
