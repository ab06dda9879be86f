#!/bin/bash
# Round 4: stream-order reproducer, batched-ingest tests, the -m gpu suite, a bench line with the batched upload timings
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c
mkdir -p $O
timeout -k 10 300 orb-slam-free-space-carving_amd/lib/ubench_big_grid > $O/big_grid.txt 2>&1 || echo "big_grid rc $?"
grep "stream order" $O/big_grid.txt
timeout -k 10 300 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q > $O/ingest.log 2>&1 || { echo "ingest FAILED"; tail -40 $O/ingest.log; exit 1; }
tail -2 $O/ingest.log
( timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -5 $O/pytest.log
timeout -k 10 300 python bench.py --no-extra --cpu-kfs 0 > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -20 $O/bench.err; }
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4c/bench.json"))
for k in ("value","ms_per_step","host_upload_ms_per_keyframe","host_upload","value_pcie_inclusive","value_pcie_inclusive_pinned","value_pcie_inclusive_per_keyframe_calls","stage_ms_per_step"):
    print(k, d.get(k))
print(d["roofline"])
PY
