#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4l
mkdir -p $O
( timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -30 $O/pytest.log | cut -c1-220
grep -q "rc=0" $O/pytest.log || exit 1
SDM_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus 2 --kfs 16 --steps 5 --warmup 1 > $O/rehearse2.json 2> $O/rehearse2.err || { echo "rehearsal failed"; tail -20 $O/rehearse2.err; }
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4l/rehearse2.json"))
for k in ("value","value_exchange","value_baseline_literal","value_allgather_full","exchange_note","exchange_ms_per_step","rehearsal"):
    print(k, d.get(k))
PY
timeout -k 10 100 python tools/latency.py 7 > $O/latency.txt 2>&1; timeout -k 10 100 python tools/latency.py 20 >> $O/latency.txt 2>&1; cat $O/latency.txt
