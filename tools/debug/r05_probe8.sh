#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/probe8
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_ingest.py tests/test_gpu_longscan.py -x -q -m gpu > $OUT/tests.txt 2>&1 || { tail -40 $OUT/tests.txt; exit 1; }
tail -3 $OUT/tests.txt
for rep in 1 2; do
for dense in 1 0; do
SDM_DENSE_HANDOVER=$dense timeout -k 10 300 python3 bench.py --no-extra --cpu-kfs 0 --no-streaming --no-live-pmc --no-stats > $OUT/bench$dense.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
python3 -c "
import json; d=json.load(open('$OUT/bench$dense.json'))
print('dense=$dense', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
done
for dense in 1 0; do
SDM_DENSE_HANDOVER=$dense timeout -k 10 300 python3 bench.py --no-extra --cpu-kfs 0 --no-streaming --no-live-pmc --no-stats --res 720p --kfs 256 --nbrs 7 > $OUT/bench720_$dense.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
python3 -c "
import json; d=json.load(open('$OUT/bench720_$dense.json'))
print('720p dense=$dense', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
