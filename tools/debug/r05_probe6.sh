#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/probe6
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/tests.txt 2>&1 || { tail -40 $OUT/tests.txt; exit 1; }
tail -3 $OUT/tests.txt
AHEAD=2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 tools/debug/streaming_probe.py > $OUT/probe.txt 2>&1
python3 tools/debug/streaming_account.py $OUT/tr 10 | tee $OUT/account.txt
python3 tools/debug/streaming_block_dump.py $OUT/tr > $OUT/block.txt 2>&1
rm -rf $OUT/tr
timeout -k 10 300 python3 bench.py --no-extra --cpu-kfs 0 > $OUT/bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
python3 -c "
import json; d=json.load(open('$OUT/bench.json'))
print({k:d[k] for k in d if k.startswith('value') or k=='ms_per_step'}); print(d.get('host_upload'))"
timeout -k 10 200 python3 tools/latency.py 7 > $OUT/latency.txt 2>&1; tail -12 $OUT/latency.txt
