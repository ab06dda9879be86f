#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/probe4
rm -rf $OUT && mkdir -p $OUT
for ps in 1 0; do
POINTSET=$ps AHEAD=2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$ps -- python3 tools/debug/streaming_probe.py > $OUT/probe$ps.txt 2>&1
echo "pointset $ps" | tee -a $OUT/account.txt
python3 tools/debug/streaming_account.py $OUT/tr$ps 10 | tee -a $OUT/account.txt
python3 tools/debug/streaming_block_dump.py $OUT/tr$ps > $OUT/block$ps.txt 2>&1
rm -rf $OUT/tr$ps
done
timeout -k 10 300 python3 bench.py --no-extra --cpu-kfs 0 > $OUT/bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
python3 -c "
import json; d=json.load(open('$OUT/bench.json'))
print({k:d[k] for k in d if k.startswith('value') or k=='ms_per_step'})"
