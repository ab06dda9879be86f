#!/bin/bash
# the other single-GPU BASELINE workloads as stand-alone bench lines (own live traffic passes): bash tools/debug/r05_standalone.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
timeout -k 10 300 python bench.py --kfs 256 --cpu-kfs 4 --no-extra > gpurun_out/final/bench_480p_256kf_n20.json 2> gpurun_out/final/sa.err || echo "256kf failed"
timeout -k 10 300 python bench.py --res 720p --kfs 256 --nbrs 7 --cpu-kfs 3 --no-extra > gpurun_out/final/bench_720p_256kf_n7.json 2>> gpurun_out/final/sa.err || echo "720p failed"
timeout -k 10 300 python bench.py --res 1080p --kfs 64 --nbrs 7 --cpu-kfs 2 --no-extra > gpurun_out/final/bench_1080p_64kf_n7.json 2>> gpurun_out/final/sa.err || echo "1080p failed"
for f in 480p_256kf_n20 720p_256kf_n7 1080p_64kf_n7; do python3 -c "
import json; d=json.load(open('gpurun_out/final/bench_$f.json')); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], 'K1', r['launch_ms'], 'frac', r['frac'], 'hbm_frac', r.get('hbm_frac'), (r.get('traffic_source') or r.get('traffic_note',''))[:30], 'streaming', d.get('value_streaming'), 'cpu', d['cpu_baseline']['value'])"; done
