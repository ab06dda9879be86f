#!/bin/bash
# the headline line three times on one box (run-to-run spread; boxes differ by more): bash tools/debug/r05_repeat.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/repeat
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --no-extra --cpu-kfs 0 > gpurun_out/repeat/bench$i.json 2> gpurun_out/repeat/bench$i.err || tail -3 gpurun_out/repeat/bench$i.err
  python3 -c "
import json; d=json.load(open('gpurun_out/repeat/bench$i.json')); r=d['roofline']
print('run $i: value %.1f  cold %.1f  ms/step %.4f  K1 %.4f ms  frac %.4f  hbm_frac %.4f  traffic %.4f GB (%s)  streaming %.1f / pinned %.1f  pcie %.1f' % (d['value'], d['value_cold'], d['ms_per_step'], r['launch_ms'], r['frac'], r.get('hbm_frac', 0), (r['traffic'] or 0)/1e9, r.get('traffic_source','')[:4], d['value_streaming'], d['value_streaming_pinned'], d['value_pcie_inclusive']))"
done
