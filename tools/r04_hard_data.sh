#!/bin/bash
# Round 4, first GPU call: the launch-size reproducer, then kernel-trace stats and PMC counters of the three hard-data
# bench lines (i.i.d.-noise images, 2 outlier neighbours, 10 px/keyframe baseline).  Run on the GPU box from the repo root.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4a
mkdir -p $O
timeout -k 10 240 orb-slam-free-space-carving_amd/lib/ubench_big_grid > $O/big_grid.txt 2>&1 || echo "big_grid rc $?"
cat $O/big_grid.txt
for cfg in "noise --noise" "outliers2 --outliers 2" "disp10 --disparity 10"; do
  set -- $cfg; tag=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 bench.py --steps 10 --warmup 2 --cpu-kfs 0 --no-extra --no-stats "$@" > $O/kt_$tag.log 2>&1 || echo "kt $tag failed"
  f=$(find $O/kt_$tag -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep "sdm::" "$f") > $O/kernel_stats_$tag.csv; cat $O/kernel_stats_$tag.csv | cut -c1-200; fi
  rm -rf $O/kt_$tag
  bash tools/pmc.sh r04_$tag "$@" > $O/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
  echo "== $tag done"
done
