#!/bin/bash
# usage: tools/pmc_custom.sh <tag> "<group1>;<group2>;..." [bench args]   (counter names space-separated)
set -u
TAG=$1; GROUPS_STR=$2; shift 2
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
IFS=';' read -ra GRPS <<< "$GROUPS_STR"
for group in "${GRPS[@]}"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $group --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 2 --warmup 1 --cpu-kfs 0 --no-stats "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i ($group) failed"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT"/p[0-9]*
cat "$OUT/summary.txt"
