#!/bin/bash
# round 5 counters (tools/pmc.sh per workload): bash tools/r05_pmc.sh a | b | c
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift; bash tools/pmc.sh $tag "$@" > gpurun_out/pmc_$tag.log 2>&1 || echo "pmc $tag failed"; echo $tag; head -c 260 gpurun_out/pmc_$tag/traffic.json; echo; }
case "${1:-a}" in
a) run r05; run r05_480p_256kf --kfs 256; run r05_720p_256kf --res 720p --kfs 256 --nbrs 7 ;;
b) run r05_noise --noise; run r05_outliers2 --outliers 2; run r05_disp10 --disparity 10 ;;
c) run r05_spread03 --prior-spread 0.3; run r05_disp10_spread03 --disparity 10 --prior-spread 0.3; run r05_strip --scene strip --roll 5 --prior-spread 0.3 ;;
esac
