#!/bin/bash
# Round 4 counters, first half: headline workload and the two 256-keyframe workloads (tools/r04_pmc_b.sh: the hard-data lines)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc.sh r04 > gpurun_out/pmc_r04.log 2>&1 || echo "pmc r04 failed"
bash tools/pmc.sh r04_480p_256kf --kfs 256 > gpurun_out/pmc_r04_256.log 2>&1 || echo "pmc 256 failed"
bash tools/pmc.sh r04_720p_256kf --res 720p --kfs 256 --nbrs 7 > gpurun_out/pmc_r04_720.log 2>&1 || echo "pmc 720 failed"
for d in gpurun_out/pmc_r04 gpurun_out/pmc_r04_480p_256kf gpurun_out/pmc_r04_720p_256kf; do echo $d; head -c 300 $d/traffic.json; echo; done
