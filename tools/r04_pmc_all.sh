#!/bin/bash
# Round 4 counters: tools/pmc.sh for the headline workload, the two 256-keyframe workloads and the three hard-data lines
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc.sh r04 > gpurun_out/pmc_r04.log 2>&1 || echo "pmc r04 failed"
bash tools/pmc.sh r04_480p_256kf --kfs 256 > gpurun_out/pmc_r04_256.log 2>&1 || echo "pmc 256 failed"
bash tools/pmc.sh r04_720p_256kf --res 720p --kfs 256 --nbrs 7 > gpurun_out/pmc_r04_720.log 2>&1 || echo "pmc 720 failed"
bash tools/pmc.sh r04_noise --noise > gpurun_out/pmc_r04_noise.log 2>&1 || echo "pmc noise failed"
bash tools/pmc.sh r04_outliers2 --outliers 2 > gpurun_out/pmc_r04_out.log 2>&1 || echo "pmc outliers failed"
bash tools/pmc.sh r04_disp10 --disparity 10 > gpurun_out/pmc_r04_disp.log 2>&1 || echo "pmc disp10 failed"
ls gpurun_out/pmc_r04*/
for d in gpurun_out/pmc_r04*/; do echo $d; cat $d/traffic.json | head -c 400; echo; done
