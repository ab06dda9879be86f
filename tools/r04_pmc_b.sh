#!/bin/bash
# Round 4 counters, second half: i.i.d.-noise images, 2 outlier neighbours, long baseline
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc.sh r04_noise --noise > gpurun_out/pmc_r04_noise.log 2>&1 || echo "pmc noise failed"
bash tools/pmc.sh r04_outliers2 --outliers 2 > gpurun_out/pmc_r04_out.log 2>&1 || echo "pmc outliers failed"
bash tools/pmc.sh r04_disp10 --disparity 10 > gpurun_out/pmc_r04_disp.log 2>&1 || echo "pmc disp10 failed"
for d in gpurun_out/pmc_r04_noise gpurun_out/pmc_r04_outliers2 gpurun_out/pmc_r04_disp10; do echo $d; head -c 300 $d/traffic.json; echo; done
