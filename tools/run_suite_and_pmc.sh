# final pass A: the whole -m gpu suite, then the PMC counter runs of the three profiled configurations
set -u
bash tools/run_gpu_suite.sh || exit 1
bash tools/pmc.sh r03 > gpurun_out/pmc_r03.log 2>&1 && tail -5 gpurun_out/pmc_r03.log
bash tools/pmc.sh r03_480p_256kf --kfs 256 > gpurun_out/pmc_r03_480p_256kf.log 2>&1 && tail -3 gpurun_out/pmc_r03_480p_256kf.log
bash tools/pmc.sh r03_720p_256kf --res 720p --kfs 256 --nbrs 7 > gpurun_out/pmc_r03_720p_256kf.log 2>&1 && tail -3 gpurun_out/pmc_r03_720p_256kf.log
ls gpurun_out/pmc_r03*/
