# one-off deep fuzz of the final build, in two gpurun calls (each inside the 1200-s limit): bash tools/run_deepfuzz.sh a | b
set -u
mkdir -p gpurun_out/deep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${1:-a}" = a ]; then
SDM_FUZZ_GEOM=1200 timeout -k 10 520 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/deep/geom.log 2>&1; echo "geom rc=$?"; tail -2 gpurun_out/deep/geom.log
SDM_OPEN_QUOTA=0 SDM_OPEN_INPLACE=65 SDM_FUZZ_GEOM=600 timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/deep/geom_defer.log 2>&1; echo "geom defer-all rc=$?"; tail -2 gpurun_out/deep/geom_defer.log
SDM_OPEN_QUOTA=0 SDM_OPEN_INPLACE=65 SDM_OPEN_CAPACITY=64 SDM_FUZZ_GEOM=300 timeout -k 10 200 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/deep/geom_tiny.log 2>&1; echo "geom tiny-list rc=$?"; tail -2 gpurun_out/deep/geom_tiny.log
else
SDM_OPEN_QUOTA=0 SDM_OPEN_INPLACE=1 SDM_FUZZ_GEOM=300 timeout -k 10 200 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/deep/geom_inplace.log 2>&1; echo "geom count-in-place rc=$?"; tail -2 gpurun_out/deep/geom_inplace.log
SDM_FUZZ_SEEDS=250 timeout -k 10 420 python -m pytest tests/test_gpu_statefuzz.py -m gpu -x -q > gpurun_out/deep/state.log 2>&1; echo "state rc=$?"; tail -2 gpurun_out/deep/state.log
SDM_FUZZ_SIZES=120 timeout -k 10 250 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sizes" > gpurun_out/deep/sizes.log 2>&1; echo "sizes rc=$?"; tail -2 gpurun_out/deep/sizes.log
timeout -k 10 250 python tools/huge_check.py > gpurun_out/deep/huge.log 2>&1; echo "huge rc=$?"; tail -3 gpurun_out/deep/huge.log
fi
