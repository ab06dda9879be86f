# round 5: deep fuzz of the new K1 paths, one gpurun call each: bash tools/run_deepfuzz_r05.sh a | b
#   a: geometry fuzz (every third case under non-default thresholds) in the default scan mode and with the mask scan forced
#   b: the same with every open pixel deferred + forced mask scan, call-sequence fuzz, random sizes (whole path; the batched
#      pre-pass in both launch shapes), the 16384 x 8192 check
set -u
mkdir -p gpurun_out/deep5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${1:-a}" = a ]; then
SDM_FUZZ_GEOM=700 timeout -k 10 480 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/deep5/geom.log 2>&1; echo "geom rc=$?"; tail -2 gpurun_out/deep5/geom.log
SDM_SCAN_MODE=2 SDM_FUZZ_GEOM=700 timeout -k 10 560 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/deep5/geom_mask.log 2>&1; echo "geom forced-mask rc=$?"; tail -2 gpurun_out/deep5/geom_mask.log
else
SDM_SCAN_MODE=2 SDM_OPEN_QUOTA=0 SDM_OPEN_INPLACE=65 SDM_FUZZ_GEOM=300 timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/deep5/geom_mask_defer.log 2>&1; echo "geom forced-mask defer-all rc=$?"; tail -2 gpurun_out/deep5/geom_mask_defer.log
SDM_FUZZ_SEEDS=150 timeout -k 10 300 python -m pytest tests/test_gpu_statefuzz.py -m gpu -x -q > gpurun_out/deep5/state.log 2>&1; echo "state rc=$?"; tail -2 gpurun_out/deep5/state.log
SDM_SCAN_MODE=2 SDM_FUZZ_SIZES=120 timeout -k 10 250 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sizes" > gpurun_out/deep5/sizes_mask.log 2>&1; echo "sizes forced-mask rc=$?"; tail -2 gpurun_out/deep5/sizes_mask.log
SDM_FUZZ_INGEST=300 timeout -k 10 300 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q -k "random_sizes" > gpurun_out/deep5/ingest_sizes.log 2>&1; echo "ingest sizes rc=$?"; tail -2 gpurun_out/deep5/ingest_sizes.log
SDM_SCAN_MODE=2 timeout -k 10 250 python tools/huge_check.py > gpurun_out/deep5/huge_mask.log 2>&1; echo "huge forced-mask rc=$?"; tail -3 gpurun_out/deep5/huge_mask.log
fi
