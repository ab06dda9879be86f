# Round 5: parity of K1's two scans + their K1 times on short and long ranges, one GPU box.
# usage (GPU box): bash tools/run_longscan_ab.sh [tag ...]   (tags under lib/variants; none = the shipped library only)
set -u
out=gpurun_out/longscan
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_longscan.py -m gpu -x -q > $out/tests_longscan.log 2>&1 || { echo LONGSCAN FAILED; tail -30 $out/tests_longscan.log; exit 1; }
tail -1 $out/tests_longscan.log
SDM_SCAN_MODE=2 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -m gpu -x -q > $out/tests_forced_mask.log 2>&1 || { echo FORCED-MASK FAILED; tail -30 $out/tests_forced_mask.log; exit 1; }
tail -1 $out/tests_forced_mask.log
fi
: > $out/k1.txt
for tag in default "$@"; do
  if [ "$tag" = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_$tag.so; fi
  for cfg in "2.6 0.1" "10 0.1" "2.6 0.3" "10 0.3"; do
    set -- $cfg
    for mode in ${MODES:-1 0 2}; do
      timeout -k 10 200 python tools/k1_time.py --disparity $1 --spread $2 --scan-mode $mode --rounds 5 --check 2>&1 | grep -E "K1 median|maps sha|Error|error" >> $out/k1.txt || exit 1
    done
  done
  set --
done
cat $out/k1.txt
