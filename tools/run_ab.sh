# A/B of engine builds on one box: arithmetic self-tests + parity on the default build, then K1 / stage times per build.
# usage (GPU box): bash tools/run_ab.sh tagA tagB ...   (tags under lib/variants; "default" = the shipped library)
set -u
mkdir -p gpurun_out/ab
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_arith.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/ab/tests.log 2>&1 || { echo FAILED; tail -20 gpurun_out/ab/tests.log; exit 1; }
tail -1 gpurun_out/ab/tests.log
for round in 1 2; do
for tag in "$@"; do
  if [ "$tag" = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_$tag.so; fi
  echo "== $tag" >> gpurun_out/ab/k1.txt
  timeout -k 10 120 python tools/k1_time.py --check 2>&1 | grep -E "K1 median|identical|differ|Error|error" >> gpurun_out/ab/k1.txt || exit 1
  timeout -k 10 120 python tools/k1_time.py --outliers 2 2>&1 | grep -E "K1 median|Error|error" >> gpurun_out/ab/k1.txt || exit 1
done
done
cat gpurun_out/ab/k1.txt
