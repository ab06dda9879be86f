set -u
mkdir -p gpurun_out/r3a
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log ) 
tail -5 gpurun_out/r3a/pytest.log
for v in o00 o01 o02 o04 o08 o10 o20 o40 o80 o7f off; do
  SDM_LIB_PATH=orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_$v.so timeout -k 10 120 python tools/k1_time.py --check 2>&1 | grep -E "K1 median|maps sha|Error|error" | tr '\n' ' ' >> gpurun_out/r3a/k1_variants.txt
  echo >> gpurun_out/r3a/k1_variants.txt
done
cat gpurun_out/r3a/k1_variants.txt
