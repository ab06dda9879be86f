set -u
mkdir -p gpurun_out/r3c
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest.log )
tail -5 gpurun_out/r3c/pytest.log
for v in v1 v2 v3 v4 v6 v7 v8 v9 v10 v11 v12 v13 v14 v15 v1; do
  SDM_LIB_PATH=orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_$v.so timeout -k 10 120 python tools/k1_time.py --check 2>&1 | grep -E "K1 median|maps sha|Error|error" | tr '\n' ' ' >> gpurun_out/r3c/k1_variants.txt
  echo >> gpurun_out/r3c/k1_variants.txt
done
cat gpurun_out/r3c/k1_variants.txt
