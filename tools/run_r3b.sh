set -u
mkdir -p gpurun_out/r3b
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest.log )
tail -5 gpurun_out/r3b/pytest.log
for v in n17f f17f o7f o7d f17d n7f; do
  SDM_LIB_PATH=orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_$v.so timeout -k 10 120 python tools/k1_time.py --check 2>&1 | grep -E "K1 median|maps sha|Error|error" | tr '\n' ' ' >> gpurun_out/r3b/k1_variants.txt
  echo >> gpurun_out/r3b/k1_variants.txt
done
for k in 1 2; do
  timeout -k 10 120 python tools/k1_time.py --outliers $k 2>&1 | grep -E "K1 median|Error|error" >> gpurun_out/r3b/k1_variants.txt
done
cat gpurun_out/r3b/k1_variants.txt
timeout -k 10 400 python bench.py > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err; tail -c 1500 gpurun_out/r3b/bench.json
