set -u
mkdir -p gpurun_out/r3k
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3k/pytest.log )
tail -4 gpurun_out/r3k/pytest.log
for k in 0 1 2 4 0; do
  timeout -k 10 120 python tools/k1_time.py --outliers $k 2>&1 | grep -E "K1 median|Error|error" >> gpurun_out/r3k/k1.txt
done
cat gpurun_out/r3k/k1.txt
