set -u
mkdir -p gpurun_out/r3g
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r3g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3g/pytest.log )
tail -25 gpurun_out/r3g/pytest.log
for k in 0 1 2 4; do
  timeout -k 10 120 python tools/k1_time.py --outliers $k 2>&1 | grep -E "K1 median|Error|error" >> gpurun_out/r3g/k1.txt
done
cat gpurun_out/r3g/k1.txt
