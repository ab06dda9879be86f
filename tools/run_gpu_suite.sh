set -u
mkdir -p gpurun_out/suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the parity / fuzz files first, each under its own short timeout: a kernel fault must not cost minutes
for f in tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_statefuzz.py; do
  timeout -k 10 300 python -m pytest $f -m gpu -x -q > gpurun_out/suite/$(basename $f).log 2>&1 || { echo "FAILED $f"; tail -15 gpurun_out/suite/$(basename $f).log; exit 1; }
  tail -1 gpurun_out/suite/$(basename $f).log
done
( timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/suite/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/suite/pytest.log )
tail -4 gpurun_out/suite/pytest.log
grep -q "rc=0" gpurun_out/suite/pytest.log || exit 1
for k in 0 1 2 4 0; do
  timeout -k 10 120 python tools/k1_time.py --outliers $k 2>&1 | grep -E "K1 median|Error|error" >> gpurun_out/suite/k1.txt
done
cat gpurun_out/suite/k1.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/suite/kt -- python3 bench.py --steps 10 --warmup 2 --cpu-kfs 0 --no-extra > gpurun_out/suite/kt.log 2>&1
python3 tools/kstats.py gpurun_out/suite/kt | head -8; rm -rf gpurun_out/suite/kt
