set -u
mkdir -p gpurun_out/r3i
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3i/pytest.log )
tail -6 gpurun_out/r3i/pytest.log
for n in 2 3; do
  SDM_BENCH_REHEARSE=1 timeout -k 10 400 python bench.py --gpus $n --kfs 16 --steps 5 --warmup 2 --cpu-kfs 0 > gpurun_out/r3i/rehearse$n.json 2> gpurun_out/r3i/rehearse$n.err
  tail -c 900 gpurun_out/r3i/rehearse$n.json; echo; tail -3 gpurun_out/r3i/rehearse$n.err
done
timeout -k 10 500 python bench.py > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err; tail -c 300 gpurun_out/r3i/bench.json; tail -3 gpurun_out/r3i/bench.err
