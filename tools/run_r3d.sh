set -u
mkdir -p gpurun_out/r3d
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d/pytest.log )
tail -25 gpurun_out/r3d/pytest.log
timeout -k 10 500 python bench.py > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err; tail -c 600 gpurun_out/r3d/bench.json; tail -5 gpurun_out/r3d/bench.err
