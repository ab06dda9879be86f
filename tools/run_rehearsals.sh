set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4r
for n in 2 3; do
  SDM_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus $n --kfs 16 --steps 5 --warmup 1 2> gpurun_out/r4r/rehearse$n.err | grep '^{"metric"' > gpurun_out/r4r/rehearsal_${n}ranks_one_gpu.json || { echo "rehearsal $n failed"; tail -20 gpurun_out/r4r/rehearse$n.err; }
done
python3 - <<'PY'
import json
for n in (2,3):
    d=json.load(open("gpurun_out/r4r/rehearsal_%dranks_one_gpu.json"%n))
    print(n, d["value"], d.get("value_baseline_literal"), d.get("exchange_ms_per_step"), d["config"].get("exchange_wire"), d.get("staged_refused_maps"))
PY
timeout -k 10 300 python -m pytest tests/test_gpu_shard.py tests/test_gpu_comm.py -m gpu -x -q 2>&1 | tail -2
