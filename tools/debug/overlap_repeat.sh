# repeats the overlapped-ingest tests N times in fresh processes, keeping the full log of any failing run
# usage (GPU box): bash tools/debug/overlap_repeat.sh [N]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/overlap
n=${1:-10}; bad=0
for i in $(seq 1 $n); do
  timeout -k 10 200 python -m pytest tests/test_gpu_ingest.py -m gpu -q -x -k "streaming_order or slots_in_use" > gpurun_out/overlap/run_$i.log 2>&1
  if [ $? -ne 0 ]; then bad=$((bad+1)); echo "run $i FAILED"; else rm -f gpurun_out/overlap/run_$i.log; fi
done
echo "$bad of $n runs failed"
