# round 5: bench lines + kernel stats of the final build (one gpurun call): bash tools/r05_final.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
( time timeout -k 10 700 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err ) 2> gpurun_out/final/bench.time || echo "bench failed"
tail -c 300 gpurun_out/final/bench.json; cat gpurun_out/final/bench.time
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt -- python3 bench.py --steps 10 --warmup 2 --cpu-kfs 0 --no-extra --no-streaming --no-live-pmc > gpurun_out/final/kt.log 2>&1 || echo "kernel trace failed"
f=$(find gpurun_out/final/kt -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then (head -1 "$f"; grep "sdm::" "$f") > gpurun_out/final/kernel_stats.csv; fi
python3 tools/kstats.py gpurun_out/final/kt || true
rm -rf gpurun_out/final/kt
timeout -k 10 100 python tools/latency.py 7 > gpurun_out/final/latency.txt 2>&1 || echo "latency failed"
timeout -k 10 100 python tools/latency.py 20 >> gpurun_out/final/latency.txt 2>&1 || echo "latency failed"
grep -v amdgpu gpurun_out/final/latency.txt
for cfg in "disp10 --disparity 10" "spread03 --prior-spread 0.3" "disp10_spread03 --disparity 10 --prior-spread 0.3" "strip --scene strip --roll 5 --prior-spread 0.3"; do
  set -- $cfg; tag=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt_$tag -- python3 bench.py --steps 10 --warmup 2 --cpu-kfs 0 --no-extra --no-stats --no-streaming --no-live-pmc "$@" > gpurun_out/final/kt_$tag.log 2>&1 || echo "kt $tag failed"
  f=$(find gpurun_out/final/kt_$tag -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep "sdm::" "$f") > gpurun_out/final/kernel_stats_$tag.csv; fi
  rm -rf gpurun_out/final/kt_$tag
done
for n in 2 3; do
  SDM_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus $n --kfs 16 --steps 5 --warmup 1 2> gpurun_out/final/rehearse$n.err | grep '^{"metric"' > gpurun_out/final/rehearsal_${n}ranks_one_gpu.json || echo "rehearsal $n failed"
done
echo done benches
