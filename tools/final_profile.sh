set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
timeout -k 10 500 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || echo "bench failed"
tail -c 400 gpurun_out/final/bench.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt -- python3 bench.py --steps 10 --warmup 2 --cpu-kfs 0 --no-extra > gpurun_out/final/kt.log 2>&1 || echo "kernel trace failed"
f=$(find gpurun_out/final/kt -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then (head -1 "$f"; grep "sdm::" "$f") > gpurun_out/final/kernel_stats.csv; fi
python3 tools/kstats.py gpurun_out/final/kt || true
rm -rf gpurun_out/final/kt
timeout -k 10 200 python bench.py --kfs 256 --cpu-kfs 4 > gpurun_out/final/bench_480p_256kf_n20.json 2>> gpurun_out/final/bench.err || echo "256kf failed"
timeout -k 10 200 python bench.py --res 720p --kfs 256 --nbrs 7 --cpu-kfs 3 > gpurun_out/final/bench_720p_256kf_n7.json 2>> gpurun_out/final/bench.err || echo "720p failed"
timeout -k 10 200 python bench.py --res 1080p --kfs 64 --nbrs 7 --cpu-kfs 2 > gpurun_out/final/bench_1080p_64kf_n7.json 2>> gpurun_out/final/bench.err || echo "1080p failed"
timeout -k 10 100 python tools/latency.py 7 > gpurun_out/final/latency.txt 2>&1 || echo "latency failed"
timeout -k 10 100 python tools/latency.py 20 >> gpurun_out/final/latency.txt 2>&1 || echo "latency failed"
cat gpurun_out/final/latency.txt
# kernel-trace stats of the hard-data lines (i.i.d.-noise images, 2 outlier neighbours, long baseline)
for cfg in "noise --noise" "outliers2 --outliers 2" "disp10 --disparity 10"; do
  set -- $cfg; tag=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt_$tag -- python3 bench.py --steps 10 --warmup 2 --cpu-kfs 0 --no-extra --no-stats "$@" > gpurun_out/final/kt_$tag.log 2>&1 || echo "kt $tag failed"
  f=$(find gpurun_out/final/kt_$tag -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep "sdm::" "$f") > gpurun_out/final/kernel_stats_$tag.csv; fi
  rm -rf gpurun_out/final/kt_$tag
done
# control-flow rehearsals of the multi-rank bench on this one GPU (gloo, host-staged exchange: NOT measurements)
for n in 2 3; do
  SDM_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus $n --kfs 16 --steps 5 --warmup 1 2> gpurun_out/final/rehearse$n.err | grep '^{"metric"' > gpurun_out/final/rehearsal_${n}ranks_one_gpu.json || echo "rehearsal $n failed"
done
echo done benches
