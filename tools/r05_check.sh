# round 5: new fixture / ingest / streaming tests, the two-plane scene report, the streaming figure of the bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout -k 10 300 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_ingest.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 400 python tools/scene_report.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/scene_report.txt
timeout -k 10 500 python bench.py --no-extra --cpu-kfs 0 > gpurun_out/r05/bench_noextra.json 2> gpurun_out/r05/bench_noextra.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_noextra.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_streaming", "value_streaming_pinned", "value_streaming_error",
                             "value_pcie_inclusive", "host_upload_ms_per_keyframe")})
PY
