#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4k
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_golden.py -m gpu -x -q > $O/tests.log 2>&1 || { echo FAILED; tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
V=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants
for round in 1 2; do
for tag in default k4b128 k4b64 k4b512; do
  if [ "$tag" = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=$V/libsdm_hip_$tag.so; fi
  echo "== $tag" >> $O/stage.txt
  timeout -k 10 120 python tools/stage_time.py 2>&1 | grep -v "amdgpu.ids\|^\[" >> $O/stage.txt || exit 1
  timeout -k 10 200 python tools/stage_time.py --res 720p --kfs 256 --nbrs 7 --reps 5 --rounds 5 2>&1 | grep -v "amdgpu.ids\|^\[" >> $O/stage.txt || exit 1
done
done
unset SDM_LIB_PATH
cat $O/stage.txt
timeout -k 10 300 python bench.py --no-extra --cpu-kfs 0 > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -20 $O/bench.err; }
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4k/bench.json"))
for k in ("value","ms_per_step","host_upload","value_pcie_inclusive","value_pcie_inclusive_pinned","value_pcie_inclusive_per_keyframe_calls"):
    print(k, d.get(k))
PY
