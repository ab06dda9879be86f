#!/bin/bash
# Round 4: the -m gpu suite on the current build, K4 pipelining A/B (stage times), noise K1 with the counter-free dense path, bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4j
mkdir -p $O
( timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -5 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
V=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants
for round in 1 2; do
for tag in default k4pipe; do
  if [ "$tag" = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=$V/libsdm_hip_$tag.so; fi
  echo "== $tag" >> $O/stage.txt
  timeout -k 10 120 python tools/stage_time.py 2>&1 | grep -v "amdgpu.ids\|^\[" >> $O/stage.txt || exit 1
  timeout -k 10 200 python tools/stage_time.py --res 720p --kfs 256 --nbrs 7 --reps 5 --rounds 5 2>&1 | grep -v "amdgpu.ids\|^\[" >> $O/stage.txt || exit 1
done
done
unset SDM_LIB_PATH
cat $O/stage.txt
timeout -k 10 120 python tools/k1_time.py --noise --rounds 5 2>&1 | grep -E "K1 median|Error|error"
timeout -k 10 300 python bench.py --no-extra --cpu-kfs 0 > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -20 $O/bench.err; }
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4j/bench.json"))
for k in ("value","ms_per_step","host_upload","value_pcie_inclusive","value_pcie_inclusive_pinned","value_pcie_inclusive_per_keyframe_calls","stage_ms_per_step"):
    print(k, d.get(k))
PY
