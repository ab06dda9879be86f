#!/bin/bash
# Round 4: (1) the over-2^32 launch in the ubench; (2) parity of the open-list regimes; (3) K1 on the hard-data workloads:
# in-place threshold / k_fuse_open grid A/B, and the scan-only time on noise images (fusion compiled out);
# (4) ingest timings in the bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4i
mkdir -p $O
timeout -k 10 400 orb-slam-free-space-carving_amd/lib/ubench_big_grid > $O/big_grid.txt 2>&1 || echo "big_grid rc $?"
grep "2^32" $O/big_grid.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ingest.py -m gpu -x -q > $O/tests.log 2>&1 || { echo FAILED; tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
K=$O/k1.txt
for cfg in "--outliers 2" "--noise" ""; do
  for env in "SDM_OPEN_INPLACE=65 SDM_OPEN_GRID=512" "SDM_OPEN_INPLACE=65 SDM_OPEN_GRID=2048" "SDM_OPEN_INPLACE=40 SDM_OPEN_GRID=2048" "SDM_OPEN_INPLACE=24 SDM_OPEN_GRID=2048" "SDM_OPEN_INPLACE=56 SDM_OPEN_GRID=2048"; do
    echo "== [$cfg] $env" >> $K
    env $env timeout -k 10 120 python tools/k1_time.py $cfg --rounds 5 2>&1 | grep -E "K1 median|Error|error" >> $K || exit 1
  done
done
echo "== [--noise] fusion compiled out (SDM_ABLATE=1)" >> $K
SDM_LIB_PATH=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_nofuse.so timeout -k 10 120 python tools/k1_time.py --noise --rounds 5 2>&1 | grep -E "K1 median|Error|error" >> $K
echo "== [] fusion compiled out (SDM_ABLATE=1)" >> $K
SDM_LIB_PATH=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_nofuse.so timeout -k 10 120 python tools/k1_time.py --rounds 5 2>&1 | grep -E "K1 median|Error|error" >> $K
cat $K
timeout -k 10 300 python bench.py --no-extra --cpu-kfs 0 > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -20 $O/bench.err; }
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4i/bench.json"))
for k in ("value","ms_per_step","host_upload","value_pcie_inclusive","value_pcie_inclusive_pinned","value_pcie_inclusive_per_keyframe_calls","stage_ms_per_step"):
    print(k, d.get(k))
PY
