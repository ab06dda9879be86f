#!/bin/bash
# chunked fusion phase: arithmetic self-tests + parity + fuzz, then K1 A/B (clean, outliers, noise) against the one-row form
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4m
mkdir -p $O
for f in tests/test_gpu_arith.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py; do
  timeout -k 10 300 python -m pytest $f -m gpu -x -q > $O/$(basename $f).log 2>&1 || { echo "FAILED $f"; tail -30 $O/$(basename $f).log; exit 1; }
  tail -1 $O/$(basename $f).log
done
SDM_OPEN_QUOTA=0 SDM_OPEN_INPLACE=65 SDM_FUZZ_GEOM=300 timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_defer.log 2>&1 || { echo "FAILED fuzz defer"; tail -30 $O/fuzz_defer.log; exit 1; }
tail -1 $O/fuzz_defer.log
SDM_OPEN_QUOTA=0 SDM_OPEN_INPLACE=1 SDM_FUZZ_GEOM=300 timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_inplace.log 2>&1 || { echo "FAILED fuzz inplace"; tail -30 $O/fuzz_inplace.log; exit 1; }
tail -1 $O/fuzz_inplace.log
V=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants
for round in 1 2; do
for tag in default fuse1row; do
  if [ "$tag" = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=$V/libsdm_hip_$tag.so; fi
  echo "== $tag" >> $O/k1.txt
  for cfg in "" "--outliers 2" "--noise --rounds 5" "--disparity 10"; do
    timeout -k 10 120 python tools/k1_time.py --check $cfg 2>&1 | grep -E "K1 median|maps sha|Error|error" | sed "s/^/[$cfg] /" >> $O/k1.txt || exit 1
  done
done
done
cat $O/k1.txt | cut -c1-200
