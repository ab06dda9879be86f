# parity / fuzz / state fuzz on the default build, then per-stage times (tools/stage_time.py) per build at 480p and 720p
# usage (GPU box): bash tools/run_stage_ab.sh tagA tagB ...   ("default" = the shipped library)
set -u
mkdir -p gpurun_out/stage_ab
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_statefuzz.py; do
  timeout -k 10 300 python -m pytest $f -m gpu -x -q > gpurun_out/stage_ab/$(basename $f).log 2>&1 || { echo "FAILED $f"; tail -25 gpurun_out/stage_ab/$(basename $f).log; exit 1; }
  tail -1 gpurun_out/stage_ab/$(basename $f).log
done
V=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants
for round in 1 2; do
for tag in "$@"; do
  if [ "$tag" = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=$V/libsdm_hip_$tag.so; fi
  timeout -k 10 120 python tools/stage_time.py 2>&1 | grep -v "amdgpu.ids\|^\[" >> gpurun_out/stage_ab/t.txt || exit 1
  timeout -k 10 200 python tools/stage_time.py --res 720p --kfs 256 --nbrs 7 --reps 5 --rounds 5 2>&1 | grep -v "amdgpu.ids\|^\[" >> gpurun_out/stage_ab/t.txt || exit 1
done
done
cat gpurun_out/stage_ab/t.txt
