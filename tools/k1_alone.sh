set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/k1alone
V=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants
for round in 1 2; do
for tag in default tapskip; do
  if [ "$tag" = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=$V/libsdm_hip_$tag.so; fi
  timeout -k 10 120 python tools/k1_time.py 2>&1 | grep "K1 median" >> gpurun_out/k1alone/t.txt
  timeout -k 10 200 python tools/k1_time.py --res 720p --kfs 256 --nbrs 7 --reps 5 --rounds 5 2>&1 | grep "K1 median" >> gpurun_out/k1alone/t.txt
  timeout -k 10 200 python tools/stage_time.py --res 720p --kfs 256 --nbrs 7 --reps 5 --rounds 5 2>&1 | grep -v "amdgpu.ids\|^\[" >> gpurun_out/k1alone/t.txt
done
done
cat gpurun_out/k1alone/t.txt
