set -u
mkdir -p gpurun_out/r3h
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3h/pytest.log )
tail -6 gpurun_out/r3h/pytest.log
for v in k4old k4new k4old k4new; do
  SDM_LIB_PATH=orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_$v.so timeout -k 10 120 python tools/stage_time.py 2>&1 | grep -E "K1 |Error|error" >> gpurun_out/r3h/stages.txt
done
SDM_LIB_PATH=orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_k4new.so timeout -k 10 120 python tools/stage_time.py --res 720p --kfs 256 --nbrs 7 2>&1 | grep -E "K1 |Error|error" >> gpurun_out/r3h/stages.txt
SDM_LIB_PATH=orb-slam-free-space-carving_amd/lib/variants/libsdm_hip_k4old.so timeout -k 10 120 python tools/stage_time.py --res 720p --kfs 256 --nbrs 7 2>&1 | grep -E "K1 |Error|error" >> gpurun_out/r3h/stages.txt
cat gpurun_out/r3h/stages.txt
