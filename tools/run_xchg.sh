set -u
mkdir -p gpurun_out/xchg
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_comm.py tests/test_gpu_shard.py -m gpu -x -q > gpurun_out/xchg/tests.log 2>&1 || { echo FAILED; tail -20 gpurun_out/xchg/tests.log; exit 1; }
tail -1 gpurun_out/xchg/tests.log
V=$GRAFT_REPO_ROOT/orb-slam-free-space-carving_amd/lib/variants
for r in 1 2; do
timeout -k 10 120 python tools/xchg_copy_time.py >> gpurun_out/xchg/t.txt 2>&1 || exit 1
SDM_LIB_PATH=$V/libsdm_hip_old.so timeout -k 10 120 python tools/xchg_copy_time.py >> gpurun_out/xchg/t.txt 2>&1 || exit 1
SDM_COMM_SINGLE_RANK_RCCL=1 timeout -k 10 120 python tools/xchg_copy_time.py >> gpurun_out/xchg/t.txt 2>&1 || exit 1
done
timeout -k 10 120 python tools/xchg_copy_time.py --res 1080p --nbrs 7 >> gpurun_out/xchg/t.txt 2>&1 || exit 1
SDM_LIB_PATH=$V/libsdm_hip_old.so timeout -k 10 120 python tools/xchg_copy_time.py --res 1080p --nbrs 7 >> gpurun_out/xchg/t.txt 2>&1 || exit 1
grep -v "^\[" gpurun_out/xchg/t.txt
