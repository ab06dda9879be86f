set -u
mkdir -p gpurun_out/ovh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python tools/step_overhead.py > gpurun_out/ovh/480p.txt 2>&1 || { tail -20 gpurun_out/ovh/480p.txt; exit 1; }
grep -v "^\[\|amdgpu.ids" gpurun_out/ovh/480p.txt
timeout -k 10 200 python tools/step_overhead.py --res 1080p --kfs 32 --nbrs 7 --steps 20 > gpurun_out/ovh/1080p.txt 2>&1 || { tail -20 gpurun_out/ovh/1080p.txt; exit 1; }
grep -v "^\[\|amdgpu.ids" gpurun_out/ovh/1080p.txt
