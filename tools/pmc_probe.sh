#!/bin/bash
# rocprofv3 PMC counters (one group per pass, no trace domains) of any python script that drives the engine -- tools/pmc.sh is
# the same for bench.py workloads.  usage (GPU box, repo root): tools/pmc_probe.sh <tag> <script.py> [args...]
# Leaves gpurun_out/pmc_<tag>/summary.txt (per-kernel averages of this repo's kernels).
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
while IFS= read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $group --output-format csv -d "$OUT/p$i" -- python3 "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i ($group) failed"
done <<'GROUPS'
TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32
GROUPS
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT"/p[0-9]*
grep -A 30 "k_prepass_batch" "$OUT/summary.txt" | head -60
