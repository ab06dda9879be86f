#!/bin/bash
# Collects rocprofv3 PMC counters for one bench configuration, one counter group per pass
# (gpurun refuses --pmc combined with trace domains; FETCH_SIZE and WRITE_SIZE cannot share a pass).
# usage (on the GPU box, from the repo root): tools/pmc.sh <tag> [bench args...]
# Leaves gpurun_out/pmc_<tag>/summary.txt (+ traffic.json); the raw per-dispatch CSVs are deleted
# (they exceed what gpurun merges back).
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
while IFS= read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $group --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 2 --warmup 1 --cpu-kfs 0 --no-stats --no-extra --no-streaming --no-live-pmc "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i ($group) failed"
done <<'GROUPS'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum
FETCH_SIZE
WRITE_SIZE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_WAVES GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
GROUPS
PMC_BENCH_ARGS="$*" python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT"/p[0-9]*
cat "$OUT/summary.txt"
