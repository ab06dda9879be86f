#!/bin/bash
# VALU / SALU / lane-utilisation counters of k_search_fuse for one engine build (SDM_LIB_PATH), one rocprofv3 pass.
# usage (on the GPU box): SDM_LIB_PATH=... tools/pmc_valu.sh <tag>
set -u
TAG=$1
OUT=gpurun_out/pv_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/p1" -- python3 tools/k1_time.py --rounds 1 --reps 3 > "$OUT/log" 2>&1 || echo "pass failed"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(root + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_search_fuse<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(tag, " ".join("%s=%.1fM" % (k.replace("SQ_", ""), sum(v) / len(v) / 1e6) for k, v in sorted(agg.items())))
PY
rm -rf "$OUT/p1"
