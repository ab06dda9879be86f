#!/bin/bash
# Issue / lane-utilisation / wait counters of k_search_fuse for one K1 workload and scan mode, two rocprofv3 passes.
# usage (on the GPU box): [SDM_LIB_PATH=...] tools/pmc_k1.sh <tag> [tools/k1_time.py args, e.g. --disparity 10 --scan-mode 2]
set -u
TAG=$1; shift
OUT=gpurun_out/pk1_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/p1" -- python3 tools/k1_time.py --rounds 1 --reps 3 "$@" > "$OUT/log1" 2>&1 || echo "pass 1 failed"
timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT/p2" -- python3 tools/k1_time.py --rounds 1 --reps 3 "$@" > "$OUT/log2" 2>&1 || echo "pass 2 failed"
python3 - "$OUT" "$TAG $*" <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(root + "/p[12]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_search_fuse<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print(tag)
print("   " + " ".join("%s=%.1fM" % (k.replace("SQ_", ""), v / 1e6) for k, v in sorted(m.items())))
if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
    print("   lanes of 64: %.1f" % (m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"] * 1.0))
PY
rm -rf "$OUT/p1" "$OUT/p2"
