#!/bin/bash
# Cross-compiles the engine for gfx950 (no GPU needed) and prints, per kernel matching $1 (default k_search_fuse): registers,
# spills, scratch, occupancy and the static instruction count of its ISA.  Extra arguments are passed to hipcc (-DSDM_...).
# usage: bash tools/k1_resources.sh [kernel-name-substring] [-DFLAG ...]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
pat=${1:-k_search_fuse}
shift || true
out=$(mktemp -d /tmp/k1res.XXXXXX)
cd "$out"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -I"$ROOT/include" \
    -I"$ROOT/orb-slam-free-space-carving_amd/csrc" "$@" "$ROOT/orb-slam-free-space-carving_amd/csrc/sdm_engine.hip" -o x.so \
    -Rpass-analysis=kernel-resource-usage --save-temps 2> remarks.txt || { grep -E "error|warning" remarks.txt | head -40; exit 1; }
grep -E "error" remarks.txt | head || true
python3 - "$pat" <<'EOF'
import re, sys
pat = sys.argv[1]
cur = None
for line in open("remarks.txt"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1) if pat in m.group(1) else None
        if cur:
            print(cur[:70])
        continue
    if cur:
        m = re.search(r"(VGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m:
            print("   %-28s %s" % (m.group(1), m.group(2)))
# static instruction counts per kernel body
name = None
counts = {}
for line in open("sdm_engine-hip-amdgcn-amd-amdhsa-gfx950.s"):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1) if pat in m.group(1) else None
        continue
    if name and re.match(r"^\s+(v_|s_|ds_|global_|buffer_|flat_|scratch_)", line):
        op = line.split()[0]
        c = counts.setdefault(name, {"total": 0, "valu": 0, "salu": 0, "mem": 0, "scratch": 0})
        c["total"] += 1
        c["valu"] += op.startswith("v_")
        c["salu"] += op.startswith("s_")
        c["mem"] += op.startswith(("global_", "buffer_", "flat_", "ds_"))
        c["scratch"] += op.startswith("scratch_")
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        name = None
for k, c in counts.items():
    print(k[:70], c)
EOF
rm -rf "$out"
