"""Per-block GPU-time accounting of the streaming run from a rocprofv3 --kernel-trace of tools/debug/streaming_probe.py:
for the last N blocks of the overlapped run (a block = from one k_pair_setup/first K1 to the next), the summed duration of
every kernel by name, the union busy time and the idle time.  usage: python tools/debug/streaming_account.py <dir> [blocks]"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sdm::", "")))
rows.sort()
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 10
# the overlapped run's timed loop = the last 20 K1 launches but three (one "alone" step follows, and set-up ones precede)
k1 = [i for i, r in enumerate(rows) if r[2].startswith("k_search_fuse<false")]
k1 = k1[-(nb + 2):-1]  # nb blocks, delimited by nb+1 K1 starts; drops the lone step at the end
lo, hi = rows[k1[0]][0], rows[k1[-1]][0]
sel = [r for r in rows if lo <= r[0] < hi]
acc = defaultdict(float)
cnt = defaultdict(int)
busy = 0.0
end = lo
for s, e, name in sel:
    acc[name] += (e - s) / 1e3
    cnt[name] += 1
    if e > end:
        busy += (e - max(s, end)) / 1e3
        end = e
n = len(k1) - 1
print("blocks: %d, wall per block %.1f us, busy %.1f us, idle %.1f us" % (n, (hi - lo) / 1e3 / n, busy / n, ((hi - lo) / 1e3 - busy) / n))
for name in sorted(acc, key=lambda k: -acc[k]):
    print("  %-44s %6.1f launches/block  %8.1f us/block  (avg %.1f us)" % (name[:44], cnt[name] / n, acc[name] / n, acc[name] / cnt[name]))
