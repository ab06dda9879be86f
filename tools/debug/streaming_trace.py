"""Reads a rocprofv3 --kernel-trace CSV of tools/debug/streaming_probe.py and prints, for the last few blocks of the overlapped
run, every kernel with its queue, start (us, relative) and duration, and the idle gaps of the GPU.  usage: python
tools/debug/streaming_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sdm::", ""), r.get("Queue_Id", "?")))
rows.sort()
# the last 2 % of the trace = the tail of the overlapped run
t_end = rows[-1][1]
tail = [r for r in rows if r[0] > t_end - 12_000_000]  # last 12 ms
t0 = tail[0][0]
busy_until = tail[0][0]
for s, e, name, q in tail:
    gap = s - busy_until
    if name.startswith("k_search_fuse") or name.startswith("k_prepass_batch") or name.startswith("k_inter_check") or gap > 20_000:
        print("%9.1f us  %-34s q%-3s %8.1f us%s" % ((s - t0) / 1e3, name[:34], q, (e - s) / 1e3, ("   <-- GPU idle %.0f us before" % (gap / 1e3)) if gap > 20_000 else ""))
    busy_until = max(busy_until, e)
