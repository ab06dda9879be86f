"""One steady-state block of the streaming probe's kernel trace, every dispatch: queue, start (us), duration.
usage: python tools/debug/streaming_block_dump.py <dir>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sdm::", ""), r.get("Queue_Id", "?"),
                     r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
rows.sort()
k1 = [i for i, r in enumerate(rows) if r[2].startswith("k_search_fuse<false")]
a, b = k1[-4], k1[-3]
t0 = rows[a][0]
end = t0
for s, e, name, q, g in rows[a:b + 1]:
    print("%9.1f us  %-36s q%-3s grid %-10s %8.1f us   gap %6.1f" % ((s - t0) / 1e3, name[:36], q, g, (e - s) / 1e3, (s - end) / 1e3))
    end = max(end, e)
