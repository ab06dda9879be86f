"""Averages rocprofv3 counter_collection.csv values per kernel and counter (this repo's kernels)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "sdm::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("    %-34s %16.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))

# HBM-side traffic of the dominant kernel from the size-specific request counters (the guide's
# "calibrate on your own access pattern": FETCH_SIZE tallies every request at 64 B, but K1's gathers
# leave the L2 as 128-B requests, so bytes = 32*n32 + 64*n64 + 128*n128; writes likewise).
import json
k1 = [k for k in agg if "k_search_fuse<false" in k]
if k1:
    c = {n: (sum(v) / len(v)) for n, v in agg[k1[0]].items()}
    need = ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]
    if all(n in c for n in need):
        rd = 32 * c[need[0]] + 64 * c[need[1]] + 128 * c[need[2]]
        wr = None
        if "TCC_EA0_WRREQ_sum" in c and "TCC_EA0_WRREQ_64B_sum" in c:
            wr = 64 * c["TCC_EA0_WRREQ_64B_sum"] + 32 * (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"])
        out = {"kernel": "k_search_fuse", "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
               "traffic_bytes_per_launch": rd + (wr or 0),
               "FETCH_SIZE_KB": c.get("FETCH_SIZE"), "WRITE_SIZE_KB": c.get("WRITE_SIZE"),
               "method": "32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B (+ 64*WRREQ_64B + 32*other WRREQ), "
                         "TCC_EA0 counters summed over channels, averaged over launches"}
        # tie the figure to the build and workload it was measured on (bench.py only reports a matching one)
        import argparse, os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        ap = argparse.ArgumentParser()
        ap.add_argument("--kfs", type=int, default=64)
        ap.add_argument("--nbrs", type=int, default=20)
        ap.add_argument("--res", default="480p")
        ap.add_argument("--disparity", type=float, default=2.6)
        ap.add_argument("--noise", action="store_true")
        ap.add_argument("--outliers", type=int, default=0)
        ap.add_argument("--prior-spread", type=float, default=0.1)
        ap.add_argument("--scene", default="plane")
        ap.add_argument("--roll", type=float, default=1.0)
        a, _ = ap.parse_known_args(os.environ.get("PMC_BENCH_ARGS", "").split())
        out["workload"] = {"res": a.res, "kfs": a.kfs, "nbrs": a.nbrs, "disparity": a.disparity, "noise": a.noise,
                           "outliers": a.outliers, "spread": a.prior_spread, "scene": a.scene, "roll": a.roll}
        out["src_hash"] = bench.source_hash()
        out["source"] = "tools/pmc.sh (rocprofv3 --pmc, one counter group per pass)"
        json.dump(out, open(root + "/traffic.json", "w"), indent=1)
        print("traffic.json:", json.dumps(out))
