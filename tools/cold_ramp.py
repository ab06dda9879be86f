"""How long after an idle period does K1 run slow?  Per-launch K1 times (HIP events on the engine's stream) of the first
launches after a 0.5 s pause, on the clean configs[1] workload and with 2 outlier neighbours: attributes the bench's
cold / steady-state gap (a time-based clock ramp, or a first-use cost of the open-pixel path?).
usage: python tools/cold_ramp.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sdm_pkg  # noqa: E402
import bench  # noqa: E402

pkg = sdm_pkg.load()
for outliers in (0, 2, 0):
    wl = bench.Workload(pkg, torch, "480p", 64, 20, 2.6, 1, 0, 0, outliers=outliers)
    eng, pl = wl.eng, wl.pl
    eng.enable_timing(True)
    for trial in range(2):
        eng.synchronize()
        time.sleep(0.5)
        ms, t_wall = [], []
        t0 = time.perf_counter()
        for i in range(40):
            eng.get_timing(reset=True)
            eng.search_fuse(pl["own_slots"], pl["nbr_slots"], wl.min_d, wl.max_d)
            eng.synchronize()
            t = eng.get_timing()
            ms.append(t["search_fuse"][0])
            t_wall.append((time.perf_counter() - t0) * 1e3)
        steady = sorted(ms[20:])[10]
        slow = [i for i, m in enumerate(ms) if m > 1.03 * steady]
        print("outliers %d, trial %d: steady %.4f ms; launches 1-12: %s; last launch more than 3 %% slow: #%d at %.1f ms after the pause" % (
            outliers, trial, steady, " ".join("%.3f" % m for m in ms[:12]), (slow[-1] + 1) if slow else 0,
            t_wall[slow[-1]] if slow else 0.0), flush=True)
    wl.close()
