"""Where does a streaming iteration (step block i | upload block i+1) spend its wall time?  Host-side split of the two calls,
with and without overlapped ingest.  usage: python tools/debug/streaming_probe.py"""
import gc
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sdm_pkg  # noqa: E402
import bench  # noqa: E402

pkg = sdm_pkg.load()
wl = bench.Workload(pkg, torch, "480p", 64, 20, 2.6, 1, 0, 0, keep_images=64)
pl = wl.pl
ks = pl["own"]
n_slots = pl["n_slots"]
ims = [wl.images[k] for k in ks]
poses = [wl.scene.Tcw(k) for k in ks]
gc.disable()
AHEAD = int(os.environ.get("AHEAD", "3"))
for overlap in (False, True):
    eng = pkg.Engine(wl.W, wl.H, AHEAD * n_slots, max_neighbours=wl.N, batch_capacity=64, with_pointset=os.environ.get("POINTSET", "1") == "1")
    eng.set_ingest_overlap(overlap)
    shift = lambda lst, off: [s_ + off for s_ in lst]
    pls = [dict(pl, own_slots=shift(pl["own_slots"], h * n_slots), nbr_slots=[shift(r, h * n_slots) for r in pl["nbr_slots"]]) for h in range(AHEAD)]
    slots = [[pl["slot"][k] + h * n_slots for k in ks] for h in range(AHEAD)]
    step = lambda h: pkg.shard.pipeline_step(eng, None, pls[h], wl.min_d, wl.max_d, "none", None, "torch")
    for _ in range(3):
        for h in range(AHEAD):
            eng.upload_images_batch(slots[h], ims, wl.K, poses); step(h)
    eng.synchronize()
    ts, tu = [], []
    t0 = time.perf_counter()
    eng.upload_images_batch(slots[0], ims, wl.K, poses)
    for i in range(20):
        a = time.perf_counter()
        eng.upload_images_batch(slots[(i + 1) % AHEAD], ims, wl.K, poses)  # the next block first ...
        b = time.perf_counter()
        step(i % AHEAD)                                                     # ... then this one's step
        c = time.perf_counter()
        tu.append(b - a); ts.append(c - b)
    eng.synchronize()
    dt = time.perf_counter() - t0
    # the pieces on their own
    eng.synchronize(); a = time.perf_counter(); step(0); eng.synchronize(); t_step = time.perf_counter() - a
    a = time.perf_counter(); eng.upload_images_batch(slots[1], ims, wl.K, poses); b = time.perf_counter(); eng.synchronize(); c = time.perf_counter()
    print("overlap %d: %.3f ms per block | step call %.3f ms, upload call %.3f ms | alone: step to completion %.3f ms, upload call %.3f + drain %.3f ms" % (
        overlap, dt / 20 * 1e3, np.mean(ts) * 1e3, np.mean(tu) * 1e3, t_step * 1e3, (b - a) * 1e3, (c - b) * 1e3), flush=True)
    eng.close()
