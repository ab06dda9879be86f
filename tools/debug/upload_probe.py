"""Where does a 64-keyframe batch upload spend its wall time?  Call and drain, pageable and pinned sources, both ingest modes;
SDM_DEBUG_INGEST_TIMING=<ms> makes the engine print every chunk slower than that.  usage: python tools/debug/upload_probe.py"""
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
ims = [wl.images[k] for k in ks]
poses = [wl.scene.Tcw(k) for k in ks]
slots = [pl["slot"][k] for k in ks]
gc.disable()
eng = pkg.Engine(wl.W, wl.H, pl["n_slots"], max_neighbours=wl.N, batch_capacity=64, with_pointset=True)
block = eng.host_alloc((len(ks), wl.H, wl.W))
for i, im in enumerate(ims):
    block[i][...] = im
pinned = [block[i] for i in range(len(ks))]
for overlap in (False, True):
    eng.set_ingest_overlap(overlap)
    for name, src in (("pageable", ims), ("pinned", pinned)):
        for _ in range(3):
            eng.upload_images_batch(slots, src, wl.K, poses)
        eng.synchronize()
        tc, td = [], []
        for _ in range(15):
            a = time.perf_counter()
            eng.upload_images_batch(slots, src, wl.K, poses)
            b = time.perf_counter()
            eng.synchronize()
            c = time.perf_counter()
            tc.append(b - a)
            td.append(c - b)
        print("overlap %d, %-8s: call %.3f ms (min %.3f), drain %.3f ms, total %.3f ms = %.1f GB/s of images" % (
            overlap, name, np.median(tc) * 1e3, min(tc) * 1e3, np.median(td) * 1e3, (np.median(tc) + np.median(td)) * 1e3,
            len(ks) * wl.W * wl.H / (np.median(tc) + np.median(td)) / 1e9), flush=True)
# the copy alone, for scale: one 19.7-MB pinned -> device copy
src = torch.from_numpy(np.ascontiguousarray(block)).pin_memory() if False else None
t = torch.empty(len(ks) * wl.W * wl.H, dtype=torch.uint8).pin_memory()
d = torch.empty_like(t, device="cuda")
for _ in range(3):
    d.copy_(t, non_blocking=True)
torch.cuda.synchronize()
a = time.perf_counter()
for _ in range(10):
    d.copy_(t, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - a) / 10
print("plain pinned H2D of %.1f MB: %.3f ms = %.1f GB/s" % (t.numel() / 1e6, dt * 1e3, t.numel() / dt / 1e9))
# and the host's staging copy alone (this thread)
buf = np.empty((len(ks), wl.H, wl.W), np.uint8)
a = time.perf_counter()
for _ in range(10):
    for i, im in enumerate(ims):
        buf[i][...] = im
dt = (time.perf_counter() - a) / 10
print("one thread's memcpy of the 64 images: %.3f ms = %.1f GB/s" % (dt * 1e3, buf.nbytes / dt / 1e9))
eng.host_free(block)
eng.close()
