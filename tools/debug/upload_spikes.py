"""Where do the multi-millisecond outliers of a single-keyframe sdm_upload_image from pageable memory come from: the call
itself (staging copy, API calls) or the wait for the device?  usage: python tools/debug/upload_spikes.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sdm_pkg  # noqa: E402

pkg = sdm_pkg.load()
cam = pkg.synth.TUM1
W, H = cam["W"], cam["H"]
scene = pkg.synth.Scene(cam, 0x5EED0002)
eng = pkg.Engine(W, H, 8, max_neighbours=7)
host = scene.render(5, device="cuda")[0].cpu().numpy()
pin = eng.host_alloc((H, W))
pin[...] = host
K, T = scene.K(), scene.Tcw(5)
for name, src in (("pageable", host), ("pinned", pin), ("pageable again", host)):
    call, wait = [], []
    for rep in range(400):
        t0 = time.perf_counter()
        eng.upload_image(3, src, K, T)
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
        if rep >= 20:
            call.append((t1 - t0) * 1e3)
            wait.append((t2 - t1) * 1e3)
    c, w = np.array(call), np.array(wait)
    print("%-15s call: p50 %.3f p99 %.3f max %.3f ms (%d above 1 ms)   wait: p50 %.3f p99 %.3f max %.3f ms (%d above 1 ms)" % (
        name, np.percentile(c, 50), np.percentile(c, 99), c.max(), int((c > 1).sum()), np.percentile(w, 50), np.percentile(w, 99),
        w.max(), int((w > 1).sum())), flush=True)
eng.close()
