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

# ---- the alternating pattern of tools/latency.py (pageable, pinned, pageable, ...): which call pays, and where ------------------
eng = pkg.Engine(W, H, 8, max_neighbours=7)
pin = eng.host_alloc((H, W))
pin[...] = host
host2 = host.copy()
for label, seq in (("pageable/pinned alternating", (("pageable", host), ("pinned", pin))),
                   ("pageable A / pageable B alternating", (("pageable A", host), ("pageable B", host2))),
                   ("pinned only", (("pinned", pin),)),
                   ("pageable/pinned alternating, 2 ms host sleep between calls", (("pageable", host), ("pinned", pin)))):
    acc = {name: ([], []) for name, _ in seq}
    for rep in range(300):
        for name, src in seq:
            if "sleep" in label:
                time.sleep(0.002)
            t0 = time.perf_counter()
            eng.upload_image(3, src, K, T)
            t1 = time.perf_counter()
            eng.synchronize()
            t2 = time.perf_counter()
            if rep >= 10:
                acc[name][0].append((t1 - t0) * 1e3)
                acc[name][1].append((t2 - t1) * 1e3)
    print(label, flush=True)
    for name, (c, w) in acc.items():
        c, w = np.array(c), np.array(w)
        slow = [(i, round(float(c[i]), 2), round(float(w[i]), 2)) for i in range(len(c)) if c[i] + w[i] > 1.0]
        print("   %-11s call: mean %.3f p50 %.3f p99 %.3f max %.3f | wait: mean %.3f p50 %.3f p99 %.3f max %.3f | > 1 ms (rep, call, wait): %s" % (
            name, c.mean(), np.percentile(c, 50), np.percentile(c, 99), c.max(), w.mean(), np.percentile(w, 50), np.percentile(w, 99),
            w.max(), slow[:12]), flush=True)
eng.close()
