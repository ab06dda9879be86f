"""Rare ~40 ms outliers of sdm_upload_image: at which call indices do they occur, and in which phase of the call
(SDM_DEBUG_INGEST_TIMING=1 prints the engine's own split of any chunk slower than 2 ms).  usage: python tools/debug/upload_spikes2.py"""
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
import gc
for name, src, nogc in (("pageable", host, False), ("pinned", pin, False), ("pageable, gc disabled", host, True)):
    if nogc:
        gc.disable()
    slow = []
    t_all0 = time.perf_counter()
    for rep in range(4000):
        t0 = time.perf_counter()
        eng.upload_image(3, src, K, T)
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
        if t2 - t0 > 1e-3:
            slow.append((rep, round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2), round((t0 - t_all0) * 1e3, 1)))
    print("%-22s 4000 calls in %.1f ms; slower than 1 ms (call index, call ms, wait ms, at ms): %s" % (
        name, (time.perf_counter() - t_all0) * 1e3, slow), flush=True)
    gc.enable()
eng.close()
