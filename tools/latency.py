"""Per-keyframe latency of the online use (one SemiDenseRecon call per new keyframe, PM.cc:137-256), through the
C ABI: recon([k]) -> synchronize, then inter_check([k]) + pointset([k]) -> synchronize.  Prints mean / p50 / p99 (ms)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdm_pkg  # noqa: E402

pkg = sdm_pkg.load()
synth = pkg.synth
n_kf, N = 40, int(sys.argv[1]) if len(sys.argv) > 1 else 7
cam = synth.TUM1
scene = synth.Scene(cam, 0x5EED0002)
W, H = cam["W"], cam["H"]
eng = pkg.Engine(W, H, n_kf, max_neighbours=N, with_pointset=True)
for k in range(n_kf):
    im, _ = scene.render(k, device="cuda")
    torch.cuda.synchronize()
    eng.upload_image_device(k, im.data_ptr(), scene.K(), scene.Tcw(k))
min_d, max_d = scene.depth_prior()
nb = {k: scene.neighbours(k, n_kf, N) for k in range(n_kf)}
eng.recon(list(range(n_kf)), [nb[k] for k in range(n_kf)], min_d, max_d)  # everything reconstructed once
eng.synchronize()
# The timed loops run with Python's cyclic garbage collector off: a full (generation 2) collection of this interpreter's heap --
# torch is loaded -- takes ~40 ms and lands deterministically after a few hundred binding calls; round 4's "p99 25 ms" of the
# single-keyframe upload was exactly one such collection inside its 50 samples (tools/debug/upload_spikes2.py: the engine's own
# whole-call timer stays below 2 ms for that call, and the outlier disappears with gc.disable()).  A C++ caller has no such pause.
import gc
gc.collect()
gc.disable()
t_recon, t_rest = [], []
for rep in range(3):
    for k in range(n_kf):
        t0 = time.perf_counter()
        eng.recon([k], [nb[k]], min_d, max_d)
        eng.synchronize()
        t1 = time.perf_counter()
        eng.inter_check([k], [nb[k]], commit=False)
        eng.pointset([k], source=1)
        eng.synchronize()
        t2 = time.perf_counter()
        if rep:
            t_recon.append((t1 - t0) * 1e3)
            t_rest.append((t2 - t1) * 1e3)
# ingest of ONE new keyframe from host memory (what the fork's Tracking -> Modeler::AddFrameImage hand-over costs per
# keyframe): H2D + device pre-pass + records + pixel list, wall clock until the device has finished; pageable and pinned
host_im = scene.render(5, device="cuda")[0].cpu().numpy()
pin_im = eng.host_alloc((H, W))
pin_im[...] = host_im
t_up, t_up_pin = [], []
for rep in range(60):
    for src, acc in ((host_im, t_up), (pin_im, t_up_pin)):
        t0 = time.perf_counter()
        eng.upload_image(n_kf - 1, src, scene.K(), scene.Tcw(5))
        eng.synchronize()
        if rep >= 10:
            acc.append((time.perf_counter() - t0) * 1e3)
for name, t in (("SemiDenseRecon (K1-K3), 1 keyframe x %d neighbours" % N, t_recon), ("inter-check + point set (K4-K5)", t_rest),
                ("sdm_upload_image, 1 keyframe (pageable source)", t_up), ("sdm_upload_image, 1 keyframe (pinned source)", t_up_pin)):
    t = np.array(t)
    print("%-52s mean %.3f ms  p50 %.3f  p99 %.3f" % (name, t.mean(), np.percentile(t, 50), np.percentile(t, 99)))
eng.close()
