"""Host-side cost of one boundary all-gather pass on ONE GPU (no peers: what remains is the packing of the boundary maps,
the gather itself -- a device copy, or RCCL over a one-rank communicator with SDM_COMM_SINGLE_RANK_RCCL=1 -- and the fetch
of the halo maps into their slots): the part of the multi-GPU step that is not transfer time.
usage: python tools/xchg_copy_time.py [--res 480p --nbrs 20 --iters 200]   (SDM_LIB_PATH selects the build)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdm_pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="480p")
ap.add_argument("--kfs", type=int, default=64)
ap.add_argument("--nbrs", type=int, default=20)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--compact", action="store_true",
                help="the compact wire format (sdm_exchange_compact): keyframes rendered and resident twice, as on two ranks")
a = ap.parse_args()
W, H = {"480p": (640, 480), "720p": (1280, 720), "1080p": (1920, 1080)}[a.res]
pkg = sdm_pkg.load()
h = a.nbrs // 2
eng = pkg.Engine(W, H, a.kfs + a.nbrs, max_neighbours=a.nbrs)
if os.environ.get("SDM_COMM_SINGLE_RANK_RCCL") == "1":
    eng.comm_init(eng.comm_unique_id(), 1, 0)
eng.mark_depth_present(list(range(a.kfs)))
boundary = list(range(h)) + list(range(a.kfs - (a.nbrs - h), a.kfs))          # two runs: packed through the staging buffer
fetch = [(i, a.kfs + i) for i in range(a.nbrs)]
wire = "whole maps"
if a.compact:
    import numpy as np
    cam = {"480p": pkg.synth.TUM1, "720p": pkg.synth.HD720, "1080p": pkg.synth.HD1080}[a.res]
    scene = pkg.synth.Scene(cam, 0x5EED0002, disparity_px=2.6)
    for i, s in enumerate(boundary):  # the same image in the source slot and in the slot that receives its map
        im, _ = scene.render(s, device="cpu")
        im = im.numpy()
        eng.upload_image(s, im, scene.K(), scene.Tcw(s))
        eng.upload_image(a.kfs + i, im, scene.K(), scene.Tcw(s))
    eng.assume_pipeline_maps(boundary)  # all-zero maps are pipeline maps; the copies do not depend on the values
    entries = (max(eng.active_count(s) for s in boundary) + 63) // 64 * 64
    eng.exchange_compact(entries)
    wire = "%d list entries per map (%.2f of %.2f MB)" % (entries, 8e-6 * entries, 8e-6 * W * H)
for label, n in (("warm-up", 20), ("timed", a.iters)):
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.allgather_begin(a.nbrs)
        eng.allgather_piece(boundary)
        eng.allgather_finish(fetch)
    t_enq = time.perf_counter() - t0
    eng.synchronize()
    t_all = time.perf_counter() - t0
    if label == "timed":
        print("%s %s N=%d: %.1f us per pass on the GPU queues, %.1f us of host time to enqueue it  (%s)" % (
            a.res, os.path.basename(pkg.lib_path()), a.nbrs, t_all / n * 1e6, t_enq / n * 1e6,
            ("RCCL one-rank all-gather" if os.environ.get("SDM_COMM_SINGLE_RANK_RCCL") == "1" else "device-copy gather") + ", " + wire))
if a.compact:
    assert eng.exchange_mismatches() == 0
eng.close()
