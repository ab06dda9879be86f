"""Offline model of a skip-distance scan for K1 (EXPERIMENTS.md): for one reference keyframe of the bench scene and its N neighbours,
how many gathers would a lane need if every record carried the distance to the next gradient-passing column of its row -- against the
slots the 4-wide batched scan walks now.  Wave = 64 consecutive active-list entries, one neighbour.  CPU only (numpy + oracle)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sdm_pkg  # noqa: E402
import np_pm  # noqa: E402
from pm_oracle import Oracle  # noqa: E402

pkg = sdm_pkg.load()
synth = pkg.synth
res = sys.argv[1] if len(sys.argv) > 1 else "480p"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
disp = float(sys.argv[3]) if len(sys.argv) > 3 else 2.6
cam = {"480p": synth.TUM1, "720p": synth.HD720}[res]
seed = {"480p": 0x5EED0010, "720p": 0x5EED0020}.get(res, 1)
try:
    import bench
    seed = bench.SEEDS[res]
except Exception:
    pass
scene = synth.Scene(cam, seed, disparity_px=disp)
oracle = Oracle("strict")
W, H = cam["W"], cam["H"]
k0 = 32
nbrs = scene.neighbours(k0, 64, N)
K = scene.K()


def make(k):
    im, _ = scene.render(k)
    im = im.numpy()
    grad, theta, istd = oracle.gradient_prepass(im)
    return np_pm.KF(im, grad, theta, istd, K, scene.Tcw(k))


ref = make(k0)
lamG = 8.0
ys, xs = np.nonzero(ref.grad[2:H - 2, 2:W - 2] >= lamG)  # (inset as PM.cc:198; order = raster)
ys += 2
xs += 2
mind, maxd = scene.depth_prior()
tot_slots = tot_iters = tot_iters_rowchk = waves = 0
lane_gathers = lane_cands = 0
hist = np.zeros(40, dtype=np.int64)
for j in nbrs:
    nb = make(j)
    pair = np_pm.Pair(ref, nb)
    F = pair.F12
    x, y = xs.astype(np.float32), ys.astype(np.float32)
    a = x * F[0, 0] + y * F[1, 0] + F[2, 0]
    b = x * F[0, 1] + y * F[1, 1] + F[2, 1]
    c = x * F[0, 2] + y * F[1, 2] + F[2, 2]
    with np.errstate(all="ignore"):
        ab = (a / b).astype(np.float32)
        cb = (c / b).astype(np.float32)
    umin, umax = np_pm.search_range(ref, pair, xs, ys, mind, maxd)
    lo = np.ceil(umin).astype(np.int64)
    hi = np.minimum(np.floor(umax).astype(np.int64), W - 1)
    live = (np.abs(ab) <= 4) & (hi >= lo)
    act = ~(nb.grad < lamG)  # the scan's gate: GradImg(vj, uj) < lambdaG -> skip
    # next active column strictly after x in each row
    nxt = np.full((H, W), 1 << 20, dtype=np.int64)
    for r in range(H):
        last = 1 << 20
        row = act[r]
        for cx in range(W - 1, -1, -1):
            nxt[r, cx] = last
            if row[cx]:
                last = cx
    gathers = np.zeros(len(xs), dtype=np.int64)
    L = np.where(live, hi - lo + 1, 0)
    for i in np.nonzero(live)[0]:
        u = lo[i]
        g = 0
        while u <= hi[i]:
            g += 1
            yf = -(np.float32(ab[i] * np.float32(u)) + cb[i])
            if not (yf >= 1 and yf < H - 1):
                u += 1
                continue
            r = int(np.floor(yf))
            un = min(nxt[r, u], hi[i] + 1)
            if un > u + 1:
                t = un - 1
                yt = -(np.float32(ab[i] * np.float32(t)) + cb[i])
                if int(np.floor(yt)) != r:
                    un = u + 1
            u = un
        gathers[i] = g
    lane_gathers += gathers.sum()
    lane_cands += L.sum()
    for w0 in range(0, len(xs), 64):
        Lw = L[w0:w0 + 64].max()
        gw = gathers[w0:w0 + 64].max()
        tot_slots += int(np.ceil(Lw / 4) * 4)
        tot_iters += int(gw)
        hist[min(int(gw), 39)] += 1
        waves += 1
print("%s N=%d disparity %.1f: %d active pixels, %d waves x neighbours" % (res, N, disp, len(xs), waves))
print("per lane: candidates %.2f, gathers with skip %.2f" % (lane_cands / (len(xs) * N), lane_gathers / (len(xs) * N)))
print("per wave-search: slots walked now %.2f (batches %.2f), iterations with skip %.2f" % (tot_slots / waves, tot_slots / waves / 4, tot_iters / waves))
print("iterations histogram:", hist[:16].tolist())
