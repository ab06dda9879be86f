"""Offline model of K1's scan on a bench scene (EXPERIMENTS.md round 5): for one reference keyframe and its N neighbours, per
lane and per wave (64 consecutive active-list entries x one neighbour) -- the range length L, the candidates that pass the row
test + gradient gate, the candidates that pass all three gates, the rows a lane's segment crosses -- i.e. how many iterations each
scan design would walk:
  now          slots = ceil(Lmax / 4) * 4 per wave-search, every block of the body executed per slot
  defer-cost   phase 1: slots of gates only; phase 2: max over the lanes of #gate-passing candidates
  grad-mask    phase 1: one mask word per (row run, 64 columns); phase 2: max over the lanes of #gradient-passing candidates
CPU only (numpy + the oracle's pre-pass).  usage: long_scan_model.py [res] [N] [disparity] [prior_spread]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sdm_pkg  # noqa: E402
import np_pm  # noqa: E402
from np_pm import f32  # noqa: E402
from pm_oracle import Oracle  # noqa: E402

pkg = sdm_pkg.load()
synth = pkg.synth
res = sys.argv[1] if len(sys.argv) > 1 else "480p"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
disp = float(sys.argv[3]) if len(sys.argv) > 3 else 2.6
spread = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
cam = {"480p": synth.TUM1, "720p": synth.HD720}[res]
seed = {"480p": 0x5EED0002, "720p": 0x5EED0003}[res]
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
ys, xs = np.nonzero(ref.grad[2:H - 2, 2:W - 2] >= 8.0)
ys += 2
xs += 2
mu = np.float32(scene.z0)
sd = np.float32(spread) * mu
maxd = float(np.float32(1) / (mu + np.float32(2) * sd))
mind = float(np.float32(1) / (mu - np.float32(2) * sd))
n = len(xs)
NB_BINS = int(os.environ.get("NB_BINS", "6"))
acc = dict(p2_bin=0, nbin=0, waves=0, slots=0, p2_pass=0, p2_grad=0, steps_mask=0, L=0, ngrad=0, npass=0, lanes=0, absab=[], runs=0)
for j in nbrs:
    nb = make(j)
    pair = np_pm.Pair(ref, nb)
    F = pair.F12
    x, y = xs.astype(f32), ys.astype(f32)
    with np.errstate(all="ignore"):
        a = x * F[0, 0] + y * F[1, 0] + F[2, 0]
        b = x * F[0, 1] + y * F[1, 1] + F[2, 1]
        c = x * F[0, 2] + y * F[1, 2] + F[2, 2]
        ab = (a / b).astype(f32)
        cb = (c / b).astype(f32)
    umin, umax = np_pm.search_range(ref, pair, xs, ys, mind, maxd)
    lo = np.ceil(umin).astype(np.int64)
    hi = np.minimum(np.floor(umax).astype(np.int64), W - 1)
    alive = (np.abs(ab) <= 4) & (hi >= lo)
    L = np.where(alive, hi - lo + 1, 0)
    th_line = np_pm.fast_atan2_x1(-ab)
    apr = ref.theta[ys, xs]
    ngrad = np.zeros(n, np.int64)
    nbin = np.zeros(n, np.int64)
    npass = np.zeros(n, np.int64)
    b0 = np.floor(np.mod(apr - 45.0 - 0.01, 360.0) * (16.0 / 360.0)).astype(np.int64)
    rows_lo = np.zeros(n, np.int64)
    rows_hi = np.zeros(n, np.int64)
    for t in range(int(L.max())):
        uj = lo + t
        act = alive & (uj <= hi)
        ujc = np.where(act, uj, 0)
        inner = ab * ujc.astype(f32) + cb
        vj = -np.trunc(np.where(np.isfinite(inner), inner, 0)).astype(np.int64)
        if t == 0:
            rows_lo = vj.copy()
        rows_hi = np.where(act, vj, rows_hi)
        ok = act & (vj >= 1) & (vj <= H - 2)
        vjc = np.where(ok, vj, 1)
        ok &= ~(nb.grad[vjc, ujc] < 8.0)
        ngrad += ok
        th2 = nb.theta[vjc, ujc]
        bn = np.floor(th2 * (16.0 / 360.0)).astype(np.int64)
        nbin += ok & (((bn - b0) & 15) < NB_BINS)
        d = np_pm._wrap_diff(th2 - th_line)
        d = np.where(d > 90, f32(180) - d, d)
        ok &= ~(d > 80)
        ok &= ~(np_pm._wrap_diff(th2 - apr) > 45)
        npass += ok
    runs = np.where(alive, np.abs(rows_hi - rows_lo) + 1, 0)
    # a lane's mask steps: one per (row run x 64-column word): ~ runs + words crossed
    words = np.where(alive, (hi >> 6) - (lo >> 6) + 1, 0)
    steps = runs + words - 1
    steps = np.where(alive, steps, 0)
    acc["absab"].append(np.abs(ab[alive]))
    for w0 in range(0, n, 64):
        s = slice(w0, w0 + 64)
        Lw = int(L[s].max())
        acc["waves"] += 1
        acc["slots"] += (Lw + 3) // 4 * 4
        acc["p2_pass"] += int(npass[s].max())
        acc["p2_grad"] += int(ngrad[s].max())
        acc["p2_bin"] += int(nbin[s].max())
        acc["steps_mask"] += int(steps[s].max())
    acc["L"] += int(L.sum())
    acc["ngrad"] += int(ngrad.sum())
    acc["npass"] += int(npass.sum())
    acc["nbin"] += int(nbin.sum())
    acc["lanes"] += n
    acc["runs"] += int(runs.sum())
w = acc["waves"]
absab = np.concatenate(acc["absab"])
print("%s N=%d disparity %.1f prior spread %.2f: %d active pixels, %d wave-searches" % (res, N, disp, spread, n, w))
print("per lane: L %.2f, gradient-gate pass %.2f (%.1f %%), all gates %.2f (%.1f %%), rows crossed %.2f" % (
    acc["L"] / acc["lanes"], acc["ngrad"] / acc["lanes"], 100.0 * acc["ngrad"] / max(acc["L"], 1),
    acc["npass"] / acc["lanes"], 100.0 * acc["npass"] / max(acc["L"], 1), acc["runs"] / acc["lanes"]))
print("|a/b|: median %.4f p90 %.4f p99 %.4f max %.4f" % tuple(np.percentile(absab, [50, 90, 99, 100])))
print("per wave-search: slots now %.2f | defer-cost phase 2 iterations %.2f | grad-mask: mask steps %.2f, phase 2 iterations %.2f" % (
    acc["slots"] / w, acc["p2_pass"] / w, acc["steps_mask"] / w, acc["p2_grad"] / w))
print("orientation bins (16 x 22.5 deg, %d from ang-45): listed per lane %.2f, phase 2 iterations per wave-search %.2f" % (
    NB_BINS, acc["nbin"] / acc["lanes"], acc["p2_bin"] / w))
# instruction model (wave-instructions per wave-search; EXPERIMENTS.md: skeleton ~15, gates ~17, cost ~25 per slot)
SK, GA, CO = 15, 17, 25
now = acc["slots"] / w * (SK + GA + CO)
defer = acc["slots"] / w * (SK + GA + 3) + acc["p2_pass"] / w * (CO + 14)
mask = acc["steps_mask"] / w * 30 + acc["p2_grad"] / w * (14 + GA + CO)
print("scan instruction model: now %.0f | defer-cost %.0f | grad-mask %.0f" % (now, defer, mask))
