"""K1 (k_search_fuse) launch time on the configs[1] workload for an engine build (SDM_LIB_PATH selects the .so):
kernel experiments / ablation builds.  usage: python tools/k1_time.py [--res 480p --kfs 64 --nbrs 20 --reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sdm_pkg  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="480p")
ap.add_argument("--kfs", type=int, default=64)
ap.add_argument("--nbrs", type=int, default=20)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--disparity", type=float, default=2.6)
ap.add_argument("--spread", type=float, default=0.1, help="depth prior s = spread * mu (PM.cc:381-382)")
ap.add_argument("--scan-mode", type=int, default=-1, help="sdm_set_scan_mode: 0 per wave, 1 batched, 2 gradient mask")
ap.add_argument("--params", default="", help="sdm_set_params overrides, e.g. lambdaL=70,lambdaTheta=40,theta_var=0.25")
ap.add_argument("--theta360", action="store_true",
                help="keyframes uploaded as planes (sdm_upload_keyframe) with GradTheta = 360.0 at some pixels: pairs are not "
                     "'clean' (PairConst::clean), the scan keeps the per-candidate precondition of the closed-form gates")
ap.add_argument("--check", action="store_true", help="compare K1 maps with the default library's (bit-exact)")
ap.add_argument("--outliers", type=int, default=0,
                help="replace this many of every keyframe's neighbours by a copy with a wrong pose (baseline stretched by "
                     "40 %%): their hypotheses are outliers, so no pixel is settled by the fusion shortcut")
ap.add_argument("--noise", action="store_true", help="i.i.d. uniform u8 images (nothing fuses cleanly: every pixel is open)")
a = ap.parse_args()
pkg = sdm_pkg.load()
wl = bench.Workload(pkg, torch, a.res, a.kfs, a.nbrs, a.disparity, 1, 0, 0, noise=a.noise, spread=a.spread)
eng, pl = wl.eng, wl.pl
if a.scan_mode >= 0:
    eng.set_scan_mode(a.scan_mode)
if a.theta360:
    import numpy as np
    for k in pl["inputs"]:
        im, g, th, istd = eng.download_inputs(pl["slot"][k])
        th = th.copy()
        th[::7, ::5] = np.float32(360.0)
        eng.upload_keyframe(pl["slot"][k], im, g, th, istd, wl.K, wl.scene.Tcw(k))
if a.params:
    kw = {}
    for item in a.params.split(","):
        k_, v_ = item.split("=")
        kw[k_] = int(v_) if k_ == "lambdaN" else float(v_)
    eng.set_params(**kw)
if a.outliers:
    # a second engine with kfs extra slots holding wrong-pose copies; neighbour j of keyframe k at list position
    # 3, 7, 11, ... is redirected to the copy of j
    import numpy as np
    eng.close()
    n_slots = pl["n_slots"]
    eng = pkg.Engine(wl.W, wl.H, 2 * n_slots, max_neighbours=a.nbrs, batch_capacity=64)
    for k in pl["inputs"]:
        im, _ = wl.scene.render(k, device="cuda")
        torch.cuda.synchronize()
        eng.upload_image_device(pl["slot"][k], im.data_ptr(), wl.K, wl.scene.Tcw(k))
        T = wl.scene.Tcw(k).copy()
        T[:, 3] *= np.float32(1.4)
        eng.upload_image_device(n_slots + pl["slot"][k], im.data_ptr(), wl.K, T)
    nb = [list(r) for r in pl["nbr_slots"]]
    for r in nb:
        for i in range(a.outliers):
            pos = min(3 + 4 * i, len(r) - 1)
            r[pos] = n_slots + r[pos]
    pl = dict(pl, nbr_slots=nb)
for _ in range(3):
    eng.search_fuse(pl["own_slots"], pl["nbr_slots"], wl.min_d, wl.max_d)
eng.enable_timing(True)
rounds = []
for _ in range(a.rounds):  # per-round means: the minimum / median over rounds is robust against clock drift
    eng.get_timing(reset=True)
    for _ in range(a.reps):
        eng.search_fuse(pl["own_slots"], pl["nbr_slots"], wl.min_d, wl.max_d)
    eng.synchronize()
    t = eng.get_timing()
    rounds.append(t["search_fuse"][0] / t["search_fuse"][1])
rounds.sort()
ms = rounds[len(rounds) // 2]
alg = wl.P * (17 + 9 * a.nbrs) * a.kfs
print("%s disp %.1f spread %.2f mode %d: K1 median %.4f ms (min %.4f, max %.4f over %d rounds x %d)  frac %.4f" % (
    os.path.basename(os.environ.get("SDM_LIB_PATH", "default")), a.disparity, a.spread, a.scan_mode, ms, rounds[0], rounds[-1],
    a.rounds, a.reps, alg / (ms * 1e-3) / 1e9 / 8000.0))
if a.check:
    import hashlib
    h = hashlib.sha256()
    for k in (0, a.kfs // 2, a.kfs - 1):
        r, s = eng.download_depth(k)
        h.update(r.tobytes())
        h.update(s.tobytes())
    print("maps sha", h.hexdigest()[:16])
