"""Per-stage launch times (K1, K2+K3, K4+K5) of one step on a bench workload for an engine build (SDM_LIB_PATH selects the
.so): medians over rounds of HIP-event stage times.  usage: python tools/stage_time.py [--res 480p --kfs 64 --nbrs 20]"""
import argparse
import hashlib
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
ap.add_argument("--noise", action="store_true", help="i.i.d. uniform u8 images")
a = ap.parse_args()
pkg = sdm_pkg.load()
wl = bench.Workload(pkg, torch, a.res, a.kfs, a.nbrs, a.disparity, 1, 0, 0, noise=a.noise)
eng = wl.eng
for _ in range(3):
    wl.step("halo", "torch")
eng.enable_timing(True)
acc = {}
for _ in range(a.rounds):
    eng.get_timing(reset=True)
    for _ in range(a.reps):
        wl.step("halo", "torch")
    eng.synchronize()
    for s, (ms, n) in eng.get_timing().items():
        acc.setdefault(s, []).append(ms / max(n, 1))
med = {s: sorted(v)[len(v) // 2] for s, v in acc.items()}
h = hashlib.sha256()
for k in (0, a.kfs // 2, a.kfs - 1):
    r, s = eng.download_depth(k)
    h.update(r.tobytes()); h.update(s.tobytes()); h.update(eng.download_checked(k).tobytes()); h.update(eng.download_pointset(k).tobytes())
print("%s  K1 %.4f  K2+K3 %.4f  K4+K5 %.4f  step %.4f ms  sha %s" % (
    os.path.basename(os.environ.get("SDM_LIB_PATH", "default")), med["search_fuse"], med["intra"], med["inter"],
    sum(med.values()), h.hexdigest()[:12]))
