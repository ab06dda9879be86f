"""The two-plane scene with rolled keyframes (synth.Scene(strip=True, roll_deg=5); SURVEY.md App. D's optional second plane) on
the engine: how the harder data behaves -- scan statistics, fusion's open-pixel rate, the share of reconstructed pixels the
inter-keyframe check rejects (PM.cc:762-765), and the absolute accuracy against the analytic ground truth per plane -- next to
the one-plane scene.  usage (GPU box): python tools/scene_report.py [--res 480p --kfs 64 --nbrs 20 --spread 0.3]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sdm_pkg  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="480p")
ap.add_argument("--kfs", type=int, default=64)
ap.add_argument("--nbrs", type=int, default=20)
ap.add_argument("--spread", type=float, default=0.3)
ap.add_argument("--roll", type=float, default=5.0)
a = ap.parse_args()
pkg = sdm_pkg.load()
for label, strip, roll in (("one plane, roll +-1 deg (App. D)", False, 1.0), ("plane + foreground strip, roll +-%g deg" % a.roll, True, a.roll)):
    wl = bench.Workload(pkg, torch, a.res, a.kfs, a.nbrs, 2.6, 1, 0, 0, spread=a.spread, strip=strip, roll=roll)
    eng, pl = wl.eng, wl.pl
    st = wl.scan_stats()
    wl.step("halo", "torch")
    eng.synchronize()
    fused = kept = 0
    err = {"background": [], "strip": []}
    for k in pl["own"][8:-8:6]:  # a sample of keyframes away from the ends of the sequence
        _, gt = wl.scene.render(k, device="cuda")
        fg = wl.scene.last_fg
        gt = gt.cpu().numpy()
        fg = np.zeros_like(gt, bool) if fg is None else fg.cpu().numpy()
        rho, _ = eng.download_depth(pl["slot"][k])
        chk = eng.download_checked(pl["slot"][k])
        fused += int((rho > 1e-6).sum())
        kept += int((chk > 1e-6).sum())
        m = chk > 1e-6
        err["background"].append(np.abs(chk - gt)[m & ~fg])
        err["strip"].append(np.abs(chk - gt)[m & fg])
    print(label)
    print("   scan: %.2f candidates per search, gate pass %.1f %%, hypotheses per search %.3f; mask-scan waves %d" % (
        st["candidates"] / max(st["searches"], 1), 100.0 * st["gate_pass"] / max(st["candidates"], 1),
        st["hypotheses"] / max(st["searches"], 1), st["mask_waves"]))
    print("   fusion: %d pixels fused, open pixels (all-pairs count) %d = %.2f %% of the fused ones" % (
        st["fused"], st["open_pixels"], 100.0 * st["open_pixels"] / max(st["fused"], 1)))
    print("   inter-keyframe check: %d of %d reconstructed pixels rejected (%.1f %%) on the sampled keyframes" % (
        fused - kept, fused, 100.0 * (fused - kept) / max(fused, 1)))
    for name, e in err.items():
        e = np.concatenate(e) if e else np.zeros(0)
        if len(e):
            print("   accuracy, %-10s: %7d pixels, |rho - rho_gt| median %.2e, p90 %.2e, share above 0.05: %.2f %%" % (
                name, len(e), np.median(e), np.percentile(e, 90), 100.0 * (e > 0.05).mean()))
    wl.close()
