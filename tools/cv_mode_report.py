"""How far could OpenCV's cv::Mat rounding move the results?  ("parity unpinned")

Runs the NumPy restatement of PM.cc (tests/np_pm.py) in its two arithmetic modes on the golden fixtures and
on a keyframe of BASELINE.json configs[1] (640x480, N = 20):
  n1  float left-to-right small-matrix algebra -- what the oracle and the engine implement (bit-equal to both);
  cv  OpenCV-3.x MatExpr / cv::gemm semantics restated from memory (OpenCV is absent from the image).
Reports support-mask flips and the distribution of relative differences per stage.  Test infrastructure.

    python tools/cv_mode_report.py            # prints a markdown table
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def rel(a, b):
    with np.errstate(all="ignore"):
        return np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.maximum(np.abs(a.astype(np.float64)), 1e-30)


def stage_stats(a, b):
    """a, b: lists of maps (n1, cv)"""
    sup = flips = ndiff = 0
    rels = []
    for x, y in zip(a, b):
        ma, mb = x > 1e-6, y > 1e-6
        sup += int(ma.sum())
        flips += int((ma != mb).sum())
        both = ma & mb
        r = rel(x[both], y[both])
        ndiff += int((x[both].view(np.uint32) != y[both].view(np.uint32)).sum())
        rels.append(r)
    r = np.concatenate(rels) if rels else np.zeros(0)
    q = lambda p: float(np.quantile(r, p)) if len(r) else 0.0
    return dict(support=sup, mask_flips=flips, values_differ=ndiff, rel_p50=q(0.5), rel_p99=q(0.99), rel_max=q(1.0),
                over_1e4=int((r > 1e-4).sum()))


def run(seq, n, refs, with_inter=True):
    import np_pm
    kfs = [np_pm.KF(seq.im[k], seq.grad[k], seq.theta[k], seq.istd[k], seq.K, seq.Tcw[k]) for k in range(seq.n_kf)]
    need = sorted(set(refs) | ({j for k in refs for j in seq.neighbours(k, n)} if with_inter else set()))
    out = {}
    for mode in ("n1", "cv"):
        maps, pairs = {}, {}
        for k in need:
            nb = seq.neighbours(k, n)
            pairs[k] = [np_pm.Pair(kfs[k], kfs[j], mode) for j in nb]
            r, s, _ = np_pm.recon_search_fuse(kfs[k], [kfs[j] for j in nb], pairs[k], seq.min_depth, seq.max_depth)
            r2, s2 = np_pm.intra_check(r, s)
            maps[k] = (r, s, r2, s2)
        out[mode] = (maps, pairs)
    a, b = out["n1"][0], out["cv"][0]
    res = {"K1 rho (search+fusion)": stage_stats([a[k][0] for k in need], [b[k][0] for k in need]),
           "K1 sigma": stage_stats([a[k][1] for k in need], [b[k][1] for k in need]),
           "K2 rho (intra check)": stage_stats([a[k][2] for k in need], [b[k][2] for k in need])}
    if with_inter:  # both modes on the SAME (n1) input maps: isolates K4's own arithmetic
        ca, cb = [], []
        for k in refs:
            nb = seq.neighbours(k, n)
            args = ([kfs[j] for j in nb],)
            maps_r, maps_s = [a[j][2] for j in nb], [a[j][3] for j in nb]
            ca.append(np_pm.inter_check(kfs[k], a[k][2], args[0], out["n1"][1][k], maps_r, maps_s))
            cb.append(np_pm.inter_check(kfs[k], a[k][2], args[0], out["cv"][1][k], maps_r, maps_s))
        res["K4 rho (inter check)"] = stage_stats(ca, cb)
    return res


def main():
    import sdm_pkg
    from pm_oracle import Oracle
    import golden_util as gu
    from common import Sequence
    pkg, o = sdm_pkg.load(), Oracle("strict")
    rows = []
    for name in gu.fixture_names():
        g = gu.load(name)
        seq = gu.sequence_from(pkg, o, g)
        rows.append((name, run(seq, g["n"], list(range(seq.n_kf)))))
    seq = Sequence(pkg, o, 640, 480, 41, 0x5EED0002)
    rows.append(("configs[1] sample: 640x480, N=20, keyframe 20 (+ its 20 neighbours for K4)", run(seq, 20, [20])))
    print("| workload | stage | support px | mask flips | values differ | rel. diff p50 | p99 | max | > 1e-4 |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, res in rows:
        for stage, s in res.items():
            print("| %s | %s | %d | %d | %d | %.1e | %.1e | %.1e | %d |" % (
                name, stage, s["support"], s["mask_flips"], s["values_differ"], s["rel_p50"], s["rel_p99"], s["rel_max"],
                s["over_1e4"]))


if __name__ == "__main__":
    main()
