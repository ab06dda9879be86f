"""CPU: the SECOND independent restatement of PM.cc's loops (tests/np_pm.py, NumPy, written from the reference
text) against the C oracle (oracle/pm_oracle.c) on the committed golden fixtures -- bit for bit.

The reference holds no vectors for this path and cannot be built here (parity unpinned), so what
can be checked is that two restatements made by different routes -- scalar C following PM.cc line by line, and
vectorised NumPy following PM.cc's text -- agree exactly on: the epipolar scan + sub-pixel refinement + Eq. 8/9
hypothesis (PM.cc:385-465, 806-875), the search range (877-910), the hypothesis fusion (598-626, 912-970), the
intra-keyframe check (486-547) and growing (549-596), and the inter-keyframe check with its Gauss-Newton step
(628-799).  The closed-form pieces have their own cross-check in test_oracle_crosscheck.py.

The last test bounds what "unpinned" can cost: the same restatement with OpenCV's cv::Mat rounding semantics
(restated from memory) instead of the build's normative float algebra; tools/cv_mode_report.py prints the full
table quoted in DESIGN.md §3."""
import numpy as np
import pytest

import golden_util as gu
import np_pm
from common import assert_bit_equal


def _kfs(seq):
    return [np_pm.KF(seq.im[k], seq.grad[k], seq.theta[k], seq.istd[k], seq.K, seq.Tcw[k]) for k in range(seq.n_kf)]


@pytest.mark.parametrize("name", gu.fixture_names())
def test_second_restatement_matches_oracle_and_golden(pkg, oracle, name):
    g = gu.load(name)
    seq = gu.sequence_from(pkg, oracle, g)
    n, kfs = g["n"], _kfs(seq)
    for k in range(seq.n_kf):
        nb = seq.neighbours(k, n)
        pairs = [np_pm.Pair(kfs[k], kfs[j], "n1") for j in nb]
        # pair geometry against the oracle
        for j, pr in zip(nb, pairs):
            po = oracle.pair_geometry(seq.okf[k], seq.okf[j])
            assert_bit_equal(pr.R21, np.array(po.R21[:]).reshape(3, 3), "R21")
            assert_bit_equal(pr.t21, np.array(po.t21[:]), "t21")
            assert_bit_equal(pr.F12, np.array(po.F12[:]).reshape(3, 3), "F12")
        r, s, st = np_pm.recon_search_fuse(kfs[k], [kfs[j] for j in nb], pairs, seq.min_depth, seq.max_depth,
                                           rots=seq.rot(k, n))  # (None for the one-plane fixtures: rot = 0)
        assert_bit_equal(r, g["k1_rho"][k], "%s K1 rho kf %d" % (name, k))      # scan + refine + Eq. 8/9 + fusion
        assert_bit_equal(s, g["k1_sigma"][k], "%s K1 sigma kf %d" % (name, k))
        assert st["searches"] == g["searches"][k] and st["candidates"] == g["candidates"][k]  # same scan loops
        r2, s2 = np_pm.intra_check(r, s)
        r3, s3 = np_pm.intra_grow(r2, s2, seq.grad[k])
        assert_bit_equal(r3, g["rho"][k], "%s intra rho kf %d" % (name, k))
        assert_bit_equal(s3, g["sigma"][k], "%s intra sigma kf %d" % (name, k))
        c = np_pm.inter_check(kfs[k], g["rho"][k], [kfs[j] for j in nb], pairs, [g["rho"][j] for j in nb],
                              [g["sigma"][j] for j in nb])
        assert_bit_equal(c, g["chk"][k], "%s inter-keyframe check kf %d" % (name, k))
        assert (c > 1e-6).sum() > 100


def test_second_restatement_per_pixel_search(pkg, oracle):
    """per-pixel outputs of EpipolarSearch incl. in-plane rotation and a hostile depth prior, against the oracle"""
    g = gu.load("plane_160x120_n7")
    seq = gu.sequence_from(pkg, oracle, g)
    kfs = _kfs(seq)
    ys, xs = np.nonzero(seq.grad[3][2:-2, 2:-2] >= 8)
    rng = np.random.default_rng(5)
    pick = rng.choice(len(xs), 150, replace=False)
    xs, ys = xs[pick] + 2, ys[pick] + 2
    n_sup = 0
    for nbr, rot, (mn, mx) in [(4, 0.0, (seq.min_depth, seq.max_depth)), (0, 7.5, (seq.min_depth, seq.max_depth)),
                               (7, 350.0, (seq.min_depth, seq.max_depth)), (2, 0.0, (0.5, 2.0)), (5, 0.0, (-1.0, 1.0))]:
        pr = np_pm.Pair(kfs[3], kfs[nbr], "n1")
        r, s, sup, _ = np_pm.epipolar_search(kfs[3], kfs[nbr], pr, xs, ys, mn, mx, rot)
        for i in range(len(xs)):
            ref = oracle.epipolar_search(seq.okf[3], seq.okf[nbr], int(xs[i]), int(ys[i]), mn, mx, rot)
            assert bool(sup[i]) == bool(ref["supported"]), (nbr, xs[i], ys[i])
            assert_bit_equal(np.array([r[i], s[i]]), np.array([ref["rho"], ref["sigma"]]), "search %d,%d nbr %d" % (xs[i], ys[i], nbr))
            n_sup += int(sup[i])
    assert n_sup > 200


def test_second_restatement_growing_on_crafted_maps(oracle):
    """IntraKeyFrameDepthGrowing is a no-op on pipeline maps; exercise it with centres that
    have rho < 1e-6 but sigma > 0, NaN/Inf/zero sigmas included"""
    rng = np.random.default_rng(11)
    H, W = 40, 56
    for trial in range(6):
        rho = np.where(rng.random((H, W)) < 0.6, 1.0 + 0.02 * rng.standard_normal((H, W)), 0).astype(np.float32)
        sig = (0.01 + 0.05 * rng.random((H, W))).astype(np.float32)
        sig[rng.random((H, W)) < 0.05] = 0
        if trial >= 3:
            sig[rng.random((H, W)) < 0.02] = np.inf
            rho[rng.random((H, W)) < 0.01] = np.nan
        grad = (16 * rng.random((H, W))).astype(np.float32)
        wr, ws = oracle.intra_grow(rho, sig, grad)
        gr, gs = np_pm.intra_grow(rho, sig, grad)
        assert_bit_equal(gr, wr, "grown rho %d" % trial)
        assert_bit_equal(gs, ws, "grown sigma %d" % trial)
        cr, cs = oracle.intra_check(rho, sig)
        nr, ns = np_pm.intra_check(rho, sig)
        assert_bit_equal(nr, cr, "checked rho %d" % trial)
        assert_bit_equal(ns, cs, "checked sigma %d" % trial)
    assert (wr != rho).sum() > 10, "the crafted maps must make the growing step do something"


def test_cv_rounding_mode_moves_no_support(pkg, oracle):
    """What "parity unpinned" can cost: OpenCV cv::Mat / cv::gemm rounding (restated from memory, np_pm mode="cv")
    instead of the normative float algebra (N1/N2).  On this fixture: no support-mask flip at any stage, and the
    values move by about one float ulp (stated bounds: p99 of the relative difference < 1e-5, fewer than 0.1 % of
    the supported pixels move by more than 1e-4 -- those are arg-min / tap choices that tip over)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cv_mode_report
    g = gu.load("plane_64x48_n7")
    seq = gu.sequence_from(pkg, oracle, g)
    res = cv_mode_report.run(seq, g["n"], list(range(seq.n_kf)))
    for stage, s in res.items():
        assert s["support"] > 5000, stage
        assert s["mask_flips"] <= 0.001 * s["support"], (stage, s)
        assert s["rel_p99"] < 1e-5, (stage, s)
        assert s["over_1e4"] <= 0.001 * s["support"], (stage, s)
    assert res["K1 rho (search+fusion)"]["values_differ"] > 1000, "the two modes must actually differ"
