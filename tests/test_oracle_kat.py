"""CPU: hand-derived known-answer tests of the oracle for the semantics SURVEY.md App. A singles
out (threshold promotion, tie-breaking, NaN/Inf behaviour, border rules)."""
import numpy as np
import pytest

from common import assert_bit_equal

f32 = np.float32


def u2f(u):
    return np.array([u], np.uint32).view(np.float32)[0]


def _operands_for_chi(target):
    """float32 (d, sa) with fl(fl(d*d) / fl(sa*sa)) == target exactly (sigma_b = Inf kills the 2nd term)"""
    for k in range(3, 64):
        sa = f32(k) / f32(8)
        s2 = f32(sa * sa)
        centre = f32(np.sqrt(np.float64(target) * np.float64(s2)))
        ds = centre + np.arange(-4000, 4000, dtype=np.float32) * np.spacing(centre)
        chi = ((ds * ds).astype(f32) / s2).astype(f32)
        hit = np.nonzero(chi == target)[0]
        for i in hit:
            d = ds[i]
            if f32((f32(1.0) + d) - f32(1.0)) == d:  # rho_b = 1 + d must give back d exactly
                return d, sa
    return None


def test_chi_threshold_is_double_599(oracle):
    """(double)chi < 5.99  <=>  chi <= 0x40BFAE14 (App. A.0); `chi < 5.99f` would be wrong there."""
    lo, hi = u2f(0x40BFAE14), u2f(0x40BFAE15)
    assert float(lo) < 5.99 < float(hi) and f32(5.99) == lo
    for target, compatible in [(lo, True), (hi, False)]:
        ops = _operands_for_chi(target)
        assert ops is not None, "no exact operands found for the boundary value"
        d, sa = ops
        # hypotheses 0,1,2 identical (always mutually compatible); #3 differs by d with sigma_b = Inf:
        # it joins their set iff chi(0,3) = d^2/sa^2 < 5.99.  Its presence changes the fused sigma.
        rho = np.array([1, 1, 1, f32(1.0) + d], f32)
        sig = np.array([sa, sa, sa, np.inf], f32)
        r, s, ok = oracle.fuse(rho, sig)
        base = oracle.fuse(rho[:3], sig[:3])
        assert ok == 1
        # an Inf-sigma member adds 0 weight: the fused values equal the 3-set's either way, so observe
        # the decision through intra_check instead: centre + 2 neighbours needed (>= 3 incl. itself)
        H = W = 7
        R = np.zeros((H, W), f32)
        S = np.zeros((H, W), f32)
        R[3, 3], S[3, 3] = 1.0, sa            # centre p
        R[2, 2], S[2, 2] = f32(1.0) + d, np.inf  # neighbour tested as ChiTest(n, p, sigma_n=Inf, sigma_p=sa)
        R[2, 3], S[2, 3] = 1.0, sa            # a second, trivially compatible neighbour
        r2, s2 = oracle.intra_check(R, S)
        # with the boundary neighbour compatible the centre has 3 members and survives; otherwise 2 -> zeroed
        assert (r2[3, 3] > 0) == compatible, (target, compatible, r2[3, 3])


def test_fusion_first_largest_set_wins(oracle):
    # two disjoint clusters of 3: the first (a = 0) must win (strict '>' at PM.cc:616)
    r, s, ok = oracle.fuse(f32([1, 1, 1, 2, 2, 2]), f32([0.01] * 6))
    assert ok == 1 and abs(r - 1.0) < 1e-6
    r, s, ok = oracle.fuse(f32([2, 2, 2, 1, 1, 1]), f32([0.01] * 6))
    assert ok == 1 and abs(r - 2.0) < 1e-6
    # fused sigma = sqrt(1/sum(1/s^2)) = 0.01/sqrt(3)
    assert abs(s - 0.01 / np.sqrt(3)) < 1e-8


def test_fusion_needs_three_and_zero_sigma_is_nan(oracle):
    assert oracle.fuse(f32([1, 1]), f32([0.1, 0.1]))[2] == 0
    # sigma = 0 -> self test 0/0 = NaN -> never compatible (App. A.5)
    assert oracle.fuse(f32([1, 1, 1, 1]), f32([0, 0, 0, 0]))[2] == 0
    r, s, ok = oracle.fuse(f32([1, 1, 1, 1]), f32([0.1, 0, 0.1, 0.1]))
    assert ok == 1 and abs(r - 1) < 1e-6


def test_fusion_weighted_mean(oracle):
    rho, sig = f32([1.0, 1.1, 1.2]), f32([0.1, 0.2, 0.1])
    r, s, ok = oracle.fuse(rho, sig)
    w = 1 / sig.astype(np.float64) ** 2
    assert ok == 1
    assert abs(r - (rho * w).sum() / w.sum()) < 1e-6
    assert abs(s - np.sqrt(1 / w.sum())) < 1e-7


def test_intra_check_rules(oracle):
    H, W = 9, 9
    rho = np.zeros((H, W), f32)
    sig = np.zeros((H, W), f32)
    # centre + 2 compatible neighbours -> kept: rho = weighted mean, sigma = MIN sigma (PM.cc:531)
    rho[4, 4], sig[4, 4] = 1.0, 0.2
    rho[3, 3], sig[3, 3] = 1.1, 0.1
    rho[4, 5], sig[4, 5] = 0.9, 0.3
    # isolated pixel -> zeroed (PM.cc:535-536)
    rho[6, 2], sig[6, 2] = 1.0, 0.1
    # border ring (row 1) is outside the 2-px inset: untouched
    rho[1, 4], sig[1, 4] = 1.0, 0.1
    r, s = oracle.intra_check(rho, sig)
    w = 1 / np.float64([0.1, 0.3, 0.2]) ** 2  # raster order of neighbours, then itself
    assert abs(r[4, 4] - (np.float64([1.1, 0.9, 1.0]) * w).sum() / w.sum()) < 1e-6
    assert s[4, 4] == f32(0.1)
    assert r[6, 2] == 0 and s[6, 2] == 0
    assert r[1, 4] == 1.0 and s[1, 4] == f32(0.1)
    # Jacobi: (3,3) sees the ORIGINAL centre, not the updated one
    w2 = 1 / np.float64([0.2, 0.1]) ** 2
    assert r[3, 3] == 0  # only 1 neighbour + itself = 2 < 3


def test_growing_noop_when_centre_sigma_zero(oracle):
    """App. A.6: rho_c < 1e-6 with sigma_c = 0 -> ChiTest gives Inf/NaN -> nothing grows"""
    H, W = 9, 9
    rho = np.full((H, W), 1.0, f32)
    sig = np.full((H, W), 0.5, f32)
    rho[4, 4], sig[4, 4] = 0.0, 0.0
    grad = np.full((H, W), 20, f32)
    r, s = oracle.intra_grow(rho, sig, grad)
    assert_bit_equal(r, rho)
    assert_bit_equal(s, sig)
    # live when sigma_c > 0 and neighbours are close enough: chi = d^2/sn^2 + d^2/sc^2
    rho[:] = 0.5
    sig[:] = 1.0
    rho[4, 4], sig[4, 4] = 0.0, 1.0
    r, s = oracle.intra_grow(rho, sig, grad)
    assert r[4, 4] == f32(0.5) and s[4, 4] == f32(1.0)
    # and blocked by the gradient gate (PM.cc:562)
    grad[4, 4] = 7.9
    r, s = oracle.intra_grow(rho, sig, grad)
    assert r[4, 4] == 0


def test_gradient_prepass_known_values(oracle):
    """horizontal ramp of slope 3 gray levels/px: Scharr/32 gives |g| = 3, theta = 0 deg"""
    H, W = 12, 40
    im = np.tile((np.arange(W) * 3).astype(np.uint8), (H, 1))
    g, t, s = oracle.gradient_prepass(im)
    assert (g[:, 1:-1] == 3.0).all() and (t[:, 1:-1] == 0.0).all()
    # replicated border halves the central difference at the edge
    assert (g[:, 0] == 1.5).all()
    assert abs(s - np.std(im.astype(np.float64))) < 1e-4
    g2, t2, _ = oracle.gradient_prepass(im.T.copy())
    assert (g2[1:-1] == 3.0).all()
    assert np.allclose(t2[1:-1], 90.0, atol=0.01)


def test_search_needs_interior_match(pkg, oracle):
    """N3/N4: a best match on the outermost column of the neighbour image yields no hypothesis"""
    from common import Sequence
    seq = Sequence(pkg, oracle, 64, 48, 4, 0x5EED0D01, disparity_px=2.0)
    any_sup = 0
    for y in range(2, 46):
        for x in (2, 3, 60, 61):
            h = oracle.epipolar_search(seq.okf[1], seq.okf[2], x, y, seq.min_depth, seq.max_depth)
            if h["supported"]:
                any_sup += 1
                assert 1.0 <= h["best_u"] + 1.5 and h["best_u"] <= seq.W - 0.5
    # degenerate line: identical poses -> F12 = 0 -> a/b = NaN -> no hypothesis (N5)
    h = oracle.epipolar_search(seq.okf[1], seq.okf[1], 30, 20, seq.min_depth, seq.max_depth)
    assert h["supported"] == 0 and h["rho"] == 0


def test_median_rot(oracle):
    # shared map points 5 and 9; angle<0 pairs skipped; median = rot[(n-1)/2] of sorted diffs
    mp1, a1 = [5, -1, 9, 7], [10.0, 0.0, 350.0, -1.0]
    mp2, a2 = [9, 5, 7, 3], [5.0, 40.0, 12.0, 0.0]
    # pairs: (5: 40-10=30), (9: 5-350=-345), (7: skipped, angle1<0)
    assert oracle.median_rot_in_plane(mp1, a1, mp2, a2) == -345.0  # sorted [-345, 30], index (2-1)//2 = 0
    assert oracle.median_rot_in_plane([1], [3.0], [2], [4.0]) == 0.0
