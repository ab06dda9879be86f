"""GPU: randomised geometry fuzzing of the whole path against the oracle -- large rotations, vertical and
forward baselines (|a/b| > 4 gate, epipoles inside the image), per-keyframe intrinsics, random in-plane
rotations and depth priors (including swapped / negative / huge bounds), textured and noise images.
Everything must stay bit-identical, including the NaN/Inf propagation rules of the reference."""
import math
import os

import numpy as np
import pytest

from common import assert_bit_equal

pytestmark = pytest.mark.gpu


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_case(rng, oracle, W, H, n_kf, mode):
    """returns dict(im, K, Tcw, okf) with random but reproducible content"""
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.zeros((H, W))
    for _ in range(12):
        fx_, fy_ = rng.uniform(0.02, 0.5, 2)
        base += rng.uniform(10, 40) * np.sin(fx_ * xx + fy_ * yy + rng.uniform(0, 6.28))
    ims, Ks, Ts = [], [], []
    for k in range(n_kf):
        shift = rng.integers(-4, 5, 2)
        im = np.roll(base, tuple(shift), axis=(0, 1)) + 127 + rng.normal(0, 3, (H, W))
        if mode == "noise":
            im = rng.integers(0, 256, (H, W)).astype(np.float64)
        ims.append(np.clip(np.rint(im), 0, 255).astype(np.uint8))
        f = rng.uniform(0.7, 1.6) * W
        Ks.append(np.float32([f, f * rng.uniform(0.9, 1.1), W / 2 + rng.uniform(-5, 5), H / 2 + rng.uniform(-5, 5)]))
        amp = {"small": 0.02, "large": 0.25, "noise": 0.05, "forward": 0.03, "vertical": 0.03}[mode]
        R = rot(*rng.uniform(-amp, amp, 3))
        if mode == "forward":
            t = np.array([rng.normal(0, 0.002), rng.normal(0, 0.002), rng.uniform(-0.1, 0.1)])
        elif mode == "vertical":
            t = np.array([rng.normal(0, 0.003), rng.uniform(-0.08, 0.08), rng.normal(0, 0.003)])
        else:
            t = rng.uniform(-0.06, 0.06, 3)
        Ts.append(np.concatenate([R, t[:, None]], axis=1).astype(np.float32))
    okf = []
    derived = []
    for k in range(n_kf):
        g, th, s = oracle.gradient_prepass(ims[k])
        derived.append((g, th, s))
        okf.append(oracle.keyframe(ims[k], g, th, s, Ks[k], Ts[k]))
    return dict(im=ims, K=Ks, Tcw=Ts, okf=okf, derived=derived)


_EXTRA = int(os.environ.get("SDM_FUZZ_GEOM", "0"))  # deeper one-off runs: that many more seeds per mode
_CASES = [("small", 1), ("small", 2), ("large", 3), ("large", 4), ("forward", 5), ("vertical", 6), ("noise", 7),
          ("small", 8), ("large", 9), ("forward", 10)]
_CASES += [(mode, 100 + i) for i in range(_EXTRA) for mode in ("small", "large", "forward", "vertical", "noise")]


@pytest.mark.parametrize("mode,seed", _CASES)
def test_fuzz_whole_path(pkg, oracle, gpu_ok, mode, seed):
    rng = np.random.default_rng(1000 + seed)
    W, H, n_kf, n = 72, 56, 7, 5
    case = make_case(rng, oracle, W, H, n_kf, mode)
    eng = pkg.Engine(W, H, n_kf, max_neighbours=n)
    if seed % 3 == 0:
        # every third case runs under thresholds other than PM.h:38-49's defaults: the closed-form gates / approximate arg-min
        # with run-time constants (validated on the device by sdm_set_params) or, where they do not hold, the reference statements
        from pm_oracle import Oracle
        oracle = Oracle("strict")
        prm = dict(lambdaG=float(rng.choice([6.0, 8.0, 11.5])), lambdaL=float(rng.choice([80.0, 70.0, 45.5, 89.0, 95.0, -3.0])),
                   lambdaTheta=float(rng.choice([45.0, 40.0, 12.25, 56.0, 120.0, 181.0, 0.0])),
                   theta_var=float(rng.choice([0.23, 0.25, 0.05, 7.0])), lambdaN=int(rng.choice([3, 2, 4])))
        for key, val in prm.items():
            setattr(oracle.params, key, val)
        eng.set_params(**prm)
    for k in range(n_kf):
        if k % 2:
            eng.upload_image(k, case["im"][k], case["K"][k], case["Tcw"][k])
        else:
            g, th, s = case["derived"][k]
            eng.upload_keyframe(k, case["im"][k], g, th, s, case["K"][k], case["Tcw"][k])
    refs = list(range(n_kf))
    nbrs = [[int(j) for j in rng.permutation([j for j in range(n_kf) if j != k])[:n]] for k in refs]
    rots = rng.uniform(-30, 390, (n_kf, n)).astype(np.float32)
    rots[rng.random((n_kf, n)) < 0.4] = 0
    priors = [(1.25, 0.83), (0.5, 2.0), (2.0, 0.5), (-1.0, 1.0), (1e-3, 1e3), (1.0, 1.0), (0.9, 1.1)]
    mind = np.float32([priors[(k + seed) % len(priors)][0] for k in refs])
    maxd = np.float32([priors[(k + seed) % len(priors)][1] for k in refs])
    eng.recon(refs, nbrs, mind, maxd, rot=rots)
    rho, sig = {}, {}
    fused = 0
    for k in refs:
        r, s, st = oracle.semi_dense_recon(case["okf"][k], [case["okf"][j] for j in nbrs[k]], rots[k], float(mind[k]),
                                           float(maxd[k]))
        gr, gs = eng.download_depth(k)
        assert_bit_equal(gr, r, "%s/%d rho kf %d" % (mode, seed, k))
        assert_bit_equal(gs, s, "%s/%d sigma kf %d" % (mode, seed, k))
        rho[k], sig[k] = r, s
        fused += st["fused"]
    eng.inter_check(refs, nbrs)
    eng.pointset(refs, source=1)
    for k in refs:
        c = oracle.inter_check(case["okf"][k], rho[k], [case["okf"][j] for j in nbrs[k]], [rho[j] for j in nbrs[k]],
                               [sig[j] for j in nbrs[k]])
        assert_bit_equal(eng.download_checked(k), c, "%s/%d checked kf %d" % (mode, seed, k))
        assert_bit_equal(eng.download_pointset(k), oracle.pointset(case["okf"][k], c), "%s/%d xyz kf %d" % (mode, seed, k))
    # per-pixel entry point on random pixels, including the image border
    for _ in range(40):
        a, b = rng.choice(n_kf, 2, replace=False)
        x, y = int(rng.integers(0, W)), int(rng.integers(0, H))
        got = eng.epipolar_search(int(a), int(b), x, y, float(mind[a]), float(maxd[a]), float(rots[a, 0]))
        ref = oracle.epipolar_search(case["okf"][a], case["okf"][b], x, y, float(mind[a]), float(maxd[a]), float(rots[a, 0]))
        assert got["supported"] == ref["supported"]
        assert_bit_equal(np.float32([got["rho"], got["sigma"], got["best_u"], got["best_v"]]),
                         np.float32([ref["rho"], ref["sigma"], ref["best_u"], ref["best_v"]]), "pixel search")
    eng.close()


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_fuzz_hostile_planes(pkg, oracle, gpu_ok, seed):
    """caller-supplied GradImg / GradTheta / I_stddev with NaN, Inf, negative, huge and out-of-range values
    (the reference never validates them): gates, costs, fusion and the checks must still agree bit for bit"""
    rng = np.random.default_rng(2000 + seed)
    W, H, n_kf, n = 96, 64, 6, 5
    case = make_case(rng, oracle, W, H, n_kf, "small")
    bad_g = np.float32([np.nan, np.inf, -np.inf, -5.0, 1e30, 8.0, 7.9999995, 0.0, 3e38])
    bad_t = np.float32([np.nan, np.inf, -np.inf, -10.0, 360.0, 725.5, 1e9, -1e9, 359.99997, -0.0, 1080.0, -720.25])
    odd_istd = {1: [0.0, float("nan"), float("inf"), 1e-30][seed % 4], 4: -3.0}  # two of the six keyframes
    eng = pkg.Engine(W, H, n_kf, max_neighbours=n)
    okf = []
    for k in range(n_kf):
        g, th, s = case["derived"][k]
        g, th = g.copy(), th.copy()
        m = rng.random((H, W)) < 0.04
        g[m] = rng.choice(bad_g, int(m.sum()))
        m = rng.random((H, W)) < 0.04
        th[m] = rng.choice(bad_t, int(m.sum()))
        # angles a little outside [0, 360) exercise the reference form of the wrap (d >= 360 / d < -360)
        m = rng.random((H, W)) < 0.05
        th[m] = th[m] + np.float32(360.0) * rng.integers(-2, 3, int(m.sum())).astype(np.float32)
        sk = odd_istd.get(k, s)
        eng.upload_keyframe(k, case["im"][k], g, th, sk, case["K"][k], case["Tcw"][k])
        okf.append(oracle.keyframe(case["im"][k], g, th, sk, case["K"][k], case["Tcw"][k]))
    refs = list(range(n_kf))
    nbrs = [[j for j in range(n_kf) if j != k][:n] for k in refs]
    rots = rng.uniform(-400, 400, (n_kf, n)).astype(np.float32)
    rots[rng.random((n_kf, n)) < 0.7] = 0
    rots[0, 0], rots[1, 1] = np.nan, np.inf
    eng.recon(refs, nbrs, 0.2, 5.0, rot=rots)
    rho, sig = {}, {}
    for k in refs:
        r, s, _ = oracle.semi_dense_recon(okf[k], [okf[j] for j in nbrs[k]], rots[k], 0.2, 5.0)
        gr, gs = eng.download_depth(k)
        assert_bit_equal(gr, r, "hostile/%d rho kf %d" % (seed, k))
        assert_bit_equal(gs, s, "hostile/%d sigma kf %d" % (seed, k))
        rho[k], sig[k] = r, s
    assert sum(int((rho[k] > 1e-6).sum()) for k in refs) > 200, "the case must still fuse something"
    eng.inter_check(refs, nbrs)
    eng.pointset(refs, source=1)
    for k in refs:
        c = oracle.inter_check(okf[k], rho[k], [okf[j] for j in nbrs[k]], [rho[j] for j in nbrs[k]],
                               [sig[j] for j in nbrs[k]])
        assert_bit_equal(eng.download_checked(k), c, "hostile/%d checked kf %d" % (seed, k))
        assert_bit_equal(eng.download_pointset(k), oracle.pointset(okf[k], c), "hostile/%d xyz kf %d" % (seed, k))
    eng.close()
