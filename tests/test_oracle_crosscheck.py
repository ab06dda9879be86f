"""CPU: an independent float32 NumPy restatement of the closed-form pieces of PM.cc, written from
the reference text (not from oracle/pm_oracle.c), checked against the oracle.  This is the
transcription check SURVEY.md §8c asks for: the reference holds no vectors (parity unpinned), so
two independent restatements must agree bit for bit.

NumPy float32 scalars round every operation to float32 exactly like the C oracle built with
-ffp-contract=off; float64 is used where PM.cc's promotion rules give double (App. A.0)."""
import math

import numpy as np
import pytest

from common import Sequence, assert_bit_equal

f32 = np.float32
f64 = np.float64


# ---- restatements (reference line numbers: /root/reference/src/Modeler/ProbabilityMapping.cc) ----
def np_fast_atan2(y, x):
    """cv::fastAtan2, OpenCV 3.x atan_f32"""
    y, x = f32(y), f32(x)
    scale = f32(180.0 / math.pi)
    p1 = f32(0.9997878412794807) * scale
    p3 = f32(-0.3258083974640975) * scale
    p5 = f32(0.1555786518463281) * scale
    p7 = f32(-0.04432655554792128) * scale
    ax, ay = abs(x), abs(y)
    eps = f32(np.finfo(np.float64).eps)
    if ax >= ay:
        c = ay / (ax + eps)
        c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps)
        c2 = c * c
        a = f32(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = f32(180.0) - a
    if y < 0:
        a = f32(360.0) - a
    return f32(a)


def mm(A, B):
    """3x3 float32 product, left-to-right accumulation (normative choice N1)"""
    C = np.zeros((3, 3), f32)
    for i in range(3):
        for k in range(3):
            C[i, k] = (A[i, 0] * B[0, k] + A[i, 1] * B[1, k]) + A[i, 2] * B[2, k]
    return C


def mv(A, v):
    return np.array([(A[i, 0] * v[0] + A[i, 1] * v[1]) + A[i, 2] * v[2] for i in range(3)], f32)


def np_pair(T1, T2, K1, K2):
    """PM.cc:859-860 (R21, t21) and :972-986 (F12); skew as LocalMapping.cc:711-716"""
    R1, t1, R2, t2 = T1[:, :3], T1[:, 3], T2[:, :3], T2[:, 3]
    R21 = mm(R2, R1.T.copy())
    t21 = (-mv(R21, t1)) + t2
    R12 = mm(R1, R2.T.copy())
    t12 = (-mv(R12, t2)) + t1
    z = f32(0)
    t12x = np.array([[z, -t12[2], t12[1]], [t12[2], z, -t12[0]], [-t12[1], t12[0], z]], f32)
    fx1, fy1, cx1, cy1 = K1
    fx2, fy2, cx2, cy2 = K2
    one = f32(1)
    K1ti = np.array([[one / fx1, z, z], [z, one / fy1, z], [-cx1 / fx1, -cy1 / fy1, one]], f32)
    K2i = np.array([[one / fx2, z, -cx2 / fx2], [z, one / fy2, -cy2 / fy2], [z, z, one]], f32)
    F = mm(mm(mm(K1ti, t12x), R12), K2i)
    return R21, t21.astype(f32), F


def np_xp(K, px, py):
    fx, fy, cx, cy = K
    return (f32(px) - cx) / fx, (f32(py) - cy) / fy


def rowdot(r, xp0, xp1):
    return (r[0] * xp0 + r[1] * xp1) + r[2] * f32(1)


def np_search_range(K, R21, t21, px, py, mind, maxd, W):
    """PM.cc:877-910"""
    fx, fy, cx, cy = K
    xp0, xp1 = np_xp(K, px, py)
    rx, rz = rowdot(R21[0], xp0, xp1), rowdot(R21[2], xp0, xp1)
    with np.errstate(all="ignore"):
        umin = fx * (rx * f32(mind) + t21[0]) / (rz * f32(mind) + t21[2]) + cx
        umax = fx * (rx * f32(maxd) + t21[0]) / (rz * f32(maxd) + t21[2]) + cx
    if umin > umax:
        umin, umax = umax, umin
    if umin < 0:
        umin = f32(0)
    if umax < 0:
        umax = f32(0)
    if umin > W:
        umin = f32(W)
    if umax > W:
        umax = f32(W)
    return f32(umin), f32(umax)


def np_pixel_depth(K, R21, t21, uj, px, py):
    """PM.cc:845-875, Eq. 8"""
    fx, fy, cx, cy = K
    ucx = f32(uj) - cx
    xp0, xp1 = np_xp(K, px, py)
    num1 = rowdot(R21[2], xp0, xp1) * ucx
    num2 = fx * rowdot(R21[0], xp0, xp1)
    den1 = -t21[2] * ucx
    den2 = fx * t21[0]
    with np.errstate(all="ignore"):
        return f32((num1 - num2) / (den1 + den2))


def np_chi(a, b, sa, sb):
    """PM.cc:912-924: float arithmetic, compared against the DOUBLE literal 5.99"""
    a, b, sa, sb = f32(a), f32(b), f32(sa), f32(sb)
    with np.errstate(all="ignore"):
        num = (a - b) * (a - b)
        chi = num / (sa * sa) + num / (sb * sb)
    return bool(f64(chi) < 5.99)


def np_fusion_b(rho, sig):
    """GetFusion overload B, PM.cc:947-970: pow() promotes each term to double, sums are float"""
    pjsj, rsj, tmin = f32(0), f32(0), f32(sig[0])
    with np.errstate(all="ignore"):
        for r, s in zip(rho, sig):
            s2 = f64(s) * f64(s)
            pjsj = f32(f64(pjsj) + f64(r) / s2)
            rsj = f32(f64(rsj) + f64(1.0) / s2)
            if s2 < f64(tmin) * f64(tmin):
                tmin = f32(s)
        return f32(pjsj / rsj), f32(np.sqrt(f32(1) / rsj)), tmin


def np_fuse(rho, sig, lambdaN=3):
    """PM.cc:598-626"""
    n = len(rho)
    best = []
    for a in range(n):
        cur = [b for b in range(n) if np_chi(rho[a], rho[b], sig[a], sig[b])]
        if len(cur) > len(best):
            best = cur
    if len(best) >= lambdaN:
        r, s, _ = np_fusion_b([rho[b] for b in best], [sig[b] for b in best])
        return float(r), float(s), 1
    return 0.0, 0.0, 0


def np_pointset_pixel(K, Tcw, x, y, inv_d):
    """PM.cc:345-363 with Twc of KeyFrame.cc:70-84"""
    fx, fy, cx, cy = K
    R, t = Tcw[:, :3], Tcw[:, 3]
    Rwc = R.T.copy()
    Ow = -mv(Rwc, t)
    if f64(inv_d) < 0.000001:
        return np.zeros(3, f32)
    Z = f32(1) / f32(inv_d)
    X = Z * (f32(x) - cx) / fx
    Y = Z * (f32(y) - cy) / fy
    return np.array([((Rwc[i, 0] * X + Rwc[i, 1] * Y) + Rwc[i, 2] * Z) + Ow[i] * f32(1) for i in range(3)], f32)


# ---- tests ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def seq(pkg, oracle):
    return Sequence(pkg, oracle, 96, 72, 8, 0x5EED0C01)


def test_fast_atan2(oracle):
    rng = np.random.default_rng(0)
    pts = [(0, 1), (1, 1), (-1, 1), (1, 0), (0, -1), (1, -1), (-3, -4), (1e-20, 1), (4.0, 1.0), (-4.0, 1.0), (0, 0)]
    pts += [tuple(v) for v in rng.standard_normal((500, 2)) * 10]
    for y, x in pts:
        got = oracle.fast_atan2(y, x)
        want = np_fast_atan2(y, x)
        assert f32(got) == want or (math.isnan(got) and math.isnan(want)), (y, x, got, want)
        if x != 0 or y != 0:
            true = math.degrees(math.atan2(y, x)) % 360.0
            d = abs(got - true)
            assert min(d, 360 - d) < 0.35, (y, x, got, true)  # the documented ~0.3 deg accuracy


def test_pair_geometry_range_and_depth(oracle, seq):
    rng = np.random.default_rng(1)
    for (a, b) in [(2, 3), (2, 0), (7, 1), (4, 4)]:
        p = oracle.pair_geometry(seq.okf[a], seq.okf[b])
        R21, t21, F = np_pair(seq.Tcw[a], seq.Tcw[b], seq.K, seq.K)
        assert_bit_equal(np.array(p.R21[:]).reshape(3, 3), R21, "R21")
        assert_bit_equal(np.array(p.t21[:]), t21, "t21")
        assert_bit_equal(np.array(p.F12[:]).reshape(3, 3), F, "F12")
        for _ in range(25):
            x, y = int(rng.integers(0, seq.W)), int(rng.integers(0, seq.H))
            for (mn, mx) in [(seq.min_depth, seq.max_depth), (0.5, 0.5), (-2.0, 3.0)]:
                got = oracle.search_range(seq.okf[a], p, x, y, mn, mx)
                want = np_search_range(seq.K, R21, t21, x, y, mn, mx, seq.W)
                assert_bit_equal(np.array(got), np.array(want), "search range")
            uj = float(rng.uniform(0, seq.W))
            assert_bit_equal(np.array([oracle.pixel_depth(seq.okf[a], p, uj, x, y)]),
                             np.array([np_pixel_depth(seq.K, R21, t21, f32(uj), x, y)]), "pixel depth")


def test_fundamental_matrix_is_epipolar(oracle, seq):
    """property: a world point's projections satisfy x1^T F12 x2 ~ 0 (PM.cc:389-391 uses F12^T x1)"""
    rng = np.random.default_rng(2)
    fx, fy, cx, cy = [float(v) for v in seq.K]
    for (a, b) in [(2, 5), (0, 7)]:
        F = np.array(oracle.pair_geometry(seq.okf[a], seq.okf[b]).F12[:], np.float64).reshape(3, 3)
        for _ in range(20):
            Xw = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.8, 1.2), 1.0])
            xs = []
            for k in (a, b):
                Xc = seq.Tcw[k].astype(np.float64) @ Xw
                xs.append(np.array([fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy, 1.0]))
            line = F.T @ xs[0]  # (a, b, c) of the epipolar line in image 2
            dist = abs(line @ xs[1]) / math.hypot(line[0], line[1])
            assert dist < 0.05, dist  # pixels


def test_chi_and_fusion(oracle):
    rng = np.random.default_rng(3)
    for n in [1, 3, 4, 6, 9, 15]:
        for _ in range(20):
            rho = (1 + 0.04 * rng.standard_normal(n)).astype(f32)
            sig = (0.01 + 0.05 * rng.random(n)).astype(f32)
            got = oracle.fuse(rho, sig)
            want = np_fuse(rho, sig)
            assert got[2] == want[2]
            assert_bit_equal(np.array(got[:2]), np.array(want[:2]), "fusion")


def test_pointset(oracle, seq):
    rng = np.random.default_rng(4)
    rho = np.where(rng.random((seq.H, seq.W)) < 0.4, 0.5 + rng.random((seq.H, seq.W)), 0).astype(f32)
    rho[5, 5] = f32(5e-7)
    xyz = oracle.pointset(seq.okf[1], rho).reshape(seq.H, seq.W, 3)
    assert not xyz[:2].any() and not xyz[:, :2].any() and not xyz[-2:].any() and not xyz[:, -2:].any()
    for _ in range(200):
        x, y = int(rng.integers(2, seq.W - 2)), int(rng.integers(2, seq.H - 2))
        assert_bit_equal(xyz[y, x], np_pointset_pixel(seq.K, seq.Tcw[1], x, y, rho[y, x]), "xyz")


def test_pointset_roundtrip_property(oracle, seq):
    """back-projected points re-project onto their own pixel with the stored inverse depth"""
    rho = np.full((seq.H, seq.W), 0.9, f32)
    xyz = oracle.pointset(seq.okf[4], rho).reshape(seq.H, seq.W, 3).astype(np.float64)
    fx, fy, cx, cy = [float(v) for v in seq.K]
    T = seq.Tcw[4].astype(np.float64)
    ys, xs = np.mgrid[2:seq.H - 2, 2:seq.W - 2]
    P = xyz[2:-2, 2:-2]
    Xc = P @ T[:, :3].T + T[:, 3]
    assert np.abs(fx * Xc[..., 0] / Xc[..., 2] + cx - xs).max() < 1e-3
    assert np.abs(fy * Xc[..., 1] / Xc[..., 2] + cy - ys).max() < 1e-3
    assert np.abs(1 / Xc[..., 2] - 0.9).max() < 1e-5


def test_stereo_search_constraints(oracle):
    rng = np.random.default_rng(5)
    for n in [1, 2, 17, 1000]:
        d = (1.0 + 0.2 * rng.standard_normal(n)).astype(f32)
        got = oracle.stereo_search_constraints(d)
        acc = f64(0)
        for v in d:  # std::accumulate(..., 0.0): double accumulator, PM.cc:373
            acc = acc + f64(v)
        mean = f32(acc) / f32(n)
        acc2 = f64(0)
        for v in d:  # std::inner_product(..., 0.0) over float diffs, PM.cc:378
            diff = f32(v) - mean
            acc2 = acc2 + f64(diff * diff)
        std = np.sqrt(f32(acc2 / f64(n)))
        with np.errstate(all="ignore"):
            want = (f32(1) / (mean - f32(2) * std), f32(1) / (mean + f32(2) * std))
        assert_bit_equal(np.array(got), np.array(want), "StereoSearchConstraints")
