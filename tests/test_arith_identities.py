"""CPU checks of arithmetic identities the HIP kernels rely on (host logic; no GPU, no oracle)."""
import numpy as np


def test_double_1em6_compare_is_a_float_compare():
    """sdm_device.h gt_1em6/lt_1em6: (double)x > 1e-6  <=>  x > 0x358637bd,  (double)x < 1e-6  <=>  x <= 0x358637bd"""
    f0 = np.uint32(0x358637BD).view(np.float32)
    assert float(f0) < 1e-6 < float(np.nextafter(f0, np.float32(1)))
    bits = np.arange(0x358637BD - 4096, 0x358637BD + 4096, dtype=np.uint32)
    xs = np.concatenate([bits.view(np.float32), -bits.view(np.float32),
                         np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 1.0], dtype=np.float32)])
    wide = xs.astype(np.float64)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(wide > 1e-6, xs > f0)
        assert np.array_equal(wide < 1e-6, xs <= f0)


def test_reciprocal_sign_test_is_a_range_test():
    """k_search_fuse: (1.0f/rho) > 0  <=>  bits(rho) < bits(+Inf), i.e. rho in [+0, +Inf)  (denormals on), PM.cc:216"""
    xs = np.array([0.0, -0.0, 1e-45, 1e-38, 1.0, 3.4028235e38, np.inf, -np.inf, np.nan, -1.0, -1e-45], dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        want = (np.float32(1.0) / xs) > 0
        got = xs.view(np.uint32) < np.uint32(0x7F800000)
    assert np.array_equal(want, got)


def _fma32(a, b, c):
    """float32 fma through float64 (the product of two float32 is exact in float64; the sum's second rounding moves the
    result by far less than the slack of the bounds checked here)"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_scan_argmin_may_compare_approximate_costs():
    """sdm_device.h match_cost_approx (K1, SDM_K1_OPT bit 13): fma(ge2, (float)(1/0.23), pe2) stays within 4 float steps of
    PM.cc:436's err = (float)((double)pe2 + (double)ge2 / 0.23), so two costs more than COST_BAND = 16 steps apart order like
    the exact ones; the initial state (pe = 1000, ge = 0) evaluates to old_err = 1e6 exactly (PM.cc:396)"""
    rng = np.random.default_rng(0x5EED13)
    n = 400000
    pe = (rng.random(n) * 255).astype(np.float32)
    ge = (rng.random(n) * 60).astype(np.float32)
    sc = np.exp2(rng.integers(-60, 40, n)).astype(np.float32)  # magnitudes far beyond image data as well
    pe2 = np.concatenate([pe * pe, pe * pe * sc, np.zeros(8, np.float32)])
    ge2 = np.concatenate([ge * ge, ge * ge * sc, np.array([0, 1e-30, 1e-38, 1e-44, 1.0, 4e6, 1e30, 0.23], np.float32)])
    ref = (pe2.astype(np.float64) + ge2.astype(np.float64) / 0.23).astype(np.float32)
    approx = _fma32(ge2, np.full_like(ge2, np.float32(1.0 / 0.23)), pe2)
    fin = np.isfinite(ref)
    steps = np.abs(approx.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
    assert steps[fin].max() <= 4
    one = np.array([1000.0], np.float32), np.array([0.0], np.float32)
    assert (one[0] * one[0]).astype(np.float64) + (one[1] * one[1]).astype(np.float64) / 0.23 == 1000000.0
    # ordering: wherever the approximations are more than 16 steps apart, the exact costs compare the same way (strictly)
    i, j = rng.integers(0, len(ref), 2 * 10 ** 6).reshape(2, -1)
    far = np.abs(approx.view(np.int32).astype(np.int64)[i] - approx.view(np.int32).astype(np.int64)[j]) > 16
    ok = fin[i] & fin[j] & far
    assert np.array_equal((approx[i] < approx[j])[ok], (ref[i] < ref[j])[ok])


def test_line_coefficients_keep_the_grid_of_F12():
    """sdm_device.h line_quot_safe (K1, SDM_K1_OPT bit 15): with every entry of F12 either +0 or of magnitude in
    [2^-36, 2^20] and integer pixel coordinates below 2^16, a = (x*F0 + y*F3) + F6 in float is never -0, is zero or at least
    2^-59 in magnitude and at most 3 * 2^36 -- the range in which the reciprocal-form quotients need no per-lane guard"""
    rng = np.random.default_rng(0x5EED15)
    n = 300000
    ex = rng.integers(-36, 20, (3, n))
    F = (np.exp2(ex) * (1 + rng.random((3, n))) * rng.choice([-1.0, 1.0], (3, n))).astype(np.float32)
    F = np.minimum(np.abs(F), np.float32(2.0 ** 20)) * np.sign(F)
    F[rng.random((3, n)) < 0.15] = 0.0  # +0 entries
    x = rng.integers(0, 65536, n).astype(np.float32)
    y = rng.integers(0, 65536, n).astype(np.float32)
    # cancellation on purpose: F6 = -(x*F0 + y*F3) for a part of the cases
    p = x * F[0] + y * F[1]
    hit = (rng.random(n) < 0.2) & (np.abs(p) >= 2.0 ** -36) & (np.abs(p) <= 2.0 ** 20)
    F[2, hit] = -p[hit]
    a = (x * F[0] + y * F[1]) + F[2]
    assert not np.any((a == 0) & np.signbit(a))
    nz = a != 0
    assert np.abs(a[nz]).min() >= 2.0 ** -59 and np.abs(a).max() <= 3 * 2.0 ** 36
    assert np.all(np.mod(np.abs(a[nz]).astype(np.float64) * 2.0 ** 59, 1.0) == 0)  # multiples of 2^-59


def test_inter_check_approximate_projection_bound():
    """sdm_kernels.h k4_proj_bounds / inter_project_approx (K4): the approximate chain t_i = fma(n_i, 1/rho, T_i),
    xj = u * (1/t2) lies within eps = H + G |xj| of the reference chain (PM.cc:677-680: three divisions by rho, two by z),
    H = (pb0/rho + pb1)|r|, G = pb2 H + 3 * 2^-23 with the per-pair constants -- so a value further than eps from every
    integer names the reference chain's cell and validity (the device statement is sdm_selftest(9))"""
    rng = np.random.default_rng(0x5EED17)
    n = 500000
    f32 = np.float32

    def u(lo, hi, *shape):
        return (lo + (hi - lo) * rng.random(shape)).astype(f32)

    fx, fy, cx, cy = f32(517.3), f32(516.5), f32(318.6), f32(255.3)
    W, H = 640, 480
    X0 = f32(max(abs(cx), abs(W - 1 - cx)) / fx * (1 + 2.0 ** -20))
    X1 = f32(max(abs(cy), abs(H - 1 - cy)) / fy * (1 + 2.0 ** -20))
    R = np.eye(3, dtype=f32)[:, :, None] + u(-0.05, 0.05, 3, 3, n)
    t = u(-0.08, 0.08, 3, n)
    xp0 = ((rng.integers(0, W, n).astype(f32) - cx) / fx).astype(f32)
    xp1 = ((rng.integers(0, H, n).astype(f32) - cy) / fy).astype(f32)
    rho = np.exp2(u(-1.5, 1.5, n)).astype(f32)
    d, s = f32(2.0 ** -23), f32(1 + 2.0 ** -18)
    N = [(np.abs(R[i, 0]) * X0 + np.abs(R[i, 1]) * X1 + np.abs(R[i, 2])) * s for i in range(3)]
    C1 = np.maximum(fx * N[0] + cx * N[2], fy * N[1] + cy * N[2]) * s
    C2 = np.maximum(fx * np.abs(t[0]) + cx * np.abs(t[2]), fy * np.abs(t[1]) + cy * np.abs(t[2])) * s
    pb0, pb1, pb2 = f32(8) * d * C1 * s, f32(4) * d * C2 * s, f32(0.3125) / min(cx, cy) * s
    n_ = [(R[i, 0] * xp0 + R[i, 1] * xp1) + R[i, 2] * f32(1) for i in range(3)]
    # reference chain (float32 throughout, PM.cc:677-680)
    te = [(n_[i] / rho).astype(f32) + t[i] for i in range(3)]
    xe = ((fx * te[0] + cx * te[2]) / te[2]).astype(f32)
    ye = ((fy * te[1] + cy * te[2]) / te[2]).astype(f32)
    # approximate chain
    dp = (f32(1) / rho).astype(f32)
    ta = [_fma32(n_[i], dp, t[i]) for i in range(3)]
    r = (1.0 / ta[2].astype(np.float64)).astype(f32)  # v_rcp_f32: within 1 ulp; take the rounded value and its neighbours
    for r_ in (r, np.nextafter(r, f32(np.inf)), np.nextafter(r, f32(-np.inf))):
        xa, ya = (fx * ta[0] + cx * ta[2]) * r_, (fy * ta[1] + cy * ta[2]) * r_
        Hh = _fma32(pb0, dp, pb1) * np.abs(r_)
        Gg = _fma32(Hh, np.full_like(Hh, pb2), np.full_like(Hh, f32(3) * d))
        ex, ey = _fma32(Gg, np.abs(xa), Hh), _fma32(Gg, np.abs(ya), Hh)
        assert np.all(np.abs(xa.astype(np.float64) - xe) <= ex) and np.all(np.abs(ya.astype(np.float64) - ye) <= ey)
        clear = (np.abs(xa - np.rint(xa)) > ex) & (np.abs(ya - np.rint(ya)) > ey)
        assert clear.mean() > 0.99  # the exact chain is the exception
        assert np.array_equal(np.floor(xa)[clear], np.floor(xe)[clear]) and np.array_equal(np.floor(ya)[clear], np.floor(ye)[clear])
        # and the measured error uses only a part of the bound (the spare the derivation keeps)
        assert (np.abs(xa.astype(np.float64) - xe) / ex).max() < 0.8
