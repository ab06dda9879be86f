"""A SECOND, independent restatement of the ProbabilityMapping loops in NumPy -- test infrastructure.

Written from the text of /root/reference/src/Modeler/ProbabilityMapping.cc ("PM.cc"), NOT from
oracle/pm_oracle.c: the scan loop + sub-pixel refinement (PM.cc:385-465), the hypothesis (806-829,
845-875), the search range (877-910), the fusion (598-626, 912-970), the intra-keyframe check
(486-547) and the inter-keyframe check with its Gauss-Newton step (628-799).  Vectorised over the
pixels of one keyframe; every operation is an elementwise IEEE float32/float64 NumPy operation in the
order and precision the C++ promotion rules give, so it can be compared with the
C oracle bit for bit (tests/test_oracle_second_restatement.py).  The reference holds no vectors
(parity unpinned); two restatements by different routes agreeing bit for bit is what bounds
transcription error.

Two arithmetic modes for the pieces that live inside OpenCV in the reference:

  mode="n1"  the build's normative choice N1/N2: 3x3 / 3x1 products in float,
             accumulated left to right; `A*B/s`, `A*B*s + C` as separate float operations.
  mode="cv"  OpenCV-3.x cv::Mat expression semantics, RESTATED FROM MEMORY (OpenCV is absent from the
             image, so this cannot be checked against the library):
             * a product with a transposed operand (`Rcw2*Rcw1.t()`, PM.cc:859) and any product whose
               result is 1x1 (`R21.row(2)*xp`, `J.t()*r0`) take cv::gemm's general path: products and
               sums in double, scaled by alpha in double, rounded to float once;
             * a plain 3x3*3x3 or 3x3*3x1 product takes cv::gemm's small-matrix path: the dot product
               is accumulated in FLOAT left to right, then `(float)(t*alpha + c*beta)` is evaluated
               in double -- so `R21*xp*mind + t21` (PM.cc:894) and `Rji*xp/depthp + tji` (PM.cc:678)
               fold their scale (mind, resp. the DOUBLE reciprocal 1.0/depthp) and the addition into
               one rounding;
             * `Xj/Xj(2)` (PM.cc:680) is a multiplication by (float)(1.0/(double)Xj(2));
             * K.inv() of a 3x3 float matrix is the adjugate evaluated in double, rounded once.
The "cv" mode exists to MEASURE how much the un-pinnable OpenCV rounding could move results
(tests/test_oracle_second_restatement.py reports mask flips and ulp distances); "n1" is what the
oracle and the engine implement.
"""
import math

import numpy as np

f32, f64 = np.float32, np.float64
THETA = 0.23          # PM.h:47, a double literal
LAMBDA_G, LAMBDA_L, LAMBDA_THETA, LAMBDA_N = f32(8), f32(80), f32(45), 3  # PM.h:40-43


class KF:
    """the KeyFrame members PM.cc reads"""

    def __init__(self, im, grad, theta, istd, K, Tcw):
        self.im = np.ascontiguousarray(im, np.uint8)
        self.grad = np.ascontiguousarray(grad, f32)
        self.theta = np.ascontiguousarray(theta, f32)
        self.istd = f32(istd)
        self.fx, self.fy, self.cx, self.cy = [f32(v) for v in K]
        T = np.asarray(Tcw, f32).reshape(3, 4)
        self.R = T[:, :3].copy()
        self.t = T[:, 3].copy()
        self.H, self.W = self.im.shape


# ---- small matrix algebra in the two modes -------------------------------------------------------------
def _dot_f32(r, v):
    """cv::gemm small-matrix path / N1: float products, float sums, left to right"""
    s = r[0] * v[0]
    for k in range(1, len(r)):
        s = s + r[k] * v[k]
    return s


def _dot_f64(r, v):
    """cv::gemm general path: double products (exact for float inputs) and double sums; NOT yet rounded"""
    s = f64(0)
    for k in range(len(r)):
        s = s + f64(r[k]) * f64(v[k])
    return s


def _matmul(A, B, mode, transposed_operand=False):
    """C = A*B for small float matrices.  In "cv" mode a product with a transposed operand goes
    through double accumulation (B is passed already transposed; only the flag matters)."""
    A, B = np.asarray(A, f32), np.asarray(B, f32)
    C = np.zeros((A.shape[0], B.shape[1]), f32)
    for i in range(A.shape[0]):
        for k in range(B.shape[1]):
            if mode == "cv" and transposed_operand:
                C[i, k] = f32(_dot_f64(A[i], B[:, k]))
            else:
                C[i, k] = _dot_f32(A[i], B[:, k])
    return C


def _kinv(fx, fy, cx, cy, mode, transpose=False):
    """K^-1 (PM.cc:986) or (K^T)^-1"""
    one, z = f32(1), f32(0)
    if mode == "cv":  # cv::invert, 3x3 float: adjugate / determinant in double, rounded once
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], f64)
        if transpose:
            K = K.T.copy()
        S = K
        det = S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0]) + \
            S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0])
        d = 1.0 / det
        t = np.empty((3, 3), f64)
        t[0, 0] = (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d
        t[0, 1] = (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d
        t[0, 2] = (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d
        t[1, 0] = (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d
        t[1, 1] = (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d
        t[1, 2] = (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d
        t[2, 0] = (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d
        t[2, 1] = (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d
        t[2, 2] = (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d
        return t.astype(f32)
    Ki = np.array([[one / fx, z, -cx / fx], [z, one / fy, -cy / fy], [z, z, one]], f32)  # N2: closed form
    return Ki.T.copy() if transpose else Ki


class Pair:
    """per (kf1, kf2): R21, t21 (PM.cc:859-860 == 890-891 == 643-644) and F12 (PM.cc:972-986)"""

    def __init__(self, kf1, kf2, mode="n1"):
        self.mode = mode
        self.R21, self.t21 = self._rel(kf2, kf1, mode)   # R21 = Rcw2*Rcw1.t(); t21 = -Rcw2*Rcw1.t()*tcw1 + tcw2
        R12, t12 = self._rel(kf1, kf2, mode)             # R12 = R1w*R2w.t();   t12 = -R1w*R2w.t()*t2w + t1w
        z = f32(0)
        t12x = np.array([[z, -t12[2], t12[1]], [t12[2], z, -t12[0]], [-t12[1], t12[0], z]], f32)  # LocalMapping.cc:711-716
        K1ti = _kinv(kf1.fx, kf1.fy, kf1.cx, kf1.cy, mode, transpose=True)
        K2i = _kinv(kf2.fx, kf2.fy, kf2.cx, kf2.cy, mode)
        self.F12 = _matmul(_matmul(_matmul(K1ti, t12x, mode), R12, mode), K2i, mode)  # left to right, PM.cc:986

    @staticmethod
    def _rel(kfa, kfb, mode):
        """Rab = Ra*Rb.t();  tab = -Ra*Rb.t()*tb + ta"""
        Rab = _matmul(kfa.R, kfb.R.T.copy(), mode, transposed_operand=True)
        neg = -Rab  # (-Ra)*Rb.t(): alpha = -1 folded into the product; the negation of a float is exact
        tab = np.zeros(3, f32)
        for i in range(3):
            d = _dot_f32(neg[i], kfb.t)  # (3x3)*(3x1), no transposed operand: float accumulation in both modes
            if mode == "cv":
                tab[i] = f32(f64(d) * 1.0 + f64(kfa.t[i]) * 1.0)  # (float)(t*alpha + c*beta), double
            else:
                tab[i] = d + kfa.t[i]
        return Rab, tab


# ---- cv::fastAtan2 (PM.cc:414), OpenCV 3.x polynomial, vectorised for x = 1 -----------
def fast_atan2_x1(y):
    y = np.asarray(y, f32)
    scale = f32(180.0 / math.pi)
    p1 = f32(0.9997878412794807) * scale
    p3 = f32(-0.3258083974640975) * scale
    p5 = f32(0.1555786518463281) * scale
    p7 = f32(-0.04432655554792128) * scale
    eps = f32(np.finfo(f64).eps)
    ax, ay = f32(1), np.abs(y)
    with np.errstate(all="ignore"):
        c_lo = ay / (ax + eps)
        c_hi = ax / (ay + eps)
        first = ax >= ay
        c = np.where(first, c_lo, c_hi)
        c2 = c * c
        poly = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
        a = np.where(first, poly, f32(90) - poly)
        a = np.where(y < 0, f32(360) - a, a)  # x = 1 > 0: no 180-a step
    return a.astype(f32)


# ---- bilinear<T>, PM.cc:40-59, all four taps as written ---------------------------------------------------
def bilinear(img, y, x_int):
    """y float32 array, x_int integer array (every call site passes an integer column, PM.cc:433-434,452-453).
    The zero-weight column x1 is clamped into the image (the reference reads one past the row end there)."""
    H, W = img.shape
    x = x_int.astype(f32)
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0w = x1.astype(f32) - x
    y0w = y1.astype(f32) - y
    x1w = f32(1) - x0w
    y1w = f32(1) - y0w
    x1c = np.minimum(x1, W - 1)
    v = lambda yy, xx: img[yy, xx].astype(f32)
    return ((v(y0, x0) * x0w * y0w + v(y0, x1c) * x1w * y0w) + v(y1, x0) * x0w * y1w) + v(y1, x1c) * x1w * y1w


def _wrap_diff(d):
    """PM.cc:416-418 == 428-430: if (d >= 360) d -= 360; if (d < 0) d += 360; if (d > 180) d = 360 - d"""
    d = np.where(d >= 360, d - f32(360), d)
    d = np.where(d < 0, d + f32(360), d)
    return np.where(d > 180, f32(360) - d, d)


def _ray(kf, xs, ys):
    """xp = ((px-cx)/fx, (py-cy)/fy, 1), PM.cc:862 == 893 == 677"""
    return (xs.astype(f32) - kf.cx) / kf.fx, (ys.astype(f32) - kf.cy) / kf.fy


def _rows_dot_xp(pair, xp0, xp1):
    """R21.row(i)*xp for i = 0..2: float-accumulated (the 3x3*3x1 product) and double-accumulated (the 1x1 products)"""
    one = f32(1)
    R = pair.R21
    f = [(R[i, 0] * xp0 + R[i, 1] * xp1) + R[i, 2] * one for i in range(3)]
    d = [(f64(R[i, 0]) * f64(xp0) + f64(R[i, 1]) * f64(xp1)) + f64(R[i, 2]) * f64(one) for i in range(3)]
    return f, d


def search_range(kf1, pair, xs, ys, mind, maxd):
    """GetSearchRange, PM.cc:877-910 (fed INVERSE depths)"""
    xp0, xp1 = _ray(kf1, xs, ys)
    rf, _ = _rows_dot_xp(pair, xp0, xp1)
    t = pair.t21
    mind, maxd = f32(mind), f32(maxd)
    with np.errstate(all="ignore"):
        if pair.mode == "cv":  # gemm(R21, xp1, alpha = mind, t21, beta = 1): (float)(t*alpha + c*beta) in double
            x_min = (f64(rf[0]) * f64(mind) + f64(t[0])).astype(f32)
            z_min = (f64(rf[2]) * f64(mind) + f64(t[2])).astype(f32)
            x_max = (f64(rf[0]) * f64(maxd) + f64(t[0])).astype(f32)
            z_max = (f64(rf[2]) * f64(maxd) + f64(t[2])).astype(f32)
        else:
            x_min, z_min = rf[0] * mind + t[0], rf[2] * mind + t[2]
            x_max, z_max = rf[0] * maxd + t[0], rf[2] * maxd + t[2]
        umin = kf1.fx * x_min / z_min + kf1.cx
        umax = kf1.fx * x_max / z_max + kf1.cx
    swap = umin > umax
    umin, umax = np.where(swap, umax, umin), np.where(swap, umin, umax)
    cols = f32(kf1.W)
    umin = np.where(umin < 0, f32(0), umin)
    umax = np.where(umax < 0, f32(0), umax)
    umin = np.where(umin > cols, cols, umin)
    umax = np.where(umax > cols, cols, umax)
    return umin.astype(f32), umax.astype(f32)


def pixel_depth(kf1, pair, uj, xs, ys):
    """GetPixelDepth, PM.cc:845-875 (Eq. 8)"""
    ucx = uj - kf1.cx
    xp0, xp1 = _ray(kf1, xs, ys)
    rf, rd = _rows_dot_xp(pair, xp0, xp1)
    t = pair.t21
    with np.errstate(all="ignore"):
        if pair.mode == "cv":  # 1x1 products: double accumulation, alpha = ucx resp. fx folded in, one rounding
            num1 = (rd[2] * f64(ucx)).astype(f32)
            num2 = (rd[0] * f64(kf1.fx)).astype(f32)
        else:
            num1 = rf[2] * ucx
            num2 = kf1.fx * rf[0]
        den1 = -t[2] * ucx
        den2 = kf1.fx * t[0]
        return ((num1 - num2) / (den1 + den2)).astype(f32)


def epipolar_search(kf1, kf2, pair, xs, ys, min_depth, max_depth, rot=0.0):
    """EpipolarSearch PM.cc:385-465 + ComputeInvDepthHypothesis PM.cc:806-829 for the pixels (xs, ys) of kf1.
    Returns rho, sigma, supported, candidates (scan iterations per pixel).  Normative choices N3-N5
    where the reference has undefined behaviour."""
    n = len(xs)
    W2, H2 = kf2.W, kf2.H
    F = pair.F12
    x, y = xs.astype(f32), ys.astype(f32)
    with np.errstate(all="ignore"):
        a = x * F[0, 0] + y * F[1, 0] + F[2, 0]
        b = x * F[0, 1] + y * F[1, 1] + F[2, 1]
        c = x * F[0, 2] + y * F[1, 2] + F[2, 2]
        ab = a / b
        cb = c / b
    alive = ~((ab < -4) | (ab > 4)) & ~np.isnan(ab)  # PM.cc:393; N5: a NaN line gives no hypothesis
    pixel = kf1.im[ys, xs].astype(f32)               # PM.cc:202
    grad1 = kf1.grad[ys, xs]
    th_pi = kf1.theta[ys, xs]
    umin, umax = search_range(kf1, pair, xs, ys, min_depth, max_depth)
    alive &= ~np.isnan(umin) & ~np.isnan(umax)       # N5
    lo = np.where(alive, np.ceil(np.where(alive, umin, 0)), 0).astype(np.int64)
    hi = np.where(alive, np.floor(np.where(alive, umax, -1)), -1).astype(np.int64)
    hi = np.minimum(hi, W2 - 1)                      # N3: the inclusive clamp at PM.cc:908 would read column W

    with np.errstate(all="ignore"):
        th_line = fast_atan2_x1(-a / b)              # PM.cc:414, loop invariant
        apr = th_pi + f32(rot)                       # PM.cc:424-426
        apr = np.where(apr >= 360, apr - f32(360), apr)
        apr = np.where(apr < 0, apr + f32(360), apr)

    old_err = np.full(n, f32(1000000.0), f32)
    best_pe = np.zeros(n, f32)
    best_ge = np.zeros(n, f32)
    best_px = np.zeros(n, np.int64)
    cand = np.zeros(n, np.int64)
    span = int((hi - lo)[alive].max()) + 1 if alive.any() else 0
    for t in range(max(span, 0)):
        uj = lo + t
        act = alive & (uj <= hi)
        if not act.any():
            break
        cand += act
        ujc = np.where(act, uj, 0)
        with np.errstate(all="ignore"):
            inner = ab * ujc.astype(f32) + cb         # (a/b)*uj + (c/b)
        ok = act & np.isfinite(inner)
        tr = np.trunc(np.where(ok, inner, 0)).astype(np.int64)
        vj = -tr                                      # vj = -(int)(...), PM.cc:407
        ok &= (vj >= 1) & (vj <= H2 - 2)              # PM.cc:408 (vj<=0 || vj>=rows) tightened by N3 (row vj+1 is read)
        vjc = np.where(ok, vj, 1)
        ok &= ~(kf2.grad[vjc, ujc] < LAMBDA_G)        # condition 1, PM.cc:411
        th2 = kf2.theta[vjc, ujc]
        d = _wrap_diff(th2 - th_line)                 # condition 2, PM.cc:415-421
        d = np.where(d > 90, f32(180) - d, d)
        ok &= ~(d > LAMBDA_L)
        ok &= ~(_wrap_diff(th2 - apr) > LAMBDA_THETA)  # condition 3, PM.cc:427-431
        if not ok.any():
            continue
        yf = np.where(ok, -inner, f32(1))
        pe = pixel - bilinear(kf2.im, yf, ujc)        # PM.cc:433
        ge = grad1 - bilinear(kf2.grad, yf, ujc)      # PM.cc:434
        err = (f64(pe * pe) + f64(ge * ge) / THETA).astype(f32)  # PM.cc:436: float + double -> double -> float err
        better = ok & (err < old_err)                 # strict: lowest uj wins ties
        best_px = np.where(better, uj, best_px)
        old_err = np.where(better, err, old_err)
        best_pe = np.where(better, pe, best_pe)
        best_ge = np.where(better, ge, best_ge)

    rho = np.zeros(n, f32)
    sigma = np.zeros(n, f32)
    found = old_err < f32(1000000.0)                  # PM.cc:446
    up, um = best_px + 1, best_px - 1                 # PM.cc:449-450
    found &= (um >= 0) & (up <= W2 - 1)               # N4
    upc, umc = np.where(found, up, 1), np.where(found, um, 0)
    with np.errstate(all="ignore"):
        yfp = -(ab * upc.astype(f32) + cb)
        yfm = -(ab * umc.astype(f32) + cb)
    fyp, fym = np.floor(yfp), np.floor(yfm)
    found &= (fyp >= 0) & (fyp <= H2 - 2) & (fym >= 0) & (fym <= H2 - 2)  # N4
    yfp, yfm = np.where(found, yfp, f32(0)), np.where(found, yfm, f32(0))
    upc, umc = np.where(found, up, 1), np.where(found, um, 0)
    with np.errstate(all="ignore"):
        g = (bilinear(kf2.im, yfp, upc) - bilinear(kf2.im, yfm, umc)) / f32(2)     # PM.cc:452
        q = (bilinear(kf2.grad, yfp, upc) - bilinear(kf2.grad, yfm, umc)) / f32(2)  # PM.cc:453
        inv_t = 1 / THETA
        den = (f64(g * g) + inv_t * f64(q) * f64(q)).astype(f32)                    # PM.cc:455
        ustar = (best_px.astype(f64) + (f64(g * best_pe) + inv_t * f64(q) * f64(best_ge)) / f64(den)).astype(f32)
        ustar_var = f32(2) * kf2.istd * kf2.istd / den                               # PM.cc:457
        # ComputeInvDepthHypothesis, PM.cc:806-829
        d0 = pixel_depth(kf1, pair, ustar, xs, ys)
        s = np.sqrt(ustar_var)
        dmin = pixel_depth(kf1, pair, ustar - s, xs, ys)
        dmax = pixel_depth(kf1, pair, ustar + s, xs, ys)
        e1, e2 = np.abs(dmax - d0), np.abs(dmin - d0)
        sg = np.where(e1 < e2, e2, e1)                # cv::max(a, b) = (a < b) ? b : a
    rho = np.where(found, d0, f32(0)).astype(f32)
    sigma = np.where(found, sg, f32(0)).astype(f32)
    return rho, sigma, found, cand


# ---- fusion -------------------------------------------------------------------------------------------------
def chi_matrix(ra, rb, sa, sb):
    """ChiTest PM.cc:912-924 elementwise: float arithmetic, compared with the double literal 5.99"""
    with np.errstate(all="ignore"):
        num = (ra - rb) * (ra - rb)
        chi = num / (sa * sa) + num / (sb * sb)
    return chi.astype(f64) < 5.99


def fusion_b(rho, sig, member):
    """GetFusion overload B (PM.cc:947-970) over the members of each row's set, in hypothesis order.
    rho, sig, member: [P, N].  Returns fused rho, sqrt(1/rsj), min sigma (by sigma^2 compare, first wins)."""
    P, N = rho.shape
    pjsj = np.zeros(P, f32)
    rsj = np.zeros(P, f32)
    first = np.argmax(member, axis=1)  # compatible_ho[0]
    tmin = sig[np.arange(P), first]
    with np.errstate(all="ignore"):
        for j in range(N):
            m = member[:, j]
            s2 = f64(sig[:, j]) * f64(sig[:, j])  # pow(sigma, 2) -> double
            pjsj = np.where(m, (f64(pjsj) + f64(rho[:, j]) / s2).astype(f32), pjsj)
            rsj = np.where(m, (f64(rsj) + 1.0 / s2).astype(f32), rsj)
            tmin = np.where(m & (s2 < f64(tmin) * f64(tmin)), sig[:, j], tmin)
        return (pjsj / rsj).astype(f32), np.sqrt(f32(1) / rsj).astype(f32), tmin.astype(f32)


def hypothesis_fusion(rho, sig, valid):
    """InverseDepthHypothesisFusion PM.cc:598-626 for every row of [P, N] hypothesis arrays (valid = the
    hypotheses PM.cc:216 accepted; the reference's vector holds only those, in neighbour order)."""
    P, N = rho.shape
    comp = chi_matrix(rho[:, :, None], rho[:, None, :], sig[:, :, None], sig[:, None, :])  # [P, a, b]
    comp &= valid[:, :, None] & valid[:, None, :]
    size = comp.sum(axis=2)                       # |S_a|
    size = np.where(valid, size, -1)
    besta = np.argmax(size, axis=1)               # strict '>': the first largest set wins
    best = size[np.arange(P), besta]
    member = comp[np.arange(P), besta, :]
    nh = valid.sum(axis=1)
    fuse = (nh > LAMBDA_N) & (best >= LAMBDA_N)   # PM.cc:221 and :623
    member = member & fuse[:, None]
    safe = np.where(member.any(axis=1)[:, None], member, np.eye(1, N, dtype=bool))  # rows that do not fuse: any member will do
    r, s, _ = fusion_b(rho, sig, safe)
    return np.where(fuse, r, f32(0)).astype(f32), np.where(fuse, s, f32(0)).astype(f32), fuse


def recon_search_fuse(kf1, nbrs, pairs, min_depth, max_depth, rots=None):
    """hot loop 1, PM.cc:197-231, for one reference keyframe: returns rho, sigma [H, W] and the scan statistics"""
    H, W = kf1.H, kf1.W
    ys, xs = np.nonzero(~(kf1.grad[2:H - 2, 2:W - 2] < LAMBDA_G))  # PM.cc:198-201
    ys, xs = ys + 2, xs + 2
    N = len(nbrs)
    P = len(xs)
    R = np.zeros((P, N), f32)
    S = np.ones((P, N), f32)
    V = np.zeros((P, N), bool)
    cands = 0
    for j, (kf2, pr) in enumerate(zip(nbrs, pairs)):
        r, s, sup, cand = epipolar_search(kf1, kf2, pr, xs, ys, min_depth, max_depth, 0.0 if rots is None else rots[j])
        with np.errstate(all="ignore"):
            ok = sup & ((f32(1) / r).astype(f64) > 0.0)  # PM.cc:216: dh.supported && 1/dh.depth > 0.0
        R[:, j] = np.where(ok, r, f32(0))
        S[:, j] = np.where(ok, s, f32(1))
        V[:, j] = ok
        cands += int(cand.sum())
    rho = np.zeros((H, W), f32)
    sigma = np.zeros((H, W), f32)
    if P:
        r, s, fused = hypothesis_fusion(R, S, V)
        rho[ys[fused], xs[fused]] = r[fused]
        sigma[ys[fused], xs[fused]] = s[fused]
    return rho, sigma, dict(searches=P * N, candidates=cands, hypotheses=int(V.sum()))


# ---- IntraKeyFrameDepthChecking, PM.cc:486-547 ------------------------------------------------------------------
def intra_check(rho, sigma):
    H, W = rho.shape
    out_r, out_s = rho.copy(), sigma.copy()        # the clones of PM.cc:488-489
    ys, xs = np.nonzero(rho[2:H - 2, 2:W - 2].astype(f64) > 0.000001)
    ys, xs = ys + 2, xs + 2
    P = len(xs)
    if not P:
        return out_r, out_s
    dp, sp = rho[ys, xs], sigma[ys, xs]
    R = np.zeros((P, 9), f32)
    S = np.ones((P, 9), f32)
    M = np.zeros((P, 9), bool)
    k = 0
    for dy in (-1, 0, 1):                          # raster order y, x (PM.cc:504-507)
        for dx in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            dn, sn = rho[ys + dy, xs + dx], sigma[ys + dy, xs + dx]
            ok = (dn.astype(f64) > 0.000001) & chi_matrix(dn, dp, sn, sp)  # PM.cc:510-512
            R[:, k], S[:, k], M[:, k] = dn, sn, ok
            k += 1
    R[:, 8], S[:, 8], M[:, 8] = dp, sp, True       # itself, last (PM.cc:522)
    enough = M.sum(axis=1) >= 3                    # PM.cc:524
    r, _, tmin = fusion_b(R, S, M)
    out_r[ys, xs] = np.where(enough, r, f32(0))    # PM.cc:530-531, 535-536: sigma := MIN sigma, not the fused one
    out_s[ys, xs] = np.where(enough, tmin, f32(0))
    return out_r, out_s


# ---- InterKeyFrameDepthChecking, PM.cc:628-799 -----------------------------------------------------------------------
def inter_check(cur, cur_rho, nbrs, pairs, nbr_rho, nbr_sigma):
    """returns the new depth map of `cur` (the reference writes it in place; each pixel reads only itself)"""
    H, W = cur.H, cur.W
    out = cur_rho.copy()
    sel = ~(cur_rho[2:H - 2, 2:W - 2].astype(f64) < 0.000001)   # PM.cc:662
    ys, xs = np.nonzero(sel)
    ys, xs = ys + 2, xs + 2
    P = len(xs)
    if not P:
        return out
    depthp = cur_rho[ys, xs]
    xp0, xp1 = _ray(cur, xs, ys)
    mode = pairs[0].mode if pairs else "n1"
    count = np.zeros(P, np.int64)
    sum_Jr32, sum_JJ32 = np.zeros(P, f32), np.zeros(P, f32)   # N1: float accumulation in (j, n) order
    sum_Jr64, sum_JJ64 = np.zeros(P, f64), np.zeros(P, f64)   # cv: J.t()*r0 is a 1x1 product -> double accumulation
    with np.errstate(all="ignore"):
        dp = f32(1) / depthp                                    # PM.cc:769
        for kj, pr, rj, sj in zip(nbrs, pairs, nbr_rho, nbr_sigma):
            rf, rd = _rows_dot_xp(pr, xp0, xp1)
            t = pr.t21
            if mode == "cv":
                inv = 1.0 / f64(depthp)                          # `A*B/s`: alpha = 1./s, a DOUBLE reciprocal
                tmp = [(f64(rf[i]) * inv + f64(t[i])).astype(f32) for i in range(3)]
                rzxp = rd[2].astype(f32)                         # Rji.row(2)*xp: 1x1 -> double accumulation
            else:
                tmp = [rf[i] / depthp + t[i] for i in range(3)]  # PM.cc:678
                rzxp = rf[2]
            # Xj = K*temp (PM.cc:679): 3x3*3x1, float accumulation; the zero entries of K contribute exact zeros
            u = kj.fx * tmp[0] + kj.cx * tmp[2]
            v = kj.fy * tmp[1] + kj.cy * tmp[2]
            if mode == "cv":
                sc = (1.0 / f64(tmp[2])).astype(f32)             # Xj/Xj(2): convertTo with (float)(1./s)
                xj, yj = u * sc, v * sc
            else:
                xj, yj = u / tmp[2], v / tmp[2]                  # PM.cc:680
            depthj = depthp / (rzxp + depthp * t[2])             # PM.cc:684-688
            inb = ~((xj < 0) | (xj >= W - 1) | (yj < 0) | (yj >= H - 1)) & ~np.isnan(xj) & ~np.isnan(yj)  # PM.cc:695, N8
            x0 = np.where(inb, np.floor(np.where(inb, xj, 0)), 0).astype(np.int64)
            y0 = np.where(inb, np.floor(np.where(inb, yj, 0)), 0).astype(np.int64)
            nj = np.zeros(P, np.int64)
            for (yy, xx) in ((y0, x0), (y0 + 1, x0), (y0, x0 + 1), (y0 + 1, x0 + 1)):  # PM.cc:705,717,729,741
                d = rj[yy, xx]
                sg = sj[yy, xx]
                dd = depthj - d
                test = ((f64(dd) * f64(dd)) / (f64(sg) * f64(sg))).astype(f32)  # pow(.,2)/pow(.,2) in double -> float test
                ok = inb & (d.astype(f64) > 0.000001) & (test.astype(f64) < 3.84)
                nj += ok
                djn = f32(1) / d                                  # PM.cc:777-783
                d2s = djn * djn * sg
                J = -rzxp / d2s
                r0 = (djn - dp * rzxp - t[2]) / d2s
                sum_Jr32 = np.where(ok, sum_Jr32 + J * r0, sum_Jr32)
                sum_JJ32 = np.where(ok, sum_JJ32 + J * J, sum_JJ32)
                sum_Jr64 = np.where(ok, sum_Jr64 + f64(J) * f64(r0), sum_Jr64)
                sum_JJ64 = np.where(ok, sum_JJ64 + f64(J) * f64(J), sum_JJ64)
            count += (nj >= 1)
        if mode == "cv":
            Jtr0 = (sum_Jr64 * -1.0).astype(f32)                 # -J.t()*r0: alpha = -1
            JtJ = sum_JJ64.astype(f32)
        else:
            Jtr0, JtJ = -sum_Jr32, sum_JJ32
        new = f32(1) / (dp + Jtr0 / JtJ)                         # PM.cc:791-793
    out[ys, xs] = np.where(count < LAMBDA_N, f32(0), new)        # PM.cc:762-765: sigma untouched
    return out


# ---- IntraKeyFrameDepthGrowing, PM.cc:549-596 (GetFusion overload A, PM.cc:926-945) ---------------------------------
def intra_grow(rho, sigma, grad):
    H, W = rho.shape
    out_r, out_s = rho.copy(), sigma.copy()
    inner = np.zeros((H, W), bool)
    inner[2:H - 2, 2:W - 2] = True
    sel = inner & (rho.astype(f64) < 0.000001) & ~(grad < LAMBDA_G)   # PM.cc:560, 562
    ys, xs = np.nonzero(sel)
    P = len(xs)
    if not P:
        return out_r, out_s
    dp, sp = rho[ys, xs], sigma[ys, xs]
    pjsj, rsj = np.zeros(P, f32), np.zeros(P, f32)
    cnt = np.zeros(P, np.int64)
    mins = np.zeros(P, f32)
    with np.errstate(all="ignore"):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx == 0 and dy == 0:
                    continue
                dn, sn = rho[ys + dy, xs + dx], sigma[ys + dy, xs + dx]
                ok = chi_matrix(dn, dp, sn, sp)                          # PM.cc:571: no depth-present test here
                s2 = f64(sn) * f64(sn)
                pjsj = np.where(ok, (f64(pjsj) + f64(dn) / s2).astype(f32), pjsj)
                rsj = np.where(ok, (f64(rsj) + 1.0 / s2).astype(f32), rsj)
                mins = np.where(ok & ((cnt == 0) | (sn < mins)), sn, mins)  # min_sigma starts at supported[0].second
                cnt += ok
        grown = cnt >= 2                                                 # PM.cc:581
        out_r[ys, xs] = np.where(grown, pjsj / rsj, out_r[ys, xs])
        out_s[ys, xs] = np.where(grown, mins, out_s[ys, xs])
    return out_r, out_s
