"""ctypes wrapper of the CPU oracle (oracle/pm_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (orb-slam-free-space-carving_amd/, include/) never does.  PARITY UNPINNED -- see
pm_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_NBR = 64


class KeyFrameC(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int),
                ("im", C.POINTER(C.c_uint8)), ("grad", C.POINTER(C.c_float)),
                ("theta", C.POINTER(C.c_float)), ("I_stddev", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("Tcw", C.c_float * 12)]


class HypoC(C.Structure):
    _fields_ = [("rho", C.c_float), ("sigma", C.c_float), ("supported", C.c_int)]


class ParamsC(C.Structure):
    _fields_ = [("lambdaG", C.c_float), ("lambdaL", C.c_float), ("lambdaTheta", C.c_float),
                ("lambdaN", C.c_int), ("theta_var", C.c_double)]


class PairC(C.Structure):
    _fields_ = [("R21", C.c_float * 9), ("t21", C.c_float * 3), ("F12", C.c_float * 9)]


class StatsC(C.Structure):
    _fields_ = [("searches", C.c_longlong), ("candidates", C.c_longlong),
                ("gate_pass", C.c_longlong), ("hypotheses", C.c_longlong),
                ("fused", C.c_longlong)]


def build(variant="strict", out_dir=None):
    """Compile the oracle.  variant: 'strict' (bit-defined checker) or 'omp' (timed baseline,
    -O3 -march=native + OpenMP; must be built on the machine that runs it)."""
    out_dir = out_dir or os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    name = "libpm_oracle.so" if variant == "strict" else "libpm_oracle_omp.so"
    out = os.path.join(out_dir, name)
    src = os.path.join(_HERE, "pm_oracle.c")
    hdr = os.path.join(_HERE, "pm_oracle.h")
    if os.path.exists(out) and variant == "strict" and \
            os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return out
    base = ["gcc", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"]
    if variant == "strict":
        flags = ["-O2"]
    else:
        flags = ["-O3", "-march=native", "-fopenmp"]
    subprocess.check_call(base + flags + [src, "-o", out, "-lm"])
    return out


class Oracle:
    def __init__(self, variant="strict", out_dir=None):
        self.path = build(variant, out_dir)
        L = self.lib = C.CDLL(self.path)
        f32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        L.pmo_default_params.argtypes = [C.POINTER(ParamsC)]
        L.pmo_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.pmo_fast_atan2.restype = C.c_float
        L.pmo_gradient_prepass.argtypes = [u8p, C.c_int, C.c_int, f32p, f32p, f32p]
        L.pmo_pair_geometry.argtypes = [C.POINTER(KeyFrameC), C.POINTER(KeyFrameC), C.POINTER(PairC)]
        L.pmo_stereo_search_constraints.argtypes = [f32p, C.c_int, f32p, f32p]
        L.pmo_median_rot_in_plane.argtypes = [C.POINTER(C.c_int), f32p, C.c_int,
                                              C.POINTER(C.c_int), f32p, C.c_int]
        L.pmo_median_rot_in_plane.restype = C.c_float
        L.pmo_search_range.argtypes = [C.POINTER(KeyFrameC), C.POINTER(PairC), C.c_int, C.c_int,
                                       C.c_float, C.c_float, f32p, f32p]
        L.pmo_pixel_depth.argtypes = [C.POINTER(KeyFrameC), C.POINTER(PairC), C.c_float, C.c_int, C.c_int]
        L.pmo_pixel_depth.restype = C.c_float
        L.pmo_epipolar_search.argtypes = [C.POINTER(KeyFrameC), C.POINTER(KeyFrameC), C.POINTER(PairC),
                                          C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                          C.POINTER(ParamsC), C.POINTER(HypoC), f32p, f32p,
                                          C.POINTER(StatsC)]
        L.pmo_fuse.argtypes = [C.POINTER(HypoC), C.c_int, C.POINTER(ParamsC), C.POINTER(HypoC)]
        L.pmo_recon_search_fuse.argtypes = [C.POINTER(KeyFrameC), C.POINTER(KeyFrameC), f32p, C.c_int,
                                            C.c_float, C.c_float, C.POINTER(ParamsC), f32p, f32p,
                                            C.POINTER(StatsC)]
        L.pmo_semi_dense_recon.argtypes = L.pmo_recon_search_fuse.argtypes
        L.pmo_intra_check.argtypes = [f32p, f32p, C.c_int, C.c_int]
        L.pmo_intra_grow.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, C.POINTER(ParamsC)]
        L.pmo_inter_check.argtypes = [C.POINTER(KeyFrameC), f32p, C.POINTER(KeyFrameC),
                                      C.POINTER(f32p), C.POINTER(f32p), C.c_int, C.POINTER(ParamsC)]
        L.pmo_pointset.argtypes = [C.POINTER(KeyFrameC), f32p, f32p]
        L.pmo_num_threads.restype = C.c_int
        L.pmo_ingest.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, u8p]
        ip = C.POINTER(C.c_int)
        L.pmo_recon_batch.argtypes = [C.POINTER(KeyFrameC), ip, C.c_int, ip, C.c_int, C.c_float, C.c_float,
                                      C.POINTER(ParamsC), f32p, f32p, C.POINTER(StatsC)]
        L.pmo_inter_pointset_batch.argtypes = [C.POINTER(KeyFrameC), ip, C.c_int, ip, C.c_int,
                                               C.POINTER(f32p), C.POINTER(f32p), C.POINTER(ParamsC),
                                               f32p, f32p, f32p]
        self.params = ParamsC()
        L.pmo_default_params(C.byref(self.params))

    # -- helpers -----------------------------------------------------------------------------
    @staticmethod
    def _f32(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return a, a.ctypes.data_as(C.POINTER(C.c_float))

    def keyframe(self, im, grad, theta, I_stddev, K, Tcw):
        """Returns (KeyFrameC, keepalive) for host arrays. K=(fx,fy,cx,cy); Tcw 3x4."""
        im = np.ascontiguousarray(im, dtype=np.uint8)
        grad = np.ascontiguousarray(grad, dtype=np.float32)
        theta = np.ascontiguousarray(theta, dtype=np.float32)
        H, W = im.shape
        kf = KeyFrameC()
        kf.W, kf.H = W, H
        kf.im = im.ctypes.data_as(C.POINTER(C.c_uint8))
        kf.grad = grad.ctypes.data_as(C.POINTER(C.c_float))
        kf.theta = theta.ctypes.data_as(C.POINTER(C.c_float))
        kf.I_stddev = float(I_stddev)
        kf.fx, kf.fy, kf.cx, kf.cy = [float(np.float32(v)) for v in K]
        t = np.asarray(Tcw, dtype=np.float32).reshape(12)
        for i in range(12):
            kf.Tcw[i] = float(t[i])
        kf._keep = (im, grad, theta)
        return kf

    def kf_array(self, kfs):
        arr = (KeyFrameC * len(kfs))()
        for i, k in enumerate(kfs):
            C.memmove(C.byref(arr, i * C.sizeof(KeyFrameC)), C.byref(k), C.sizeof(KeyFrameC))
        arr._keep = list(kfs)
        return arr

    # -- API ---------------------------------------------------------------------------------
    def num_threads(self):
        return int(self.lib.pmo_num_threads())

    def fast_atan2(self, y, x):
        return float(self.lib.pmo_fast_atan2(float(y), float(x)))

    def ingest(self, pixels, order, K, dist):
        """camera frame -> undistorted gray (Tracking.cc:244-271, Modeler.cc:154-155); order in rgb/bgr/rgba/bgra/gray"""
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        ch, r, g, b = dict(rgb=(3, 0, 1, 2), bgr=(3, 2, 1, 0), rgba=(4, 0, 1, 2), bgra=(4, 2, 1, 0), gray=(1, 0, 0, 0))[order]
        H, W = px.shape[:2]
        assert px.size == H * W * ch
        k, kp = self._f32(K)
        dp = None
        if dist is not None:
            d, dp = self._f32(dist)
        out = np.empty((H, W), np.uint8)
        u8p = C.POINTER(C.c_uint8)
        self.lib.pmo_ingest(px.ctypes.data_as(u8p), W, H, ch, r, g, b, kp, dp, out.ctypes.data_as(u8p))
        return out

    def gradient_prepass(self, im):
        im = np.ascontiguousarray(im, dtype=np.uint8)
        H, W = im.shape
        grad = np.empty((H, W), np.float32)
        theta = np.empty((H, W), np.float32)
        s = C.c_float()
        self.lib.pmo_gradient_prepass(im.ctypes.data_as(C.POINTER(C.c_uint8)), W, H,
                                      grad.ctypes.data_as(C.POINTER(C.c_float)),
                                      theta.ctypes.data_as(C.POINTER(C.c_float)), C.byref(s))
        return grad, theta, float(s.value)

    def pair_geometry(self, kf1, kf2):
        p = PairC()
        self.lib.pmo_pair_geometry(C.byref(kf1), C.byref(kf2), C.byref(p))
        return p

    def stereo_search_constraints(self, depths):
        d, dp = self._f32(depths)
        mn, mx = C.c_float(), C.c_float()
        self.lib.pmo_stereo_search_constraints(dp, len(d), C.byref(mn), C.byref(mx))
        return float(mn.value), float(mx.value)

    def median_rot_in_plane(self, mp1, ang1, mp2, ang2):
        mp1 = np.ascontiguousarray(mp1, dtype=np.int32)
        mp2 = np.ascontiguousarray(mp2, dtype=np.int32)
        a1, a1p = self._f32(ang1)
        a2, a2p = self._f32(ang2)
        ip = C.POINTER(C.c_int)
        return float(self.lib.pmo_median_rot_in_plane(mp1.ctypes.data_as(ip), a1p, len(mp1),
                                                      mp2.ctypes.data_as(ip), a2p, len(mp2)))

    def search_range(self, kf1, pair, px, py, mind, maxd):
        a, b = C.c_float(), C.c_float()
        self.lib.pmo_search_range(C.byref(kf1), C.byref(pair), px, py, mind, maxd, C.byref(a), C.byref(b))
        return float(a.value), float(b.value)

    def pixel_depth(self, kf1, pair, uj, px, py):
        return float(self.lib.pmo_pixel_depth(C.byref(kf1), C.byref(pair), float(uj), px, py))

    def epipolar_search(self, kf1, kf2, x, y, min_depth, max_depth, rot=0.0, pair=None):
        pair = pair or self.pair_geometry(kf1, kf2)
        h = HypoC()
        bu, bv = C.c_float(), C.c_float()
        self.lib.pmo_epipolar_search(C.byref(kf1), C.byref(kf2), C.byref(pair), x, y, min_depth,
                                     max_depth, rot, C.byref(self.params), C.byref(h),
                                     C.byref(bu), C.byref(bv), None)
        return dict(rho=float(h.rho), sigma=float(h.sigma), supported=int(h.supported),
                    best_u=float(bu.value), best_v=float(bv.value))

    def fuse(self, rho, sigma):
        n = len(rho)
        arr = (HypoC * max(n, 1))()
        for i in range(n):
            arr[i].rho, arr[i].sigma, arr[i].supported = float(rho[i]), float(sigma[i]), 1
        out = HypoC()
        self.lib.pmo_fuse(arr, n, C.byref(self.params), C.byref(out))
        return float(out.rho), float(out.sigma), int(out.supported)

    def _recon(self, fn, ref, nbrs, rot, min_depth, max_depth):
        n = len(nbrs)
        arr = self.kf_array(nbrs)
        rot, rotp = self._f32(rot if rot is not None else np.zeros(n, np.float32))
        rho = np.zeros((ref.H, ref.W), np.float32)
        sigma = np.zeros((ref.H, ref.W), np.float32)
        st = StatsC()
        fn(C.byref(ref), arr, rotp, n, min_depth, max_depth, C.byref(self.params),
           rho.ctypes.data_as(C.POINTER(C.c_float)), sigma.ctypes.data_as(C.POINTER(C.c_float)),
           C.byref(st))
        stats = {k: int(getattr(st, k)) for k, _ in StatsC._fields_}
        return rho, sigma, stats

    def recon_search_fuse(self, ref, nbrs, rot, min_depth, max_depth):
        return self._recon(self.lib.pmo_recon_search_fuse, ref, nbrs, rot, min_depth, max_depth)

    def semi_dense_recon(self, ref, nbrs, rot, min_depth, max_depth):
        return self._recon(self.lib.pmo_semi_dense_recon, ref, nbrs, rot, min_depth, max_depth)

    def intra_check(self, rho, sigma):
        rho = np.array(rho, dtype=np.float32, order="C")
        sigma = np.array(sigma, dtype=np.float32, order="C")
        H, W = rho.shape
        self.lib.pmo_intra_check(rho.ctypes.data_as(C.POINTER(C.c_float)),
                                 sigma.ctypes.data_as(C.POINTER(C.c_float)), W, H)
        return rho, sigma

    def intra_grow(self, rho, sigma, grad):
        rho = np.array(rho, dtype=np.float32, order="C")
        sigma = np.array(sigma, dtype=np.float32, order="C")
        g, gp = self._f32(grad)
        H, W = rho.shape
        self.lib.pmo_intra_grow(rho.ctypes.data_as(C.POINTER(C.c_float)),
                                sigma.ctypes.data_as(C.POINTER(C.c_float)), gp, W, H,
                                C.byref(self.params))
        return rho, sigma

    def inter_check(self, cur, cur_rho, nbrs, nbr_rho, nbr_sigma):
        n = len(nbrs)
        arr = self.kf_array(nbrs)
        out = np.array(cur_rho, dtype=np.float32, order="C")
        rr = [np.ascontiguousarray(r, dtype=np.float32) for r in nbr_rho]
        ss = [np.ascontiguousarray(s, dtype=np.float32) for s in nbr_sigma]
        f32p = C.POINTER(C.c_float)
        rp = (f32p * max(n, 1))(*[r.ctypes.data_as(f32p) for r in rr])
        sp = (f32p * max(n, 1))(*[s.ctypes.data_as(f32p) for s in ss])
        self.lib.pmo_inter_check(C.byref(cur), out.ctypes.data_as(f32p), arr, rp, sp, n,
                                 C.byref(self.params))
        return out

    def pointset(self, kf, rho):
        r, rp = self._f32(rho)
        xyz = np.zeros((kf.H, 3 * kf.W), np.float32)
        self.lib.pmo_pointset(C.byref(kf), rp, xyz.ctypes.data_as(C.POINTER(C.c_float)))
        return xyz

    # -- keyframe-batched drivers (timed CPU baseline; rows of all keyframes shared out to the threads) --
    def recon_batch(self, kfs, ref_idx, nbr_idx, min_depth, max_depth):
        """kfs: list of KeyFrameC; ref_idx [n_ref], nbr_idx [n_ref][n] index into it.
        Returns rho, sigma [n_ref,H,W] and the scan statistics."""
        arr = self.kf_array(kfs)
        ri = np.ascontiguousarray(ref_idx, dtype=np.int32)
        ni = np.ascontiguousarray(nbr_idx, dtype=np.int32).reshape(len(ri), -1)
        H, W = kfs[0].H, kfs[0].W
        rho = np.empty((len(ri), H, W), np.float32)
        sigma = np.empty((len(ri), H, W), np.float32)
        st = StatsC()
        ip, f32p = C.POINTER(C.c_int), C.POINTER(C.c_float)
        self.lib.pmo_recon_batch(arr, ri.ctypes.data_as(ip), len(ri), ni.ctypes.data_as(ip), ni.shape[1],
                                 min_depth, max_depth, C.byref(self.params), rho.ctypes.data_as(f32p),
                                 sigma.ctypes.data_as(f32p), C.byref(st))
        return rho, sigma, {k: int(getattr(st, k)) for k, _ in StatsC._fields_}

    def inter_pointset_batch(self, kfs, ref_idx, nbr_idx, map_rho, map_sigma, rho_in, with_xyz=True):
        """map_rho/map_sigma: per entry of kfs, the finished map (None where no reference reads it)."""
        arr = self.kf_array(kfs)
        ri = np.ascontiguousarray(ref_idx, dtype=np.int32)
        ni = np.ascontiguousarray(nbr_idx, dtype=np.int32).reshape(len(ri), -1)
        H, W = kfs[0].H, kfs[0].W
        ip, f32p = C.POINTER(C.c_int), C.POINTER(C.c_float)
        keep = []

        def ptrs(maps):
            out = (f32p * len(kfs))()
            for i, m in enumerate(maps):
                if m is not None:
                    a = np.ascontiguousarray(m, dtype=np.float32)
                    keep.append(a)
                    out[i] = a.ctypes.data_as(f32p)
            return out
        rp, sp = ptrs(map_rho), ptrs(map_sigma)
        rin = np.ascontiguousarray(rho_in, dtype=np.float32)
        chk = np.empty((len(ri), H, W), np.float32)
        xyz = np.zeros((len(ri), H, 3 * W), np.float32) if with_xyz else None
        self.lib.pmo_inter_pointset_batch(arr, ri.ctypes.data_as(ip), len(ri), ni.ctypes.data_as(ip), ni.shape[1],
                                          rp, sp, C.byref(self.params), rin.ctypes.data_as(f32p),
                                          chk.ctypes.data_as(f32p),
                                          xyz.ctypes.data_as(f32p) if with_xyz else None)
        return chk, xyz
