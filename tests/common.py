"""Shared helpers for the parity tests: build a synthetic keyframe sequence once and hand the SAME
inputs to the CPU oracle (checker) and to the HIP engine (product)."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        idx = np.argwhere(~same)
        first = tuple(idx[0])
        raise AssertionError("%s: %d of %d values differ; first at %s: %r vs %r" %
                             (what, len(idx), a.size, first, a[first], b[first]))


class Sequence:
    """n_kf synthetic keyframes (images + oracle-derived gradient inputs + poses)."""

    def __init__(self, pkg, oracle, W, H, n_kf, seed, disparity_px=2.6, base="TUM1", noise=False,
                 images=None, strip=False, roll_deg=1.0):
        synth = pkg.synth
        cam0 = getattr(synth, base)
        cam = cam0 if (W, H) == (cam0["W"], cam0["H"]) else synth.scaled_intrinsics(cam0, W, H)
        self.scene = synth.Scene(cam, seed, disparity_px=disparity_px, noise_images=noise, strip=strip, roll_deg=roll_deg)
        self.with_rot = roll_deg != 1.0  # pairs carry PM.cc:170-179's median in-plane rotation (else 0, as App. D says)
        self.W, self.H, self.n_kf = W, H, n_kf
        self.K = self.scene.K()
        self.Tcw = [self.scene.Tcw(k) for k in range(n_kf)]
        self.im, self.gt, self.fg = [], [], []
        for k in range(n_kf):
            if images is not None:
                self.im.append(np.ascontiguousarray(images[k], dtype=np.uint8))
                self.gt.append(None)
                self.fg.append(None)
            else:
                im, gt = self.scene.render(k)
                self.im.append(im.numpy())
                self.gt.append(gt.numpy())
                fg = self.scene.last_fg
                self.fg.append(np.zeros((H, W), bool) if fg is None else fg.numpy())
        self.grad, self.theta, self.istd = [], [], []
        for k in range(n_kf):
            g, t, s = oracle.gradient_prepass(self.im[k])
            self.grad.append(g)
            self.theta.append(t)
            self.istd.append(s)
        self.min_depth, self.max_depth = self.scene.depth_prior()
        self.okf = [oracle.keyframe(self.im[k], self.grad[k], self.theta[k], self.istd[k], self.K, self.Tcw[k])
                    for k in range(n_kf)]

    def neighbours(self, k, n):
        return self.scene.neighbours(k, self.n_kf, n)

    def rot(self, k, n):
        """median in-plane rotation of every (k, neighbour) pair, float32 degrees; None when the scene leaves it at 0"""
        if not self.with_rot:
            return None
        return np.float32([self.scene.rot_deg(k, j) for j in self.neighbours(k, n)])

    def rots(self, refs, n):
        return None if not self.with_rot else np.stack([self.rot(k, n) for k in refs])

    def upload(self, eng, device_prepass=False):
        for k in range(self.n_kf):
            if device_prepass:
                eng.upload_image(k, self.im[k], self.K, self.Tcw[k])
            else:
                eng.upload_keyframe(k, self.im[k], self.grad[k], self.theta[k], self.istd[k], self.K, self.Tcw[k])


def oracle_pipeline(oracle, seq, n, refs=None, rot=None):
    """Whole path on the CPU oracle with SNAPSHOT inter-keyframe semantics:
    returns dict of per-keyframe arrays."""
    refs = list(range(seq.n_kf)) if refs is None else refs
    out = dict(k1_rho={}, k1_sigma={}, rho={}, sigma={}, chk={}, xyz={}, stats={})
    for k in refs:
        nb = seq.neighbours(k, n)
        rk = seq.rot(k, n) if rot is None else rot
        r, s, st = oracle.recon_search_fuse(seq.okf[k], [seq.okf[j] for j in nb], rk, seq.min_depth, seq.max_depth)
        out["k1_rho"][k], out["k1_sigma"][k], out["stats"][k] = r, s, st
        r2, s2 = oracle.intra_check(r, s)
        r3, s3 = oracle.intra_grow(r2, s2, seq.grad[k])
        out["rho"][k], out["sigma"][k] = r3, s3
    return out


def oracle_inter(oracle, seq, n, maps, refs=None):
    refs = list(range(seq.n_kf)) if refs is None else refs
    chk, xyz = {}, {}
    for k in refs:
        nb = seq.neighbours(k, n)
        chk[k] = oracle.inter_check(seq.okf[k], maps["rho"][k], [seq.okf[j] for j in nb],
                                    [maps["rho"][j] for j in nb], [maps["sigma"][j] for j in nb])
        xyz[k] = oracle.pointset(seq.okf[k], chk[k])
    return chk, xyz
