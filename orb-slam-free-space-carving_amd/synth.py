"""Synthetic keyframe sequences for the ProbabilityMapping path.

The reference's dataset (TUM fr1_xyz), ORB vocabulary and OpenCV are not available offline, so
every BASELINE.json config is realised with this generator: a textured plane
``Z = Z0 + aX + bY`` (Z0 = 1: ORB-SLAM's monocular map is normalised to median depth 1,
/root/reference/src/Tracking.cc:698-720, and PM.cc:877-910's search range only brackets the true
match at that scale), seen by cameras that move predominantly along +X so that epipolar lines
pass the |a/b| <= 4 gate of PM.cc:393.

All parameter draws come from SplitMix64 in a fixed order; the per-pixel texture is evaluated in
float64 with torch (CPU or GPU), so images are reproducible up to libm/GPU `sin` rounding at
exact .5 quantisation ties -- golden fixtures therefore commit their images.
"""
import math

import numpy as np
import torch

TUM1 = dict(W=640, H=480, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989)
HD720 = dict(W=1280, H=720, fx=1101.107438, fy=1095.367233, cx=604.565580, cy=365.568908)
HD1080 = dict(W=1920, H=1080, fx=1101.107438 * 1.5, fy=1095.367233 * 1.5,
              cx=604.565580 * 1.5, cy=365.568908 * 1.5)


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next() >> 11) * (1.0 / (1 << 53)))


def scaled_intrinsics(base, W, H):
    """Intrinsics of `base` rescaled to a W x H image (used for the small test fixtures)."""
    sx, sy = W / base["W"], H / base["H"]
    return dict(W=W, H=H, fx=base["fx"] * sx, fy=base["fy"] * sy, cx=base["cx"] * sx,
                cy=base["cy"] * sy)


class Scene:
    """Plane + texture + camera path.  Keyframe k can be rendered independently of the others
    (ranks of a multi-GPU run render only their own block plus halo)."""

    N_WAVES = 32

    def __init__(self, cam, seed, disparity_px=2.6, z0=1.0, noise_images=False, strip=False, roll_deg=1.0):
        """disparity_px: disparity between adjacent keyframes at depth Z0 (App. D's b = 0.005*Z0
        at TUM1's fx is 2.6 px); it is the knob that sets the mean scan length of PM.cc:405.
        Texture frequencies are App. D's [2,40] cycles/unit at TUM1's fx, rescaled with fx so the
        texture has the same period in PIXELS (13..260 px) at every resolution.
        strip: App. D's optional second plane -- a fronto-parallel foreground strip at 0.75 Z0 with its own texture, a quarter
        of the view wide: a depth discontinuity with an occluding edge on either side (the background next to it is seen by
        some keyframes and hidden from others as the camera moves along X).
        roll_deg: keyframe k's in-plane roll is drawn from [-roll_deg, roll_deg] (1 degree in App. D); with larger values a
        pair's images are rotated against each other and PM.cc:170-179's median rotation (rot_deg) is no longer ~0."""
        self.cam = dict(cam)
        self.seed = seed
        self.z0 = z0
        self.b = disparity_px * z0 / cam["fx"]
        self.noise_images = noise_images
        self.strip = bool(strip)
        self.roll_deg = float(roll_deg)
        fscale = cam["fx"] / TUM1["fx"]
        r = SplitMix64(seed)
        self.alpha = r.uniform(-0.1, 0.1)
        self.beta = r.uniform(-0.1, 0.1)
        self.freq = [fscale * math.exp(r.uniform(math.log(2.0), math.log(40.0)))
                     for _ in range(self.N_WAVES)]
        self.phi = [r.uniform(0.0, 2 * math.pi) for _ in range(self.N_WAVES)]
        self.psi = [r.uniform(0.0, 2 * math.pi) for _ in range(self.N_WAVES)]
        self.wgt = [r.uniform(0.5, 1.0) for _ in range(self.N_WAVES)]
        self._rot_seed = r.next()
        # the foreground strip (drawn AFTER everything the one-plane scene draws: the same seed gives the same background)
        self.strip_z = 0.75 * z0
        half_view = 0.5 * cam["W"] / cam["fx"] * z0
        self.strip_x0 = -0.25 * half_view            # world X extent, about a quarter of the view at keyframe 0
        self.strip_x1 = self.strip_x0 + 0.5 * half_view
        # amplitude: ~20 % of pixels should pass the lambdaG = 8 gate (Scharr/32 magnitude).  The
        # analytic gradient of the texture in pixels is |dT/dX| / f (one pixel = 1/fx world units
        # at Z0); pick A so that the 80th percentile of that magnitude is 8 gray levels per pixel.
        g = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFF)
        XY = (torch.rand(20000, 2, generator=g, dtype=torch.float64) - 0.5) * 1.2 / fscale
        gx = torch.zeros(20000, dtype=torch.float64)
        gy = torch.zeros(20000, dtype=torch.float64)
        for k in range(self.N_WAVES):
            w = 2 * math.pi * self.freq[k]
            c, s = math.cos(self.phi[k]), math.sin(self.phi[k])
            ph = w * (XY[:, 0] * c + XY[:, 1] * s) + self.psi[k]
            gx += self.wgt[k] * w * c * torch.cos(ph)
            gy += self.wgt[k] * w * s * torch.cos(ph)
        mag = torch.sqrt(gx * gx + gy * gy) / (cam["fx"] / z0)
        self.amp = 8.0 / float(torch.quantile(mag, 0.80))

    # -- camera path ---------------------------------------------------------------------------
    def pose(self, k):
        """Returns (Rwc 3x3, C 3) float64 for keyframe k."""
        b = self.b
        C = np.array([k * b, 0.1 * b * math.sin(0.7 * k), 0.05 * b * math.cos(0.3 * k)])
        r = SplitMix64(self._rot_seed ^ (0xA5A5A5A5 + 0x9E3779B97F4A7C15 * (k + 1)))
        roll = math.radians(r.uniform(-1.0, 1.0) * self.roll_deg)
        pitch = math.radians(r.uniform(-0.3, 0.3))
        yaw = math.radians(r.uniform(-0.3, 0.3))
        cz, sz = math.cos(roll), math.sin(roll)
        cx_, sx_ = math.cos(pitch), math.sin(pitch)
        cy_, sy_ = math.cos(yaw), math.sin(yaw)
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
        Rx = np.array([[1.0, 0, 0], [0, cx_, -sx_], [0, sx_, cx_]])
        Ry = np.array([[cy_, 0, sy_], [0, 1.0, 0], [-sy_, 0, cy_]])
        return Ry @ Rx @ Rz, C

    def roll(self, k):
        """keyframe k's in-plane roll in degrees (the first draw of pose(k))"""
        r = SplitMix64(self._rot_seed ^ (0xA5A5A5A5 + 0x9E3779B97F4A7C15 * (k + 1)))
        return r.uniform(-1.0, 1.0) * self.roll_deg

    def rot_deg(self, k_ref, k_nbr):
        """What PM.cc:170-179 would measure for the pair: the median over shared ORB features of (angle in the neighbour -
        angle in the reference).  A camera rolled by +r about its optical axis sees the scene rotated by -r, so image
        directions in the neighbour are those of the reference plus roll(ref) - roll(nbr) (pitch / yaw of 0.3 degrees do
        not matter at this precision).  float32, as GetRotInPlane returns it."""
        return float(np.float32(self.roll(k_ref) - self.roll(k_nbr)))

    def Tcw(self, k):
        Rwc, C = self.pose(k)
        Rcw = Rwc.T
        t = -Rcw @ C
        return np.concatenate([Rcw, t[:, None]], axis=1).astype(np.float32)

    def K(self):
        c = self.cam
        return np.array([c["fx"], c["fy"], c["cx"], c["cy"]], dtype=np.float32)

    # -- rendering -------------------------------------------------------------------------------
    def render(self, k, device="cpu"):
        """Returns (im uint8 HxW tensor on `device`, gt_rho float32 HxW tensor)."""
        c = self.cam
        W, H = c["W"], c["H"]
        Rwc, C = self.pose(k)
        dev = torch.device(device)
        xs = (torch.arange(W, dtype=torch.float64, device=dev) - c["cx"]) / c["fx"]
        ys = (torch.arange(H, dtype=torch.float64, device=dev) - c["cy"]) / c["fy"]
        yn, xn = torch.meshgrid(ys, xs, indexing="ij")
        dx = Rwc[0, 0] * xn + Rwc[0, 1] * yn + Rwc[0, 2]
        dy = Rwc[1, 0] * xn + Rwc[1, 1] * yn + Rwc[1, 2]
        dz = Rwc[2, 0] * xn + Rwc[2, 1] * yn + Rwc[2, 2]
        s = (self.z0 + self.alpha * C[0] + self.beta * C[1] - C[2]) / (dz - self.alpha * dx - self.beta * dy)
        fg = None
        if self.strip:  # ray / foreground-plane intersection; the strip wins where the hit lies inside its X extent
            s2 = (self.strip_z - C[2]) / dz
            X2 = C[0] + s2 * dx
            fg = (X2 >= self.strip_x0) & (X2 <= self.strip_x1) & (s2 > 0) & (s2 < s)
            s = torch.where(fg, s2, s)
        self.last_fg = fg  # (H, W) bool of the last rendered keyframe's foreground pixels, or None
        gt_rho = (1.0 / s).to(torch.float32)
        if self.noise_images:
            g = torch.Generator(device="cpu").manual_seed((int(self.seed) * 1000003 + k) & 0x7FFFFFFF)
            im = torch.randint(0, 256, (H, W), generator=g, dtype=torch.uint8).to(dev)
            return im, gt_rho
        X = C[0] + s * dx
        Y = C[1] + s * dy
        T = torch.zeros_like(X)
        for i in range(self.N_WAVES):
            w = 2 * math.pi * self.freq[i]
            T += self.wgt[i] * torch.sin(w * (X * math.cos(self.phi[i]) + Y * math.sin(self.phi[i])) + self.psi[i])
        if fg is not None:  # the strip's own texture: the same spectrum, directions and phases permuted
            T2 = torch.zeros_like(X)
            for i in range(self.N_WAVES):
                j = (7 * i + 3) % self.N_WAVES
                w = 2 * math.pi * self.freq[i] * (self.z0 / self.strip_z)
                T2 += self.wgt[j] * torch.sin(w * (X * math.cos(self.phi[j] + 1.0) + Y * math.sin(self.phi[j] + 1.0)) + self.psi[i] + 2.0)
            T = torch.where(fg, T2, T)
        im = torch.clamp(torch.floor(127.5 + self.amp * T + 0.5), 0, 255).to(torch.uint8)
        return im, gt_rho

    # -- covisibility stand-in --------------------------------------------------------------------
    @staticmethod
    def neighbours(k, n_kf, n):
        """The n nearest keyframe indices ordered by |dk| (then +dk before -dk): stands in for
        GetVectorCovisibleKeyFrames()'s descending-covisibility order (src/KeyFrame.cc:168-172)."""
        if n_kf - 1 < n:
            raise ValueError("sequence too short for %d neighbours" % n)
        out = []
        d = 1
        while len(out) < n:
            for cand in (k + d, k - d):
                if 0 <= cand < n_kf and len(out) < n:
                    out.append(cand)
            d += 1
        return out

    def depth_prior(self, spread=0.1):
        """(min_depth, max_depth) exactly as StereoSearchConstraints names them (PM.cc:381-382):
        inverse depths 1/(mu-2s), 1/(mu+2s) with mu = Z0, s = spread*Z0 (App. D: 0.1; the ORB depths of a real keyframe
        spread more like 0.3 ... 0.5 of their mean, and the reference does not guard mu - 2s <= 0), computed in float32."""
        mu = np.float32(self.z0)
        sd = np.float32(spread) * mu
        max_depth = np.float32(1) / (mu + np.float32(2) * sd)
        min_depth = np.float32(1) / (mu - np.float32(2) * sd)
        return float(min_depth), float(max_depth)
