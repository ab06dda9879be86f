"""One-off check at the largest image the engine accepts (W*H = 2^27): 32-bit byte offsets, 24-bit multiplies and
the (y<<16|x) pixel lists at their limits.  16384 x 8192, 3 keyframes (about 17 GB of device memory); per-pixel
searches at a few hundred scattered pixels (including the last rows/columns) are compared with the oracle, which
receives the same three full-size images; the batched kernels are then run over everything and spot-checked
against those per-pixel results."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sdm_pkg  # noqa: E402
from pm_oracle import Oracle  # noqa: E402

pkg = sdm_pkg.load()
oracle = Oracle("omp", out_dir=os.path.join(ROOT, "gpurun_out", "_oracle_build"))
W, H, n_kf = 16384, 8192, 3
rng = np.random.default_rng(1)
t0 = time.time()
# smooth random texture (so that gates pass and matches exist), shifted by one pixel per keyframe
small = rng.integers(0, 256, (H // 8 + 2, W // 8 + 4)).astype(np.float32)
base = np.kron(small, np.ones((8, 8), np.float32))
base += rng.normal(0, 6, base.shape).astype(np.float32)
base = np.clip(base, 0, 255).astype(np.uint8)
K = np.float32([1.2 * W, 1.2 * W, (W - 1) / 2.0, (H - 1) / 2.0])
eng = pkg.Engine(W, H, n_kf, max_neighbours=2, with_pointset=False)
okf = []
for k in range(n_kf):
    im = np.ascontiguousarray(base[:H, 2 * k:2 * k + W])
    Tcw = np.concatenate([np.eye(3), np.float32([[-2.0 * k / (1.2 * W)], [0.0], [0.0]])], axis=1).astype(np.float32)
    eng.upload_image(k, im, K, Tcw)
    g, th, s = oracle.gradient_prepass(im)
    okf.append(oracle.keyframe(im, g, th, s, K, Tcw))
    dim, dg, dth, dstd = eng.download_inputs(k)
    assert np.array_equal(dg.view(np.uint32), g.view(np.uint32)) and np.array_equal(dth.view(np.uint32), th.view(np.uint32))
    assert np.float32(dstd).view(np.uint32) == np.float32(s).view(np.uint32)
print("uploaded + pre-pass bit-equal to the oracle at %dx%d (%.0f s)" % (W, H, time.time() - t0), flush=True)
pts = [(int(rng.integers(2, W - 2)), int(rng.integers(2, H - 2))) for _ in range(300)]
pts += [(W - 3, H - 3), (W - 3, 2), (2, H - 3), (W - 3, H // 2), (W // 2, H - 3), (W - 4, H - 4)]
nsup = 0
for (x, y) in pts:
    for (a, b) in ((0, 1), (1, 2), (2, 0)):
        got = eng.epipolar_search(a, b, x, y, 0.25, 4.0, 0.0)
        ref = oracle.epipolar_search(okf[a], okf[b], x, y, 0.25, 4.0, 0.0)
        assert got["supported"] == ref["supported"], (x, y, a, b)
        ga = np.float32([got["rho"], got["sigma"], got["best_u"], got["best_v"]])
        ra = np.float32([ref["rho"], ref["sigma"], ref["best_u"], ref["best_v"]])
        assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32)), (x, y, a, b, ga, ra)
        nsup += int(ref["supported"])
print("per-pixel searches bit-equal at %d pixel-pairs (%d supported)" % (3 * len(pts), nsup), flush=True)
refs = [0, 1, 2]
nbrs = [[1, 2], [0, 2], [0, 1]]
eng.set_params(lambdaN=1)  # two neighbours only: let two hypotheses fuse so that the later stages have input
oracle.params.lambdaN = 1
t1 = time.time()
eng.search_fuse(refs, nbrs, 0.25, 4.0)
eng.synchronize()
print("batched K1 over 3 x 2^27 pixels: %.2f s" % (time.time() - t1), flush=True)
r0, s0 = eng.download_depth(0)
_, g0, _, _ = eng.download_inputs(0)
nchk = 0
for (x, y) in pts:
    hyp = [oracle.epipolar_search(okf[0], okf[b], x, y, 0.25, 4.0, 0.0) for b in (1, 2)]
    hyp = [h for h in hyp if h["supported"] and 1.0 / h["rho"] > 0]
    want = (0.0, 0.0)
    if g0[y, x] >= 8 and len(hyp) > 1:
        fr, fs, ok = oracle.fuse(np.float32([h["rho"] for h in hyp]), np.float32([h["sigma"] for h in hyp]))
        if ok:
            want = (fr, fs)
    got = np.float32([r0[y, x], s0[y, x]])
    assert np.array_equal(got.view(np.uint32), np.float32(want).view(np.uint32)), (x, y, got, want)
    nchk += int(want[0] != 0)
print("batched K1 agrees with per-pixel search + fusion at %d sample pixels (%d fused); %d supported pixels in keyframe 0"
      % (len(pts), nchk, int((r0 > 1e-6).sum())), flush=True)
t1 = time.time()
eng.recon(refs, nbrs, 0.25, 4.0)
eng.inter_check(refs, nbrs)
eng.synchronize()
c0 = eng.download_checked(0)
print("batched K1-K4: %.2f s; keyframe 0 keeps %d checked pixels (last rows: %d)" %
      (time.time() - t1, int((c0 > 1e-6).sum()), int((c0[H - 6:] > 1e-6).sum())), flush=True)
print("huge check ok", flush=True)
eng.close()
