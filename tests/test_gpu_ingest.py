"""GPU: sdm_upload_image_rgb (device colour->gray + lens undistortion feeding the gradient pre-pass; SURVEY.md §8f-1,
src/Tracking.cc:244-257, 266-271, src/Modeler/Modeler.cc:154-155) against the oracle's restatement, bit for bit --
the gray image, the derived GradImg / GradTheta / I_stddev, and the whole path run from colour frames.
PARITY UNPINNED for the OpenCV pieces (absent from the image): both sides state the same published algorithm."""
import numpy as np
import pytest

from common import Sequence, assert_bit_equal, oracle_inter, oracle_pipeline
from test_oracle_ingest import TUM1_DIST, TUM1_K

pytestmark = pytest.mark.gpu
EYE = np.float32([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]])


@pytest.mark.parametrize("W,H", [(160, 120), (75, 53)])
def test_ingest_matches_oracle(pkg, oracle, gpu_ok, W, H):
    rng = np.random.default_rng(W)
    eng = pkg.Engine(W, H, 2)
    K = TUM1_K * np.float32(W / 640.0)
    for order, ch in (("rgb", 3), ("bgr", 3), ("rgba", 4), ("bgra", 4), ("gray", 1)):
        # smooth + noisy content so that both the interpolation and the rounding are exercised
        yy, xx = np.mgrid[0:H, 0:W]
        base = (127 + 100 * np.sin(xx / 7.0) * np.cos(yy / 5.0))[..., None] + rng.integers(-20, 20, (H, W, ch))
        px = np.clip(base, 0, 255).astype(np.uint8)
        if ch == 1:
            px = px[..., 0]
        for dist in (TUM1_DIST, None, np.float32([-0.35, 0.12, 0.002, -0.001, 0.0])):
            eng.upload_image_rgb(0, px, order, K, dist, EYE)
            im, g, t, s = eng.download_inputs(0)
            want = oracle.ingest(px, order, K, dist)
            assert (im == want).all(), (order, dist, int((im != want).sum()))
            wg, wt, ws = oracle.gradient_prepass(want)
            assert_bit_equal(g, wg, "GradImg")
            assert_bit_equal(t, wt, "GradTheta")
            assert np.float32(s) == np.float32(ws)
    # strong distortion pulls source positions outside the frame: the constant zero border shows up in the corners
    eng.upload_image_rgb(1, np.full((H, W, 3), 200, np.uint8), "rgb", K, np.float32([0.9, 0, 0, 0, 0]), EYE)
    im = eng.download_inputs(1)[0]
    assert im[0, 0] == 0 and im[H // 2, W // 2] == 200
    assert (im == oracle.ingest(np.full((H, W, 3), 200, np.uint8), "rgb", K, np.float32([0.9, 0, 0, 0, 0]))).all()
    assert eng.lib.sdm_upload_image_rgb(eng.ctx, 0, None, 0, None, None, None) == 1  # SDM_EINVAL: null input
    with pytest.raises(KeyError):
        eng.upload_image_rgb(0, np.zeros((H, W, 3), np.uint8), "yuv", K, None, EYE)
    eng.close()


def test_whole_path_from_colour_frames(pkg, oracle, gpu_ok):
    """frames arrive as distorted RGB (as Tracking gets them); the engine ingests them on the device and the maps equal
    the oracle run on the oracle-ingested gray images"""
    base = Sequence(pkg, oracle, 160, 120, 8, 0x5EED0D01)
    rng = np.random.default_rng(3)
    K = base.K
    dist = TUM1_DIST
    frames = []
    for k in range(base.n_kf):  # colour frames whose gray value is close to the synthetic texture
        g = base.im[k].astype(np.int32)
        rgb = np.stack([np.clip(g + rng.integers(-6, 6, g.shape), 0, 255) for _ in range(3)], axis=2).astype(np.uint8)
        frames.append(rgb)
    grays = [oracle.ingest(f, "rgb", K, dist) for f in frames]
    seq = Sequence(pkg, oracle, 160, 120, 8, 0x5EED0D01, images=grays)
    n = 7
    eng = pkg.Engine(seq.W, seq.H, seq.n_kf, max_neighbours=n)
    for k in range(seq.n_kf):
        eng.upload_image_rgb(k, frames[k], "rgb", K, dist, seq.Tcw[k])
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
    eng.inter_check_pointset(refs, nbrs)
    maps = oracle_pipeline(oracle, seq, n)
    chk, xyz = oracle_inter(oracle, seq, n, maps)
    kept = 0
    for k in refs:
        r, s = eng.download_depth(k)
        assert_bit_equal(r, maps["rho"][k], "rho kf %d" % k)
        assert_bit_equal(s, maps["sigma"][k], "sigma kf %d" % k)
        assert_bit_equal(eng.download_checked(k), chk[k], "checked kf %d" % k)
        assert_bit_equal(eng.download_pointset(k), xyz[k], "xyz kf %d" % k)
        kept += int((chk[k] > 1e-6).sum())
    assert kept > 500
    eng.close()
