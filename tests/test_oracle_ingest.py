"""CPU: the image-ingest restatement (oracle/pm_oracle.c pmo_ingest: lens undistortion + colour->gray as the fork
applies them before the path, src/Tracking.cc:244-257, 266-271 and src/Modeler/Modeler.cc:154-155).

OpenCV is absent from the image, so this is PARITY UNPINNED: cv::undistort and cvtColor's published 8-bit
algorithms restated from memory.  What is checked: known answers of the gray weights (the values OpenCV documents
for pure primaries), identity behaviour, an independent vectorised NumPy restatement bit for bit, and the geometry
(an undistorted straight-line scene is straight)."""
import numpy as np
import pytest

TUM1_K = np.float32([517.306408, 516.469215, 318.643040, 255.313989])
TUM1_DIST = np.float32([0.262383, -0.953104, -0.005358, 0.002628, 1.163314])  # k1 k2 p1 p2 k3, Examples/Monocular/TUM1.yaml:13-17


def np_ingest(px, order, K, dist):
    """independent restatement: float64 NumPy, vectorised"""
    px = np.asarray(px, np.uint8)
    if order == "gray":
        chans = [px.astype(np.int64)]
    else:
        r, g, b = dict(rgb=(0, 1, 2), bgr=(2, 1, 0), rgba=(0, 1, 2), bgra=(2, 1, 0))[order]
        chans = [px[..., r].astype(np.int64), px[..., g].astype(np.int64), px[..., b].astype(np.int64)]
    H, W = chans[0].shape

    def gray(c):
        return c[0] if len(c) == 1 else (c[0] * 4899 + c[1] * 9617 + c[2] * 1868 + (1 << 13)) >> 14
    if dist is None:
        return gray(chans).astype(np.uint8)
    fx, fy, cx, cy = [np.float64(v) for v in K]
    k1, k2, p1, p2, k3 = [np.float64(v) for v in dist]
    v, u = np.mgrid[0:H, 0:W]
    x = (u.astype(np.float64) - cx) / fx
    y = (v.astype(np.float64) - cy) / fy
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy
    iu = np.rint((fx * xd + cx) * 32.0).astype(np.int64)
    iv = np.rint((fy * yd + cy) * 32.0).astype(np.int64)
    sx, sy, a, b = iu >> 5, iv >> 5, iu & 31, iv & 31
    out = []
    for c in chans:
        acc = np.zeros((H, W), np.int64)
        for (dy, dx, w) in ((0, 0, (32 - a) * (32 - b) * 32), (0, 1, a * (32 - b) * 32), (1, 0, (32 - a) * b * 32),
                            (1, 1, a * b * 32)):
            yy, xx = sy + dy, sx + dx
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            acc += np.where(ok, w * c[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0)
        out.append((acc + (1 << 14)) >> 15)
    return gray(out).astype(np.uint8)


def test_gray_weights_known_answers(oracle):
    """OpenCV's 8-bit RGB2GRAY of the pure primaries: R -> 76, G -> 150, B -> 29, white -> 255"""
    px = np.uint8([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [10, 20, 30]]])
    g = oracle.ingest(px, "rgb", TUM1_K, None)
    assert g.tolist() == [[76, 150, 29, 255, 0, (10 * 4899 + 20 * 9617 + 30 * 1868 + 8192) >> 14]]
    assert oracle.ingest(px[..., ::-1].copy(), "bgr", TUM1_K, None).tolist() == g.tolist()
    rgba = np.concatenate([px, np.full((1, 6, 1), 77, np.uint8)], axis=2)
    assert oracle.ingest(rgba, "rgba", TUM1_K, None).tolist() == g.tolist()  # alpha is ignored
    assert oracle.ingest(rgba[..., [2, 1, 0, 3]].copy(), "bgra", TUM1_K, None).tolist() == g.tolist()


def test_zero_distortion_is_identity(oracle):
    rng = np.random.default_rng(0)
    px = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    a = oracle.ingest(px, "rgb", TUM1_K / 10, np.zeros(5, np.float32))
    assert (a == oracle.ingest(px, "rgb", TUM1_K / 10, None)).all()
    g = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    assert (oracle.ingest(g, "gray", TUM1_K / 10, np.zeros(5, np.float32)) == g).all()


@pytest.mark.parametrize("order", ["rgb", "bgr", "rgba", "bgra", "gray"])
def test_independent_numpy_restatement(oracle, order):
    rng = np.random.default_rng(1)
    H, W = 120, 160
    ch = dict(rgb=3, bgr=3, rgba=4, bgra=4, gray=1)[order]
    px = rng.integers(0, 256, (H, W, ch) if ch > 1 else (H, W), dtype=np.uint8)
    K = TUM1_K / 4
    for dist in (TUM1_DIST, np.float32([-0.3, 0.1, 0.001, -0.002, 0.0]), None):
        got = oracle.ingest(px, order, K, dist)
        want = np_ingest(px, order, K, dist)
        assert (got == want).all(), (order, dist)


def test_undistortion_straightens_a_distorted_line(oracle):
    """geometry: render a straight vertical edge through the TUM1 distortion model (forward model, float64), ingest
    it, and the edge must come back straight (within the 1/32-pixel map quantisation + bilinear blur: 1 px)"""
    H, W = 240, 320
    K = TUM1_K / 2
    fx, fy, cx, cy = [float(v) for v in K]
    k1, k2, p1, p2, k3 = [float(v) for v in TUM1_DIST]
    # a distorted image of the scene "white where undistorted x_u >= 296": invert the model per DISTORTED pixel by
    # fixed-point iteration (cv::undistortPoints' scheme)
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    xd, yd = (u - cx) / fx, (v - cy) / fy
    x, y = xd.copy(), yd.copy()
    for _ in range(30):
        r2 = x * x + y * y
        kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (xd - dx) / kr, (yd - dy) / kr
    und_u = fx * x + cx
    img = np.where(und_u >= 296.0, 255, 0).astype(np.uint8)
    out = oracle.ingest(img, "gray", K, TUM1_DIST)
    for row in (20, 60, 120, 180, 220):
        edge = int(np.argmax(out[row] >= 128))
        assert abs(edge - 296) <= 1, (row, edge)
    # without undistortion the edge is visibly bent between the centre row and the top rows
    raw = [int(np.argmax(img[r] >= 128)) for r in (20, 120)]
    assert abs(raw[0] - raw[1]) >= 2
