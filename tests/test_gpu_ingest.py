"""GPU: sdm_upload_image_rgb (device colour->gray + lens undistortion feeding the gradient pre-pass; SURVEY.md §8f-1,
src/Tracking.cc:244-257, 266-271, src/Modeler/Modeler.cc:154-155) against the oracle's restatement, bit for bit --
the gray image, the derived GradImg / GradTheta / I_stddev, and the whole path run from colour frames.
PARITY UNPINNED for the OpenCV pieces (absent from the image): both sides state the same published algorithm."""
import ctypes as C

import numpy as np
import pytest

from common import Sequence, assert_bit_equal, oracle_inter, oracle_pipeline
from test_oracle_ingest import TUM1_DIST, TUM1_K

pytestmark = pytest.mark.gpu


def _fp(a):
    a = np.ascontiguousarray(a, np.float32)
    _fp.keep = a
    return a.ctypes.data_as(C.POINTER(C.c_float))


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


def np_active_list(grad, lambdaG=8.0):
    """PM.cc:198-201 restated: inset pixels with GradImg >= lambdaG, raster order, as (y << 16 | x)"""
    H, W = grad.shape
    m = ~(grad < np.float32(lambdaG))
    m[:2] = m[-2:] = False
    m[:, :2] = m[:, -2:] = False
    ys, xs = np.nonzero(m)
    return (ys.astype(np.uint32) << 16) | xs.astype(np.uint32)


def np_list_hash(lst):
    """the compact wire header's hash (csrc/sdm_ingest.h seg_hash_term), restated with plain integers: per 64-pixel row segment
    with listed pixels, SplitMix64's finaliser of mask ^ ((y << 16 | x0) * golden), summed mod 2^64"""
    M = (1 << 64) - 1
    segs = {}
    for v in lst.tolist():
        segs[v & ~63] = segs.get(v & ~63, 0) | (1 << (v & 63))
    tot = 0
    for key, mask in segs.items():
        z = mask ^ ((key * 0x9E3779B97F4A7C15) & M)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        tot = (tot + (z ^ (z >> 31))) & M
    return tot


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("W,H,n_kf", [(160, 120, 8), (75, 53, 5), (640, 480, 70)])
def test_batch_upload_equals_single_uploads(pkg, oracle, gpu_ok, W, H, n_kf, overlap):
    """sdm_upload_images_batch (one launch of each pre-pass kernel over (keyframe, tile), chunked, H2D on the upload
    stream; pageable AND pinned sources, non-consecutive slots) against one sdm_upload_image call per keyframe and
    against the oracle's pre-pass: records, I_stddev, the active-pixel lists and their hashes, bit for bit.
    overlap: the streaming ingest's launch shape (one pre-pass launch per group of four chunks; 67 keyframes = five chunks =
    a full group and a one-chunk group)."""
    rng = np.random.default_rng(W * 31 + n_kf)
    yy, xx = np.mgrid[0:H, 0:W]
    ims = [np.clip(127 + 90 * np.sin(xx / (5.0 + k)) * np.cos(yy / 4.0) + rng.integers(-25, 25, (H, W)), 0, 255).astype(np.uint8)
           for k in range(n_kf)]
    K = TUM1_K * np.float32(W / 640.0)
    poses = [EYE + np.float32(0.01 * k) * np.float32([[0, 0, 0, 1], [0, 0, 0, 0], [0, 0, 0, 0]]) for k in range(n_kf)]
    n_slots = 2 * n_kf + 3
    one, bat = pkg.Engine(W, H, n_slots), pkg.Engine(W, H, n_slots)
    slots = [(7 * k + 3) % n_slots for k in range(n_kf)]  # scattered, not consecutive
    assert len(set(slots)) == n_kf
    for k in range(n_kf):
        one.upload_image(slots[k], ims[k], K, poses[k])
    # dirty planes first: the batch path clears depth / checked / point planes inside its pre-pass kernel
    for s in slots[:3]:
        bat.upload_image(s, ims[0], K, EYE)
        bat.upload_depth(s, np.full((H, W), 0.7, np.float32), np.full((H, W), 0.1, np.float32))
        bat.pointset([s], source=0)  # every pixel of the point-set plane non-zero
        assert bat.download_pointset(s).any()
    pinned = [bat.host_alloc((H, W)) for _ in range(n_kf)]
    for k in range(n_kf):
        pinned[k][...] = ims[k]
    bat.set_ingest_overlap(overlap)
    half = min(3, n_kf // 2) if overlap else n_kf // 2
    bat.upload_images_batch(slots[:half], pinned[:half], K, poses[:half])      # read in place by the copy engine
    bat.upload_images_batch(slots[half:], ims[half:], K, poses[half:])         # pageable: through the pinned ring
    for k in list(range(min(n_kf, 6))) + [n_kf - 1]:
        a, b = one.download_inputs(slots[k]), bat.download_inputs(slots[k])
        assert (a[0] == b[0]).all() and (a[0] == ims[k]).all()
        assert_bit_equal(a[1], b[1], "GradImg kf %d" % k)
        assert_bit_equal(a[2], b[2], "GradTheta kf %d" % k)
        assert np.float32(a[3]) == np.float32(b[3])
        wg, wt, ws = oracle.gradient_prepass(ims[k])
        assert_bit_equal(b[1], wg, "GradImg vs oracle kf %d" % k)
        assert_bit_equal(b[2], wt, "GradTheta vs oracle kf %d" % k)
        assert np.float32(b[3]) == np.float32(ws)
        la, ha = one.active_list(slots[k])
        lb, hb = bat.active_list(slots[k])
        want = np_active_list(wg)
        assert (la == want).all() and (lb == want).all() and la.size == want.size
        assert ha == hb == np_list_hash(want)
        r, s = bat.download_depth(slots[k])
        assert not r.any() and not s.any(), "a new keyframe starts with zero maps"
        assert not bat.download_pointset(slots[k]).any()
    # lists follow lambdaG: rebuilt in one batch from the records
    one.set_params(lambdaG=14.0)
    bat.set_params(lambdaG=14.0)
    for k in (0, n_kf - 1):
        want = np_active_list(oracle.gradient_prepass(ims[k])[0], 14.0)
        for e in (one, bat):
            l, h = e.active_list(slots[k])
            assert (l == want).all() and h == np_list_hash(want)
    assert bat.lib.sdm_upload_images_batch(bat.ctx, 2, (C.c_int * 2)(1, 1), (C.c_void_p * 2)(pinned[0].ctypes.data, pinned[1].ctypes.data),
                                           _fp(np.tile(K, 2)), _fp(np.tile(EYE.reshape(12), 2))) == 1  # duplicate slot
    for p in pinned:
        bat.host_free(p)
    one.close()
    bat.close()


def test_rgb_batch_equals_single_uploads(pkg, oracle, gpu_ok):
    W, H, n_kf = 160, 120, 9
    rng = np.random.default_rng(5)
    K = TUM1_K * np.float32(W / 640.0)
    frames = [rng.integers(0, 255, (H, W, 3)).astype(np.uint8) for _ in range(n_kf)]
    poses = [EYE] * n_kf
    one, bat = pkg.Engine(W, H, n_kf), pkg.Engine(W, H, n_kf)
    for dist in (TUM1_DIST, None):
        for k in range(n_kf):
            one.upload_image_rgb(k, frames[k], "bgr", K, dist, EYE)
        bat.upload_images_rgb_batch(list(range(n_kf)), frames, "bgr", K, dist, poses)
        for k in range(n_kf):
            a, b = one.download_inputs(k), bat.download_inputs(k)
            assert (a[0] == b[0]).all() and (b[0] == oracle.ingest(frames[k], "bgr", K, dist)).all()
            assert_bit_equal(a[1], b[1])
            assert_bit_equal(a[2], b[2])
            assert np.float32(a[3]) == np.float32(b[3])
            assert one.active_list(k)[1] == bat.active_list(k)[1]
    one.close()
    bat.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_streaming_order_equals_serial_order(pkg, oracle, gpu_ok, overlap):
    """blocks of keyframes arrive continuously (src/Tracking.cc:266-271): block i is stepped while block i+1 is uploaded
    into the other half of the slot pool (bench.py's value_streaming).  Every block's maps equal the ones an engine gets
    that uploads and steps the same block on its own."""
    W, H, n_blk, n = 160, 120, 8, 5
    seqs = [Sequence(pkg, oracle, W, H, n_blk, 0x5EED0E00 + b, disparity_px=2.6 + b) for b in range(4)]
    refs = list(range(n_blk))
    nbrs = [seqs[0].neighbours(k, n) for k in refs]

    def serial(seq):
        eng = pkg.Engine(W, H, n_blk, max_neighbours=n)
        eng.upload_images_batch(refs, seq.im, seq.K, seq.Tcw)
        eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
        eng.inter_check_pointset(refs, nbrs, commit=False)
        out = [(eng.download_depth(k), eng.download_checked(k), eng.download_pointset(k)) for k in refs]
        eng.close()
        return out

    want = [serial(s) for s in seqs]
    eng = pkg.Engine(W, H, 2 * n_blk, max_neighbours=n)
    eng.set_ingest_overlap(overlap)  # (8 keyframes of 160x120 go through in four chunks: the overlapped path is taken)
    half = lambda b: [k + (b & 1) * n_blk for k in refs]
    hn = lambda b: [[j + (b & 1) * n_blk for j in row] for row in nbrs]
    got = {}
    eng.upload_images_batch(half(0), seqs[0].im, seqs[0].K, seqs[0].Tcw)
    for b in range(len(seqs)):
        eng.recon(half(b), hn(b), seqs[b].min_depth, seqs[b].max_depth)         # queued, not awaited ...
        eng.inter_check_pointset(half(b), hn(b), commit=False)
        if b + 1 < len(seqs):                                                   # ... while the next block comes in
            eng.upload_images_batch(half(b + 1), seqs[b + 1].im, seqs[b + 1].K, seqs[b + 1].Tcw)
        # block b's results are read back after block b+1's upload was queued: the upload went to the OTHER half
        got[b] = [(eng.download_depth(k), eng.download_checked(k), eng.download_pointset(k)) for k in half(b)]
    # the bench's pattern: uploads run TWO blocks ahead, into the half whose block has just been queued for its step (the
    # engine orders them behind that step); results are read back one block late
    eng.synchronize()
    seqs2 = seqs + seqs[:2]
    got2 = {}
    eng.upload_images_batch(half(0), seqs2[0].im, seqs2[0].K, seqs2[0].Tcw)
    eng.upload_images_batch(half(1), seqs2[1].im, seqs2[1].K, seqs2[1].Tcw)
    out_dev = {}
    for b in range(len(seqs2)):
        eng.recon(half(b), hn(b), seqs2[b].min_depth, seqs2[b].max_depth)
        eng.inter_check_pointset(half(b), hn(b), commit=False)
        got2[b] = [(eng.download_depth(k), eng.download_checked(k), eng.download_pointset(k)) for k in half(b)] if not overlap else None
        if overlap:
            # (a download would drain the stream: keep the pipeline asynchronous and compare through a device-side copy)
            out_dev[b] = None
        if b + 2 < len(seqs2):
            eng.upload_images_batch(half(b), seqs2[b + 2].im, seqs2[b + 2].K, seqs2[b + 2].Tcw)
        if overlap and b + 2 >= len(seqs2):
            got2[b] = [(eng.download_depth(k), eng.download_checked(k), eng.download_pointset(k)) for k in half(b)]
    for b, val in got2.items():
        if val is None:
            continue
        w = want[b % len(seqs)]
        for i, k in enumerate(refs):
            assert_bit_equal(val[i][0][0], w[i][0][0], "two-ahead block %d rho kf %d" % (b, k))
            assert_bit_equal(val[i][1], w[i][1], "two-ahead block %d checked kf %d" % (b, k))
            assert_bit_equal(val[i][2], w[i][2], "two-ahead block %d xyz kf %d" % (b, k))
    for b in range(len(seqs)):
        for i, k in enumerate(refs):
            assert_bit_equal(got[b][i][0][0], want[b][i][0][0], "block %d rho kf %d" % (b, k))
            assert_bit_equal(got[b][i][0][1], want[b][i][0][1], "block %d sigma kf %d" % (b, k))
            assert_bit_equal(got[b][i][1], want[b][i][1], "block %d checked kf %d" % (b, k))
            assert_bit_equal(got[b][i][2], want[b][i][2], "block %d xyz kf %d" % (b, k))
    eng.close()


def test_overlapped_ingest_into_slots_in_use(pkg, oracle, gpu_ok):
    """overlapped ingest must still be ordered behind compute calls that READ the slots it overwrites: a block is re-uploaded
    with different images into the very slots a queued reconstruction uses; the reconstruction sees the old images, the next
    one the new ones"""
    W, H, n_blk, n = 160, 120, 8, 5
    a = Sequence(pkg, oracle, W, H, n_blk, 0x5EED0E10)
    b = Sequence(pkg, oracle, W, H, n_blk, 0x5EED0E11, disparity_px=4.0)
    refs = list(range(n_blk))
    nbrs = [a.neighbours(k, n) for k in refs]
    eng = pkg.Engine(W, H, n_blk, max_neighbours=n)
    eng.set_ingest_overlap(True)
    eng.upload_images_batch(refs, a.im, a.K, a.Tcw)
    for rep in range(3):
        eng.recon(refs, nbrs, a.min_depth, a.max_depth)          # queued on the compute stream ...
        eng.upload_images_batch(refs, b.im, b.K, b.Tcw)           # ... and its inputs overwritten right behind it
        eng.recon(refs, nbrs, b.min_depth, b.max_depth)
        got_b = [eng.download_depth(k) for k in refs]
        eng.upload_images_batch(refs, a.im, a.K, a.Tcw)
        eng.recon(refs, nbrs, a.min_depth, a.max_depth)
        got_a = [eng.download_depth(k) for k in refs]
        for k in refs:
            for seq, got in ((a, got_a), (b, got_b)):
                r, s, _ = oracle.semi_dense_recon(seq.okf[k], [seq.okf[j] for j in nbrs[k]], None, seq.min_depth, seq.max_depth)
                assert_bit_equal(got[k][0], r, "rep %d rho kf %d" % (rep, k))
                assert_bit_equal(got[k][1], s, "rep %d sigma kf %d" % (rep, k))
    eng.close()


_INGEST_SIZES = int(__import__("os").environ.get("SDM_FUZZ_INGEST", "8"))


def test_prepass_random_sizes(pkg, oracle, gpu_ok):
    """The batched pre-pass on random image sizes (widths that are no multiple of the 64-pixel tile, images smaller than a
    tile), random images (flat patches, saturated patches, noise), both launch shapes, into slots whose planes are dirty:
    records, I_stddev, list, hash against the oracle's pre-pass; maps, checked plane and point set start as zeros.
    SDM_FUZZ_INGEST = number of cases (deep runs: tools/run_deepfuzz_r05.sh)."""
    rng = np.random.default_rng(20261005)
    for case in range(_INGEST_SIZES):
        W, H = int(rng.integers(8, 300)), int(rng.integers(8, 200))  # (the engine takes 8 x 8 and up)
        n_kf = int(rng.integers(1, 10))
        ims = []
        for k in range(n_kf):
            im = rng.integers(0, 256, (H, W)).astype(np.uint8)
            kind = int(rng.integers(0, 4))
            if kind == 0:  # smooth ramp + a little noise: the usual 20 %-dense list
                yy, xx = np.mgrid[0:H, 0:W]
                im = np.clip(127 + 100 * np.sin(xx / 7.0 + k) * np.cos(yy / 5.0) + rng.integers(-6, 6, (H, W)), 0, 255).astype(np.uint8)
            elif kind == 1:  # flat and saturated patches (zero gradients, maximal gradients)
                im[: H // 2, : W // 2] = int(rng.integers(0, 256))
                im[H // 2:, W // 2:] = 255 * (np.indices((H - H // 2, W - W // 2)).sum(0) & 1)
            elif kind == 2:
                im[...] = int(rng.integers(0, 256))  # nothing listed
            ims.append(im)
        K = TUM1_K * np.float32(W / 640.0)
        poses = [EYE] * n_kf
        eng = pkg.Engine(W, H, n_kf + 2)
        overlap = bool(case & 1)
        eng.set_ingest_overlap(overlap)
        slots = list(rng.permutation(n_kf + 2)[:n_kf])
        dirty = slots[0]
        eng.upload_image(dirty, ims[-1], K, EYE)
        eng.upload_depth(dirty, np.full((H, W), 0.7, np.float32), np.full((H, W), 0.1, np.float32))
        eng.pointset([dirty], source=0)
        eng.upload_images_batch(slots, ims, K, poses)
        for k in range(n_kf):
            im_b, g, t, sd = eng.download_inputs(slots[k])
            wg, wt, ws = oracle.gradient_prepass(ims[k])
            tag = "case %d (%dx%d, %d kf, overlap %d) kf %d" % (case, W, H, n_kf, overlap, k)
            assert (im_b == ims[k]).all(), tag
            assert_bit_equal(g, wg, "GradImg " + tag)
            assert_bit_equal(t, wt, "GradTheta " + tag)
            assert np.float32(sd) == np.float32(ws), tag
            lst, h = eng.active_list(slots[k])
            want = np_active_list(wg)
            assert lst.size == want.size and (lst == want).all(), tag
            assert h == np_list_hash(want) == pkg.shard.list_hash(want), tag
            r, s = eng.download_depth(slots[k])
            assert not r.any() and not s.any(), tag
            assert not eng.download_pointset(slots[k]).any(), tag
        eng.close()
