"""GPU parity tests proper: the HIP engine (through the C ABI, include/sdm_c.h) against the CPU
oracle on the same seeded inputs.  Bar: BIT-EXACT for every output -- support masks, rho, sigma,
checked rho and the point set -- because both sides evaluate the same IEEE operation sequence with
FMA contraction off (the stated float tolerance of BASELINE.json is therefore 0 ulp here; the
full-size tests in test_gpu_fullsize.py state their own tolerances)."""
import os

import numpy as np
import pytest

from common import Sequence, assert_bit_equal, oracle_inter, oracle_pipeline

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seq_small(pkg, oracle):
    return Sequence(pkg, oracle, 64, 48, 8, 0x5EED0A01)


@pytest.fixture(scope="module")
def seq_mid(pkg, oracle):
    return Sequence(pkg, oracle, 160, 120, 8, 0x5EED0A02)


@pytest.fixture(scope="module")
def seq_ragged(pkg, oracle):
    # W, H not multiples of the 64x16 tile; N = 20 neighbours
    return Sequence(pkg, oracle, 200, 75, 21, 0x5EED0A03, disparity_px=1.5)


def make_engine(pkg, seq, n, **kw):
    eng = pkg.Engine(seq.W, seq.H, seq.n_kf, max_neighbours=n, **kw)
    seq.upload(eng)
    return eng


def test_device_is_gfx950(pkg, gpu_ok):
    eng = pkg.Engine(64, 48, 2)
    assert eng.arch().startswith("gfx950"), eng.arch()
    eng.close()


def test_prepass_parity(pkg, oracle, gpu_ok, seq_mid):
    """device Scharr/magnitude/phase/sigma_I == oracle pre-pass, bit for bit"""
    eng = pkg.Engine(seq_mid.W, seq_mid.H, 2)
    for k in range(2):
        eng.upload_image(k, seq_mid.im[k], seq_mid.K, seq_mid.Tcw[k])
        im, g, t, s = eng.download_inputs(k)
        assert (im == seq_mid.im[k]).all()
        assert_bit_equal(g, seq_mid.grad[k], "GradImg")
        assert_bit_equal(t, seq_mid.theta[k], "GradTheta")
        assert np.float32(s) == np.float32(seq_mid.istd[k])
    eng.close()


def test_upload_roundtrip(pkg, oracle, gpu_ok, seq_small):
    eng = make_engine(pkg, seq_small, 7)
    im, g, t, s = eng.download_inputs(3)
    assert (im == seq_small.im[3]).all()
    assert_bit_equal(g, seq_small.grad[3])
    assert_bit_equal(t, seq_small.theta[3])
    eng.close()


def test_pair_geometry_and_range(pkg, oracle, gpu_ok, seq_mid):
    eng = make_engine(pkg, seq_mid, 7)
    rng = np.random.default_rng(1)
    for (a, b) in [(3, 4), (3, 0), (0, 7), (5, 2)]:
        F, R, t = eng.pair_geometry(a, b)
        p = oracle.pair_geometry(seq_mid.okf[a], seq_mid.okf[b])
        assert_bit_equal(F, np.array(p.F12[:]), "F12")
        assert_bit_equal(R, np.array(p.R21[:]), "R21")
        assert_bit_equal(t, np.array(p.t21[:]), "t21")
        for _ in range(20):
            x, y = int(rng.integers(0, seq_mid.W)), int(rng.integers(0, seq_mid.H))
            got = eng.search_range(a, b, x, y, seq_mid.min_depth, seq_mid.max_depth)
            ref = oracle.search_range(seq_mid.okf[a], p, x, y, seq_mid.min_depth, seq_mid.max_depth)
            assert_bit_equal(np.array(got), np.array(ref), "GetSearchRange")
    eng.close()


def test_epipolar_search_pixels(pkg, oracle, gpu_ok, seq_mid):
    """per-pixel EpipolarSearch incl. rot != 0 and degenerate depth bounds"""
    eng = make_engine(pkg, seq_mid, 7)
    ys, xs = np.nonzero(seq_mid.grad[3][2:-2, 2:-2] >= 8)
    rng = np.random.default_rng(2)
    pick = rng.choice(len(xs), 60, replace=False)
    n_sup = 0
    for i in pick:
        x, y = int(xs[i]) + 2, int(ys[i]) + 2
        for nbr, rot in [(4, 0.0), (0, 0.0), (7, 3.0), (2, 350.0)]:
            got = eng.epipolar_search(3, nbr, x, y, seq_mid.min_depth, seq_mid.max_depth, rot)
            ref = oracle.epipolar_search(seq_mid.okf[3], seq_mid.okf[nbr], x, y, seq_mid.min_depth,
                                         seq_mid.max_depth, rot)
            assert got["supported"] == ref["supported"]
            for key in ("rho", "sigma", "best_u", "best_v"):
                assert_bit_equal(np.array([got[key]]), np.array([ref[key]]), "%s at %d,%d nbr %d" % (key, x, y, nbr))
            n_sup += got["supported"]
    assert n_sup > 30
    # degenerate bounds: zero-width, swapped, negative, huge
    for (mn, mx) in [(1.0, 1.0), (0.5, 2.0), (-1.0, 1.0), (1e30, 1e-30), (0.0, 0.0)]:
        got = eng.epipolar_search(3, 4, int(xs[pick[0]]) + 2, int(ys[pick[0]]) + 2, mn, mx)
        ref = oracle.epipolar_search(seq_mid.okf[3], seq_mid.okf[4], int(xs[pick[0]]) + 2, int(ys[pick[0]]) + 2, mn, mx)
        assert got["supported"] == ref["supported"]
        assert_bit_equal(np.array([got["rho"], got["sigma"]]), np.array([ref["rho"], ref["sigma"]]))
    eng.close()


def test_fuse_hypotheses(pkg, oracle, gpu_ok):
    eng = pkg.Engine(64, 48, 2, max_neighbours=32)
    rng = np.random.default_rng(3)
    cases = []
    for n in [0, 1, 3, 4, 5, 7, 12, 20, 32]:
        for _ in range(6):
            rho = (1.0 + 0.05 * rng.standard_normal(n)).astype(np.float32)
            sig = (0.02 + 0.05 * rng.random(n)).astype(np.float32)
            cases.append((rho, sig))
    # edge cases: zero sigma (0/0 NaN self test), NaN, Inf, exact ties, two equal-size clusters
    cases.append((np.float32([1, 1, 1, 1]), np.float32([0, 0, 0, 0])))
    cases.append((np.float32([1, 1, 1, 1, 1]), np.float32([0.1, 0, 0.1, 0.1, np.inf])))
    cases.append((np.float32([1, np.nan, 1, 1, 1]), np.float32([0.1, 0.1, 0.1, np.nan, 0.1])))
    cases.append((np.float32([1, 1, 1, 2, 2, 2]), np.float32([0.01] * 6)))
    cases.append((np.float32([2, 2, 2, 1, 1, 1]), np.float32([0.01] * 6)))
    cases.append((np.float32([1.0, 1.1, 1.2, 1.3, 1.4]), np.float32([0.05] * 5)))
    for rho, sig in cases:
        got = eng.fuse(rho, sig)
        ref = oracle.fuse(rho, sig)
        assert got[2] == ref[2], (rho, sig)
        assert_bit_equal(np.array(got[:2]), np.array(ref[:2]), "fusion %r %r" % (rho, sig))
    eng.close()


@pytest.mark.parametrize("which,n", [("small", 7), ("mid", 7), ("ragged", 20)])
def test_search_fuse_parity(pkg, oracle, gpu_ok, seq_small, seq_mid, seq_ragged, which, n):
    """K1 == PM.cc:197-231 on every keyframe of a sequence, batched in one call"""
    seq = dict(small=seq_small, mid=seq_mid, ragged=seq_ragged)[which]
    eng = make_engine(pkg, seq, n)
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    eng.search_fuse(refs, nbrs, seq.min_depth, seq.max_depth)
    total = 0
    for k in refs:
        r, s, st = oracle.recon_search_fuse(seq.okf[k], [seq.okf[j] for j in nbrs[k]], None, seq.min_depth, seq.max_depth)
        gr, gs = eng.download_depth(k)
        assert ((gr > 1e-6) == (r > 1e-6)).all(), "support mask"
        assert_bit_equal(gr, r, "rho kf %d" % k)
        assert_bit_equal(gs, s, "sigma kf %d" % k)
        total += st["fused"]
    assert total > 100, "scene must exercise the fusion path"
    eng.close()


def test_search_fuse_stats(pkg, oracle, gpu_ok, seq_mid):
    eng = make_engine(pkg, seq_mid, 7)
    eng.enable_stats(True)
    eng.get_stats(reset=True)
    nb = seq_mid.neighbours(3, 7)
    eng.search_fuse([3], [nb], seq_mid.min_depth, seq_mid.max_depth)
    got = eng.get_stats()
    _, _, ref = oracle.recon_search_fuse(seq_mid.okf[3], [seq_mid.okf[j] for j in nb], None, seq_mid.min_depth,
                                         seq_mid.max_depth)
    assert {k: got[k] for k in ref} == ref  # (the mask_* fields say how the ranges were walked, not what was found)
    assert got["mask_row_mismatch"] == 0
    eng.close()


def test_rot_and_params(pkg, oracle, gpu_ok, seq_mid):
    """non-zero in-plane rotation per neighbour, and non-default gates"""
    n = 7
    eng = make_engine(pkg, seq_mid, n)
    nb = seq_mid.neighbours(4, n)
    rot = np.float32([0, 2, -3, 10, 355, 180, 44])
    eng.search_fuse([4], [nb], seq_mid.min_depth, seq_mid.max_depth, rot=[rot])
    r, s, _ = oracle.recon_search_fuse(seq_mid.okf[4], [seq_mid.okf[j] for j in nb], rot, seq_mid.min_depth, seq_mid.max_depth)
    gr, gs = eng.download_depth(4)
    assert_bit_equal(gr, r)
    assert_bit_equal(gs, s)
    eng.set_params(lambdaG=12.0, lambdaL=60.0, lambdaTheta=30.0, lambdaN=2, theta_var=0.5)
    oracle.params.lambdaG, oracle.params.lambdaL, oracle.params.lambdaTheta = 12.0, 60.0, 30.0
    oracle.params.lambdaN, oracle.params.theta_var = 2, 0.5
    try:
        eng.recon([4], [nb], seq_mid.min_depth, seq_mid.max_depth)
        r, s, _ = oracle.semi_dense_recon(seq_mid.okf[4], [seq_mid.okf[j] for j in nb], None, seq_mid.min_depth,
                                          seq_mid.max_depth)
    finally:
        oracle.lib.pmo_default_params(oracle.params)
    gr, gs = eng.download_depth(4)
    assert (gr > 1e-6).sum() > 50
    assert_bit_equal(gr, r)
    assert_bit_equal(gs, s)
    eng.close()


def crafted_maps(rng, H, W, density=0.3, zero_sigma_frac=0.1, orphan_sigma_frac=0.1):
    rho = np.where(rng.random((H, W)) < density, 1.0 + 0.02 * rng.standard_normal((H, W)), 0.0).astype(np.float32)
    sig = np.where(rho > 0, 0.01 + 0.05 * rng.random((H, W)), 0.0).astype(np.float32)
    sig[(rng.random((H, W)) < zero_sigma_frac) & (rho > 0)] = 0.0          # supported pixel with sigma 0
    orphan = (rng.random((H, W)) < orphan_sigma_frac) & (rho == 0)
    # rho 0 & sigma > 0 next to wide-sigma neighbours: the only inputs for which growing is live
    sig[orphan] = (1.0 + 2.0 * rng.random((H, W)))[orphan].astype(np.float32)
    wide = (rng.random((H, W)) < 0.3) & (rho > 0)
    sig[wide] = (1.0 + 2.0 * rng.random((H, W)))[wide].astype(np.float32)
    rho[rng.random((H, W)) < 0.01] = np.nan
    rho[rng.random((H, W)) < 0.01] = -0.5
    sig[rng.random((H, W)) < 0.005] = np.inf
    return rho, sig


@pytest.mark.parametrize("shape", [(48, 64), (75, 200), (16, 64), (33, 130)])
def test_intra_check_and_grow_maps(pkg, oracle, gpu_ok, shape):
    """K2/K3 through the PM.h:85-86 signatures on crafted maps (NaN/Inf/zero-sigma/border cases,
    and rho=0 & sigma>0 pixels for which IntraKeyFrameDepthGrowing is NOT a no-op, App. A.6)"""
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    eng = pkg.Engine(W, H, 2)
    for rep in range(3):
        rho, sig = crafted_maps(rng, H, W, density=[0.3, 0.7, 0.05][rep])
        grad = (16 * rng.random((H, W))).astype(np.float32)
        r1, s1 = eng.intra_check_maps(rho, sig)
        o1, p1 = oracle.intra_check(rho, sig)
        assert_bit_equal(r1, o1, "check rho")
        assert_bit_equal(s1, p1, "check sigma")
        r2, s2 = eng.intra_grow_maps(rho, sig, grad)
        o2, p2 = oracle.intra_grow(rho, sig, grad)
        assert_bit_equal(r2, o2, "grow rho")
        assert_bit_equal(s2, p2, "grow sigma")
        assert not np.array_equal(o2.view(np.uint32), rho.view(np.uint32)), "growing must be exercised"
    eng.close()


@pytest.mark.parametrize("grow_list", ["default", "tiny"])
def test_list_kernels_on_declared_pipeline_maps(pkg, oracle, gpu_ok, seq_mid, grow_list, monkeypatch):
    """K2/K3/K4 list kernels (one thread per active-list entry) on crafted maps that are zero outside
    the active set but otherwise adversarial: NaN/Inf/zero sigma, rho~0 with sigma>0 (growing live).  K3 on lists is a
    candidate list collected by K2 (or by a detection pass) and grown by two small kernels; "tiny" = a 4-entry list, so
    the grow kernels fall back to walking the whole lists."""
    if grow_list == "tiny":
        monkeypatch.setenv("SDM_GROW_CAPACITY", "4")
    seq, n = seq_mid, 7
    eng = make_engine(pkg, seq, n)
    rng = np.random.default_rng(21)
    rho, sig = {}, {}
    for k in range(seq.n_kf):
        active = np.zeros((seq.H, seq.W), bool)
        active[2:-2, 2:-2] = seq.grad[k][2:-2, 2:-2] >= 8
        r, s = crafted_maps(rng, seq.H, seq.W, density=0.7)
        r[r > 0] = (seq.gt[k] * (1 + 0.01 * rng.standard_normal((seq.H, seq.W))).astype(np.float32))[r > 0]
        r[~active] = 0
        s[~active] = 0
        rho[k], sig[k] = r, s
        eng.upload_depth(k, r, s)
    refs = list(range(seq.n_kf))
    eng.assume_pipeline_maps(refs)
    eng.intra_check(refs)
    grown = 0
    for k in refs:
        o1, p1 = oracle.intra_check(rho[k], sig[k])
        g = eng.download_depth(k)
        assert_bit_equal(g[0], o1, "list K2 rho kf %d" % k)
        assert_bit_equal(g[1], p1, "list K2 sigma kf %d" % k)
        eng.upload_depth(k, rho[k], sig[k])
    eng.assume_pipeline_maps(refs)
    eng.intra_grow(refs)
    for k in refs:
        o2, p2 = oracle.intra_grow(rho[k], sig[k], seq.grad[k])
        g = eng.download_depth(k)
        assert_bit_equal(g[0], o2, "list K3 rho kf %d" % k)
        assert_bit_equal(g[1], p2, "list K3 sigma kf %d" % k)
        grown += int((o2.view(np.uint32) != rho[k].view(np.uint32)).sum())
        eng.upload_depth(k, rho[k], sig[k])
    assert grown > 10, "growing must be exercised by the list kernel"
    eng.assume_pipeline_maps(refs)
    nbrs = [seq.neighbours(k, n) for k in refs]
    eng.inter_check(refs, nbrs)
    for k in refs:
        ref = oracle.inter_check(seq.okf[k], rho[k], [seq.okf[j] for j in nbrs[k]], [rho[j] for j in nbrs[k]],
                                 [sig[j] for j in nbrs[k]])
        assert_bit_equal(eng.download_checked(k), ref, "list K4 kf %d" % k)
    eng.close()


def test_mixed_generic_and_list_paths(pkg, oracle, gpu_ok, seq_mid):
    """alternate pipeline maps (list kernels; zero-fill / copy passes skipped) with arbitrary uploaded
    maps (generic kernels) on the same slots: the sparse-plane bookkeeping must never leak stale data"""
    seq, n = seq_mid, 7
    eng = make_engine(pkg, seq, n)
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    rng = np.random.default_rng(33)
    maps = oracle_pipeline(oracle, seq, n)
    chk_ref, xyz_ref = oracle_inter(oracle, seq, n, maps)

    def check_pipeline(tag):
        eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
        eng.inter_check(refs, nbrs)
        eng.pointset(refs, source=1)
        for k in refs:
            r, s = eng.download_depth(k)
            assert_bit_equal(r, maps["rho"][k], tag + " rho")
            assert_bit_equal(s, maps["sigma"][k], tag + " sigma")
            assert_bit_equal(eng.download_checked(k), chk_ref[k], tag + " chk")
            assert_bit_equal(eng.download_pointset(k), xyz_ref[k], tag + " xyz")

    check_pipeline("first")
    check_pipeline("repeat (no zero-fill, no copy)")
    # arbitrary dense maps on every slot -> generic K4/K5, planes become non-sparse
    rho, sig = {}, {}
    for k in refs:
        rho[k], sig[k] = crafted_maps(rng, seq.H, seq.W, density=0.9, orphan_sigma_frac=0.0)
        rho[k][rho[k] > 0] = (seq.gt[k] * (1 + 0.01 * rng.standard_normal((seq.H, seq.W))).astype(np.float32))[rho[k] > 0]
        eng.upload_depth(k, rho[k], sig[k])
    eng.inter_check(refs, nbrs)
    eng.pointset(refs, source=1)
    eng.pointset(refs[:3], source=0)
    for k in refs:
        ref = oracle.inter_check(seq.okf[k], rho[k], [seq.okf[j] for j in nbrs[k]], [rho[j] for j in nbrs[k]],
                                 [sig[j] for j in nbrs[k]])
        assert_bit_equal(eng.download_checked(k), ref, "generic chk")
        want = oracle.pointset(seq.okf[k], rho[k] if k < 3 else ref)
        assert_bit_equal(eng.download_pointset(k), want, "generic xyz")
    # back to the pipeline: stale dense data in pool/chk/xyz must be cleared
    check_pipeline("after generic")
    check_pipeline("after generic, repeat")
    eng.close()


def test_growing_is_noop_on_pipeline_maps(pkg, oracle, gpu_ok, seq_mid):
    """after K1+K2 every rho<1e-6 pixel has sigma 0, so K3 changes nothing"""
    eng = make_engine(pkg, seq_mid, 7)
    nb = seq_mid.neighbours(3, 7)
    eng.search_fuse([3], [nb], seq_mid.min_depth, seq_mid.max_depth)
    eng.intra_check([3])
    a = eng.download_depth(3)
    eng.intra_grow([3])
    b = eng.download_depth(3)
    assert_bit_equal(a[0], b[0])
    assert_bit_equal(a[1], b[1])
    eng.close()


@pytest.mark.parametrize("which,n", [("small", 7), ("mid", 7), ("ragged", 20)])
def test_full_path_parity(pkg, oracle, gpu_ok, seq_small, seq_mid, seq_ragged, which, n):
    """SemiDenseRecon -> InterKeyFrameDepthChecking (snapshot) -> UpdateSemiDensePointSet"""
    seq = dict(small=seq_small, mid=seq_mid, ragged=seq_ragged)[which]
    eng = make_engine(pkg, seq, n, batch_capacity=3)  # forces several scratch chunks
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
    maps = oracle_pipeline(oracle, seq, n)
    for k in refs:
        gr, gs = eng.download_depth(k)
        assert_bit_equal(gr, maps["rho"][k], "recon rho kf %d" % k)
        assert_bit_equal(gs, maps["sigma"][k], "recon sigma kf %d" % k)
    eng.inter_check(refs, nbrs, commit=False)
    eng.pointset(refs, source=1)
    chk, xyz = oracle_inter(oracle, seq, n, maps)
    kept = 0
    for k in refs:
        g = eng.download_checked(k)
        assert ((g > 1e-6) == (chk[k] > 1e-6)).all()
        assert_bit_equal(g, chk[k], "checked rho kf %d" % k)
        assert_bit_equal(eng.download_pointset(k), xyz[k], "xyz kf %d" % k)
        # sigma is untouched by the inter-keyframe check (PM.cc:764)
        assert_bit_equal(eng.download_depth(k)[1], maps["sigma"][k])
        kept += int((g > 1e-6).sum())
    assert kept > 100
    eng.close()


def test_inter_check_sequential_commit(pkg, oracle, gpu_ok, seq_mid):
    """the reference's in-place, keyframe-by-keyframe order (PM.cc:262-315): later keyframes see
    the already-checked maps of earlier ones"""
    n = 7
    seq = seq_mid
    eng = make_engine(pkg, seq, n)
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
    maps = oracle_pipeline(oracle, seq, n)
    rho = {k: maps["rho"][k].copy() for k in refs}
    for k in refs:
        eng.inter_check([k], [nbrs[k]], commit=True)
        rho[k] = oracle.inter_check(seq.okf[k], rho[k], [seq.okf[j] for j in nbrs[k]], [rho[j] for j in nbrs[k]],
                                    [maps["sigma"][j] for j in nbrs[k]])
    for k in refs:
        assert_bit_equal(eng.download_depth(k)[0], rho[k], "sequential inter-check kf %d" % k)
    eng.close()


def test_inter_check_crafted(pkg, oracle, gpu_ok, seq_small):
    """crafted neighbour maps: NaN / negative / zero-sigma taps, projections leaving the image"""
    n = 7
    seq = seq_small
    eng = make_engine(pkg, seq, n)
    rng = np.random.default_rng(9)
    rho, sig = {}, {}
    for k in range(seq.n_kf):
        rho[k], sig[k] = crafted_maps(rng, seq.H, seq.W, density=0.8, orphan_sigma_frac=0.0)
        rho[k][rho[k] > 0] = (seq.gt[k] * (1 + 0.01 * rng.standard_normal((seq.H, seq.W))).astype(np.float32))[rho[k] > 0]
        eng.upload_depth(k, rho[k], sig[k])
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    eng.inter_check(refs, nbrs)
    eng.pointset(refs, source=1)
    n_kept = 0
    for k in refs:
        ref = oracle.inter_check(seq.okf[k], rho[k], [seq.okf[j] for j in nbrs[k]], [rho[j] for j in nbrs[k]],
                                 [sig[j] for j in nbrs[k]])
        got = eng.download_checked(k)
        assert_bit_equal(got, ref, "crafted inter-check kf %d" % k)
        assert_bit_equal(eng.download_pointset(k), oracle.pointset(seq.okf[k], ref), "crafted xyz")
        n_kept += int((ref > 1e-6).sum())
    assert n_kept > 200
    eng.close()


def test_pointset_pose_update(pkg, oracle, gpu_ok, seq_small):
    """UpdateAllSemiDensePointSet: re-project after kf->poseChanged (PM.cc:321-334)"""
    seq = seq_small
    eng = make_engine(pkg, seq, 7)
    rng = np.random.default_rng(4)
    rho = np.where(rng.random((seq.H, seq.W)) < 0.5, 0.5 + rng.random((seq.H, seq.W)), 0).astype(np.float32)
    eng.upload_depth(2, rho, np.zeros_like(rho))
    eng.pointset([2], source=0)
    assert_bit_equal(eng.download_pointset(2), oracle.pointset(seq.okf[2], rho))
    T2 = seq.Tcw[5]
    eng.set_pose(2, T2)
    eng.pointset([2], source=0)
    kf = oracle.keyframe(seq.im[2], seq.grad[2], seq.theta[2], seq.istd[2], seq.K, T2)
    assert_bit_equal(eng.download_pointset(2), oracle.pointset(kf, rho))
    eng.close()


def test_batch_invariance(pkg, oracle, gpu_ok, seq_mid):
    """a keyframe's result does not depend on what else is in the batch or on slot numbering"""
    n = 7
    seq = seq_mid
    eng = make_engine(pkg, seq, n)
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
    all_maps = [eng.download_depth(k) for k in refs]
    eng.recon([5], [nbrs[5]], seq.min_depth, seq.max_depth)
    one = eng.download_depth(5)
    assert_bit_equal(one[0], all_maps[5][0])
    assert_bit_equal(one[1], all_maps[5][1])
    eng.recon(refs[::-1], nbrs[::-1], seq.min_depth, seq.max_depth)
    for k in refs:
        m = eng.download_depth(k)
        assert_bit_equal(m[0], all_maps[k][0])
    eng.close()


@pytest.mark.parametrize("n", [1, 3, 4, 5, 33, 64])
def test_neighbour_count_edges(pkg, oracle, gpu_ok, n):
    """covisN from 1 (nothing can fuse: PM.cc:221 needs > 3 hypotheses) to the 64-neighbour maximum
    (validity masks beyond 32 bits, > 64 KB of LDS hypotheses)"""
    n_kf = max(n + 1, 8)
    seq = Sequence(pkg, oracle, 64, 48, n_kf, 0x5EED0B00 + n, disparity_px=0.4 if n > 8 else 2.6)
    eng = make_engine(pkg, seq, n)
    refs = [0, n_kf // 2, n_kf - 1]
    nbrs = [seq.neighbours(k, n) for k in refs]
    eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
    with pytest.raises(pkg.SdmError) as ei:  # neighbours without a depth map: the reference's gate, PM.cc:292-298
        eng.inter_check(refs, nbrs)
    assert ei.value.code == 4
    fused = 0
    for i, k in enumerate(refs):
        r, s, _ = oracle.semi_dense_recon(seq.okf[k], [seq.okf[j] for j in nbrs[i]], None, seq.min_depth, seq.max_depth)
        g = eng.download_depth(k)
        assert_bit_equal(g[0], r, "rho n=%d" % n)
        assert_bit_equal(g[1], s, "sigma n=%d" % n)
        fused += int((r > 1e-6).sum())
    if n <= 3:
        assert fused == 0
    if n >= 5:
        assert fused > 50
    eng.close()


def test_noise_images(pkg, oracle, gpu_ok):
    """adversarial i.i.d. uniform images: nearly nothing fuses, scan-only path"""
    seq = Sequence(pkg, oracle, 96, 64, 8, 0x5EED0A04, noise=True)
    eng = make_engine(pkg, seq, 7)
    refs = [2, 3]
    nbrs = [seq.neighbours(k, 7) for k in refs]
    eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
    for i, k in enumerate(refs):
        r, s, _ = oracle.semi_dense_recon(seq.okf[k], [seq.okf[j] for j in nbrs[i]], None, seq.min_depth, seq.max_depth)
        g = eng.download_depth(k)
        assert_bit_equal(g[0], r)
        assert_bit_equal(g[1], s)
    eng.close()


def test_error_behaviour(pkg, gpu_ok):
    eng = pkg.Engine(64, 48, 4, max_neighbours=3)
    with pytest.raises(pkg.SdmError) as e:
        eng.recon([0], [[1, 2, 3]], 1.25, 0.83)
    assert e.value.code == 4  # SDM_ESTATE: nothing uploaded
    with pytest.raises(pkg.SdmError) as e:
        eng.download_depth(9)
    assert e.value.code == 1
    with pytest.raises(pkg.SdmError):
        pkg.Engine(64, 48, 4, max_neighbours=65)
    with pytest.raises(pkg.SdmError):
        pkg.Engine(64, 48, 4, device=99)
    z = np.zeros((48, 64), np.uint8)
    K = np.float32([50, 50, 32, 24])
    T = np.eye(4, dtype=np.float32)[:3]
    for k in range(4):
        eng.upload_image(k, z, K, T)
    with pytest.raises(pkg.SdmError) as e:
        eng.recon([0], [[1, 2, 3, 0]], 1.25, 0.83)  # n > max_neighbours
    assert e.value.code == 1
    eng.recon([0], [[1, 2, 3]], 1.25, 0.83)  # empty image: runs, yields an all-zero map
    r, s = eng.download_depth(0)
    assert not r.any() and not s.any()
    with pytest.raises(pkg.SdmError) as e:
        eng.download_checked(0)
    assert e.value.code == 4
    eng.close()


def test_inter_check_threshold_sweep(pkg, oracle, gpu_ok):
    """the 3.84 tap test (PM.cc:709-710) swept through its threshold with ulp-dense sigma values: coincident
    cameras make every pixel project onto itself, rho_j is constant, so each neighbour tap's statistic is
    (1 - 0.98)^2 / sigma^2 with sigma stepping by one part in 2^25 around the critical value; also sigma / rho
    outside the kernel's fast window (tiny, huge, zero, Inf, NaN).  Every decision must match the oracle."""
    W, H, n_kf, n = 320, 96, 4, 3
    rng = np.random.default_rng(77)
    K = np.float32([300.0, 300.0, 159.5, 47.5])
    Tcw = np.concatenate([np.eye(3), np.zeros((3, 1))], axis=1).astype(np.float32)
    eng = pkg.Engine(W, H, n_kf, max_neighbours=n)
    okf = []
    for k in range(n_kf):
        im = rng.integers(0, 256, (H, W)).astype(np.uint8)
        eng.upload_image(k, im, K, Tcw)
        g, th, s = oracle.gradient_prepass(im)
        okf.append(oracle.keyframe(im, g, th, s, K, Tcw))
    dd = float(np.float32(1.0) - np.float32(0.98))
    s_thr = dd / np.sqrt(3.84)
    idx = np.arange(W * H, dtype=np.float64).reshape(H, W)
    rho, sig = {}, {}
    rho[0] = np.full((H, W), 1.0, np.float32)
    sig[0] = np.full((H, W), 0.01, np.float32)
    for k in range(1, n_kf):
        rho[k] = np.full((H, W), 0.98, np.float32)
        off = (k - 2) * 1000.0  # each neighbour crosses the threshold at another pixel
        sig[k] = (s_thr * (1.0 + (idx - W * H / 2 + off) * 2.0 ** -25)).astype(np.float32)
        # rows of odd operands: outside the fast window or degenerate; the reference must still be matched
        odd = np.float32([1e-5, 3e-5, 1.3e-4, 9000.0, 0.0, np.inf, np.nan, 1e-20, 1e20, -0.01])
        sig[k][5 + k, :] = np.resize(odd, W)
        rho[k][9 + k, :] = np.resize(np.float32([2e-6, 1e-4, 1.3e-4, 8000.0, 9000.0, 1e9, np.inf, np.nan, 0.98, 1.0]), W)
    for k in range(n_kf):
        eng.upload_depth(k, rho[k], sig[k])
    refs, nbrs = [0], [[1, 2, 3]]
    eng.inter_check(refs, nbrs)
    ref = oracle.inter_check(okf[0], rho[0], [okf[j] for j in nbrs[0]], [rho[j] for j in nbrs[0]],
                             [sig[j] for j in nbrs[0]])
    got = eng.download_checked(0)
    assert_bit_equal(got, ref, "threshold sweep")
    kept = int((ref > 1e-6).sum())
    assert 0.2 * W * H < kept < 0.8 * W * H, kept  # the sweep really straddles the threshold
    eng.close()


def test_flat_images_have_no_work(pkg, oracle, gpu_ok):
    """constant images: no pixel passes the gradient gate (PM.cc:201), every stage is a no-op, nothing is launched
    with an empty grid; one textured keyframe among flat neighbours finds no hypotheses either"""
    W, H, n_kf, n = 96, 64, 5, 3
    rng = np.random.default_rng(5)
    K = np.float32([120.0, 120.0, 47.5, 31.5])
    ims = [np.full((H, W), 128, np.uint8) for _ in range(n_kf)]
    ims[2] = rng.integers(0, 256, (H, W)).astype(np.uint8)
    eng = pkg.Engine(W, H, n_kf, max_neighbours=n, with_pointset=True)
    okf = []
    for k in range(n_kf):
        Tcw = np.concatenate([np.eye(3), np.float32([[0.02 * k], [0.0], [0.0]])], axis=1).astype(np.float32)
        eng.upload_image(k, ims[k], K, Tcw)
        g, th, s = oracle.gradient_prepass(ims[k])
        okf.append(oracle.keyframe(ims[k], g, th, s, K, Tcw))
    refs = list(range(n_kf))
    nbrs = [[j for j in range(n_kf) if j != k][:n] for k in refs]
    eng.recon(refs, nbrs, 0.5, 2.0)
    eng.inter_check(refs, nbrs)
    eng.pointset(refs, source=1)
    for k in refs:
        r, s, _ = oracle.semi_dense_recon(okf[k], [okf[j] for j in nbrs[k]], None, 0.5, 2.0)
        gr, gs = eng.download_depth(k)
        assert_bit_equal(gr, r, "flat rho kf %d" % k)
        assert_bit_equal(gs, s, "flat sigma kf %d" % k)
        assert not gr.any() and not eng.download_checked(k).any() and not eng.download_pointset(k).any()
    eng.close()


_SIZE_RNG = np.random.default_rng(99)
_SIZES = [(8, 8), (9, 17), (131, 67), (64, 8)] + [
    (int(_SIZE_RNG.integers(8, 300)), int(_SIZE_RNG.integers(8, 200))) for _ in range(int(os.environ.get("SDM_FUZZ_SIZES", "0")))]


@pytest.mark.parametrize("W,H", _SIZES)
def test_tiny_and_odd_sizes(pkg, oracle, gpu_ok, W, H):
    """the smallest image the engine accepts and odd sizes: whole path against the oracle (noise images so that
    the gradient gate passes; any hypotheses that survive must agree)"""
    n_kf, n = 4, 3
    rng = np.random.default_rng(W * 1000 + H)
    K = np.float32([1.2 * W, 1.2 * W, (W - 1) / 2.0, (H - 1) / 2.0])
    base = rng.integers(0, 256, (H, W + 8)).astype(np.uint8)
    eng = pkg.Engine(W, H, n_kf, max_neighbours=n, with_pointset=True)
    okf = []
    for k in range(n_kf):
        im = np.ascontiguousarray(base[:, k:k + W])  # one-pixel shifts: true matches exist
        Tcw = np.concatenate([np.eye(3), np.float32([[-k / (1.2 * W)], [0.0], [0.0]])], axis=1).astype(np.float32)
        eng.upload_image(k, im, K, Tcw)
        g, th, s = oracle.gradient_prepass(im)
        okf.append(oracle.keyframe(im, g, th, s, K, Tcw))
    refs = list(range(n_kf))
    nbrs = [[j for j in range(n_kf) if j != k][:n] for k in refs]
    eng.recon(refs, nbrs, 0.25, 4.0)
    rho, sig = {}, {}
    for k in refs:
        rho[k], sig[k], _ = oracle.semi_dense_recon(okf[k], [okf[j] for j in nbrs[k]], None, 0.25, 4.0)
        gr, gs = eng.download_depth(k)
        assert_bit_equal(gr, rho[k], "%dx%d rho kf %d" % (W, H, k))
        assert_bit_equal(gs, sig[k], "%dx%d sigma kf %d" % (W, H, k))
    eng.inter_check(refs, nbrs)
    eng.pointset(refs, source=1)
    for k in refs:
        c = oracle.inter_check(okf[k], rho[k], [okf[j] for j in nbrs[k]], [rho[j] for j in nbrs[k]], [sig[j] for j in nbrs[k]])
        assert_bit_equal(eng.download_checked(k), c, "%dx%d checked kf %d" % (W, H, k))
        assert_bit_equal(eng.download_pointset(k), oracle.pointset(okf[k], c), "%dx%d xyz kf %d" % (W, H, k))
    eng.close()


def test_fused_inter_check_pointset(pkg, oracle, gpu_ok, seq_mid):
    """sdm_inter_check_pointset == sdm_inter_check + sdm_pointset(source=1), on pipeline maps (one kernel) and on
    uploaded maps (falls back to the two passes), snapshot and commit forms; and == the oracle"""
    n = 7
    seq = seq_mid
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    out = {}
    for mode in ("separate", "fused"):
        eng = make_engine(pkg, seq, n, with_pointset=True)
        eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
        maps = {k: eng.download_depth(k) for k in refs}
        if mode == "fused":
            eng.inter_check_pointset(refs, nbrs)
        else:
            eng.inter_check(refs, nbrs)
            eng.pointset(refs, source=1)
        res = [(eng.download_checked(k), eng.download_pointset(k)) for k in refs]
        # arbitrary (uploaded) maps: the generic path
        for k in refs:
            eng.upload_depth(k, maps[k][0], maps[k][1])
        if mode == "fused":
            eng.inter_check_pointset(refs[:3], nbrs[:3], commit=True)
        else:
            eng.inter_check(refs[:3], nbrs[:3], commit=True)
            eng.pointset(refs[:3], source=1)
        res2 = [(eng.download_checked(k), eng.download_pointset(k), eng.download_depth(k)[0]) for k in refs[:3]]
        out[mode] = (res, res2, maps)
        eng.close()
    for k in refs:
        assert_bit_equal(out["fused"][0][k][0], out["separate"][0][k][0], "checked kf %d" % k)
        assert_bit_equal(out["fused"][0][k][1], out["separate"][0][k][1], "xyz kf %d" % k)
    for k in range(3):
        for i, what in enumerate(("checked", "xyz", "committed rho")):
            assert_bit_equal(out["fused"][1][k][i], out["separate"][1][k][i], "uploaded maps, %s kf %d" % (what, k))
    maps = out["fused"][2]
    for k in refs[:4]:
        c = oracle.inter_check(seq.okf[k], maps[k][0], [seq.okf[j] for j in nbrs[k]], [maps[j][0] for j in nbrs[k]],
                               [maps[j][1] for j in nbrs[k]])
        assert_bit_equal(out["fused"][0][k][0], c, "fused checked vs oracle kf %d" % k)
        assert_bit_equal(out["fused"][0][k][1], oracle.pointset(seq.okf[k], c), "fused xyz vs oracle kf %d" % k)
    assert sum(int((r[0] > 1e-6).sum()) for r in out["fused"][0]) > 500


def test_table_cache_alternating_calls(pkg, oracle, gpu_ok, seq_mid):
    """the engine keeps several staged table sets keyed by the call's arguments: six different calls in rotation
    (more than there are sets), with pose changes, an in-plane-rotation change and a parameter change in between,
    must give exactly what a fresh engine gives for the same call"""
    seq, n = seq_mid, 4
    rng = np.random.default_rng(3)
    eng = make_engine(pkg, seq, n, with_pointset=True)
    all_k = list(range(seq.n_kf))
    eng.mark_depth_present(all_k)  # the (still zero) maps of not-yet-reconstructed neighbours are read on purpose
    groups = [all_k[:3], all_k[3:6], all_k[1:5], all_k[::2], all_k[5:], all_k[2:4]]
    poses = {k: seq.Tcw[k].copy() for k in all_k}
    lam = 8.0

    def fresh(refs, nbrs, rot, mind, maxd, state):
        e2 = pkg.Engine(seq.W, seq.H, seq.n_kf, max_neighbours=n, with_pointset=True)
        for k in all_k:
            e2.upload_image(k, seq.im[k], seq.K, poses[k])
            if k not in refs:  # the inter-keyframe check reads the neighbours' current maps
                e2.upload_depth(k, state[k][0], state[k][1])
        e2.set_params(lambdaG=lam)
        e2.recon(refs, nbrs, mind, maxd, rot=rot)
        e2.inter_check_pointset(refs, nbrs)
        out = [(e2.download_depth(k), e2.download_checked(k), e2.download_pointset(k)) for k in refs]
        e2.close()
        return out

    for it in range(14):
        refs = groups[it % len(groups)]
        nbrs = [[j for j in all_k if j != k][(it // 6):(it // 6) + n] for k in refs]
        rot = np.zeros((len(refs), n), np.float32)
        if it % 4 == 3:
            rot[0, 0] = 7.5  # same slots, other constants: must not hit a cached set
        mind, maxd = seq.min_depth * (1.0 + 0.01 * (it % 3)), seq.max_depth
        if it == 5:  # a pose moves (bundle adjustment): every cached table is stale
            poses[2] = poses[2].copy()
            poses[2][0, 3] += np.float32(0.003)
            eng.set_pose(2, poses[2])
        if it == 9:
            lam = 12.0
            eng.set_params(lambdaG=lam)
        check = it in (0, 4, 5, 6, 7, 9, 10, 13)  # a fresh engine is slow to set up: check a subset of the rotation
        state = {k: eng.download_depth(k) for k in all_k} if check else None
        eng.recon(refs, nbrs, mind, maxd, rot=rot)
        eng.inter_check_pointset(refs, nbrs)
        got = [(eng.download_depth(k), eng.download_checked(k), eng.download_pointset(k)) for k in refs]
        if check:
            want = fresh(refs, nbrs, rot, mind, maxd, state)
            for (g, w, k) in zip(got, want, refs):
                assert_bit_equal(g[0][0], w[0][0], "it %d rho kf %d" % (it, k))
                assert_bit_equal(g[0][1], w[0][1], "it %d sigma kf %d" % (it, k))
                assert_bit_equal(g[1], w[1], "it %d checked kf %d" % (it, k))
                assert_bit_equal(g[2], w[2], "it %d xyz kf %d" % (it, k))
    eng.close()


@pytest.mark.parametrize("open_list", ["default", "defer-all", "tiny", "in-place"])
def test_search_fuse_with_outlier_hypotheses(pkg, oracle, gpu_ok, seq_mid, monkeypatch, open_list):
    """K1's fusion takes a shortcut when the FIRST accepted hypothesis is compatible with all the others (its set is
    then the first largest one, PM.cc:616), a second one when only a few hypotheses lie outside that set and none of
    them is compatible with a member or has a larger set, and hands every other pixel to the all-pairs count
    (k_fuse_open: 64 open pixels per workgroup).  Neighbours with a wrong pose produce consistent-looking but wrong
    hypotheses (outliers): placed first, in the middle and last in the neighbour order they exercise all three paths
    next to each other, against the oracle.  The open-pixel list has four regimes, all with the same result: the first
    512 reservations of a launch and workgroups with >= 40 open pixels are counted in place (default: a mix here),
    "defer-all" hands every open pixel to k_fuse_open, "tiny" gives the list one block only so that nearly every
    workgroup falls back to counting in place, "in-place" makes every workgroup with an open pixel count in place."""
    import sys
    if open_list == "defer-all":
        monkeypatch.setenv("SDM_OPEN_QUOTA", "0")
        monkeypatch.setenv("SDM_OPEN_INPLACE", "65")
    if open_list == "tiny":
        monkeypatch.setenv("SDM_OPEN_QUOTA", "0")
        monkeypatch.setenv("SDM_OPEN_INPLACE", "65")
        monkeypatch.setenv("SDM_OPEN_CAPACITY", "64")
    if open_list == "in-place":
        monkeypatch.setenv("SDM_OPEN_QUOTA", "0")
        monkeypatch.setenv("SDM_OPEN_INPLACE", "1")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import np_pm
    seq, n = seq_mid, 7
    eng = pkg.Engine(seq.W, seq.H, seq.n_kf + 2, max_neighbours=n)
    seq.upload(eng)
    # two extra slots: keyframes 2 and 5 again, but with their baseline to the others stretched by 40 % / shrunk by 30 %
    bad = {}
    for slot, (k, scale) in zip((seq.n_kf, seq.n_kf + 1), ((2, 1.4), (5, 0.7))):
        T = seq.Tcw[k].copy()
        T[:, 3] = T[:, 3] * np.float32(scale)
        eng.upload_keyframe(slot, seq.im[k], seq.grad[k], seq.theta[k], seq.istd[k], seq.K, T)
        bad[slot] = oracle.keyframe(seq.im[k], seq.grad[k], seq.theta[k], seq.istd[k], seq.K, T)
    okf = lambda s: bad[s] if s in bad else seq.okf[s]
    b0, b1 = seq.n_kf, seq.n_kf + 1
    cases = {3: [b0, 4, 1, 6, 0, 7, 5],      # an outlier source FIRST: the shortcut must not fire on its pixels
             4: [3, 6, 1, b0, 7, 0, b1],     # outliers in the middle and last
             6: [7, 4, 3, 1, 0, 2, 5]}       # clean
    refs = list(cases)
    eng.search_fuse(refs, [cases[k] for k in refs], seq.min_depth, seq.max_depth)
    open_px = 0
    for k in refs:
        r, s, st = oracle.recon_search_fuse(seq.okf[k], [okf(j) for j in cases[k]], None, seq.min_depth, seq.max_depth)
        gr, gs = eng.download_depth(k)
        assert_bit_equal(gr, r, "rho kf %d" % k)
        assert_bit_equal(gs, s, "sigma kf %d" % k)
        assert st["fused"] > 500
        if k != 6:  # how many pixels really needed the all-pairs path (second restatement, tests/np_pm.py)
            H, W = seq.H, seq.W
            ys, xs = np.nonzero(~(seq.grad[k][2:H - 2, 2:W - 2] < 8))
            ys, xs = ys + 2, xs + 2
            ref = np_pm.KF(seq.im[k], seq.grad[k], seq.theta[k], seq.istd[k], seq.K, seq.Tcw[k])
            R = np.zeros((len(xs), n), np.float32)
            S = np.ones((len(xs), n), np.float32)
            V = np.zeros((len(xs), n), bool)
            for j, sl in enumerate(cases[k]):
                src = 2 if sl == b0 else 5 if sl == b1 else sl
                nb = np_pm.KF(seq.im[src], seq.grad[src], seq.theta[src], seq.istd[src], seq.K, np.array(okf(sl).Tcw[:]).reshape(3, 4))
                rr, ss, sup, _ = np_pm.epipolar_search(ref, nb, np_pm.Pair(ref, nb), xs, ys, seq.min_depth, seq.max_depth)
                with np.errstate(all="ignore"):
                    ok = sup & ((np.float32(1) / rr) > 0)
                R[:, j], S[:, j], V[:, j] = np.where(ok, rr, 0), np.where(ok, ss, 1), ok
            comp = np_pm.chi_matrix(R[:, :, None], R[:, None, :], S[:, :, None], S[:, None, :]) & V[:, :, None] & V[:, None, :]
            nh = V.sum(1)
            first = V.argmax(1)
            full = comp[np.arange(len(xs)), first].sum(1) == nh
            open_px += int(((nh > 3) & ~full).sum())
            assert int(((nh > 3) & full).sum()) > 100, "some pixels must take the shortcut as well"
    assert open_px > 300, "the outlier neighbours must force the all-pairs path on many pixels (%d)" % open_px
    eng.close()


def test_table_cache_serves_the_calls_of_a_step(pkg, oracle, gpu_ok, seq_mid):
    """K1 -> K4 -> K5 of one step, and the same step again, find their slot / constant tables in a cached set (a step that
    re-staged them per call would still be right, only slower: this pins the host-side cache)"""
    n = 7
    eng = make_engine(pkg, seq_mid, n)
    refs = list(range(seq_mid.n_kf))
    nbrs = [seq_mid.neighbours(k, n) for k in refs]
    eng.get_stats(reset=True)
    for _ in range(3):
        eng.recon(refs, nbrs, seq_mid.min_depth, seq_mid.max_depth)
        eng.inter_check_pointset(refs, nbrs, commit=False)
        eng.pointset(refs, source=1)
    assert eng.get_stats()["table_stagings"] == 1
    eng.set_pose(2, seq_mid.Tcw[3])  # a pose change invalidates the cached constants: one more staging, then hits again
    for _ in range(2):
        eng.recon(refs, nbrs, seq_mid.min_depth, seq_mid.max_depth)
        eng.inter_check_pointset(refs, nbrs, commit=False)
    assert eng.get_stats()["table_stagings"] == 1
    eng.close()
