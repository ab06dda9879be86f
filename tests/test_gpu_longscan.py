"""GPU: K1's two scans of the candidate range (PM.cc:405-443) -- the batched record scan and the scan over the neighbour's
gradient-gate bit plane (sdm_device.h scan_masked) -- against the oracle on LONG ranges: wide depth priors (s = 0.3 mu and
wider), long baselines, a whole-row range, negative / huge priors, steep lines.  Every mode must give the oracle's maps bit for
bit AND walk the oracle's number of candidates; the mask scan's row-run self-check must stay at zero."""
import numpy as np
import pytest

from common import Sequence, assert_bit_equal

pytestmark = pytest.mark.gpu


def prior(mu, s):
    """StereoSearchConstraints' two formulas, PM.cc:381-382, in float"""
    mu, s = np.float32(mu), np.float32(s)
    return float(np.float32(1) / (mu - np.float32(2) * s)), float(np.float32(1) / (mu + np.float32(2) * s))


CASES = [
    # (name, W, H, disparity px, (min_depth, max_depth), expected mean candidates at least)
    ("spread0.3", 256, 128, 10.0, prior(1.0, 0.3), 25),
    ("spread0.45_long_baseline", 384, 96, 14.0, prior(1.0, 0.45), 55),
    ("whole_row", 160, 120, 4.0, (1e-3, 1e3), 60),
    ("negative_prior", 160, 120, 4.0, (-1.0, 1.0), 20),
    ("short", 160, 120, 2.6, prior(1.0, 0.1), 0),
]


def run_mode(pkg, seq, n, mind, maxd, mode, rots=None):
    eng = pkg.Engine(seq.W, seq.H, seq.n_kf, max_neighbours=n)
    seq.upload(eng, device_prepass=True)
    eng.set_scan_mode(mode)
    eng.enable_stats(True)
    eng.get_stats(reset=True)
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    eng.search_fuse(refs, nbrs, mind, maxd, rot=rots)
    st = eng.get_stats()
    maps = [eng.download_depth(k) for k in refs]
    eng.close()
    return maps, st


@pytest.mark.parametrize("name,W,H,disp,pr,min_mean", CASES, ids=[c[0] for c in CASES])
def test_long_ranges_all_scan_modes(pkg, oracle, gpu_ok, name, W, H, disp, pr, min_mean):
    n_kf, n = 9, 7
    seq = Sequence(pkg, oracle, W, H, n_kf, 0x5EED0A00 + len(name), disparity_px=disp)
    mind, maxd = pr
    want, tot = [], dict(searches=0, candidates=0, gate_pass=0)
    for k in range(n_kf):
        nb = seq.neighbours(k, n)
        r, s, st = oracle.recon_search_fuse(seq.okf[k], [seq.okf[j] for j in nb], None, mind, maxd)
        want.append((r, s))
        for key in tot:
            tot[key] += st[key]
    mean = tot["candidates"] / max(tot["searches"], 1)
    assert mean >= min_mean, "the case must exercise long ranges (mean %.1f candidates per search)" % mean
    for mode in (2, 0, 1):
        maps, st = run_mode(pkg, seq, n, mind, maxd, mode)
        for k in range(n_kf):
            assert_bit_equal(maps[k][0], want[k][0], "%s mode %d rho kf %d" % (name, mode, k))
            assert_bit_equal(maps[k][1], want[k][1], "%s mode %d sigma kf %d" % (name, mode, k))
        assert st["searches"] == tot["searches"], (name, mode)
        assert st["candidates"] == tot["candidates"], (name, mode, st["candidates"], tot["candidates"])
        assert st["gate_pass"] == tot["gate_pass"], (name, mode)
        assert st["mask_row_mismatch"] == 0, (name, mode)
        if mode == 2:
            assert st["mask_waves"] > 0
        if mode == 1:
            assert st["mask_waves"] == 0
        if mode == 0 and min_mean >= 50:
            assert st["mask_waves"] > 0, "long ranges must reach the mask scan in the default mode"
        if mode == 0 and min_mean == 0:
            assert st["mask_waves"] == 0, "short ranges stay on the batched scan"
    assert sum(int((w[0] > 1e-6).sum()) for w in want) > 200, "the case must still fuse something"


def test_mask_scan_steep_lines_and_rotations(pkg, oracle, gpu_ok):
    """forced mask scan on geometry it would not choose: rolled and pitched cameras (steep epipolar lines: a row run of one
    or two columns), in-plane rotations, a prior that spans the whole row"""
    from test_gpu_fuzz import make_case
    rng = np.random.default_rng(77)
    W, H, n_kf, n = 96, 80, 6, 5
    for mode_name in ("large", "vertical", "forward"):
        case = make_case(rng, oracle, W, H, n_kf, mode_name)
        refs = list(range(n_kf))
        nbrs = [[j for j in range(n_kf) if j != k][:n] for k in refs]
        rots = rng.uniform(-30, 390, (n_kf, n)).astype(np.float32)
        for mind, maxd in ((1e-3, 1e3), (-1.0, 1.0), (2.5, 0.625)):
            got = {}
            for mode in (2, 1):
                eng = pkg.Engine(W, H, n_kf, max_neighbours=n)
                for k in range(n_kf):
                    eng.upload_image(k, case["im"][k], case["K"][k], case["Tcw"][k])
                eng.set_scan_mode(mode)
                eng.enable_stats(True)
                eng.get_stats(reset=True)
                eng.search_fuse(refs, nbrs, mind, maxd, rot=rots)
                st = eng.get_stats()
                got[mode] = ([eng.download_depth(k) for k in refs], st)
                eng.close()
            cands = 0
            for k in refs:
                r, s, st = oracle.recon_search_fuse(case["okf"][k], [case["okf"][j] for j in nbrs[k]], rots[k], mind, maxd)
                cands += st["candidates"]
                for mode in (2, 1):
                    assert_bit_equal(got[mode][0][k][0], r, "%s mode %d rho kf %d" % (mode_name, mode, k))
                    assert_bit_equal(got[mode][0][k][1], s, "%s mode %d sigma kf %d" % (mode_name, mode, k))
            for mode in (2, 1):
                assert got[mode][1]["candidates"] == cands, (mode_name, mode, mind, maxd)
                assert got[mode][1]["mask_row_mismatch"] == 0


def test_mask_plane_follows_lambdaG(pkg, oracle, gpu_ok):
    """sdm_set_params changes lambdaG after the keyframes were uploaded: the neighbours' bit planes are rebuilt before K1
    reads them (a stale plane built under a LARGER lambdaG would hide candidates)"""
    W, H, n_kf, n = 160, 120, 8, 7
    seq = Sequence(pkg, oracle, W, H, n_kf, 0x5EED0A77, disparity_px=6.0)
    mind, maxd = prior(1.0, 0.3)
    eng = pkg.Engine(W, H, n_kf, max_neighbours=n)
    seq.upload(eng, device_prepass=True)
    eng.set_scan_mode(2)
    refs = list(range(n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    from pm_oracle import Oracle
    for lam in (14.0, 5.0, 8.0):
        eng.set_params(lambdaG=lam)
        o = Oracle("strict")
        o.params.lambdaG = lam
        eng.search_fuse([3], [nbrs[3]], mind, maxd)  # ONE reference keyframe: its neighbours were never references
        r, s, _ = o.recon_search_fuse(seq.okf[3], [seq.okf[j] for j in nbrs[3]], None, mind, maxd)
        gr, gs = eng.download_depth(3)
        assert_bit_equal(gr, r, "lambdaG %g rho" % lam)
        assert_bit_equal(gs, s, "lambdaG %g sigma" % lam)
    eng.close()


@pytest.mark.parametrize("lam_theta,lam_l,theta_var", [(30.0, 80.0, 0.23), (100.0, 70.0, 0.25), (170.0, 80.0, 0.23),
                                                        (200.0, 85.0, 0.1), (0.0, 80.0, 0.23), (45.0, 80.0, 0.25),
                                                        (40.0, 70.0, 0.25), (56.0, 95.0, 3.0), (-5.0, -1.0, 0.23)])
def test_other_thresholds_all_scan_modes(pkg, oracle, gpu_ok, lam_theta, lam_l, theta_var):
    """PM.h:38-49 made runtime: thresholds other than the defaults run the closed-form gates and the approximate arg-min
    with run-time constants (validated on the device when they are set, sdm_set_params), and the orientation window the
    mask scan lists follows lambdaTheta: narrower, wider than the planes resolve (the gradient plane alone), zero, negative"""
    from pm_oracle import Oracle
    W, H, n_kf, n = 192, 96, 8, 7
    seq = Sequence(pkg, oracle, W, H, n_kf, 0x5EED0A55, disparity_px=8.0)
    mind, maxd = prior(1.0, 0.3)
    o = Oracle("strict")
    o.params.lambdaTheta = lam_theta
    o.params.lambdaL = lam_l
    o.params.theta_var = theta_var
    refs = list(range(n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    rng = np.random.default_rng(5)
    rots = rng.uniform(-20, 20, (n_kf, n)).astype(np.float32)
    want = [o.recon_search_fuse(seq.okf[k], [seq.okf[j] for j in nbrs[k]], rots[k], mind, maxd) for k in refs]
    for mode in (2, 0, 1):
        eng = pkg.Engine(W, H, n_kf, max_neighbours=n)
        seq.upload(eng, device_prepass=True)
        eng.set_params(lambdaTheta=lam_theta, lambdaL=lam_l, theta_var=theta_var)
        eng.set_scan_mode(mode)
        eng.enable_stats(True)
        eng.get_stats(reset=True)
        eng.search_fuse(refs, nbrs, mind, maxd, rot=rots)
        st = eng.get_stats()
        for k in refs:
            gr, gs = eng.download_depth(k)
            assert_bit_equal(gr, want[k][0], "lambdaTheta %g mode %d rho kf %d" % (lam_theta, mode, k))
            assert_bit_equal(gs, want[k][1], "lambdaTheta %g mode %d sigma kf %d" % (lam_theta, mode, k))
        assert st["candidates"] == sum(w[2]["candidates"] for w in want)
        assert st["gate_pass"] == sum(w[2]["gate_pass"] for w in want)
        assert st["mask_row_mismatch"] == 0
        eng.close()
