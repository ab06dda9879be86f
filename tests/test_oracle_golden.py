"""CPU: the oracle reproduces the committed golden vectors bit for bit (pins the oracle build and
the scene parameters across machines; the reference itself holds no vectors -- parity unpinned)."""
import numpy as np
import pytest

import golden_util as gu
from common import assert_bit_equal, oracle_inter, oracle_pipeline


@pytest.mark.parametrize("name", gu.fixture_names())
def test_oracle_matches_golden(pkg, oracle, name):
    g = gu.load(name)
    seq = gu.sequence_from(pkg, oracle, g)
    n = g["n"]
    # scene parameters (poses, intrinsics, depth prior, neighbour order) are regenerated from the seed
    assert_bit_equal(np.stack(seq.Tcw), g["Tcw"], "Tcw")
    assert_bit_equal(seq.K, g["K"], "K")
    assert np.float32(seq.min_depth) == g["min_depth"] and np.float32(seq.max_depth) == g["max_depth"]
    assert (np.array([seq.neighbours(k, n) for k in range(g["n_kf"])]) == g["nbrs"]).all()
    # derived inputs
    for k in range(g["n_kf"]):
        assert gu.sha(seq.grad[k]) == str(g["grad_sha"][k])
        assert gu.sha(seq.theta[k]) == str(g["theta_sha"][k])
    assert_bit_equal(np.array(seq.istd, np.float32), g["istd"], "I_stddev")
    maps = oracle_pipeline(oracle, seq, n)
    chk, xyz = oracle_inter(oracle, seq, n, maps)
    for k in range(g["n_kf"]):
        assert_bit_equal(maps["k1_rho"][k], g["k1_rho"][k], "K1 rho kf %d" % k)
        assert_bit_equal(maps["k1_sigma"][k], g["k1_sigma"][k], "K1 sigma kf %d" % k)
        assert_bit_equal(maps["rho"][k], g["rho"][k], "rho kf %d" % k)
        assert_bit_equal(maps["sigma"][k], g["sigma"][k], "sigma kf %d" % k)
        assert_bit_equal(chk[k], g["chk"][k], "checked rho kf %d" % k)
        assert gu.sha(xyz[k]) == str(g["xyz_sha"][k])
        assert maps["stats"][k]["searches"] == g["searches"][k]
        assert maps["stats"][k]["candidates"] == g["candidates"][k]
        assert maps["stats"][k]["fused"] == g["fused"][k]


def test_synthetic_images_regenerate(pkg, oracle):
    """the committed images are what the generator renders here (allowing the rare +-1 gray level
    from libm `sin` differences at .5 quantisation ties)"""
    g = gu.load("plane_64x48_n7")
    from common import Sequence
    seq = Sequence(pkg, oracle, g["W"], g["H"], g["n_kf"], g["seed"], disparity_px=float(g["disparity_px"]))
    diff = np.abs(np.stack(seq.im).astype(np.int32) - g["im"].astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3


def test_strip_scene_recovers_both_planes(pkg, oracle):
    """the two-plane fixture: the checked depths recover the analytic ground truth on the background AND on the foreground
    strip, the regenerated rotations are the fixture's, and the occluding edges cost support (pixels next to the strip whose
    neighbours see the other plane) rather than accuracy"""
    g = gu.load("strip_roll_160x120_n7")
    from common import Sequence
    seq = Sequence(pkg, oracle, g["W"], g["H"], g["n_kf"], g["seed"], disparity_px=float(g["disparity_px"]), **gu.scene_options(g))
    assert_bit_equal(seq.rots(range(g["n_kf"]), g["n"]), g["rot"], "median in-plane rotations")
    assert np.abs(g["rot"]).max() > 3.0
    for k in (2, 4, 5):
        kept = g["chk"][k] > 1e-6
        for name, plane in (("background", ~g["fg"][k]), ("strip", g["fg"][k])):
            m = kept & plane
            assert m.sum() > 150, (name, k, int(m.sum()))
            err = np.abs(g["chk"][k][m] - g["gt_rho"][k][m])
            assert np.median(err) < 8e-3, (name, k, float(np.median(err)))
        assert abs(float(g["gt_rho"][k][g["fg"][k]].mean()) - 1.0 / 0.75) < 0.02


def test_golden_accuracy_vs_ground_truth(pkg, oracle):
    """sanity of the whole restatement: the fused inverse depths recover the analytic plane"""
    g = gu.load("plane_160x120_n7")
    from common import Sequence
    seq = Sequence(pkg, oracle, g["W"], g["H"], g["n_kf"], g["seed"], disparity_px=float(g["disparity_px"]))
    for k in (0, 3, 7):
        m = g["chk"][k] > 1e-6
        assert m.sum() > 1000
        err = np.abs(g["chk"][k][m] - seq.gt[k][m])
        assert np.median(err) < 5e-3, np.median(err)
