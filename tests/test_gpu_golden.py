"""GPU: the HIP engine against the committed golden vectors (no oracle call on this path)."""
import numpy as np
import pytest

import golden_util as gu
from common import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", gu.fixture_names())
@pytest.mark.parametrize("device_prepass", [False, True])
def test_engine_matches_golden(pkg, oracle, gpu_ok, name, device_prepass):
    g = gu.load(name)
    n, n_kf = g["n"], g["n_kf"]
    eng = pkg.Engine(g["W"], g["H"], n_kf, max_neighbours=n)
    if device_prepass:
        for k in range(n_kf):
            eng.upload_image(k, g["im"][k], g["K"], g["Tcw"][k])
            _, gr, th, istd = eng.download_inputs(k)
            assert gu.sha(gr) == str(g["grad_sha"][k]) and gu.sha(th) == str(g["theta_sha"][k])
            assert np.float32(istd) == g["istd"][k]
    else:
        seq = gu.sequence_from(pkg, oracle, g)  # oracle pre-pass only supplies the INPUT planes here
        seq.upload(eng)
    refs = list(range(n_kf))
    rots = gu.rots(g)
    eng.search_fuse(refs, g["nbrs"], float(g["min_depth"]), float(g["max_depth"]), rot=rots)
    for k in refs:
        r, s = eng.download_depth(k)
        assert_bit_equal(r, g["k1_rho"][k], "K1 rho kf %d" % k)
        assert_bit_equal(s, g["k1_sigma"][k], "K1 sigma kf %d" % k)
    eng.recon(refs, g["nbrs"], float(g["min_depth"]), float(g["max_depth"]), rot=rots)
    eng.inter_check(refs, g["nbrs"])
    eng.pointset(refs, source=1)
    for k in refs:
        r, s = eng.download_depth(k)
        assert_bit_equal(r, g["rho"][k], "rho kf %d" % k)
        assert_bit_equal(s, g["sigma"][k], "sigma kf %d" % k)
        assert_bit_equal(eng.download_checked(k), g["chk"][k], "checked rho kf %d" % k)
        assert gu.sha(eng.download_pointset(k)) == str(g["xyz_sha"][k])
    eng.close()
