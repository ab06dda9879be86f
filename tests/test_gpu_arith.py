"""Device-arithmetic contracts the bit-exactness of the kernels rests on."""
import pytest

pytestmark = pytest.mark.gpu


def test_div_theta_exhaustive(pkg, gpu_ok):
    """x/0.23 by reciprocal + two FMA corrections == IEEE division for every float-derived input"""
    eng = pkg.Engine(64, 48, 2)
    bad, _ = eng.selftest(0)
    assert bad == 0
    eng.close()


def test_chi_prefilter_is_exact(pkg, gpu_ok):
    """the reciprocal pre-filter of ChiTest never changes a decision; the exact path is exercised"""
    eng = pkg.Engine(64, 48, 2)
    bad, inband = eng.selftest(1)
    assert bad == 0
    assert inband > 1000, "test must hit the uncertainty band"
    eng.close()
