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


def test_match_cost_fast_path_is_exact(pkg, gpu_ok):
    """err = (float)(pe2 + ge2/THETA): reciprocal path + mid-point fallback == the reference expression,
    including operands constructed to land on float rounding midpoints"""
    eng = pkg.Engine(64, 48, 2)
    bad, risky = eng.selftest(2)
    assert bad == 0
    assert risky > 1000, "the fallback must be exercised"
    eng.close()


def test_angle_gates_closed_form(pkg, gpu_ok):
    """closed-form gates 2/3 == the reference's wrap-and-compare statement for every tested difference"""
    eng = pkg.Engine(64, 48, 2)
    bad, tested = eng.selftest(3)
    assert bad == 0
    assert tested > 2 ** 31
    eng.close()


def test_fusion_terms_shared_reciprocal(pkg, gpu_ok):
    """GetFusion's rho/sigma^2 via the reciprocal 1/sigma^2 + FMA corrections == the plain double division"""
    eng = pkg.Engine(64, 48, 2)
    bad, tested = eng.selftest(4)
    assert bad == 0
    assert tested >= 2 ** 31
    eng.close()


def test_shared_divisor_quotient(pkg, gpu_ok):
    """a/b as a*r with two FMA corrections, r = v_rcp_f32 + one FMA step (K4's quot_fast / rcp_fast) == IEEE division
    for every operand pair inside the quotient window [2^-40, 2^41), all-ones divisor significands included; and for the
    operands K1's line quotients see (zero or [2^-59, 2^38], b != 0)"""
    eng = pkg.Engine(64, 48, 2)
    bad, tested = eng.selftest(5)
    assert bad == 0
    assert tested >= 5 * 10 ** 9  # pairs that were inside the window (of 8.6e9 drawn in and around it)
    eng.close()


def test_exact_reciprocal(pkg, gpu_ok):
    """v_rcp_f32 + one FMA residual step == IEEE 1.0f/b for every float bit pattern (guards route the rest); the same walk
    checks sqrt_exact (v_rsq_f32 + one FMA step inside [2^-100, 2^127), sqrtf outside) against sqrtf"""
    eng = pkg.Engine(64, 48, 2)
    bad, fast = eng.selftest(6)
    assert bad == 0
    assert fast == 2 * 250 * 2 ** 23  # every operand with 2^-125 <= |b| < 2^125 took the reciprocal path
    eng.close()


def test_inter_check_fast_body_vs_reference_statement(pkg, gpu_ok):
    """K4's straight-line per-neighbour body (reciprocal-form quotients, window guards, division-free 3.84 test):
    wherever it does not raise its slow flag it must equal the plain-division reference statement bit for bit --
    5*10^8 random cases with magnitudes far outside the windows, taps on the threshold, zero/Inf/NaN operands"""
    eng = pkg.Engine(64, 48, 2)
    bad, accepted = eng.selftest(7)
    assert bad == 0
    total = 1024 * 256 * 2048
    assert 0.25 * total < accepted <= total, accepted  # the fast path is the common case, and it is really exercised
    eng.close()


def test_inter_check_approximate_projection(pkg, gpu_ok):
    """K4's approximate projection (one FMA per row instead of a quotient, v_rcp_f32 instead of the two divisions by z):
    wherever it does not ask for the exact chain, cell, validity and offset are the plain-division chain's -- the same
    5*10^8 random geometries as selftest 7, incl. pixel-scale intrinsics; and it decides the common case"""
    eng = pkg.Engine(64, 48, 2)
    bad, decided = eng.selftest(9)
    assert bad == 0
    total = 1024 * 256 * 2048
    assert 0.5 * total < decided <= total, decided
    eng.close()


def test_scan_identities(pkg, gpu_ok):
    """K1 scan: lerp weight 1 - fract(yf) == (floor(yf)+1) - yf for every float in [0, 2^24); the integer-mask wrap of
    PM.cc:425-426 == the compare statement for every float bit pattern (up to the sign of a zero)"""
    eng = pkg.Engine(64, 48, 2)
    bad, tested = eng.selftest(8)
    assert bad == 0
    assert tested > 2 ** 32
    eng.close()
