"""CPU checks of arithmetic identities the HIP kernels rely on (host logic; no GPU, no oracle)."""
import numpy as np


def test_double_1em6_compare_is_a_float_compare():
    """sdm_device.h gt_1em6/lt_1em6: (double)x > 1e-6  <=>  x > 0x358637bd,  (double)x < 1e-6  <=>  x <= 0x358637bd"""
    f0 = np.uint32(0x358637BD).view(np.float32)
    assert float(f0) < 1e-6 < float(np.nextafter(f0, np.float32(1)))
    bits = np.arange(0x358637BD - 4096, 0x358637BD + 4096, dtype=np.uint32)
    xs = np.concatenate([bits.view(np.float32), -bits.view(np.float32),
                         np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 1.0], dtype=np.float32)])
    wide = xs.astype(np.float64)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(wide > 1e-6, xs > f0)
        assert np.array_equal(wide < 1e-6, xs <= f0)


def test_reciprocal_sign_test_is_a_range_test():
    """k_search_fuse: (1.0f/rho) > 0  <=>  bits(rho) < bits(+Inf), i.e. rho in [+0, +Inf)  (denormals on), PM.cc:216"""
    xs = np.array([0.0, -0.0, 1e-45, 1e-38, 1.0, 3.4028235e38, np.inf, -np.inf, np.nan, -1.0, -1e-45], dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        want = (np.float32(1.0) / xs) > 0
        got = xs.view(np.uint32) < np.uint32(0x7F800000)
    assert np.array_equal(want, got)
