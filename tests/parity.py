"""Comparison helpers shared by the parity tests."""
import numpy as np

# north_star tolerance: 1e-5 relative in float64 (BASELINE.json). The tests
# hold the kernels to much tighter bounds, stated where they are used.
RTOL_NORTH_STAR = 1e-5


def rel_err(got, want):
    """max |got - want| / |want| over pixels where want is finite and != 0."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    m = np.isfinite(want) & (want != 0)
    if not m.any():
        return 0.0
    return float(np.max(np.abs(got[m] - want[m]) / np.abs(want[m])))


def assert_parity(got, want, rtol, what=''):
    """NaN masks and exact-zero masks identical, everything else within rtol."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert got.dtype == want.dtype, (what, got.dtype, want.dtype)
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), \
        '%s: NaN masks differ at %d pixels' % (what, int((nan_g != nan_w).sum()))
    zero_g, zero_w = (got == 0), (want == 0)
    assert np.array_equal(zero_g, zero_w), \
        '%s: zero masks differ at %d pixels' % (what, int((zero_g != zero_w).sum()))
    inf_g, inf_w = np.isinf(got), np.isinf(want)
    assert np.array_equal(inf_g, inf_w), '%s: inf masks differ' % what
    err = rel_err(got, want)
    assert err <= rtol, '%s: max rel err %.3e > %.1e' % (what, err, rtol)
    return err
