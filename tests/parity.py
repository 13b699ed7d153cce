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


def assert_mixed_parity(got, want, what='', rtol=1e-3, atol_of_max=1e-6):
    """The float32 mixed-precision form against the float64 arithmetic rounded to float32: NaN
    and inf masks identical; zero masks identical once float32 subnormals count as zero on both
    sides (v_exp_f32 / v_rcp_f32 flush them: a result below 1.2e-38 is 0 or a subnormal at the
    hardware's choice); every value within rtol OR within atol_of_max of the largest value (the
    form bounds the absolute error: s A + rho Cp vpd / r_a cancels at night)."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype == np.float32, (what, got.dtype, want.dtype)
    assert np.array_equal(np.isnan(got), np.isnan(want)), '%s: NaN masks differ' % what
    assert np.array_equal(np.isinf(got), np.isinf(want)), '%s: inf masks differ' % what
    tiny = np.finfo(np.float32).tiny
    g = np.where(np.abs(got) < tiny, 0, got).astype(np.float64)
    w = np.where(np.abs(want) < tiny, 0, want).astype(np.float64)
    zg, zw = g == 0, w == 0
    assert np.array_equal(zg, zw), '%s: zero masks differ at %d pixels' % (what, int((zg != zw).sum()))
    ok = np.isfinite(w) & (w != 0)
    if ok.any():
        err = np.abs(g[ok] - w[ok])
        bound = np.maximum(rtol * np.abs(w[ok]), atol_of_max * np.abs(w[ok]).max())
        worst = float((err / bound).max())
        assert worst <= 1, '%s: %.3g x the mixed form\'s tolerance' % (what, worst)


def in_a_fresh_thread(fn, env):
    """Runs fn in a new thread -- a new context (mod16_amd._lib.context is per thread), created
    with `env` in the environment (the library reads its two knobs when a context is made)."""
    import os
    import threading
    box = {}

    def body():
        try:
            box['value'] = fn()
        except BaseException as exc:      # handed to the caller
            box['error'] = exc
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        t = threading.Thread(target=body)
        t.start()
        t.join()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if 'error' in box:
        raise box['error']
    return box['value']


def same_bits(a, b):
    """Same shape, dtype, NaN positions, and the same BITS in every other value (signed zeros told
    apart). NaN payloads are not compared: which NaN an invalid pixel yields is not part of the result."""
    a, b = np.atleast_1d(np.asarray(a)), np.atleast_1d(np.asarray(b))
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    u = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return bool(np.array_equal(np.ascontiguousarray(a).view(u)[~na], np.ascontiguousarray(b).view(u)[~nb]))
