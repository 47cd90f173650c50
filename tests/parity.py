"""Shared tolerance of the floating-point parity tests, and the record of what was observed.

BASELINE.json's north_star: sampled indices bit-exact, fp32 embeddings within 1e-4 of the reference CPU path.  `close` asserts
that bar as a PLAIN absolute 1e-4 (forward quantities: encoder input, layer outputs, embeddings, link probabilities, memories);
`close_scaled` is for quantities whose magnitude is unbounded by construction (parameter gradients: sums over all rows of a
call), where the bar is 1e-4 * max(1, max|reference|).  Every comparison is recorded; tests/conftest.py prints the largest
observed error per label at the end of the run, so the margin is visible in the test output."""
import numpy as np

TOL = 1e-4
OBSERVED = {}          # label -> (max abs err, max |ref|, tolerance used)


def _record(label, err, refmax, atol):
    old = OBSERVED.get(label)
    if old is None or err > old[0]:
        OBSERVED[label] = (err, refmax, atol)


def _err(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what
    if got.size == 0:
        return 0.0, 0.0
    return float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()), float(np.abs(want).max())


def close(got, want, what="", label=None):
    """max |got - want| <= 1e-4, absolute."""
    err, refmax = _err(got, want, what)
    _record(label or what, err, refmax, TOL)
    assert err <= TOL, f"{what}: max abs err {err:.3e} > {TOL:.1e} (max |ref| {refmax:.3g})"
    return err


def close_scaled(got, want, what="", label=None):
    """max |got - want| <= 1e-4 * max(1, max |want|): for gradients and other sums of unbounded magnitude."""
    err, refmax = _err(got, want, what)
    atol = TOL * max(1.0, refmax)
    _record(label or what, err, refmax, atol)
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol:.3e} (max |ref| {refmax:.3g})"
    return err
