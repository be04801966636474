"""Shared parity helpers of the GPU tests."""
import numpy as np


def assert_cheetah_rewards(got, want, rtol, atol, max_flip_frac=0.003, what=""):
    """H-step HalfCheetah rewards of the device against the oracle's within the stated fp32 tolerance.

    The reward (tutorials/mujoco/cost_func.py:9-19) has three indicator terms, -10 each (cur[5] >= 0.2, cur[6] >= 0, cur[7] >= 0):
    a state that sits on a threshold flips one of them under ANY change of rounding (another summation order, a 1-ulp
    activation), so a trajectory in a few hundred may differ by a multiple of 10 with everything else in agreement.  Those,
    and only those, are let through: at most `max_flip_frac` of the entries (at least one), each within the tolerance of a
    multiple of 10 no larger than 30 per planning step taken.  Everything else must meet rtol / atol."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    tol = atol + rtol * np.abs(want)
    err = np.abs(got - want)
    bad = err > tol
    if not bad.any():
        return 0
    n_bad = int(bad.sum())
    assert n_bad <= max(1, int(max_flip_frac * got.size)), "%s%d of %d rewards outside the tolerance" % (what, n_bad, got.size)
    steps = np.round(err[bad] / 10.0)
    assert np.all(steps >= 1), "%sreward differences outside the tolerance that are no indicator flips: %r" % (what, err[bad])
    # a flipped indicator also moves the trajectory from that step on a little: allow the stated tolerance twice over
    assert np.all(np.abs(err[bad] - 10.0 * steps) <= 2.0 * tol[bad] + 0.05), \
        "%sreward differences that are no multiples of 10: %r" % (what, err[bad])
    return n_bad
