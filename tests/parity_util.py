"""Shared parity helpers of the GPU tests."""
import numpy as np

F = np.float32
# cost_func.py:9-17: -10 each for cur[5] >= 0.2, cur[6] >= 0, cur[7] >= 0
CHEETAH_INDICATORS = ((5, 0.2), (6, 0.0), (7, 0.0))


def cheetah_threshold_margin(ev, states, seq):
    """How close the ORACLE's trajectory of every (particle, agent) row comes to one of the reward's indicator thresholds:
    min over planning steps and the three indicators of |cur[i] - threshold| / (t + 1) -- a state error grows along the
    trajectory, so later steps are given proportionally more room.  `ev`: oracle_np.Evaluator (predict_next_state on row
    batches), states [A,S], seq [N,A,H,U].  -> [N,A] float64."""
    seq = np.asarray(seq, F)
    n, a, h, u = seq.shape
    rows = seq.reshape(n * a, h, u)
    state = np.tile(np.asarray(states, F), (n, 1))
    margin = np.full((n * a,), np.inf)
    for t in range(h):
        for i, thr in CHEETAH_INDICATORS:
            margin = np.minimum(margin, np.abs(state[:, i].astype(np.float64) - thr) / (t + 1))
        state = ev.predict_next_state(state, rows[:, t])
    return margin.reshape(n, a)


def assert_cheetah_rewards(got, want, rtol, atol, max_flip_frac=0.003, what="", horizon=None, margin=None, margin_tol=2e-4):
    """H-step HalfCheetah rewards of the device against the oracle's within the stated fp32 tolerance.

    The reward (tutorials/mujoco/cost_func.py:9-19) has three indicator terms, -10 each (cur[5] >= 0.2, cur[6] >= 0, cur[7] >= 0):
    a state that sits on a threshold flips one of them under ANY change of rounding (another summation order, a 1-ulp
    activation), so a trajectory in a few hundred may differ by a multiple of 10 with everything else in agreement.  Those,
    and only those, are let through:
      * each within the tolerance of a multiple of 10, at most 3 flips per planning step taken: 10 * 3 * horizon (every
        caller's atol is 1e-3 * H, which is where `horizon` comes from when it is not given);
      * with `margin` (cheetah_threshold_margin of the same rows): each only where the oracle's own trajectory comes within
        `margin_tol` per step taken of a threshold -- a flip anywhere else is a kernel bug -- and then their number is not capped;
      * without it: at most `max_flip_frac` of the entries -- none at all in an array too small for that fraction to
        reach one entry (fewer than 1 / max_flip_frac).
    Everything else must meet rtol / atol."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if horizon is None:
        horizon = max(1, int(round(atol / 1e-3)))
    tol = atol + rtol * np.abs(want)
    err = np.abs(got - want)
    bad = err > tol
    if not bad.any():
        return 0
    n_bad = int(bad.sum())
    if margin is not None:
        margin = np.asarray(margin() if callable(margin) else margin, np.float64)     # (a callable: computed only when there is something to explain)
        assert margin.shape == got.shape, (margin.shape, got.shape)
        assert np.all(margin[bad] <= margin_tol), \
            "%sreward differences on trajectories that stay clear of every indicator threshold (margins %r): %r" % (what, margin[bad], err[bad])
    else:
        assert n_bad <= int(max_flip_frac * got.size), "%s%d of %d rewards outside the tolerance" % (what, n_bad, got.size)
    steps = np.round(err[bad] / 10.0)
    assert np.all(steps >= 1), "%sreward differences outside the tolerance that are no indicator flips: %r" % (what, err[bad])
    assert np.all(steps <= 3 * horizon), "%smore indicator flips than a %d-step trajectory has indicators: %r" % (what, horizon, err[bad])
    # a flipped indicator also moves the trajectory from that step on a little: allow the stated tolerance twice over
    assert np.all(np.abs(err[bad] - 10.0 * steps) <= 2.0 * tol[bad] + 0.05), \
        "%sreward differences that are no multiples of 10: %r" % (what, err[bad])
    return n_bad


def pendulum_rewards_f64(states, seq, as_executed=True):
    """The H-step summed rewards of utils/pendulum.py:10-35, :58-92 (oracle_np.pendulum_dynamics / pendulum_reward) in
    float64: what the reference's recurrence gives without its float32 rounding.  states [A,3], seq [N,A,H,1] -> [N,A]."""
    states = np.asarray(states, np.float64)
    seq = np.asarray(seq, np.float64)
    N, A, H, _ = seq.shape
    th = np.arctan2(states[:, 1], states[:, 0])[None, :].repeat(N, 0)
    thd = states[:, 2][None, :].repeat(N, 0)
    R = np.zeros((N, A))
    for t in range(H):
        u = seq[:, :, t, 0]
        ang = np.mod(th + np.pi, 2.0 * np.pi) - np.pi
        nthd = thd + (15.0 * np.sin(th) + 3.0 * u) * 0.05         # -15 sin(th + pi)
        nth = th + nthd * 0.05                                     # (quirk Q9: the unclipped speed)
        nthd = np.clip(nthd, -8.0, 8.0)
        ss = 1.0 + nthd * nthd if as_executed else u * u          # quirk Q1: 0.001 * sum(next_state ** 2)
        R += -(ang * ang + 0.1 * thd * thd) - 0.001 * ss
        th, thd = nth, nthd
    return R


def assert_pendulum_rewards(got, want, states, seq, rtol, atol, max_frac=0.01, as_executed=True, what=""):
    """H-step summed pendulum rewards against the float32 oracle's.  A trajectory that lingers near the upright position
    amplifies rounding differences by e^(sqrt(15) t) -- four orders of magnitude over H = 50 -- so the ORACLE's own float32
    sum is off the exact-arithmetic value of the same recurrence by more than the tolerance there (2.7e-4 relative in
    test_evaluator_matches_oracle[64-5-50]).  An element outside the tolerance of the oracle's value therefore has to be
    at least as close to the float64 evaluation as the oracle's float32 evaluation is (+ the tolerance), i.e. inside the
    reference's own rounding uncertainty; and such elements have to be few."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    tol = atol + rtol * np.abs(want)
    bad = np.abs(got - want) > tol
    if not bad.any():
        return
    exact = pendulum_rewards_f64(states, seq, as_executed)
    assert bad.mean() <= max_frac, "%s: %d of %d sums outside the tolerance" % (what, int(bad.sum()), bad.size)
    e_got, e_want = np.abs(got - exact), np.abs(want - exact)
    worse = bad & (e_got > e_want + tol)
    assert not worse.any(), ("%s: %d sums are outside the tolerance AND further from the float64 evaluation than the oracle's float32 one "
                             "(worst: got %r want %r exact %r)" % (what, int(worse.sum()), got[worse][:3], want[worse][:3], exact[worse][:3]))
