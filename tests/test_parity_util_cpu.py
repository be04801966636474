"""The reward-comparison helper of the GPU parity tests (tests/parity_util.py) on CPU: what it lets through and what it does not."""
import numpy as np
import pytest

from oracle import oracle_np as O
from tests.parity_util import assert_cheetah_rewards, cheetah_threshold_margin

F = np.float32


def _ev():
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=42)
    stats = [np.zeros(20, F), np.ones(20, F), np.zeros(6, F), np.ones(6, F), np.zeros(20, F), np.full(20, 0.1, F)]
    return O.Evaluator("cheetah", O.Handler(O.MLP(ws, bs, ["tanh", "tanh", None]), False, True, stats))


def test_margin_is_the_distance_of_the_oracle_trajectory_to_a_threshold():
    ev = _ev()
    rng = np.random.default_rng(0)
    N, A, H, U = 6, 2, 5, 6
    states = O.cheetah_start_states(A, 20)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    m = cheetah_threshold_margin(ev, states, seq)
    assert m.shape == (N, A) and np.all(m >= 0)
    # by hand for one row
    n, a = 3, 1
    s = states[a:a + 1].copy()
    want = np.inf
    for t in range(H):
        want = min(want, abs(float(s[0, 5]) - 0.2) / (t + 1), abs(float(s[0, 6])) / (t + 1), abs(float(s[0, 7])) / (t + 1))
        s = ev.predict_next_state(s, seq[n, a, t][None])
    assert m[n, a] == pytest.approx(want, rel=1e-6)
    # a start state ON a threshold has margin 0 whatever follows
    st = states.copy()
    st[0, 6] = 0.0
    assert np.all(cheetah_threshold_margin(ev, st, seq)[:, 0] == 0.0)


def test_what_is_let_through():
    H = 30
    want = np.linspace(-50, 50, 1000).reshape(500, 2)
    assert assert_cheetah_rewards(want + 1e-3, want, 1e-3, 1e-3 * H) == 0
    flip = want.copy()
    flip[7, 1] -= 10.0
    assert assert_cheetah_rewards(flip, want, 1e-3, 1e-3 * H) == 1                # 1 of 1000 <= 0.3 %
    with pytest.raises(AssertionError):                                          # not a multiple of ten
        bad = want.copy(); bad[7, 1] -= 4.0
        assert_cheetah_rewards(bad, want, 1e-3, 1e-3 * H)
    with pytest.raises(AssertionError):                                          # too many
        bad = want.copy(); bad[:5, 0] -= 10.0
        assert_cheetah_rewards(bad, want, 1e-3, 1e-3 * H)
    with pytest.raises(AssertionError):                                          # more flips than 3 per step
        bad = want.copy(); bad[7, 1] -= 10.0 * (3 * H + 1)
        assert_cheetah_rewards(bad, want, 1e-3, 1e-3 * H)
    with pytest.raises(AssertionError):                                          # a small array has no free outlier
        assert_cheetah_rewards(flip[:24], want[:24], 1e-3, 1e-3 * H)
    # with margins: a flip is accepted exactly where the oracle's trajectory touches a threshold
    margin = np.full(want.shape, 0.1)
    with pytest.raises(AssertionError):
        assert_cheetah_rewards(flip[:24], want[:24], 1e-3, 1e-3 * H, margin=margin[:24])
    margin[7, 1] = 1e-5
    assert assert_cheetah_rewards(flip[:24], want[:24], 1e-3, 1e-3 * H, margin=lambda: margin[:24]) == 1


# ---- the pendulum helper: float64-anchored acceptance of the long-horizon sums -----------------------------------------
def _pendulum_case(N=64, A=5, H=50):
    from oracle import oracle_np as O
    rng = np.random.default_rng(N * 131 + A * 7 + H)            # tests/test_gpu_pendulum.py::test_evaluator_matches_oracle[64-5-50]
    states = O.pendulum_start_states(A)
    seq = rng.uniform(-2, 2, (N, A, H, 1)).astype(np.float32)
    want = O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))(states, seq)
    return states, seq, want


def _turn_form_rewards(states, seq):
    """csrc/models.hpp PendulumTurnModel in NumPy float32 (correctly rounded sine in place of v_sin_f32)."""
    F = np.float32
    N, A, H, _ = seq.shape
    phi = (np.arctan2(states[:, 1], states[:, 0]).astype(F) * F(0.15915494309189535))[None, :].repeat(N, 0).astype(F)
    thd = states[:, 2][None, :].repeat(N, 0).astype(F)
    R = np.zeros((N, A), F)
    c = F(F(0.05) * F(0.15915494309189535))
    for t in range(H):
        u = seq[:, :, t, 0]
        sn = np.sin(2 * np.pi * phi.astype(np.float64)).astype(F)
        acc = F(15) * sn
        acc = acc + F(3) * u
        nthd = thd + acc * F(0.05)
        nphi = phi + nthd * c
        nthd = np.clip(nthd, F(-8), F(8))
        n2 = (nthd - thd) + thd
        ang = phi * F(2 * np.pi)
        R = R + ((-(ang * ang + F(0.1) * (thd * thd))) - F(0.001) * (F(1) + n2 * n2))
        phi = (nphi - np.rint(nphi)).astype(F)
        thd = n2
    return R


def test_pendulum_helper_accepts_a_sum_inside_the_oracles_own_rounding_uncertainty():
    from tests.parity_util import assert_pendulum_rewards, pendulum_rewards_f64
    states, seq, want = _pendulum_case()
    got = _turn_form_rewards(states, seq)
    exact = pendulum_rewards_f64(states, seq)
    bad = np.abs(got - want) > 2e-3 + 2e-4 * np.abs(want)
    # the case the helper exists for: the float32 oracle is further from exact arithmetic than the other float32 form
    assert bad.sum() >= 1 and np.all(np.abs(got - exact)[bad] < np.abs(want - exact)[bad])
    assert_pendulum_rewards(got, want, states, seq, 2e-4, 2e-3)


def test_pendulum_helper_rejects_a_wrong_sum_and_too_many_outliers():
    from tests.parity_util import assert_pendulum_rewards
    states, seq, want = _pendulum_case()
    got = np.array(want, np.float64)
    got[3, 2] += 0.05                                            # outside the tolerance, and not towards the exact value
    with pytest.raises(AssertionError):
        assert_pendulum_rewards(got, want, states, seq, 2e-4, 2e-3)
    got = np.array(want, np.float64) * (1.0 + 1e-3)              # everything off
    with pytest.raises(AssertionError):
        assert_pendulum_rewards(got, want, states, seq, 2e-4, 2e-3)
