"""Pins the oracle (oracle/oracle_np.py) against known answers derived by hand from the reference
source.  The reference has no tests / golden vectors of its own and cannot be run
here, so these KATs + closed-form checks are what anchors the oracle ("parity unpinned" otherwise)."""
import math

import numpy as np
import pytest

from oracle import oracle_np as O

F = np.float32


def _pend_eval():
    return O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))


# values derived independently with math.* in float64 from utils/pendulum.py:78-91, :27-35
def _pend_f64(s, u):
    th = math.atan2(s[1], s[0])
    nthd = s[2] + (-15.0 * math.sin(th + math.pi) + 3.0 * u) * 0.05
    nth = th + nthd * 0.05
    nthd = min(max(nthd, -8.0), 8.0)
    n = (math.cos(nth), math.sin(nth), nthd)
    ang = ((th + math.pi) % (2 * math.pi)) - math.pi
    r = -(ang ** 2 + 0.1 * s[2] ** 2) - 0.001 * sum(v * v for v in n)
    return n, r


@pytest.mark.parametrize("s,u,exp_next,exp_r", [
    ((1.0, 0.0, 0.0), 2.0, (0.9998875, 0.01499944, 0.30000007), -0.00109),
    ((-1.0, 0.0, 0.0), -2.0, (-0.9998875, 0.01499945, -0.30000016), -9.870695),
    ((0.0, 1.0, 1.0), 0.5, (-0.09112353, 0.9958396, 1.825), -2.5717313),
    ((math.cos(3.0), math.sin(3.0), 7.9), 2.0, (-0.96277755, -0.27029496, 8.0), -15.306002),
])
def test_pendulum_known_answers(s, u, exp_next, exp_r):
    ev = _pend_eval()
    st, ac = np.array([s], F), np.array([[u]], F)
    nxt = ev.predict_next_state(st, ac)
    r = ev.evaluate_next_reward(st, nxt, ac)
    np.testing.assert_allclose(nxt[0], exp_next, rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(r[0], exp_r, rtol=2e-6, atol=2e-7)
    n64, r64 = _pend_f64(s, u)
    np.testing.assert_allclose(nxt[0], n64, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r[0], r64, rtol=1e-5, atol=1e-6)


def test_reward_arg_order_quirk_q1():
    cur = np.array([[0.0, 1.0, 2.0]], F)
    act = np.array([[1.5]], F)
    nxt = np.array([[0.5, 0.5, 3.0]], F)
    base = -((math.pi / 2) ** 2 + 0.1 * 4.0)
    np.testing.assert_allclose(O.pendulum_reward(cur, act, nxt)[0], base - 0.001 * (0.25 + 0.25 + 9.0), rtol=1e-6)
    np.testing.assert_allclose(O.pendulum_reward(cur, act, nxt, as_executed=False)[0], base - 0.001 * 2.25, rtol=1e-6)


def test_floormod_follows_divisor_sign():
    np.testing.assert_allclose(O.floormod32(F(-1.0), F(6.2831855)), 5.2831855, rtol=1e-6)
    np.testing.assert_allclose(O.floormod32(F(7.0), F(6.2831855)), 0.7168145, rtol=1e-5)
    assert O.floormod32(F(0.0), F(6.2831855)) == 0.0


def test_cheetah_reward_closed_form():
    cur = np.zeros((3, 20), F)
    nxt = np.zeros((3, 20), F)
    cur[0, 5] = 0.2; cur[0, 6] = -0.1; cur[0, 7] = -0.1; nxt[0, 17] = 0.05     # one penalty + 5.0 progress
    cur[1, 5] = 0.1; cur[1, 6] = 0.0; cur[1, 7] = 0.3; cur[1, 17] = 1.0; nxt[1, 17] = 0.98   # two penalties - 2.0
    cur[2, 5] = -1; cur[2, 6] = -1; cur[2, 7] = -1
    r = O.cheetah_reward(cur, np.ones((3, 6), F), nxt)
    np.testing.assert_allclose(r, [-10 + 5.0, -20 - 2.0, 0.0], rtol=1e-5, atol=1e-5)


def test_nan_reward_guard():
    ev = _pend_eval()
    seq = np.zeros((4, 1, 5, 1), F)
    r = ev(np.array([[np.nan, 0.0, 0.0]], F), seq)
    assert np.all(r == F(-1e6))


def test_evaluator_row_order_and_tiling():
    # row b = n*A + a (deterministic.py:53-57): agent a of particle n starts from current_states[a]
    ev = _pend_eval()
    rng = np.random.default_rng(0)
    states = O.pendulum_start_states(3)
    seq = rng.uniform(-2, 2, size=(4, 3, 6, 1)).astype(F)
    full = ev(states, seq)
    for a in range(3):
        single = ev(states[a:a + 1], seq[:, a:a + 1])
        np.testing.assert_array_equal(full[:, a], single[:, 0])


def test_topk_and_argmax_tie_rules():
    v = np.array([[1.0, 3.0, 3.0, 2.0, 3.0]], F)
    assert O.topk_desc(v, 3).tolist() == [[1, 2, 4]]
    assert O.argmax_first(np.array([[1, 5], [7, 5], [7, 2]], F), axis=0).tolist() == [1, 0]


class _FakeEval:
    """evaluator returning prescribed rewards so refits can be checked in closed form"""
    def __init__(self, rewards):
        self.rewards = [np.asarray(r, F) for r in rewards]
        self.i = 0

    def __call__(self, state, samples):
        r = self.rewards[self.i]
        self.i += 1
        return r

    def predict_next_state(self, s, a):
        return s

    def evaluate_next_reward(self, s, n, a):
        return np.zeros((s.shape[0],), F)


def test_cem_refit_closed_form():
    # N=4, k=2, H=1, U=1, bounds +-2: mean0=0, var0=1, sigma=min(1,1,1)=1 -> samples = xi
    xi = np.array([0.5, -1.0, 1.5, 0.25], F).reshape(4, 1, 1, 1)
    ev = _FakeEval([[[1.0], [5.0], [3.0], [5.0]]])     # tie between particles 1 and 3 -> 1 first
    cem = O.CEM(ev, [-2.0], [2.0], horizon=1, max_iterations=1, population=4, num_elite=2, num_agents=1, alpha=0.25)
    act = cem._optimize(np.zeros((1, 3), F), {"trunc": [xi]})
    assert cem.trace[0]["elites"].tolist() == [[1, 3]]
    em = (-1.0 + 0.25) / 2
    ev_ = ((-1.0 - em) ** 2 + (0.25 - em) ** 2) / 2
    np.testing.assert_allclose(act[0, 0], 0.75 * em, rtol=1e-6)
    np.testing.assert_allclose(cem.trace[0]["var"][0, 0, 0], 0.25 * 1.0 + 0.75 * ev_, rtol=1e-6)


def test_cem_constrained_variance():
    # mean near the upper bound: sigma = (hi-mean)/2
    ev = _FakeEval([[[0.0], [1.0]]])
    cem = O.CEM(ev, [-2.0], [2.0], horizon=1, max_iterations=1, population=2, num_elite=1, num_agents=1)
    cem.prev[:] = 1.0
    cem._optimize(np.zeros((1, 3), F), {"trunc": [np.array([1.0, -1.0], F).reshape(2, 1, 1, 1)]})
    np.testing.assert_allclose(cem.trace[0]["samples"].reshape(-1), [1.5, 0.5], rtol=1e-6)


def test_pi2_two_particle_softmin():
    ev = _FakeEval([[[-1.0], [-3.0]]])      # costs 1, 3 -> weights softmax(-cost/lambda)
    pi2 = O.PI2(ev, [-2.0], [2.0], horizon=2, max_iterations=1, population=2, num_agents=1, lamda=2.0)
    xi = np.array([[1.0, 0.5], [-1.0, 1.9]], F).reshape(2, 1, 2, 1)
    act = pi2._optimize(np.zeros((1, 3), F), {"trunc": [xi]})
    w0 = 1.0 / (1.0 + math.exp(-1.0))
    np.testing.assert_allclose(act[0, 0], w0 * 1.0 + (1 - w0) * -1.0, rtol=1e-6)
    # warm start = shift-left, last repeated (pi2.py:92-93)
    m1 = w0 * 0.5 + (1 - w0) * 1.9
    np.testing.assert_allclose(pi2.prev[0, :, 0], [m1, m1], rtol=1e-6)


def test_pi2_penalty_is_squared_norm_of_bound_violation():
    ev = _FakeEval([[[0.0], [0.0]]])
    pi2 = O.PI2(ev, [-2.0], [2.0], horizon=2, max_iterations=1, population=2, num_agents=1)
    pi2.prev[:] = 1.5
    xi = np.array([[1.0, 1.9], [0.0, 0.0]], F).reshape(2, 1, 2, 1)     # samples 2.5, 3.4 | 1.5, 1.5
    pi2._optimize(np.zeros((1, 3), F), {"trunc": [xi]})
    np.testing.assert_allclose(pi2.trace[0]["penalty"][:, 0], [0.5 ** 2 + 1.4 ** 2, 0.0], rtol=1e-5)
    assert pi2.trace[0]["samples"].max() <= 2.0


def test_random_search_argmax_first_and_uniform_scaling():
    ev = _FakeEval([[[1.0, 0.0], [4.0, 2.0], [4.0, 2.0]]])
    rs = O.RandomSearch(ev, [-2.0], [2.0], horizon=1, population=3, num_agents=2)
    u01 = np.array([0.0, 0.25, 0.5, 0.75, 1.0 - 2 ** -23, 0.5], F).reshape(3, 2, 1, 1)
    act = rs._optimize(np.zeros((2, 3), F), {"uniform": u01})
    assert rs.trace[0]["best"].tolist() == [1, 1]
    np.testing.assert_allclose(act[:, 0], [0.0, 1.0], atol=1e-6)


def test_cmaes_constructor_constants():
    c = O.cmaes_constants(500, 50, 50)
    np.testing.assert_allclose(c["mu_eff"], 26.9667, rtol=1e-5)
    np.testing.assert_allclose(c["c_sigma"], 0.353396, rtol=1e-5)
    np.testing.assert_allclose(c["d_sigma"], 1.353396, rtol=1e-5)
    np.testing.assert_allclose(c["cc"], 0.0824155, rtol=1e-5)
    np.testing.assert_allclose(c["c1"], 7.5226e-4, rtol=1e-4)
    np.testing.assert_allclose(c["c_mu"], 0.0183113, rtol=1e-5)
    np.testing.assert_allclose(c["e_norm"], 7.053436, rtol=1e-6)
    assert abs(float(c["weights"].sum()) - 1.0) < 1e-6 and np.all(c["weights"][50:] == 0)


def test_spsa_gain_sequences_and_gradient_sign():
    # reward increases with the action -> ghat > 0 -> solution moves up
    class Ev(_FakeEval):
        def __call__(self, state, samples):
            return samples.sum(axis=(2, 3)).astype(F)
    sp = O.SPSA(Ev([]), [-2.0], [2.0], horizon=2, max_iterations=2, population=8, num_agents=1)
    rng = np.random.default_rng(1)
    d = [np.where(rng.random((8, 1, 2, 1)) < 0.5, -1.0, 1.0).astype(F) for _ in range(2)]
    sp._optimize(np.zeros((1, 3), F), {"rademacher": d})
    np.testing.assert_allclose(sp.trace[0]["ak"], 0.01 / (1 + 0.2) ** 0.602, rtol=1e-6)
    np.testing.assert_allclose(sp.trace[1]["ck"], 0.3 / 2 ** 0.101, rtol=1e-6)
    assert np.all(sp.trace[1]["solution"] > 0)


def test_pso_ctor_state_vs_reset_state_q4():
    ev = _pend_eval()
    pso = O.PSO(ev, [-2.0], [2.0], horizon=3, max_iterations=1, population=4, num_agents=1)
    assert np.all(pso.pos == 0) and np.all(pso.pbest_r == 0)
    rng = np.random.default_rng(0)
    pso.reset({"uniform_pos": rng.random((4, 1, 3, 1)), "uniform_vel": rng.random((4, 1, 3, 1))})
    assert np.all(np.isneginf(pso.pbest_r)) and np.all(np.abs(pso.vel) <= 0.04 + 1e-6)


def test_policy_act_unbatches_1d_observation():
    ev = _pend_eval()
    rs = O.RandomSearch(ev, [-2.0], [2.0], horizon=4, population=16, num_agents=2)
    u01 = np.random.default_rng(0).random((16, 2, 4, 1)).astype(F)
    a, n, r = O.policy_act(rs, np.array([1.0, 0.0, 0.0], F), {"uniform": u01})
    assert a.shape == (1,) and n.shape == (3,) and np.ndim(r) == 0
