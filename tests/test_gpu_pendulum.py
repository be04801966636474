"""GPU parity tests (call through the C ABI): analytic pendulum path, evaluator + RandomSearch / CEM / PI2.

Tolerances (fp32, stated per SURVEY.md 8c): the oracle uses correctly-rounded transcendentals; the device
libm (ocml sin/cos/atan2/exp) is within 1-2 ulp of that and the error feeds an H-step recurrence, so
  per-step state / reward : rtol 1e-5, atol 1e-5
  H-step summed rewards   : rtol 2e-4, atol 2e-3
  refit mean / variance   : atol 2e-5 of the action range given the same elite set
Elite SETS are compared with a tie tolerance: where the HIP and oracle top-k differ, the swapped members'
rewards must sit within the reward tolerance of the k-th value, and the oracle is continued with the HIP
elite set so that later iterations stay comparable (lock-step)."""
import math

import numpy as np
import pytest

from oracle import oracle_np as O
from tests import philox_np as P

pytestmark = pytest.mark.gpu
F = np.float32
LO, HI = [-2.0], [2.0]
R_RTOL, R_ATOL = 2e-4, 2e-3


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1, "no gfx950 device visible"
    return _lib


def _oracle_eval():
    return O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))


_MODE = {"quirks": 0}


@pytest.fixture(params=["fused-angle", "fused-strict", "periter-angle", "periter-strict"], autouse=True)
def kernel_path(request, monkeypatch):
    """every test runs on both device paths (the persistent one-launch-per-control-step kernel and the
    per-iteration rollout + refit kernels) and with both formulations of the pendulum recurrence (the
    default angle-carried one and the op-for-op BBMPC_STRICT_MATH one): same tolerances for all four."""
    path, math = request.param.split("-")
    monkeypatch.setenv("BBMPC_FUSED", "1" if path == "fused" else "0")
    _MODE["quirks"] = (1 << 9) if math == "strict" else 0
    yield request.param
    _MODE["quirks"] = 0


def _engine(L, opt, A, H, N=0, iters=0, k=0, **kw):
    from blackbox_mpc_amd.engine import Engine
    kw["quirks"] = kw.get("quirks", 0) | _MODE["quirks"]
    return Engine(opt, L.DYN_PENDULUM, L.REW_PENDULUM, LO, HI, dim_s=3, num_agents=A, planning_horizon=H,
                  population_size=N, max_iterations=iters, num_elite=k, **kw)


# ------------------------------------------------------------------------------------------------
def test_single_step_known_answers(L):
    eng = _engine(L, L.OPT_NONE, 1, 1)
    kats = [((1.0, 0.0, 0.0), 2.0, (0.9998875, 0.01499944, 0.30000007), -0.00109),
            ((-1.0, 0.0, 0.0), -2.0, (-0.9998875, 0.01499945, -0.30000016), -9.870695),
            ((0.0, 1.0, 1.0), 0.5, (-0.09112353, 0.9958396, 1.825), -2.5717313),
            ((math.cos(3.0), math.sin(3.0), 7.9), 2.0, (-0.96277755, -0.27029496, 8.0), -15.306002)]
    s = np.array([k[0] for k in kats], F)
    a = np.array([[k[1]] for k in kats], F)
    nxt = eng.predict_next_state(s, a)
    rew = eng.evaluate_next_reward(s, nxt, a)
    np.testing.assert_allclose(nxt, np.array([k[2] for k in kats], F), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rew, np.array([k[3] for k in kats], F), rtol=1e-5, atol=1e-6)


def test_single_step_random_batch_vs_oracle(L):
    eng = _engine(L, L.OPT_NONE, 1, 1)
    rng = np.random.default_rng(3)
    th = rng.uniform(-np.pi, np.pi, 4096)
    s = np.stack([np.cos(th), np.sin(th), rng.uniform(-8, 8, 4096)], 1).astype(F)
    a = rng.uniform(-2, 2, (4096, 1)).astype(F)
    ev = _oracle_eval()
    nxt_o = ev.predict_next_state(s, a)
    nxt = eng.predict_next_state(s, a)
    np.testing.assert_allclose(nxt, nxt_o, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(eng.evaluate_next_reward(s, nxt_o, a), ev.evaluate_next_reward(s, nxt_o, a),
                               rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,A,H", [(200, 1, 20), (500, 1, 30), (257, 3, 7), (1, 1, 1), (64, 5, 50), (1000, 8, 30)])
def test_evaluator_matches_oracle(L, N, A, H):
    eng = _engine(L, L.OPT_NONE, A, H)
    rng = np.random.default_rng(N * 131 + A * 7 + H)
    states = O.pendulum_start_states(A)
    seq = rng.uniform(-2, 2, (N, A, H, 1)).astype(F)
    got = eng.evaluate(states, seq)
    want = _oracle_eval()(states, seq)
    assert got.shape == (N, A)
    from tests.parity_util import assert_pendulum_rewards
    assert_pendulum_rewards(got, want, states, seq, R_RTOL, R_ATOL, what="N=%d A=%d H=%d" % (N, A, H))


def test_evaluator_nan_guard_and_empty_population(L):
    eng = _engine(L, L.OPT_NONE, 2, 5)
    states = np.array([[np.nan, 0.0, 0.0], [1.0, 0.0, 0.0]], F)
    seq = np.zeros((8, 2, 5, 1), F)
    r = eng.evaluate(states, seq)
    assert np.all(r[:, 0] == F(-1e6)) and np.all(np.isfinite(r[:, 1])) and np.all(r[:, 1] > -1e5)
    assert eng.evaluate(states, np.zeros((0, 2, 5, 1), F)).shape == (0, 2)


def test_evaluator_linearity_property_full_size(L):
    # size-independent property at BASELINE config-3 size: agents are independent rows, so evaluating
    # all agents together equals evaluating each agent alone (bit-exact), and particle order is irrelevant.
    N, A, H = 1000, 64, 30
    eng = _engine(L, L.OPT_NONE, A, H)
    eng1 = _engine(L, L.OPT_NONE, 1, H)
    rng = np.random.default_rng(11)
    states = O.pendulum_start_states(A)
    seq = rng.uniform(-2, 2, (N, A, H, 1)).astype(F)
    full = eng.evaluate(states, seq)
    for a in (0, 17, 63):
        np.testing.assert_array_equal(full[:, a], eng1.evaluate(states[a:a + 1], seq[:, a:a + 1])[:, 0])
    perm = rng.permutation(N)
    np.testing.assert_array_equal(eng.evaluate(states, seq[perm]), full[perm])
    # and a 64-particle sample of it against the oracle
    sub = rng.choice(N, 64, replace=False)
    np.testing.assert_allclose(full[sub], _oracle_eval()(states, seq[sub]), rtol=R_RTOL, atol=R_ATOL)


# ------------------------------------------------------------------------------------------------
def test_engine_noise_matches_documented_philox_scheme(L):
    N, A, H = 96, 3, 7
    eng = _engine(L, L.OPT_CEM, A, H, N=N, iters=2, k=8, seed=0x1234567890ABCDEF, agent_offset=5)
    for step, it in [(0, 0), (3, 1)]:
        w = P.words(0x1234567890ABCDEF, step, L.NOISE_TRUNC_NORMAL, it, N, A, H, agent_offset=5)
        got = eng.dump_noise(L.NOISE_TRUNC_NORMAL, step, it, (N, A, H, 1))[..., 0]
        np.testing.assert_allclose(got, P.trunc_normal(w), rtol=0, atol=3e-7)        # the documented table sampler
        np.testing.assert_allclose(got, P.trunc_normal_exact(w), rtol=0, atol=2.5e-5)   # ... tracks the exact quantile
        w = P.words(0x1234567890ABCDEF, step, L.NOISE_UNIFORM, it, N, A, H, agent_offset=5)
        got = eng.dump_noise(L.NOISE_UNIFORM, step, it, (N, A, H, 1))[..., 0]
        np.testing.assert_array_equal(got, P.uniform(w))


def test_truncated_normal_distribution(L):
    eng = _engine(L, L.OPT_CEM, 4, 30, N=2000, iters=1, k=8, seed=7)
    z = eng.dump_noise(L.NOISE_TRUNC_NORMAL, 0, 0, (2000, 4, 30, 1)).astype(np.float64).ravel()
    assert np.all(np.abs(z) < 2.0)
    assert abs(z.mean()) < 0.01
    assert abs(z.std() - 0.8796) < 0.01          # std of N(0,1) truncated to |z|<2
    u = eng.dump_noise(L.NOISE_UNIFORM, 0, 0, (2000, 4, 30, 1)).astype(np.float64).ravel()
    assert u.min() > 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.005 and abs(u.var() - 1 / 12) < 0.002


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,A,H", [(200, 1, 20), (333, 4, 9)])
def test_random_search_injected_noise(L, N, A, H):
    eng = _engine(L, L.OPT_RANDOM_SEARCH, A, H, N=N)
    eng.set_trace(True)
    rng = np.random.default_rng(5)
    u01 = rng.random((N, A, H, 1)).astype(F)
    eng.inject_noise(L.NOISE_UNIFORM, u01)
    states = O.pendulum_start_states(A)
    act, nxt, rew = eng.optimize(states)
    rs = O.RandomSearch(_oracle_eval(), LO, HI, horizon=H, population=N, num_agents=A)
    act_o, nxt_o, rew_o = rs.call(states, {"uniform": u01})
    np.testing.assert_array_equal(eng.get_trace(0, L.TRACE_SAMPLES), rs.trace[0]["samples"])
    r_hip = eng.get_trace(0, L.TRACE_REWARDS)
    np.testing.assert_allclose(r_hip, rs.trace[0]["rewards"], rtol=R_RTOL, atol=R_ATOL)
    best = eng.get_trace(0, L.TRACE_ELITES)
    for a in range(A):       # argmax may differ only between near-tied maxima
        assert best[a] == np.argmax(r_hip[:, a])
        assert rs.trace[0]["rewards"][best[a], a] >= rs.trace[0]["rewards"][:, a].max() - R_ATOL * 2
    if np.array_equal(best, rs.trace[0]["best"]):
        np.testing.assert_array_equal(act, act_o)
        np.testing.assert_allclose(nxt, nxt_o, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(rew, rew_o, rtol=1e-5, atol=1e-5)


def _cem_lockstep(L, eng, states, noise, N, A, H, iters, k, alpha):
    """Oracle CEM run in lock-step with the HIP trace: rewards must agree within tolerance every iteration;
    where the elite sets differ, the swapped members must be near-ties of the k-th reward, and the oracle
    continues with the HIP elite set."""
    hip_el = [eng.get_trace(it, L.TRACE_ELITES) for it in range(iters)]
    hip_r = [eng.get_trace(it, L.TRACE_REWARDS) for it in range(iters)]

    def select(it, r_o, own):
        np.testing.assert_allclose(hip_r[it], r_o, rtol=R_RTOL, atol=R_ATOL)
        for a in range(A):
            he = hip_el[it][a]
            if set(own[a]) != set(he):
                kth = np.sort(r_o[:, a])[::-1][k - 1]
                for n in set(own[a]) ^ set(he):
                    assert abs(r_o[n, a] - kth) <= R_ATOL + R_RTOL * abs(kth), \
                        "elite sets differ beyond the tie tolerance (iter %d agent %d)" % (it, a)
            # HIP elites are the exact sorted top-k of the HIP rewards (top_k sorted=True, ties -> lower index)
            np.testing.assert_array_equal(he, O.topk_desc(hip_r[it][:, a], k))
        return hip_el[it]

    cem = O.CEM(_oracle_eval(), LO, HI, horizon=H, max_iterations=iters, population=N, num_elite=k, num_agents=A,
                alpha=alpha)
    cem._optimize(states, noise, forced_elites=select)
    return cem


@pytest.mark.parametrize("N,A,H,iters,k", [(500, 1, 30, 5, 50), (200, 3, 12, 3, 20), (64, 2, 5, 2, 64), (300, 2, 9, 3, 100),
                                           (1500, 1, 6, 2, 70)])
def test_cem_injected_noise_lockstep(L, N, A, H, iters, k):
    alpha = 0.25
    eng = _engine(L, L.OPT_CEM, A, H, N=N, iters=iters, k=k, alpha=alpha)
    eng.set_trace(True)
    rng = np.random.default_rng(17 + N)
    noise = {"trunc": [O.truncated_normal_noise(rng, (N, A, H, 1)) for _ in range(iters)]}
    eng.inject_noise(L.NOISE_TRUNC_NORMAL, np.stack(noise["trunc"]))
    states = O.pendulum_start_states(A)
    act, nxt, rew = eng.optimize(states)
    cem = _cem_lockstep(L, eng, states, noise, N, A, H, iters, k, alpha)
    for it in range(iters):
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_SAMPLES), cem.trace[it]["samples"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_MEAN), cem.trace[it]["mean"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_VAR), cem.trace[it]["var"], rtol=1e-5, atol=2e-5)
    act_o = cem.trace[-1]["mean"][:, 0]
    np.testing.assert_allclose(act, act_o, rtol=0, atol=2e-5)
    ev = _oracle_eval()
    nxt_o = ev.predict_next_state(states, act)
    np.testing.assert_allclose(nxt, nxt_o, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rew, ev.evaluate_next_reward(states, nxt_o, act), rtol=1e-5, atol=1e-5)
    # samples always inside the bounds (constrained variance + |xi|<2, cem.py:81-94)
    for it in range(iters):
        s = eng.get_trace(it, L.TRACE_SAMPLES)
        assert s.min() >= -2.0 and s.max() <= 2.0


def test_cem_no_warm_start_quirk_q2(L):
    # identical state + identical injected noise on consecutive control steps => identical action,
    # because the reference never re-assigns the mean/variance Variables (cem.py:129-134)
    N, A, H, iters, k = 128, 2, 8, 3, 16
    eng = _engine(L, L.OPT_CEM, A, H, N=N, iters=iters, k=k)
    rng = np.random.default_rng(2)
    eng.inject_noise(L.NOISE_TRUNC_NORMAL, np.stack([O.truncated_normal_noise(rng, (N, A, H, 1)) for _ in range(iters)]))
    s = O.pendulum_start_states(A)
    a1, _, _ = eng.optimize(s)
    a2, _, _ = eng.optimize(s)
    np.testing.assert_array_equal(a1, a2)
    eng.reset()
    a3, _, _ = eng.optimize(s)
    np.testing.assert_array_equal(a1, a3)


@pytest.mark.parametrize("N,A,H,iters,lam", [(300, 2, 10, 3, 1.0), (1000, 4, 30, 5, 1.0), (100, 1, 4, 2, 0.5)])
def test_pi2_injected_noise(L, N, A, H, iters, lam):
    # Lock-step (refit tolerance 1e-4 x range = 4e-4; held to 2e-5 here): every iteration's rewards are
    # compared within the rollout tolerance, then the oracle carries on with the device's values, so the exp-weighted
    # means (pi2.py:78-87) and the shifted warm start (pi2.py:92-93) are compared on identical inputs.  A second,
    # free-running oracle is compared as well: the softmin amplifies reward differences (d(omega)/omega ~ d(reward)/
    # lambda), so that comparison carries the tolerance that follows from the reward tolerance, and prints what it saw.
    eng = _engine(L, L.OPT_PI2, A, H, N=N, iters=iters, lamda=lam)
    eng.set_trace(True)
    rng = np.random.default_rng(23 + N)
    states = O.pendulum_start_states(A)
    pi2 = O.PI2(_oracle_eval(), LO, HI, horizon=H, max_iterations=iters, population=N, num_agents=A, lamda=lam)
    free = O.PI2(_oracle_eval(), LO, HI, horizon=H, max_iterations=iters, population=N, num_agents=A, lamda=lam)
    worst_free = 0.0
    for step in range(2):        # second control step exercises the shift-left warm start (pi2.py:92-93)
        noise = {"trunc": [O.truncated_normal_noise(rng, (N, A, H, 1)) for _ in range(iters)]}
        eng.inject_noise(L.NOISE_TRUNC_NORMAL, np.stack(noise["trunc"]))
        act, nxt, rew = eng.optimize(states)
        hip_r = [eng.get_trace(it, L.TRACE_REWARDS) for it in range(iters)]

        def lock(it, r_o):
            np.testing.assert_allclose(hip_r[it], r_o, rtol=R_RTOL, atol=R_ATOL)
            return hip_r[it]
        act_o = pi2._optimize(states, noise, rewards_override=lock)
        for it in range(iters):
            np.testing.assert_allclose(eng.get_trace(it, L.TRACE_SAMPLES), pi2.trace[it]["samples"], rtol=0, atol=2e-5)
            np.testing.assert_allclose(eng.get_trace(it, L.TRACE_MEAN), pi2.trace[it]["mean"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(act, act_o, rtol=0, atol=2e-5)
        np.testing.assert_allclose(eng.get_state("prev_mean"), pi2.prev, rtol=0, atol=2e-5)
        assert np.all(np.abs(eng.get_trace(iters - 1, L.TRACE_SAMPLES)) <= 2.0)
        # free-running: nothing carried over
        act_f = free._optimize(states, noise)
        for it in range(iters):
            np.testing.assert_allclose(hip_r[it], free.trace[it]["rewards"], rtol=R_RTOL, atol=R_ATOL)
            worst_free = max(worst_free, float(np.abs(eng.get_trace(it, L.TRACE_MEAN) - free.trace[it]["mean"]).max()))
        worst_free = max(worst_free, float(np.abs(act - act_f).max()), float(np.abs(eng.get_state("prev_mean") - free.prev).max()))
        np.testing.assert_allclose(act, act_f, rtol=0, atol=4e-3 / lam)
        np.testing.assert_allclose(eng.get_state("prev_mean"), free.prev, rtol=0, atol=4e-3 / lam)
    print(f"[pi2 pendulum N={N} A={A} H={H} lambda={lam}] lock-step held at 2e-5; free-running max |mean - oracle| = {worst_free:.3e}")


def test_pi2_refit_exact_given_rewards(L):
    # the softmin refit alone, checked tightly: H=1 so the rollout reward is a smooth function of one action
    N, A, H = 256, 3, 1
    eng = _engine(L, L.OPT_PI2, A, H, N=N, iters=1, lamda=1.0)
    eng.set_trace(True)
    rng = np.random.default_rng(9)
    xi = O.truncated_normal_noise(rng, (N, A, H, 1))
    eng.inject_noise(L.NOISE_TRUNC_NORMAL, xi[None])
    states = O.pendulum_start_states(A)
    act, _, _ = eng.optimize(states)
    r = eng.get_trace(0, L.TRACE_REWARDS).astype(np.float64)          # HIP rewards -> fp64 softmin reference
    x = eng.get_trace(0, L.TRACE_SAMPLES).astype(np.float64)
    c = -r
    p = np.exp(-(c - c.min(axis=0, keepdims=True)))
    w = p / p.sum(axis=0, keepdims=True)
    ref = (x[:, :, 0, 0] * w).sum(axis=0)
    np.testing.assert_allclose(act[:, 0], ref, rtol=2e-6, atol=2e-6)


def test_engine_generated_noise_end_to_end(L):
    # production mode (no injection): feed the engine's own documented draws to the oracle
    N, A, H, iters, k = 500, 2, 30, 5, 50
    seed = 42
    eng = _engine(L, L.OPT_CEM, A, H, N=N, iters=iters, k=k, seed=seed)
    eng.set_trace(True)
    states = O.pendulum_start_states(A)
    eng.optimize(states)          # control step 0
    act, nxt, rew = eng.optimize(states)   # control step 1 uses different draws
    noise = {"trunc": [eng.dump_noise(L.NOISE_TRUNC_NORMAL, 1, it, (N, A, H, 1)) for it in range(iters)]}
    cem = _cem_lockstep(L, eng, states, noise, N, A, H, iters, k, 0.25)
    np.testing.assert_allclose(act, cem.trace[-1]["mean"][:, 0], rtol=0, atol=2e-5)
    a0 = eng.dump_noise(L.NOISE_TRUNC_NORMAL, 0, 0, (N, A, H, 1))
    assert not np.array_equal(a0, noise["trunc"][0])


def test_exploration_noise_quirk_q7(L):
    N, A, H = 64, 3, 4
    eng = _engine(L, L.OPT_RANDOM_SEARCH, A, H, N=N)
    rng = np.random.default_rng(4)
    u01 = rng.random((N, A, H, 1)).astype(F)
    xi = O.truncated_normal_noise(rng, (A, 1))
    eng.inject_noise(L.NOISE_UNIFORM, u01)
    eng.inject_noise(L.NOISE_EXPLORATION, xi)
    states = O.pendulum_start_states(A)
    base, _, _ = eng.optimize(states, add_exploration_noise=False)
    act, nxt, rew = eng.optimize(states, add_exploration_noise=True)
    sd = math.sqrt((4.0 ** 2) / 16 * 0.05)
    np.testing.assert_allclose(act, np.clip(base + xi * F(sd) + 0.0, -2, 2), rtol=1e-6, atol=1e-6)   # midpoint = 0 here
    ev = _oracle_eval()
    np.testing.assert_allclose(nxt, ev.predict_next_state(states, act), rtol=1e-5, atol=1e-5)


def test_agent_sharding_is_bit_identical(L):
    # agents are independent and RNG is keyed by GLOBAL agent id: 2 shards == 1 unsharded engine
    N, A, H, iters, k = 200, 4, 10, 3, 20
    states = O.pendulum_start_states(A)
    full = _engine(L, L.OPT_CEM, A, H, N=N, iters=iters, k=k, seed=99)
    a_full, n_full, r_full = full.optimize(states)
    for off in (0, 2):
        sh = _engine(L, L.OPT_CEM, 2, H, N=N, iters=iters, k=k, seed=99, agent_offset=off, num_agents_global=A)
        a, n, r = sh.optimize(states[off:off + 2])
        np.testing.assert_array_equal(a, a_full[off:off + 2])
        np.testing.assert_array_equal(n, n_full[off:off + 2])
        np.testing.assert_array_equal(r, r_full[off:off + 2])


def test_mpc_policy_drop_in_api(L):
    from blackbox_mpc_amd.policies import MPCPolicy
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function
    act_space, obs_space = Box([-2.0], [2.0]), Box([-1, -1, -8], [1, 1, 8])
    pol = MPCPolicy(reward_function=pendulum_reward_function, env_action_space=act_space,
                    env_observation_space=obs_space, true_model=True, dynamics_function=PendulumTrueModel(),
                    optimizer_name="RandomSearch", num_agents=1, planning_horizon=20, population_size=200)
    obs = np.array([1.0, 0.0, 0.0])
    a, n, r = pol.act(obs, 0)
    assert a.shape == (1,) and n.shape == (3,) and np.ndim(r) == 0 and -2 <= a[0] <= 2
    pol.switch_optimizer(optimizer_name="CEM", planning_horizon=30, population_size=500, max_iterations=5, num_elite=50)
    pol.reset()
    ret = 0.0
    for t in range(60):        # closed loop: swing-up should make progress (reward improves over the episode)
        a, n, r = pol.act(obs, t)
        obs = O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True)).predict_next_state(
            obs[None].astype(F), a[None].astype(F))[0]
        ret += float(r)
    assert np.isfinite(ret)
    a2, n2, r2 = pol.act(np.tile(obs[None], (1, 1)), 61)
    assert a2.shape == (1, 1) and n2.shape == (1, 3) and r2.shape == (1,)
    # direct plugin calls run the same device code
    d = PendulumTrueModel()(np.array([[1.0, 0.0, 0.0, 2.0]], F))
    np.testing.assert_allclose(d[0], np.array([0.9998875 - 1, 0.01499944, 0.30000007]), atol=1e-6)
    np.testing.assert_allclose(pendulum_reward_function(np.array([[0, 1, 2]], F), np.zeros((1, 3), F),
                                                        np.array([[1.5]], F))[0],
                               -((math.pi / 2) ** 2 + 0.4) - 0.001 * 2.25, rtol=1e-5)


def test_closed_loop_harness_matches_step_by_step(L):
    # f-1: the on-device episode == the host loop policy.act -> env.step with the model as environment
    from blackbox_mpc_amd.policies import MPCPolicy
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function
    from blackbox_mpc_amd.utils.rollouts import ModelEnvironment, perform_rollouts, rollout_on_device
    A, T = 3, 12
    mk = lambda: MPCPolicy(reward_function=pendulum_reward_function, env_action_space=Box([-2.0], [2.0]),
                           env_observation_space=Box([-1, -1, -8], [1, 1, 8]), true_model=True,
                           dynamics_function=PendulumTrueModel(), optimizer_name="PI2", num_agents=A,
                           planning_horizon=15, population_size=128, max_iterations=2, seed=5,
                           quirks=_MODE["quirks"])
    start = O.pendulum_start_states(A)
    pol = mk()
    env = ModelEnvironment(pol._trajectory_evaluator, start)
    obs, acs, rews = perform_rollouts(env, 1, T, pol)
    assert obs[0].shape == (T + 1, A, 3) and acs[0].shape == (T, A, 1) and rews[0].shape == (T, A)
    pol2 = mk()
    a, o, r = rollout_on_device(pol2, start, T)
    np.testing.assert_allclose(a, acs[0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(o, obs[0], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r, rews[0], rtol=1e-5, atol=1e-5)
    assert np.all(np.isfinite(r)) and np.all(np.abs(a) <= 2.0)


@pytest.mark.parametrize("opt_name", ["CEM", "PI2", "RandomSearch"])
def test_degenerate_sizes(L, opt_name):
    # N = 1 particle, k = N, H = 1, and max_iterations = 0 (the reference's while_loop then never runs:
    # the action is the untouched initial mean's first entry, cem.py:129-136 / pi2.py:90-94)
    opt = {"CEM": L.OPT_CEM, "PI2": L.OPT_PI2, "RandomSearch": L.OPT_RANDOM_SEARCH}[opt_name]
    states = O.pendulum_start_states(2)
    ev = _oracle_eval()
    eng = _engine(L, opt, 2, 1, N=1, iters=2, k=1)
    a, n, r = eng.optimize(states)
    assert a.shape == (2, 1) and np.all(np.isfinite(a)) and np.all(np.abs(a) <= 2.0)
    np.testing.assert_allclose(n, ev.predict_next_state(states, a), rtol=1e-5, atol=1e-5)
    if opt_name != "RandomSearch":
        eng0 = _engine(L, opt, 2, 7, N=33, iters=0, k=33)
        a0, n0, _ = eng0.optimize(states)
        np.testing.assert_array_equal(a0, np.zeros((2, 1), F))        # bounds midpoint
        np.testing.assert_allclose(n0, ev.predict_next_state(states, a0), rtol=1e-5, atol=1e-5)
    # k == N with an odd population: every particle is an elite
    if opt_name == "CEM":
        N, H = 37, 5
        eng = _engine(L, opt, 1, H, N=N, iters=1, k=N)
        eng.set_trace(True)
        rng = np.random.default_rng(8)
        xi = O.truncated_normal_noise(rng, (N, 1, H, 1))
        eng.inject_noise(L.NOISE_TRUNC_NORMAL, xi[None])
        a, _, _ = eng.optimize(states[:1])
        cem = O.CEM(ev, LO, HI, horizon=H, max_iterations=1, population=N, num_elite=N, num_agents=1)
        a_o, _, _ = cem.call(states[:1], {"trunc": [xi]})
        assert sorted(eng.get_trace(0, L.TRACE_ELITES)[0].tolist()) == list(range(N))
        np.testing.assert_allclose(a, a_o, rtol=0, atol=2e-5)


def test_large_population_uses_global_samples(L):
    # N*H too large for the LDS-resident sample block (fused path falls back to the HBM scratch) and
    # N > 2048 with few agents (auto mode picks the per-iteration kernels): same answers as the oracle
    N, A, H, iters, k = 4096, 1, 40, 2, 64
    eng = _engine(L, L.OPT_CEM, A, H, N=N, iters=iters, k=k)
    eng.set_trace(True)
    rng = np.random.default_rng(12)
    noise = {"trunc": [O.truncated_normal_noise(rng, (N, A, H, 1)) for _ in range(iters)]}
    eng.inject_noise(L.NOISE_TRUNC_NORMAL, np.stack(noise["trunc"]))
    states = O.pendulum_start_states(A)
    act, nxt, rew = eng.optimize(states)
    cem = _cem_lockstep(L, eng, states, noise, N, A, H, iters, k, 0.25)
    np.testing.assert_allclose(act, cem.trace[-1]["mean"][:, 0], rtol=0, atol=2e-5)


@pytest.mark.parametrize("opt_name,N,A,H", [("CEM", 500, 1, 30), ("PI2", 300, 3, 17), ("RandomSearch", 200, 2, 20)])
def test_noise_prefetch_is_bit_identical_to_in_kernel_draws(L, monkeypatch, opt_name, N, A, H):
    # The persistent kernel reads draws that idle CUs generated one control step ahead (side stream); they come from
    # the same Philox counters, so switching the prefetch off must not change a single bit -- also across reset().
    opt = {"CEM": L.OPT_CEM, "PI2": L.OPT_PI2, "RandomSearch": L.OPT_RANDOM_SEARCH}[opt_name]
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("BBMPC_NOISE_PREFETCH", mode)
        eng = _engine(L, opt, A, H, N=N, iters=3, k=max(N // 10, 1), seed=11)
        s = O.pendulum_start_states(A)
        out = []
        for t in range(21):                # crosses two 8-step prefetch chunks
            if t == 11:
                eng.reset()
            a, s, r = eng.optimize(s)
            out.append(np.concatenate([a.ravel(), s.ravel(), r.ravel()]))
        runs[mode] = np.stack(out)
    np.testing.assert_array_equal(runs["0"], runs["1"])
