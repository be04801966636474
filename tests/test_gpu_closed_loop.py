"""f-1: closed-loop episodes against the ORACLE's closed loop (not against the engine itself).

Counterpart of the reference's utils/rollouts.py:85-102 (`policy.reset(); obs = env.reset(); for t: policy.act(obs, t)
-> env.step(action)`), with the model as the environment on both sides:

    device : bbmpc_rollout_episode (whole episode inside the engine) and, step by step, bbmpc_optimize
    oracle : oracle_np.policy_act (MPCPolicy.act + Optimizer.__call__) and the oracle's own pendulum step as env,
             FREE RUNNING from the common start state -- its observation at step t is its own state, never the device's

Both sides consume the same draws: the engine's Philox scheme is dumped per (control step, iteration) and fed to the
oracle.  Warm starts are live (PI2 shift-left of the mean, PSO re-seeded swarm), >= 20 control steps.

Tolerances (written here, as VERDICT r1 item 8 asks): actions within 5e-3, states within 1e-4 at every step.  The
sampled populations make the action a discontinuous function of the rewards for PSO (argmax, pso.py:97) and a sharply
peaked soft-min for PI2 (pi2.py:80-87), so -- as everywhere in this suite -- every iteration's rewards are compared
within the evaluator tolerance and the device's values are carried into the oracle's refit (lock-step): the comparison
then measures the refit, warm start and environment arithmetic over the episode instead of tie-breaking luck.  PI2 is
ALSO run without any carrying (free refit), with the tolerance that follows from the reward tolerance."""
import numpy as np
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
F = np.float32
LO, HI = [-2.0], [2.0]
R_RTOL, R_ATOL = 2e-4, 2e-3
ACT_TOL, STATE_TOL = 5e-3, 1e-4


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1
    return _lib


def _engine(L, opt, A, H, N, iters, seed, **kw):
    from blackbox_mpc_amd.engine import Engine
    return Engine(opt, L.DYN_PENDULUM, L.REW_PENDULUM, LO, HI, dim_s=3, num_agents=A, planning_horizon=H,
                  population_size=N, max_iterations=iters, seed=seed, **kw)


def _ev():
    return O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))


@pytest.mark.parametrize("fused", ["1", "0"])
def test_pi2_episode_against_oracle_closed_loop(L, monkeypatch, fused):
    monkeypatch.setenv("BBMPC_FUSED", fused)
    N, A, H, iters, T, seed = 256, 2, 12, 3, 24, 17
    start = O.pendulum_start_states(A)
    # whole episode on the device
    ep = _engine(L, L.OPT_PI2, A, H, N, iters, seed)
    ep.reset()
    acts, nexts, rews = ep.rollout_episode(start, T)
    # the same episode step by step (per-iteration rewards are only traceable this way), bit-identical to the above
    eng = _engine(L, L.OPT_PI2, A, H, N, iters, seed)
    eng.set_trace(True)
    eng.reset()
    lock = O.PI2(_ev(), LO, HI, horizon=H, max_iterations=iters, population=N, num_agents=A)
    free = O.PI2(_ev(), LO, HI, horizon=H, max_iterations=iters, population=N, num_agents=A)
    lock.reset()
    free.reset()
    obs_d, obs_l, obs_f = start.copy(), start.copy(), start.copy()
    worst = dict(act=0.0, state=0.0, act_free=0.0, state_free=0.0)
    for t in range(T):
        a, n, r = eng.optimize(obs_d, t)
        np.testing.assert_array_equal(a, acts[t])
        np.testing.assert_array_equal(n, nexts[t])
        np.testing.assert_array_equal(r, rews[t])
        noise = {"trunc": [eng.dump_noise(L.NOISE_TRUNC_NORMAL, t, it, (N, A, H, 1)) for it in range(iters)]}
        hip_r = [eng.get_trace(it, L.TRACE_REWARDS) for it in range(iters)]

        def carry(it, r_o):
            np.testing.assert_allclose(hip_r[it], r_o, rtol=R_RTOL, atol=R_ATOL)
            return hip_r[it]
        # oracle policy.act on ITS OWN observation, then the oracle's env step
        act_o = lock._optimize(obs_l, noise, rewards_override=carry)
        nxt_o = lock.ev.predict_next_state(obs_l, act_o)
        rew_o = lock.ev.evaluate_next_reward(obs_l, nxt_o, act_o)
        np.testing.assert_allclose(a, act_o, rtol=0, atol=ACT_TOL)
        np.testing.assert_allclose(n, nxt_o, rtol=0, atol=STATE_TOL)
        np.testing.assert_allclose(r, rew_o, rtol=1e-4, atol=1e-3)
        worst["act"] = max(worst["act"], float(np.abs(a - act_o).max()))
        worst["state"] = max(worst["state"], float(np.abs(n - nxt_o).max()))
        a_f, n_f, _ = O.policy_act(free, obs_f, noise)
        worst["act_free"] = max(worst["act_free"], float(np.abs(a - a_f).max()))
        worst["state_free"] = max(worst["state_free"], float(np.abs(n - n_f).max()))
        obs_d, obs_l, obs_f = n, nxt_o, n_f
    print("\n[f-1 PI2 fused=%s] worst deviations over %d steps: %s" % (fused, T, worst))
    # lock-step is far inside the stated tolerance; the free refit drifts with the reward tolerance (a reward error e
    # moves a soft-min weight by ~e/lambda), compounding through the warm start and the state
    assert worst["act"] < 1e-3 and worst["state"] < STATE_TOL
    assert worst["act_free"] < 5e-2 and worst["state_free"] < 1e-2


@pytest.mark.parametrize("fused", ["1", "0"])
def test_pso_episode_against_oracle_closed_loop(L, monkeypatch, fused):
    monkeypatch.setenv("BBMPC_FUSED", fused)
    N, A, H, iters, T, seed = 192, 2, 10, 3, 22, 23
    shp = (N, A, H, 1)
    start = O.pendulum_start_states(A)
    ep = _engine(L, L.OPT_PSO, A, H, N, iters, seed)
    ep.reset()                                           # utils/rollouts.py:87 policy.reset() -> pso.py:143-160
    acts, nexts, rews = ep.rollout_episode(start, T)
    eng = _engine(L, L.OPT_PSO, A, H, N, iters, seed)
    eng.set_trace(True)
    eng.reset()
    pso = O.PSO(_ev(), LO, HI, horizon=H, max_iterations=iters, population=N, num_agents=A)
    pso.reset({"uniform_pos": eng.dump_noise(L.NOISE_PSO_RESET_POS, 0, 0xFFFF, shp),
               "uniform_vel": eng.dump_noise(L.NOISE_PSO_RESET_VEL, 0, 0xFFFF, shp)})
    np.testing.assert_allclose(eng.get_state("pos", shp), pso.pos, rtol=0, atol=1e-6)
    obs_d, obs_o = start.copy(), start.copy()
    worst = dict(act=0.0, state=0.0)
    for t in range(T):
        a, n, r = eng.optimize(obs_d, t)
        np.testing.assert_array_equal(a, acts[t])
        np.testing.assert_array_equal(n, nexts[t])
        np.testing.assert_array_equal(r, rews[t])
        noise = {"normal2": np.stack([eng.dump_noise(L.NOISE_PSO_SCALARS, t, it, (2,)) for it in range(iters)]),
                 "trunc": eng.dump_noise(L.NOISE_PSO_RESEED_TRUNC, t, 0, shp),
                 "uniform": eng.dump_noise(L.NOISE_PSO_RESEED_UNIFORM, t, 0, shp)}
        hip_r = [eng.get_trace(it, L.TRACE_REWARDS) for it in range(iters)]

        def carry(it, r_o):
            fin = np.isfinite(r_o)
            np.testing.assert_allclose(hip_r[it][fin], r_o[fin], rtol=R_RTOL, atol=R_ATOL)
            return hip_r[it]
        act_o = pso._optimize(obs_o, noise, rewards_override=carry)
        for it in range(iters):
            np.testing.assert_array_equal(eng.get_trace(it, L.TRACE_ELITES), pso.trace[it]["gbest_idx"])
        nxt_o = pso.ev.predict_next_state(obs_o, act_o)
        rew_o = pso.ev.evaluate_next_reward(obs_o, nxt_o, act_o)
        np.testing.assert_allclose(a, act_o, rtol=0, atol=ACT_TOL)
        np.testing.assert_allclose(n, nxt_o, rtol=0, atol=STATE_TOL)
        np.testing.assert_allclose(r, rew_o, rtol=1e-4, atol=1e-3)
        worst["act"] = max(worst["act"], float(np.abs(a - act_o).max()))
        worst["state"] = max(worst["state"], float(np.abs(n - nxt_o).max()))
        obs_d, obs_o = n, nxt_o
    np.testing.assert_allclose(eng.get_state("pos", shp), pso.pos, rtol=0, atol=1e-4)     # the re-seeded swarm of step T
    print("\n[f-1 PSO fused=%s] worst deviations over %d steps: %s" % (fused, T, worst))
    assert worst["act"] < 1e-3 and worst["state"] < STATE_TOL
