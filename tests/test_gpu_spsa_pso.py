"""GPU parity tests for the SPSA and PSO optimizers (injected standard noise, oracle lock-step)."""
import numpy as np
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
F = np.float32
LO, HI = [-2.0], [2.0]
R_RTOL, R_ATOL = 2e-4, 2e-3


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1
    return _lib


def _ev():
    return O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))


def _engine(L, opt, A, H, N, iters, **kw):
    from blackbox_mpc_amd.engine import Engine
    return Engine(opt, L.DYN_PENDULUM, L.REW_PENDULUM, LO, HI, dim_s=3, num_agents=A, planning_horizon=H,
                  population_size=N, max_iterations=iters, **kw)


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("N,A,H,iters", [(500, 1, 30, 5), (130, 3, 7, 3), (64, 1, 5, 17)])     # 17 iterations: beyond the persistent kernel's table
def test_spsa_injected_noise(L, monkeypatch, N, A, H, iters, fused):
    # both device paths: the persistent one-launch kernel (two candidates per lane) and the per-iteration kernels
    monkeypatch.setenv("BBMPC_FUSED", fused)
    eng = _engine(L, L.OPT_SPSA, A, H, N, iters)
    eng.set_trace(True)
    rng = np.random.default_rng(N)
    sp = O.SPSA(_ev(), LO, HI, horizon=H, max_iterations=iters, population=N, num_agents=A)
    states = O.pendulum_start_states(A)
    for step in range(2):          # the second control step starts from the shifted solution (spsa.py:114-115)
        delta = [np.where(rng.random((N, A, H, 1)) < 0.5, -1.0, 1.0).astype(F) for _ in range(iters)]
        eng.inject_noise(L.NOISE_RADEMACHER, np.stack(delta))
        act, nxt, rew = eng.optimize(states)
        act_o, nxt_o, rew_o = sp.call(states, {"rademacher": delta})
        for it in range(iters):
            r = eng.get_trace(it, L.TRACE_REWARDS)
            np.testing.assert_allclose(r[:N], sp.trace[it]["rewards_plus"], rtol=R_RTOL, atol=R_ATOL)
            np.testing.assert_allclose(r[N:], sp.trace[it]["rewards_minus"], rtol=R_RTOL, atol=R_ATOL)
            np.testing.assert_allclose(eng.get_trace(it, L.TRACE_MEAN), sp.trace[it]["solution"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(act, act_o, rtol=0, atol=1e-4)
        np.testing.assert_allclose(eng.get_state("prev_mean"), sp.params, rtol=0, atol=1e-4)
        np.testing.assert_allclose(nxt, nxt_o, rtol=1e-4, atol=1e-4)
    assert np.all(np.abs(eng.get_state("prev_mean")) <= 2.0)


def test_spsa_fused_equals_per_iteration_with_engine_draws(L, monkeypatch):
    # production (Philox) draws: the persistent kernel and the per-iteration kernels agree bit for bit over several
    # control steps, including the shift-left warm start and a reset
    N, A, H, iters = 300, 2, 17, 4
    runs = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("BBMPC_FUSED", fused)
        eng = _engine(L, L.OPT_SPSA, A, H, N, iters, seed=21)
        s = O.pendulum_start_states(A)
        out = []
        for t in range(6):
            if t == 4:
                eng.reset()
            a, s, r = eng.optimize(s)
            out.append(np.concatenate([a.ravel(), s.ravel(), r.ravel(), eng.get_state("prev_mean").ravel()]))
        runs[fused] = np.stack(out)
    np.testing.assert_array_equal(runs["0"], runs["1"])


def test_spsa_engine_rademacher_draws(L):
    N, A, H = 4096, 2, 8
    eng = _engine(L, L.OPT_SPSA, A, H, N, 1, seed=5)
    d = eng.dump_noise(L.NOISE_RADEMACHER, 0, 0, (N, A, H, 1))
    assert set(np.unique(d)) == {-1.0, 1.0} and abs(d.mean()) < 0.02
    from tests import philox_np as P
    np.testing.assert_array_equal(d[..., 0], P.rademacher(P.words(5, 0, L.NOISE_RADEMACHER, 0, N, A, H)))


def _pso_noise(rng, N, A, H, iters):
    return {"normal2": rng.standard_normal((iters, 2)).astype(F),
            "trunc": O.truncated_normal_noise(rng, (N, A, H, 1)),
            "uniform": rng.random((N, A, H, 1)).astype(F)}


def _pso_inject(L, eng, noise):
    eng.inject_noise(L.NOISE_PSO_SCALARS, noise["normal2"])
    eng.inject_noise(L.NOISE_PSO_RESEED_TRUNC, noise["trunc"])
    eng.inject_noise(L.NOISE_PSO_RESEED_UNIFORM, noise["uniform"])


def _check_pso_state(L, eng, pso, N, A, H):
    np.testing.assert_allclose(eng.get_state("pos", (N, A, H, 1)), pso.pos, rtol=0, atol=2e-5)
    np.testing.assert_allclose(eng.get_state("vel", (N, A, H, 1)), pso.vel, rtol=0, atol=2e-5)
    np.testing.assert_allclose(eng.get_state("pbest", (N, A, H, 1)), pso.pbest, rtol=0, atol=2e-5)
    np.testing.assert_array_equal(eng.get_state("pbest_r", (N, A)), pso.pbest_r)        # all -inf after a re-seed
    np.testing.assert_array_equal(eng.get_state("gbest_r", (A,)), pso.gbest_r)


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("with_reset", [True, False])
def test_pso_injected_noise(L, monkeypatch, with_reset, fused):
    # both device paths: the persistent kernel (swarm positions / velocities in LDS) and the per-iteration kernels
    monkeypatch.setenv("BBMPC_FUSED", fused)
    N, A, H, iters = 160, 2, 9, 4
    eng = _engine(L, L.OPT_PSO, A, H, N, iters)
    eng.set_trace(True)
    rng = np.random.default_rng(11)
    pso = O.PSO(_ev(), LO, HI, horizon=H, max_iterations=iters, population=N, num_agents=A)
    if with_reset:
        rn = {"uniform_pos": rng.random((N, A, H, 1)).astype(F), "uniform_vel": rng.random((N, A, H, 1)).astype(F)}
        eng.inject_noise(L.NOISE_PSO_RESET_POS, rn["uniform_pos"])
        eng.inject_noise(L.NOISE_PSO_RESET_VEL, rn["uniform_vel"])
        eng.reset()
        pso.reset(rn)
        _check_pso_state(L, eng, pso, N, A, H)
    # without reset() the swarm starts from the constructor's all-zero Variables (quirk Q4)
    states = O.pendulum_start_states(A)
    for step in range(2):
        noise = _pso_noise(rng, N, A, H, iters)
        _pso_inject(L, eng, noise)
        act, nxt, rew = eng.optimize(states)
        act_o, nxt_o, rew_o = pso.call(states, noise)
        for it in range(iters):
            np.testing.assert_allclose(eng.get_trace(it, L.TRACE_REWARDS), pso.trace[it]["rewards"], rtol=R_RTOL, atol=R_ATOL)
            np.testing.assert_array_equal(eng.get_trace(it, L.TRACE_ELITES), pso.trace[it]["gbest_idx"])
            np.testing.assert_allclose(eng.get_trace(it, L.TRACE_MEAN), pso.trace[it]["gbest"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(act, act_o, rtol=0, atol=2e-5)
        np.testing.assert_allclose(nxt, nxt_o, rtol=1e-4, atol=1e-4)
        _check_pso_state(L, eng, pso, N, A, H)
        np.testing.assert_allclose(eng.get_state("gbest"), pso.gbest, rtol=0, atol=2e-5)


def test_pso_fused_equals_per_iteration_with_engine_draws(L, monkeypatch):
    # production (Philox) draws, config-2 size: the two device paths agree bit for bit over several control steps
    # (actions, predictions and the whole swarm state), with and without a reset in between
    N, A, H, iters = 500, 2, 30, 5
    runs = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("BBMPC_FUSED", fused)
        eng = _engine(L, L.OPT_PSO, A, H, N, iters, seed=31)
        eng.reset()
        s = O.pendulum_start_states(A)
        out = []
        for t in range(5):
            if t == 3:
                eng.reset()
            a, s, r = eng.optimize(s)
            out.append(np.concatenate([a.ravel(), s.ravel(), r.ravel(), eng.get_state("pos", (N, A, H, 1)).ravel(),
                                       eng.get_state("vel", (N, A, H, 1)).ravel(), eng.get_state("gbest").ravel()]))
        runs[fused] = np.stack(out)
    np.testing.assert_array_equal(runs["0"], runs["1"])


def test_pso_production_noise_is_shard_invariant(L):
    # the two scalar N(0,1) draws come from an agent-independent counter: two shards == one engine
    N, A, H, iters = 128, 4, 6, 3
    states = O.pendulum_start_states(A)
    full = _engine(L, L.OPT_PSO, A, H, N, iters, seed=3)
    full.reset()
    a_full = [full.optimize(states)[0] for _ in range(2)]
    for off in (0, 2):
        sh = _engine(L, L.OPT_PSO, 2, H, N, iters, seed=3, agent_offset=off, num_agents_global=A)
        sh.reset()
        for s in range(2):
            np.testing.assert_array_equal(sh.optimize(states[off:off + 2])[0], a_full[s][off:off + 2])


def test_spsa_and_pso_with_learned_dynamics(L):
    from blackbox_mpc_amd.engine import Engine
    S, U, N, A, H, iters = 20, 6, 64, 2, 6, 2
    ws, bs = O.make_mlp_params([26, 200, 200, 20])
    z, o = np.zeros, np.ones
    stats = [z(S, F), o(S, F), z(U, F), o(U, F), z(S, F), np.full(S, 0.1, F)]
    ev = O.Evaluator("cheetah", O.Handler(O.MLP(ws, bs, ["tanh", "tanh", None]), False, True, stats))
    lo, hi = [-1.0] * U, [1.0] * U
    states = O.cheetah_start_states(A, S)
    rng = np.random.default_rng(2)
    eng = Engine(L.OPT_SPSA, L.DYN_MLP, L.REW_CHEETAH, lo, hi, dim_s=S, num_agents=A, planning_horizon=H,
                 population_size=N, max_iterations=iters)
    eng.set_mlp(ws, bs, [1, 1, 0], stats)
    delta = [np.where(rng.random((N, A, H, U)) < 0.5, -1.0, 1.0).astype(F) for _ in range(iters)]
    eng.inject_noise(L.NOISE_RADEMACHER, np.stack(delta))
    act, _, _ = eng.optimize(states)
    sp = O.SPSA(ev, lo, hi, horizon=H, max_iterations=iters, population=N, num_agents=A)
    act_o, _, _ = sp.call(states, {"rademacher": delta})
    np.testing.assert_allclose(act, act_o, rtol=0, atol=2e-4)

    eng = Engine(L.OPT_PSO, L.DYN_MLP, L.REW_CHEETAH, lo, hi, dim_s=S, num_agents=A, planning_horizon=H,
                 population_size=N, max_iterations=iters)
    eng.set_mlp(ws, bs, [1, 1, 0], stats)
    eng.set_trace(True)
    pso = O.PSO(ev, lo, hi, horizon=H, max_iterations=iters, population=N, num_agents=A)
    rn = {"uniform_pos": rng.random((N, A, H, U)).astype(F), "uniform_vel": rng.random((N, A, H, U)).astype(F)}
    eng.inject_noise(L.NOISE_PSO_RESET_POS, rn["uniform_pos"])
    eng.inject_noise(L.NOISE_PSO_RESET_VEL, rn["uniform_vel"])
    eng.reset()
    pso.reset(rn)
    noise = {"normal2": rng.standard_normal((iters, 2)).astype(F), "trunc": O.truncated_normal_noise(rng, (N, A, H, U)),
             "uniform": rng.random((N, A, H, U)).astype(F)}
    eng.inject_noise(L.NOISE_PSO_SCALARS, noise["normal2"])
    eng.inject_noise(L.NOISE_PSO_RESEED_TRUNC, noise["trunc"])
    eng.inject_noise(L.NOISE_PSO_RESEED_UNIFORM, noise["uniform"])
    act, _, _ = eng.optimize(states)
    act_o, _, _ = pso.call(states, noise)
    for it in range(iters):
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_REWARDS), pso.trace[it]["rewards"], rtol=1e-3, atol=1e-2)
    np.testing.assert_allclose(act, act_o, rtol=0, atol=2e-5)
