"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them bit for bit (freezes the checker).  GPU: the HIP engine matches them
within the stated fp32 tolerances, through the C ABI."""
import os

import numpy as np
import pytest

from oracle import oracle_np as O

F = np.float32
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def _pend_eval():
    return O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))


# ---------------------------------------------------------------- CPU: oracle vs its frozen outputs
def test_oracle_reproduces_golden_cfg1_cfg2_cfg3():
    g = load("cfg1")
    rs = O.RandomSearch(_pend_eval(), [-2.0], [2.0], horizon=20, population=200, num_agents=1)
    a, n, r = rs.call(g["states"], {"uniform": g["uniform"]})
    np.testing.assert_array_equal(rs.trace[0]["rewards"], g["rewards"])
    np.testing.assert_array_equal(a, g["action"])
    np.testing.assert_array_equal(n, g["next_state"])
    g = load("cfg2")
    cem = O.CEM(_pend_eval(), [-2.0], [2.0], horizon=30, max_iterations=3, population=500, num_elite=50, num_agents=1)
    a, n, r = cem.call(g["states"], {"trunc": list(g["trunc"])})
    for it in range(3):
        np.testing.assert_array_equal(cem.trace[it]["rewards"], g["rewards"][it])
        np.testing.assert_array_equal(cem.trace[it]["elites"], g["elites"][it])
        np.testing.assert_array_equal(cem.trace[it]["mean"], g["mean"][it])
    np.testing.assert_array_equal(a, g["action"])
    g = load("cfg3")
    pi2 = O.PI2(_pend_eval(), [-2.0], [2.0], horizon=30, max_iterations=2, population=256, num_agents=4)
    for step in range(2):
        a, _, _ = pi2.call(g["states"], {"trunc": list(g["trunc"][step])})
        np.testing.assert_array_equal(a, g["action"][step])
        np.testing.assert_array_equal(pi2.prev, g["prev_mean"][step])


def test_oracle_reproduces_golden_cfg4():
    from tests.golden.make_golden import cheetah_problem
    g = load("cfg4")
    _, _, _, ev = cheetah_problem(int(g["mlp_seed"]))
    np.testing.assert_array_equal(ev(g["states"], g["seq"]), g["rewards"])
    np.testing.assert_array_equal(ev.predict_next_state(g["step_states"], g["step_actions"]), g["step_next"])


# ---------------------------------------------------------------- GPU: engine vs golden
@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    if _lib.device_count() < 1:
        pytest.skip("no GPU")
    return _lib


def _pend_engine(L, opt, A, H, N=0, iters=0, k=0, **kw):
    from blackbox_mpc_amd.engine import Engine
    return Engine(opt, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A, planning_horizon=H,
                  population_size=N, max_iterations=iters, num_elite=k, **kw)


@pytest.mark.gpu
def test_gpu_matches_golden_cfg1_random_search(L):
    g = load("cfg1")
    eng = _pend_engine(L, L.OPT_RANDOM_SEARCH, 1, 20, N=200)
    eng.set_trace(True)
    eng.inject_noise(L.NOISE_UNIFORM, g["uniform"])
    a, n, r = eng.optimize(g["states"])
    np.testing.assert_allclose(eng.get_trace(0, L.TRACE_REWARDS), g["rewards"], rtol=2e-4, atol=2e-3)
    assert eng.get_trace(0, L.TRACE_ELITES)[0] == g["best"][0]
    np.testing.assert_array_equal(a, g["action"])
    np.testing.assert_allclose(n, g["next_state"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r, g["reward"], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_gpu_matches_golden_cfg2_cem(L):
    g = load("cfg2")
    eng = _pend_engine(L, L.OPT_CEM, 1, 30, N=500, iters=3, k=50)
    eng.set_trace(True)
    eng.inject_noise(L.NOISE_TRUNC_NORMAL, g["trunc"])
    a, n, r = eng.optimize(g["states"])
    for it in range(3):
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_REWARDS), g["rewards"][it], rtol=2e-4, atol=2e-3)
        assert set(eng.get_trace(it, L.TRACE_ELITES)[0]) == set(g["elites"][it][0])
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_MEAN), g["mean"][it], rtol=0, atol=2e-5)
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_VAR), g["var"][it], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(a, g["action"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(n, g["next_state"], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_gpu_matches_golden_cfg3_pi2_warm_start(L):
    g = load("cfg3")
    eng = _pend_engine(L, L.OPT_PI2, 4, 30, N=256, iters=2, lamda=1.0)
    for step in range(2):
        eng.inject_noise(L.NOISE_TRUNC_NORMAL, g["trunc"][step])
        a, _, _ = eng.optimize(g["states"])
        np.testing.assert_allclose(a, g["action"][step], rtol=0, atol=5e-3)
        np.testing.assert_allclose(eng.get_state("prev_mean"), g["prev_mean"][step], rtol=0, atol=5e-3)


@pytest.mark.gpu
def test_gpu_matches_golden_cfg4_learned_dynamics(L):
    from blackbox_mpc_amd.engine import Engine
    from tests.golden.make_golden import cheetah_problem
    g = load("cfg4")
    ws, bs, stats, _ = cheetah_problem(int(g["mlp_seed"]))
    eng = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * 6, [1.0] * 6, dim_s=20, num_agents=2, planning_horizon=30)
    eng.set_mlp(ws, bs, [L.ACT_TANH, L.ACT_TANH, L.ACT_NONE], stats)
    np.testing.assert_allclose(eng.evaluate(g["states"], g["seq"]), g["rewards"], rtol=1e-3, atol=3e-2)
    np.testing.assert_allclose(eng.predict_next_state(g["step_states"], g["step_actions"]), g["step_next"],
                               rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_gpu_matches_golden_cfg5_pso_swarm(L):
    g = load("cfg5")
    eng = _pend_engine(L, L.OPT_PSO, 2, 8, N=96, iters=3)
    eng.set_trace(True)
    eng.inject_noise(L.NOISE_PSO_RESET_POS, g["reset_pos"])
    eng.inject_noise(L.NOISE_PSO_RESET_VEL, g["reset_vel"])
    eng.reset()
    eng.inject_noise(L.NOISE_PSO_SCALARS, g["normal2"])
    eng.inject_noise(L.NOISE_PSO_RESEED_TRUNC, g["trunc"])
    eng.inject_noise(L.NOISE_PSO_RESEED_UNIFORM, g["uniform"])
    a, _, _ = eng.optimize(g["states"])
    for it in range(3):
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_REWARDS), g["rewards"][it], rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(a, g["action"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(eng.get_state("pos", (96, 2, 8, 1)), g["pos"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(eng.get_state("vel", (96, 2, 8, 1)), g["vel"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(eng.get_state("gbest"), g["gbest"], rtol=0, atol=2e-5)
