"""User-supplied reward / dynamics as device functions compiled at run time (hiprtc), through the C ABI.

The reference takes any callable for `reward_function` / `dynamics_function`
(trajectory_evaluators/deterministic.py:13-18, :65-66, :99-100).  Here a user function is HIP source; the engine
evaluates trajectories step by step around it.  Checked against the oracle with the SAME function restated in NumPy:
 * the pendulum reward in its DECLARED argument order (what `pendulum_reward(as_executed=False)` computes) written by
   a "user", with the built-in pendulum model;
 * a user-written pendulum model (libm sinf/cosf/atan2f) with the built-in reward;
 * a user reward over the learned MLP (the cheetah reward restated) -- must agree with the built-in fused path;
 * optimizers on top (CEM / PI2 / PSO / RandomSearch / SPSA / CMA-ES): same draws, the user-function engine and the
   built-in per-iteration engine (strict op-for-op pendulum) must walk in step;
 * direct calls of the plug-in objects (cheetah reward, DeterministicMLP.__call__, process_input / process_output).
Tolerances: those of tests/test_gpu_pendulum.py / test_gpu_mlp.py."""
import numpy as np
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
F = np.float32
LO, HI = [-2.0], [2.0]
R_RTOL, R_ATOL = 2e-4, 2e-3

# utils/pendulum.py:10-35 with the arguments in the DECLARED order: the action cost uses the ACTIONS
INTENDED_PENDULUM_REWARD = """
__device__ float bbmpc_user_reward(const float* cur, const float* act, const float* nxt, int S, int U) {
    const float PI = 3.14159274101257324f, TWO_PI = 6.28318548202514648f;
    const float th = atan2f(cur[1], cur[0]);
    float m = fmodf(th + PI, TWO_PI);                    // TF FloorMod: result takes the divisor's sign
    if (m < 0.0f) m = m + TWO_PI;
    const float ang = m - PI;
    const float first = ang * ang + 0.1f * (cur[2] * cur[2]);
    float ss = 0.0f;
    for (int u = 0; u < U; ++u) ss = ss + act[u] * act[u];
    return (-first) - 0.001f * ss;
}
"""
# utils/pendulum.py:58-92 restated by a "user": returns the DELTA of the state
USER_PENDULUM_MODEL = """
__device__ void bbmpc_user_dynamics(const float* x, float* delta, int S, int U) {
    const float PI = 3.14159274101257324f;
    const float th = atan2f(x[1], x[0]);
    float acc = -15.0f * sinf(th + PI);
    acc = acc + 3.0f * x[3];
    float nthd = x[2] + acc * 0.05f;
    const float nth = th + nthd * 0.05f;
    nthd = fminf(fmaxf(nthd, -8.0f), 8.0f);
    delta[0] = cosf(nth) - x[0];
    delta[1] = sinf(nth) - x[1];
    delta[2] = nthd - x[2];
}
"""
USER_CHEETAH_REWARD = """
__device__ float bbmpc_user_reward(const float* cur, const float* act, const float* nxt, int S, int U) {
    float r = 0.0f;
    if (cur[5] >= 0.2f) r = r + (-10.0f);
    if (cur[6] >= 0.0f) r = r + (-10.0f);
    if (cur[7] >= 0.0f) r = r + (-10.0f);
    r = r + (nxt[17] - cur[17]) / 0.01f;
    float ss = 0.0f;
    for (int u = 0; u < U; ++u) ss = ss + act[u] * act[u];
    return r - 0.0f * ss;
}
"""


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1
    return _lib


@pytest.fixture(params=["fused", "stepwise"], autouse=True)
def user_rollout_form(request, monkeypatch):
    """every test runs on both forms of the user-function evaluator: the hiprtc-compiled fused lane-per-trajectory
    kernel (analytic dynamics; the default) and the step-wise evaluator (always used for MLP + user reward)"""
    if request.param == "stepwise":
        monkeypatch.setenv("BBMPC_USER_STEPWISE", "1")
    yield request.param


def _intended(cur, act, nxt):
    return O.pendulum_reward(cur, act, nxt, as_executed=False)


def test_user_reward_with_builtin_pendulum_model_matches_oracle(L):
    from blackbox_mpc_amd.engine import Engine
    A, H, N = 3, 25, 700
    eng = Engine(L.OPT_NONE, L.DYN_PENDULUM, L.REW_USER, LO, HI, dim_s=3, num_agents=A, planning_horizon=H)
    with pytest.raises(L.BBMPCError):                       # computing before the source is set fails loudly
        eng.evaluate(O.pendulum_start_states(A), np.zeros((4, A, H, 1), F))
    eng.set_reward_source(INTENDED_PENDULUM_REWARD)
    ev = O.Evaluator(_intended, O.Handler(O.pendulum_dynamics, True))
    rng = np.random.default_rng(5)
    states = O.pendulum_start_states(A)
    seq = rng.uniform(-2, 2, (N, A, H, 1)).astype(F)
    np.testing.assert_allclose(eng.evaluate(states, seq), ev(states, seq), rtol=R_RTOL, atol=R_ATOL)
    # single-step API: evaluate_next_reward is the user's function on the caller's rows
    s = rng.normal(0, 1, (64, 3)).astype(F)
    n = rng.normal(0, 1, (64, 3)).astype(F)
    a = rng.uniform(-2, 2, (64, 1)).astype(F)
    np.testing.assert_allclose(eng.evaluate_next_reward(s, n, a), _intended(s, a, n), rtol=1e-5, atol=1e-5)
    # ... and it differs from the as-executed built-in (quirk Q1), as it should
    builtin = Engine(L.OPT_NONE, L.DYN_PENDULUM, L.REW_PENDULUM, LO, HI, dim_s=3, num_agents=A, planning_horizon=H)
    assert np.abs(builtin.evaluate(states, seq) - eng.evaluate(states, seq)).max() > 1e-2
    # the built-in's own opt-out of Q1 computes the same thing
    fixed = Engine(L.OPT_NONE, L.DYN_PENDULUM, L.REW_PENDULUM, LO, HI, dim_s=3, num_agents=A, planning_horizon=H,
                   quirks=L.FIX_Q1_REWARD_ARG_ORDER | L.STRICT_MATH)
    np.testing.assert_allclose(eng.evaluate(states, seq), fixed.evaluate(states, seq), rtol=2e-5, atol=2e-4)
    # NaN guard (deterministic.py:75-77) on the step-wise path
    bad = states.copy()
    bad[1, 2] = np.nan
    r = eng.evaluate(bad, seq[:8])
    assert np.all(r[:, 1] == F(-1e6)) and np.all(np.isfinite(r[:, [0, 2]]))


def test_user_dynamics_with_builtin_reward_matches_oracle(L):
    from blackbox_mpc_amd.engine import Engine
    A, H, N = 2, 20, 500
    eng = Engine(L.OPT_NONE, L.DYN_USER, L.REW_PENDULUM, LO, HI, dim_s=3, num_agents=A, planning_horizon=H)
    eng.set_dynamics_source(USER_PENDULUM_MODEL)
    ev = O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))
    rng = np.random.default_rng(6)
    states = O.pendulum_start_states(A)
    seq = rng.uniform(-2, 2, (N, A, H, 1)).astype(F)
    np.testing.assert_allclose(eng.evaluate(states, seq), ev(states, seq), rtol=R_RTOL, atol=R_ATOL)
    s = O.pendulum_start_states(40)
    a = rng.uniform(-2, 2, (40, 1)).astype(F)
    np.testing.assert_allclose(eng.predict_next_state(s, a), ev.predict_next_state(s, a), rtol=1e-5, atol=1e-5)
    # both plug-ins user-supplied
    both = Engine(L.OPT_NONE, L.DYN_USER, L.REW_USER, LO, HI, dim_s=3, num_agents=A, planning_horizon=H)
    both.set_dynamics_source(USER_PENDULUM_MODEL)
    both.set_reward_source(INTENDED_PENDULUM_REWARD)
    ev2 = O.Evaluator(_intended, O.Handler(O.pendulum_dynamics, True))
    np.testing.assert_allclose(both.evaluate(states, seq), ev2(states, seq), rtol=R_RTOL, atol=R_ATOL)


def test_user_reward_over_the_learned_model_agrees_with_the_fused_path(L):
    from blackbox_mpc_amd.engine import Engine
    S, U, A, H, N = 20, 6, 2, 12, 300
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=42)
    z, o = np.zeros, np.ones
    stats = [z(S, F), o(S, F), z(U, F), o(U, F), z(S, F), np.full(S, 0.1, F)]
    mk = lambda rew: Engine(L.OPT_NONE, L.DYN_MLP, rew, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=A, planning_horizon=H)
    user, fused = mk(L.REW_USER), mk(L.REW_CHEETAH)
    for e in (user, fused):
        e.set_mlp(ws, bs, [1, 1, 0], stats)
    user.set_reward_source(USER_CHEETAH_REWARD)
    rng = np.random.default_rng(7)
    states = O.cheetah_start_states(A, S)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    ev = O.Evaluator("cheetah", O.Handler(O.MLP(ws, bs, ["tanh", "tanh", None]), False, True, stats))
    want = ev(states, seq)
    np.testing.assert_allclose(user.evaluate(states, seq), want, rtol=1e-3, atol=1e-3 * H)
    np.testing.assert_allclose(user.evaluate(states, seq), fused.evaluate(states, seq), rtol=1e-3, atol=1e-3 * H)


def test_user_reward_over_a_small_learned_pendulum_model(L, user_rollout_form):
    # the tutorials' learned Pendulum model (4-32-32-32-3) with a user reward: the small-network kernel
    # (kernels_mlp_w4.hpp) records the trajectory, the user function scores it
    from blackbox_mpc_amd.engine import Engine
    S, U, A, H, N = 3, 1, 2, 15, 150
    dims, acts = [4, 32, 32, 32, 3], ["tanh", "tanh", "tanh", None]
    ws, bs = O.make_mlp_params(dims, seed=8)
    stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]
    user = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_USER, LO, HI, dim_s=S, num_agents=A, planning_horizon=H)
    user.set_mlp(ws, bs, [1, 1, 1, 0], stats)
    user.set_reward_source(INTENDED_PENDULUM_REWARD)
    rng = np.random.default_rng(27)
    states = O.pendulum_start_states(A)
    seq = rng.uniform(-2, 2, (N, A, H, U)).astype(F)
    ev = O.Evaluator(_intended, O.Handler(O.MLP(ws, bs, acts), False, True, stats))
    user.set_profiling(True)
    got = user.evaluate(states, seq)
    if user_rollout_form != "stepwise":
        assert user.get_profile()[2] == "k_rollout_mlp_w4"           # (hidden <= 32; BBMPC_MLP_W4=0: k_rollout_mlp_wave)
    np.testing.assert_allclose(got, ev(states, seq), rtol=1e-3, atol=1e-3 * H)


def test_user_reward_over_the_learned_model_large_population(L, user_rollout_form):
    # 6000 rows per launch: the pipelined two-tile MFMA kernel records the trajectory (the quad kernel covers <= 2048)
    if user_rollout_form == "stepwise":
        pytest.skip("same arithmetic as the small-population test; the step-wise form needs 2H launches of 6000 rows")
    from blackbox_mpc_amd.engine import Engine
    S, U, A, H, N = 20, 6, 2, 20, 3000
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=42)
    stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]
    mk = lambda rew: Engine(L.OPT_NONE, L.DYN_MLP, rew, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=A, planning_horizon=H)
    user, fused = mk(L.REW_USER), mk(L.REW_CHEETAH)
    for e in (user, fused):
        e.set_mlp(ws, bs, [1, 1, 0], stats)
    user.set_reward_source(USER_CHEETAH_REWARD)
    rng = np.random.default_rng(17)
    states = O.cheetah_start_states(A, S)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    got, ref = user.evaluate(states, seq), fused.evaluate(states, seq)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-3)         # same dynamics kernel, same reward arithmetic


@pytest.mark.parametrize("opt_name", ["CEM", "PI2", "PSO"])
def test_optimizers_with_user_reward_over_the_learned_model(L, opt_name):
    # the common "blackbox" case: learned MLP dynamics, custom reward.  The user's restatement of the cheetah reward must
    # drive the optimizers exactly like the built-in one (same draws; selection steps stay in step because the rewards
    # agree to rounding -- both sides run the same MFMA dynamics)
    from blackbox_mpc_amd.engine import Engine
    S, U, A, H, N, iters, k = 20, 6, 2, 15, 400, 3, 40
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=42)
    stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]
    opt = {"CEM": L.OPT_CEM, "PI2": L.OPT_PI2, "PSO": L.OPT_PSO}[opt_name]
    mk = lambda rew: Engine(opt, L.DYN_MLP, rew, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=A, planning_horizon=H,
                            population_size=N, max_iterations=iters, num_elite=k, seed=77, lamda=1.0)
    user, ref = mk(L.REW_USER), mk(L.REW_CHEETAH)
    for e in (user, ref):
        e.set_mlp(ws, bs, [1, 1, 0], stats)
        e.reset()
    user.set_reward_source(USER_CHEETAH_REWARD)
    s_u = s_r = O.cheetah_start_states(A, S)
    for t in range(3):
        a_u, n_u, r_u = user.optimize(s_u, t)
        a_r, n_r, r_r = ref.optimize(s_r, t)
        np.testing.assert_allclose(a_u, a_r, rtol=0, atol=2e-3)
        np.testing.assert_allclose(n_u, n_r, rtol=0, atol=2e-3)
        np.testing.assert_allclose(r_u, r_r, rtol=1e-3, atol=0.3)        # (s'17 - s17)/0.01 amplifies the state tolerance x100
        s_u, s_r = n_u, n_r


@pytest.mark.parametrize("opt_name", ["RandomSearch", "CEM", "PI2", "PSO", "SPSA", "CMA-ES"])
def test_optimizers_on_user_functions_walk_in_step_with_the_builtin_path(L, monkeypatch, opt_name):
    # the user-function engine (step-wise evaluator) against the built-in per-iteration engine with the op-for-op
    # pendulum (BBMPC_STRICT_MATH) and the declared-order reward (BBMPC_FIX_Q1): same Philox draws (same seed), so
    # every control step's action / next state / reward must agree to rounding, warm starts included
    monkeypatch.setenv("BBMPC_FUSED", "0")
    from blackbox_mpc_amd.engine import Engine
    opt = {"RandomSearch": L.OPT_RANDOM_SEARCH, "CEM": L.OPT_CEM, "PI2": L.OPT_PI2, "PSO": L.OPT_PSO, "SPSA": L.OPT_SPSA,
           "CMA-ES": L.OPT_CMAES}[opt_name]
    A, H, N, iters, k = 2, 10, 160, 3, 16
    kw = dict(dim_s=3, num_agents=A, planning_horizon=H, population_size=N, max_iterations=iters, num_elite=k, seed=31)
    user = Engine(opt, L.DYN_USER, L.REW_USER, LO, HI, **kw)
    user.set_dynamics_source(USER_PENDULUM_MODEL)
    user.set_reward_source(INTENDED_PENDULUM_REWARD)
    ref = Engine(opt, L.DYN_PENDULUM, L.REW_PENDULUM, LO, HI, quirks=L.FIX_Q1_REWARD_ARG_ORDER | L.STRICT_MATH, **kw)
    user.reset()
    ref.reset()
    s_u = s_r = O.pendulum_start_states(A)
    for t in range(4):
        a_u, n_u, r_u = user.optimize(s_u, t)
        a_r, n_r, r_r = ref.optimize(s_r, t)
        # selection steps (top-k / argmax) are discontinuous: identical draws and rewards equal to ~1e-6 keep the two
        # in step; the tolerance below is the refit tolerance of the suite
        np.testing.assert_allclose(a_u, a_r, rtol=0, atol=5e-4 if opt_name != "CMA-ES" else 5e-3)
        np.testing.assert_allclose(n_u, n_r, rtol=0, atol=1e-4 if opt_name != "CMA-ES" else 1e-3)
        s_u, s_r = n_u, n_r


def test_policy_api_with_user_functions(L):
    from blackbox_mpc_amd.policies import MPCPolicy
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.utils.device_functions import HipDynamicsFunction, HipRewardFunction
    rew, dyn = HipRewardFunction(INTENDED_PENDULUM_REWARD), HipDynamicsFunction(USER_PENDULUM_MODEL, dim_s=3, dim_u=1)
    pol = MPCPolicy(reward_function=rew, env_action_space=Box(LO, HI), env_observation_space=Box([-1, -1, -8], [1, 1, 8]),
                    true_model=True, dynamics_function=dyn, optimizer_name="CEM", num_agents=2, planning_horizon=15,
                    population_size=256, max_iterations=3, num_elite=32)
    obs = O.pendulum_start_states(2)
    ret = []
    for t in range(10):
        a, n, r = pol.act(obs, t)
        assert a.shape == (2, 1) and n.shape == (2, 3) and r.shape == (2,) and np.all(np.abs(a) <= 2.0)
        ev = O.Evaluator(_intended, O.Handler(O.pendulum_dynamics, True))
        np.testing.assert_allclose(n, ev.predict_next_state(obs, a), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r, _intended(obs, a, n), rtol=1e-5, atol=1e-5)
        obs = n
        ret.append(r)
    # the plug-in objects are callable with NumPy batches and run the same device code
    x = np.concatenate([obs, a], axis=1)
    np.testing.assert_allclose(dyn(x), O.pendulum_dynamics(x), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rew(obs, a, n), _intended(obs, a, n), rtol=1e-5, atol=1e-5)
    # the evaluator built from the same objects
    seq = np.random.default_rng(2).uniform(-2, 2, (50, 2, 15, 1)).astype(F)
    got = pol._trajectory_evaluator(obs, seq)
    np.testing.assert_allclose(got, O.Evaluator(_intended, O.Handler(O.pendulum_dynamics, True))(obs, seq), rtol=R_RTOL, atol=R_ATOL)


def test_direct_calls_of_the_builtin_plugins_run_device_code(L):
    from blackbox_mpc_amd.dynamics_functions import DeterministicMLP
    from blackbox_mpc_amd.dynamics_handlers import SystemDynamicsHandler
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.utils.cheetah import reward_function
    rng = np.random.default_rng(11)
    S, U, B = 20, 6, 37
    cur, nxt = rng.normal(0, 0.5, (B, S)).astype(F), rng.normal(0, 0.5, (B, S)).astype(F)
    act = rng.uniform(-1, 1, (B, U)).astype(F)
    np.testing.assert_allclose(reward_function(cur, act, nxt), O.cheetah_reward(cur, act, nxt), rtol=1e-5, atol=1e-4)
    # DeterministicMLP.__call__(x, train): the Dense stack on processed inputs (deterministic_mlp.py:27-51)
    for layers, acts in (([26, 200, 200, 20], [np.tanh, np.tanh, None]), ([5, 32, 32, 32, 3], ["relu", "tanh", "sigmoid", None])):
        net = DeterministicMLP(layers=layers, activation_functions=acts, seed=3)
        net.set_weights(net.weights, [rng.normal(0, 0.1, b.shape).astype(F) for b in net.biases])
        x = rng.normal(0, 1, (B, layers[0])).astype(F)
        names = [None if a is None else (a if isinstance(a, str) else a.__name__) for a in acts]
        want = O.MLP(net.weights, net.biases, names)(x)
        np.testing.assert_allclose(net(x, train=False), want, rtol=2e-5, atol=2e-5)
        net.set_weights([w * F(0.5) for w in net.weights], net.biases)          # a refit re-uploads before the next call
        np.testing.assert_allclose(net(x), O.MLP(net.weights, net.biases, names)(x), rtol=2e-5, atol=2e-5)
    # process_input / process_output of the handler (system_dynamics_handler.py:97-161)
    stats = [rng.normal(0, 0.2, S).astype(F), rng.uniform(0.5, 1.5, S).astype(F), rng.normal(0, 0.1, U).astype(F),
             rng.uniform(0.5, 1.5, U).astype(F), rng.normal(0, 0.01, S).astype(F), rng.uniform(0.05, 0.15, S).astype(F)]
    asp, osp = Box([-1.0] * U, [1.0] * U), Box([-10.0] * S, [10.0] * S)
    net = DeterministicMLP(layers=[26, 200, 200, 20], activation_functions=["tanh", "tanh", None], seed=1)
    h = SystemDynamicsHandler(asp, osp, dynamics_function=net, true_model=False, is_normalized=True)
    h.set_normalization_stats(*stats)
    oh = O.Handler(None, False, True, stats)
    raw = rng.normal(0, 1, (B, S)).astype(F)
    np.testing.assert_allclose(h.process_input(cur, act), oh.process_input(cur, act), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(h.process_output(cur, raw), oh.process_output(cur, raw), rtol=1e-6, atol=1e-6)
    ht = SystemDynamicsHandler(asp, osp, dynamics_function=net, true_model=False, is_normalized=False)
    np.testing.assert_array_equal(ht.process_input(cur, act), np.concatenate([cur, act], axis=1))
    np.testing.assert_array_equal(ht.process_output(cur, raw), (raw + cur).astype(F))
    # predict_next_state == process_output(s, net(process_input(s, a)))  (deterministic.py:79-103)
    from blackbox_mpc_amd.trajectory_evaluators import DeterministicTrajectoryEvaluator
    ev = DeterministicTrajectoryEvaluator(reward_function, h)
    np.testing.assert_allclose(ev.predict_next_state(cur, act), h.process_output(cur, net(h.process_input(cur, act))),
                               rtol=2e-5, atol=2e-5)
