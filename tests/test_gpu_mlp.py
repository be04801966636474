"""GPU parity tests for the learned-dynamics (MFMA) path, through the C ABI.

Tolerances (fp32): the oracle's Dense layers accumulate in float64 and round once; the kernel accumulates in
fp32 on the matrix cores (exact fp32 products, k-ordered), and tanh is sign(x)(1 - 2/(e^{2|x|}+1)) on the hardware
exp / rcp units (v_exp_f32, v_rcp_f32: absolute error <= ~3e-7, csrc/kernels_mlp.hpp bb_tanhf).  Stated bounds:
  single model step  : rtol 2e-5 + atol 2e-5 on the next state
  H-step rewards     : rtol 1e-3 + atol 1e-3 * H   ('rtol 1e-3 (MLP fp32-MFMA)')
  refit mean/var     : atol 1e-4 given the same elite set (elite near-ties handled as in the pendulum tests)
"""
import numpy as np
import pytest

from tests.parity_util import assert_cheetah_rewards, cheetah_threshold_margin

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
F = np.float32
ACT = {"tanh": 1, "relu": 2, "sigmoid": 3, None: 0}


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1
    return _lib


def _stats(S, U, seed):
    rng = np.random.default_rng(seed)
    return [rng.normal(0, 0.2, S).astype(F), rng.uniform(0.5, 1.5, S).astype(F),
            rng.normal(0, 0.1, U).astype(F), rng.uniform(0.5, 1.5, U).astype(F),
            rng.normal(0, 0.01, S).astype(F), rng.uniform(0.05, 0.15, S).astype(F)]


def _problem(L, dims, acts, S, U, reward, normalized=True, seed=42, A=1, H=1, opt=None, N=0, iters=0, k=0, **kw):
    from blackbox_mpc_amd.engine import Engine
    ws, bs = O.make_mlp_params(dims, seed=seed)
    rng = np.random.default_rng(seed + 1)
    bs = [rng.normal(0, 0.05, b.shape).astype(F) for b in bs]
    stats = _stats(S, U, seed + 2) if normalized else None
    lo, hi = [-1.0] * U, [1.0] * U
    rk = L.REW_CHEETAH if reward == "cheetah" else L.REW_PENDULUM
    eng = Engine(opt if opt is not None else L.OPT_NONE, L.DYN_MLP, rk, lo, hi, dim_s=S, num_agents=A,
                 planning_horizon=H, population_size=N, max_iterations=iters, num_elite=k, **kw)
    eng.set_mlp(ws, bs, [ACT[a] for a in acts], stats)
    handler = O.Handler(O.MLP(ws, bs, acts), False, normalized, stats)
    ev = O.Evaluator(reward, handler)
    return eng, ev, lo, hi


CHEETAH = ([26, 200, 200, 20], ["tanh", "tanh", None], 20, 6, "cheetah")
PEND_MLP = ([4, 32, 32, 32, 3], ["tanh", "tanh", "tanh", None], 3, 1, "pendulum")


@pytest.mark.parametrize("spec,normalized", [(CHEETAH, True), (CHEETAH, False), (PEND_MLP, True),
                                             (([26, 500, 500, 500, 20], ["tanh", "relu", "sigmoid", None], 20, 6, "cheetah"), True),
                                             (([26, 20], [None], 20, 6, "cheetah"), True),
                                             (([23, 40, 18], ["tanh", None], 18, 5, "cheetah"), True),
                                             # hidden widths whose last 16-feature tile is half empty (permuted operand
                                             # packing, csrc set_mlp): a single 8-wide tile, 5 of 16, and sigmoid padding
                                             (([4, 8, 3], ["tanh", None], 3, 1, "pendulum"), True),
                                             (([4, 21, 24, 3], ["sigmoid", "tanh", None], 3, 1, "pendulum"), False)])
def test_single_step_matches_oracle(L, spec, normalized):
    dims, acts, S, U, reward = spec
    eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, normalized)
    rng = np.random.default_rng(0)
    B = 333
    s = rng.normal(0, 0.5, (B, S)).astype(F)
    if reward == "pendulum":
        th = rng.uniform(-np.pi, np.pi, B)
        s = np.stack([np.cos(th), np.sin(th), rng.uniform(-4, 4, B)], 1).astype(F)
    a = rng.uniform(-1, 1, (B, U)).astype(F)
    nxt = eng.predict_next_state(s, a)
    nxt_o = ev.predict_next_state(s, a)
    np.testing.assert_allclose(nxt, nxt_o, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(eng.evaluate_next_reward(s, nxt_o, a), ev.evaluate_next_reward(s, nxt_o, a),
                               rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("spec,N,A,H", [(CHEETAH, 100, 1, 30), (CHEETAH, 37, 3, 50), (PEND_MLP, 200, 2, 20),
                                        (CHEETAH, 1, 1, 1), (CHEETAH, 16, 1, 2)])
def test_evaluator_matches_oracle(L, spec, N, A, H):
    dims, acts, S, U, reward = spec
    eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=A, H=H)
    rng = np.random.default_rng(N + H)
    states = O.cheetah_start_states(A, S) if reward == "cheetah" else O.pendulum_start_states(A)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    got = eng.evaluate(states, seq)
    want = ev(states, seq)
    assert np.all(np.isfinite(want))
    if reward == "cheetah":
        assert_cheetah_rewards(got, want, 1e-3, 1e-3 * H, margin=lambda: cheetah_threshold_margin(ev, states, seq))
    else:
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-3 * H)


def test_cem_refit_workgroup_count_does_not_change_results(L, monkeypatch):
    # k_refit_cem_v2 shares the elite gather between G workgroups per agent (each repeats the selection and takes H*U / G
    # rows): the control step must not depend on G, bit for bit.
    dims, acts, S, U, reward = CHEETAH
    N, A, H, iters, k = 300, 3, 12, 3, 20
    outs = []
    for g in ("1", "3", "8"):
        monkeypatch.setenv("BBMPC_REFIT_WGS", g)
        eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=A, H=H, opt=L.OPT_CEM, N=N, iters=iters, k=k, alpha=0.1, seed=5)
        states = O.cheetah_start_states(A, S)
        outs.append(eng.optimize(states))
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("pair", ["0", "1"])
@pytest.mark.parametrize("dims,S,U,N,A,H,normalized", [
    ([26, 200, 200, 20], 20, 6, 75, 3, 7, True),       # ragged population, dimensions of the cheetah family (compile-time S, U)
    ([26, 200, 200, 20], 20, 6, 130, 2, 50, False),    # the compile-time-horizon instance, un-normalised
    ([24, 200, 200, 19], 19, 5, 90, 2, 9, True),       # same family, other dim_S / dim_U: run-time dimensions
    ([27, 204, 204, 21], 21, 6, 64, 1, 5, True),       # 13 hidden tiles whose last one is NOT half empty; 6 state-reduction waves (no interval-1 split)
])
def test_pipelined_tile_kernel_variants(L, monkeypatch, pair, dims, S, U, N, A, H, normalized):
    # k_rollout_mlp_pair in its one-tile (13 waves) and two-tile (12 waves: the 13th hidden tile's K loop in quarters on
    # four waves, layer 0 fused into the layer-1 loop) forms, forced through the environment switches at shapes the
    # size heuristics would give to the quad kernels; every template instance against the NumPy oracle.
    monkeypatch.setenv("BBMPC_MLP_Q4", "0")
    monkeypatch.setenv("BBMPC_MLP_PAIR", pair)
    eng, ev, lo, hi = _problem(L, dims, ["tanh", "tanh", None], S, U, "cheetah", normalized, A=A, H=H)
    rng = np.random.default_rng(N + H)
    states = O.cheetah_start_states(A, S)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    got = eng.evaluate(states, seq)
    want = ev(states, seq)
    assert np.all(np.isfinite(want))
    assert_cheetah_rewards(got, want, 1e-3, 1e-3 * H, margin=lambda: cheetah_threshold_margin(ev, states, seq))


@pytest.mark.parametrize("q4", ["0", "1"])
def test_evaluator_properties_at_config5_size(L, monkeypatch, q4):
    # BASELINE config-5 shape (per GPU): N=2000, A=4, H=50, S=20, U=6.  Size-independent properties:
    # agents are independent rows, particle order is irrelevant, a prefix of the population evaluates
    # to the same values (bit-exact), and a 48-particle sample matches the oracle.
    # The engine picks the 16-particle or the 4-particle tiling by problem size and the two sum in a different
    # order, so the bit-exact properties are stated per tiling (pinned through the env override).
    monkeypatch.setenv("BBMPC_MLP_Q4", q4)
    dims, acts, S, U, reward = CHEETAH
    N, A, H = 2000, 4, 50
    eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=A, H=H)
    eng1, _, _, _ = _problem(L, dims, acts, S, U, reward, True, A=1, H=H)
    rng = np.random.default_rng(5)
    states = O.cheetah_start_states(A, S)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    full = eng.evaluate(states, seq)
    assert np.all(np.isfinite(full))
    np.testing.assert_array_equal(full[:, 2], eng1.evaluate(states[2:3], seq[:, 2:3])[:, 0])
    perm = rng.permutation(N)
    np.testing.assert_array_equal(eng.evaluate(states, seq[perm]), full[perm])
    np.testing.assert_array_equal(eng.evaluate(states, seq[:100]), full[:100])
    sub = rng.choice(N, 48, replace=False)
    assert_cheetah_rewards(full[sub], ev(states, seq[sub]), 1e-3, 1e-3 * H, margin=lambda: cheetah_threshold_margin(ev, states, seq[sub]))


@pytest.mark.parametrize("tiling", ["auto", "pair1", "pair2"])
def test_nan_state_gives_minus_1e6(L, monkeypatch, tiling):
    # deterministic.py:75-77 on every MFMA tiling of the rollout (quads by default at this size, the pipelined tile kernel
    # in its one-tile and two-tile forms when forced)
    if tiling != "auto":
        monkeypatch.setenv("BBMPC_MLP_Q4", "0")
        monkeypatch.setenv("BBMPC_MLP_PAIR", "1" if tiling == "pair2" else "0")
    dims, acts, S, U, reward = CHEETAH
    eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=2, H=5)
    states = O.cheetah_start_states(2, S)
    states[0, 3] = np.nan
    r = eng.evaluate(states, np.zeros((20, 2, 5, U), F))
    assert np.all(r[:, 0] == F(-1e6)) and np.all(np.isfinite(r[:, 1])) and np.all(r[:, 1] > -1e5)


def test_compute_before_set_mlp_is_an_error(L):
    from blackbox_mpc_amd.engine import Engine
    eng = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * 6, [1.0] * 6, dim_s=20, num_agents=1, planning_horizon=3)
    with pytest.raises(L.BBMPCError) as ei:
        eng.evaluate(np.zeros((1, 20), F), np.zeros((4, 1, 3, 6), F))
    assert ei.value.code == L.E_STATE
    with pytest.raises(L.BBMPCError):      # wrong input width
        eng.set_mlp([np.zeros((25, 20), F)], [np.zeros(20, F)], [0], None)


def _lockstep_select(L, eng, A, k, iters, rtol, atol):
    hip_el = [eng.get_trace(it, L.TRACE_ELITES) for it in range(iters)]
    hip_r = [eng.get_trace(it, L.TRACE_REWARDS) for it in range(iters)]

    def select(it, r_o, own):
        np.testing.assert_allclose(hip_r[it], r_o, rtol=rtol, atol=atol)
        for a in range(A):
            he = hip_el[it][a]
            if set(own[a]) != set(he):
                kth = np.sort(r_o[:, a])[::-1][k - 1]
                for n in set(own[a]) ^ set(he):
                    assert abs(r_o[n, a] - kth) <= 2 * (atol + rtol * abs(kth)), "elite sets differ beyond the tie tolerance"
            np.testing.assert_array_equal(he, O.topk_desc(hip_r[it][:, a], k))
        return hip_el[it]
    return select


@pytest.mark.parametrize("N,A,H,iters,k", [(1000, 1, 30, 5, 50), (96, 2, 8, 2, 12)])
def test_cem_with_learned_dynamics_lockstep(L, N, A, H, iters, k):
    dims, acts, S, U, reward = CHEETAH
    eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=A, H=H, opt=L.OPT_CEM, N=N, iters=iters, k=k)
    eng.set_trace(True)
    rng = np.random.default_rng(N)
    noise = {"trunc": [O.truncated_normal_noise(rng, (N, A, H, U)) for _ in range(iters)]}
    eng.inject_noise(L.NOISE_TRUNC_NORMAL, np.stack(noise["trunc"]))
    states = O.cheetah_start_states(A, S)
    act, nxt, rew = eng.optimize(states)
    cem = O.CEM(ev, lo, hi, horizon=H, max_iterations=iters, population=N, num_elite=k, num_agents=A)
    cem._optimize(states, noise, forced_elites=_lockstep_select(L, eng, A, k, iters, 1e-3, 1e-3 * H))
    for it in range(iters):
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_SAMPLES), cem.trace[it]["samples"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_MEAN), cem.trace[it]["mean"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_VAR), cem.trace[it]["var"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(act, cem.trace[-1]["mean"][:, 0], rtol=0, atol=1e-4)
    nxt_o = ev.predict_next_state(states, act)
    np.testing.assert_allclose(nxt, nxt_o, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(rew, ev.evaluate_next_reward(states, nxt_o, act), rtol=1e-4, atol=1e-3)


def test_pi2_and_random_search_with_learned_dynamics(L):
    dims, acts, S, U, reward = CHEETAH
    N, A, H, iters = 256, 2, 10, 2
    states = O.cheetah_start_states(A, S)
    rng = np.random.default_rng(3)
    eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=A, H=H, opt=L.OPT_PI2, N=N, iters=iters, lamda=5.0)
    eng.set_trace(True)
    noise = {"trunc": [O.truncated_normal_noise(rng, (N, A, H, U)) for _ in range(iters)]}
    eng.inject_noise(L.NOISE_TRUNC_NORMAL, np.stack(noise["trunc"]))
    act, _, _ = eng.optimize(states)
    pi2 = O.PI2(ev, lo, hi, horizon=H, max_iterations=iters, population=N, num_agents=A, lamda=5.0)
    act_o, _, _ = pi2.call(states, noise)
    for it in range(iters):
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_REWARDS), pi2.trace[it]["rewards"], rtol=1e-3, atol=1e-2)
        np.testing.assert_allclose(eng.get_trace(it, L.TRACE_MEAN), pi2.trace[it]["mean"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(act, act_o, rtol=0, atol=2e-3)

    eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=A, H=H, opt=L.OPT_RANDOM_SEARCH, N=N)
    eng.set_trace(True)
    u01 = rng.random((N, A, H, U)).astype(F)
    eng.inject_noise(L.NOISE_UNIFORM, u01)
    act, _, _ = eng.optimize(states)
    rs = O.RandomSearch(ev, lo, hi, horizon=H, population=N, num_agents=A)
    act_o, _, _ = rs.call(states, {"uniform": u01})
    r_hip = eng.get_trace(0, L.TRACE_REWARDS)
    np.testing.assert_allclose(r_hip, rs.trace[0]["rewards"], rtol=1e-3, atol=1e-2)
    best = eng.get_trace(0, L.TRACE_ELITES)
    np.testing.assert_array_equal(best, np.argmax(r_hip, axis=0))
    if np.array_equal(best, rs.trace[0]["best"]):
        np.testing.assert_array_equal(act, act_o)


def test_host_api_with_learned_dynamics(L, tmp_path):
    from blackbox_mpc_amd.dynamics_functions import DeterministicMLP
    from blackbox_mpc_amd.dynamics_handlers import SystemDynamicsHandler
    from blackbox_mpc_amd.policies import MPCPolicy
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.utils.cheetah import reward_function
    S, U = 20, 6
    act_space, obs_space = Box([-1.0] * U, [1.0] * U), Box([-10.0] * S, [10.0] * S)
    mlp = DeterministicMLP(layers=[S + U, 200, 200, S], activation_functions=[np.tanh, np.tanh, None], seed=1)
    ws, bs = O.make_mlp_params([S + U, 200, 200, S])
    mlp.set_weights(ws, bs)
    h = SystemDynamicsHandler(act_space, obs_space, dynamics_function=mlp, true_model=False, is_normalized=True)
    stats = _stats(S, U, 9)
    h.set_normalization_stats(*stats)
    h.save(str(tmp_path))
    h2 = SystemDynamicsHandler(act_space, obs_space, true_model=False, is_normalized=True, saved_model_dir=str(tmp_path))
    pol = MPCPolicy(reward_function=reward_function, env_action_space=act_space, env_observation_space=obs_space,
                    dynamics_handler=h2, optimizer_name="CEM", num_agents=2, planning_horizon=10, population_size=128,
                    max_iterations=2, num_elite=16)
    obs = O.cheetah_start_states(2, S)
    a, n, r = pol.act(obs, 0)
    assert a.shape == (2, U) and n.shape == (2, S) and r.shape == (2,)
    ev = O.Evaluator("cheetah", O.Handler(O.MLP(ws, bs, ["tanh", "tanh", None]), False, True, stats))
    np.testing.assert_allclose(n, ev.predict_next_state(obs, a), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(pol._trajectory_evaluator.predict_next_state(obs, a), n, rtol=1e-6, atol=1e-6)


def test_deep_network_generic_kernel(L):
    # more Dense layers than the weights-stationary specialisations cover (PETS-style 4 x 200 hidden): generic kernel,
    # 5 and 8 layers (the limit)
    for dims, acts in (([26, 200, 200, 200, 200, 20], ["tanh", "tanh", "relu", "tanh", None]),
                       ([4, 24, 24, 24, 24, 24, 24, 24, 3], ["tanh"] * 7 + [None])):
        S = dims[-1]
        U = dims[0] - S
        reward = "cheetah" if S == 20 else "pendulum"
        eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=2, H=12)
        rng = np.random.default_rng(len(dims))
        states = O.cheetah_start_states(2, S) if reward == "cheetah" else O.pendulum_start_states(2)
        seq = rng.uniform(-1, 1, (60, 2, 12, U)).astype(F)
        want = ev(states, seq)
        assert np.all(np.isfinite(want))
        np.testing.assert_allclose(eng.evaluate(states, seq), want, rtol=1e-3, atol=1e-3 * 12)
        s1 = states[:1].repeat(9, 0)
        a1 = rng.uniform(-1, 1, (9, U)).astype(F)
        np.testing.assert_allclose(eng.predict_next_state(s1, a1), ev.predict_next_state(s1, a1), rtol=2e-5, atol=2e-5)
    from blackbox_mpc_amd.engine import Engine
    dims = [4] + [8] * 8 + [3]
    ws, bs = O.make_mlp_params(dims, seed=1)
    eng = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_PENDULUM, [-1.0], [1.0], dim_s=3, num_agents=1, planning_horizon=4)
    with pytest.raises(Exception):
        eng.set_mlp(ws, bs, [1] * 8 + [0], None)                                     # 9 Dense layers


def test_wide_io_network_generic_kernel(L):
    # S + U > 32 and S > 32: outside the weights-stationary specialisations -> generic streaming kernel
    S, U = 40, 9
    dims, acts = [S + U, 96, 64, S], ["relu", "tanh", None]
    from blackbox_mpc_amd.engine import Engine
    ws, bs = O.make_mlp_params(dims, seed=3)
    stats = _stats(S, U, 5)
    eng = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=2, planning_horizon=6)
    eng.set_mlp(ws, bs, [ACT[a] for a in acts], stats)
    ev = O.Evaluator("cheetah", O.Handler(O.MLP(ws, bs, acts), False, True, stats))
    rng = np.random.default_rng(1)
    states = rng.normal(0, 0.1, (2, S)).astype(F)
    seq = rng.uniform(-1, 1, (50, 2, 6, U)).astype(F)
    np.testing.assert_allclose(eng.evaluate(states, seq), ev(states, seq), rtol=1e-3, atol=6e-3)
    s1 = rng.normal(0, 0.5, (17, S)).astype(F)
    a1 = rng.uniform(-1, 1, (17, U)).astype(F)
    np.testing.assert_allclose(eng.predict_next_state(s1, a1), ev.predict_next_state(s1, a1), rtol=2e-5, atol=2e-5)


WAVE_SPECS = [
    (PEND_MLP, True),                                                                     # tutorials: 3 hidden layers x 2 tiles
    (([4, 16, 3], ["tanh", None], 3, 1, "pendulum"), True),                               # one hidden tile, one hidden layer
    (([4, 64, 64, 3], ["relu", "sigmoid", None], 3, 1, "pendulum"), False),               # 4 tiles, other activations, raw I/O
    (([26, 32, 32, 20], ["tanh", "tanh", None], 20, 6, "cheetah"), True),                 # two input tiles, two output tiles
    (([26, 48, 20], ["tanh", "tanh"], 20, 6, "cheetah"), True),                           # 3 tiles, activation on the output
    (([4, 21, 24, 3], ["sigmoid", "tanh", None], 3, 1, "pendulum"), True),                # half-empty last tiles
]


@pytest.mark.parametrize("form", ["wave", "w4"])
@pytest.mark.parametrize("spec,normalized", WAVE_SPECS)
def test_small_networks_run_the_wave_kernel_and_match(L, monkeypatch, spec, normalized, form):
    # hidden width <= 64: kernels_mlp_wave.hpp (one wave per hidden tile, the recurrence in registers, a reward wave);
    # hidden width <= 32: kernels_mlp_w4.hpp (one wave per four particles, nothing through LDS between layers), the default
    # there.  Checked against the oracle at the evaluator tolerance and against the general kernel (same arithmetic up to
    # the order in which the bias and the K-split partial sums are added) much tighter
    dims, acts, S, U, reward = spec
    w4_shape = max(dims[1:-1]) <= 32
    if form == "w4" and not w4_shape:
        pytest.skip("k_rollout_mlp_w4 takes hidden layers of at most 32 units")
    if form == "wave":
        monkeypatch.setenv("BBMPC_MLP_W4", "0")
    N, A, H = 77, 2, 25
    rng = np.random.default_rng(11)
    states = O.cheetah_start_states(A, S) if reward == "cheetah" else O.pendulum_start_states(A)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, normalized, A=A, H=H)
    eng.set_profiling(True)
    got = eng.evaluate(states, seq)
    assert eng.get_profile()[2] == "k_rollout_mlp_" + form
    want = ev(states, seq)
    assert np.all(np.isfinite(want))
    if reward == "cheetah":
        assert_cheetah_rewards(got, want, 1e-3, 1e-3 * H, margin=lambda: cheetah_threshold_margin(ev, states, seq))
    else:
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-3 * H)
    monkeypatch.setenv("BBMPC_MLP_WAVE", "0")
    monkeypatch.setenv("BBMPC_MLP_W4", "0")
    gen, _, _, _ = _problem(L, dims, acts, S, U, reward, normalized, A=A, H=H)
    monkeypatch.delenv("BBMPC_MLP_WAVE")
    monkeypatch.delenv("BBMPC_MLP_W4")
    gen.set_profiling(True)
    ref = gen.evaluate(states, seq)
    assert gen.get_profile()[2] not in ("k_rollout_mlp_wave", "k_rollout_mlp_w4")
    if reward == "cheetah":
        assert_cheetah_rewards(got, ref, 2e-5, 2e-5 * H, margin=lambda: cheetah_threshold_margin(ev, states, seq))
    else:
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * H)


@pytest.mark.parametrize("form", ["wave", "w4"])
def test_wave_kernel_in_the_optimizers(L, monkeypatch, form):
    # PI2 (clip + penalty candidates), CEM and the __call__ tail on the tutorial network: same injected draws through the
    # small-network kernel (k_rollout_mlp_wave / k_rollout_mlp_w4) and the general kernel
    dims, acts, S, U, reward = PEND_MLP
    N, A, H, iters = 200, 3, 12, 3
    states = O.pendulum_start_states(A)
    rng = np.random.default_rng(5)
    draws = np.stack([O.truncated_normal_noise(rng, (N, A, H, U)) for _ in range(iters)])
    for opt, kw in ((L.OPT_PI2, dict(lamda=2.0)), (L.OPT_CEM, dict(k=20))):
        out = []
        for small in (True, False):
            monkeypatch.setenv("BBMPC_MLP_WAVE", "1" if small else "0")
            monkeypatch.setenv("BBMPC_MLP_W4", "1" if (small and form == "w4") else "0")
            eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=A, H=H, opt=opt, N=N, iters=iters, **kw)
            eng.set_trace(True)
            eng.set_profiling(True)
            eng.inject_noise(L.NOISE_TRUNC_NORMAL, draws)
            a, n, r = eng.optimize(states)
            assert (eng.get_profile()[2] == "k_rollout_mlp_" + form) == small
            out.append((a, n, r, [eng.get_trace(it, L.TRACE_REWARDS) for it in range(iters)]))
        monkeypatch.delenv("BBMPC_MLP_WAVE")
        monkeypatch.delenv("BBMPC_MLP_W4")
        for it in range(iters):
            np.testing.assert_allclose(out[0][3][it], out[1][3][it], rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(out[0][0], out[1][0], rtol=0, atol=2e-3)
        np.testing.assert_allclose(out[0][1], out[1][1], rtol=0, atol=2e-3)


@pytest.mark.parametrize("mode,q99,frac_flip", [("3", 1e-3, 0.005), ("1", 15.0, 0.1)])
def test_optional_bf16_modes_against_oracle(L, monkeypatch, mode, q99, frac_flip):
    # BBMPC_MLP_BF16 = 3: every MFMA operand split into bf16 hi + lo, three products per K tile (~16 mantissa bits per
    # factor, fp32 accumulate); = 1: plain bf16 inputs.  Opt-in modes with their OWN tolerance (never the default path):
    #   split : 99 % of the H=50 rollout rewards within 1e-3 of the oracle (median ~3e-5);
    #   plain : median error below 0.1 on rewards of spread ~200.
    # The cheetah reward has three indicator terms (cur[5] >= 0.2 etc., -10 each): a state that sits on a threshold can
    # flip one of them under ANY rounding change, so a small fraction of trajectories may differ by a multiple of 10.
    from oracle import oracle_c as OC
    monkeypatch.setenv("BBMPC_MLP_BF16", mode)
    dims, acts, S, U, reward = CHEETAH
    N, A, H = 2000, 2, 50
    eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=A, H=H)
    assert eng is not None
    ws, bs = O.make_mlp_params(dims, seed=42)
    rng = np.random.default_rng(43)
    bs = [rng.normal(0, 0.05, b.shape).astype(F) for b in bs]
    co = OC.COracle("mlp", "cheetah", lo, hi, N, A, H, S, mlp=(ws, bs, acts), stats=_stats(S, U, 44))
    rng = np.random.default_rng(9)
    states = O.cheetah_start_states(A, S)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    got, want = eng.evaluate(states, seq), co.evaluate(states, seq)
    err = np.abs(got - want)
    assert np.quantile(err, 0.99) < q99
    big = err > (0.05 if mode == "3" else 5.0)
    assert big.mean() < frac_flip
    if mode == "3":
        np.testing.assert_allclose(err[big] / 10.0, np.round(err[big] / 10.0), atol=0.02)     # flips are multiples of 10
        assert np.median(err) < 2e-4
    else:
        assert np.median(err) < 0.1


@pytest.mark.parametrize("pair", ["0", "1"])
@pytest.mark.parametrize("dims,S,U,reward", [
    ([17, 200, 200, 16], 16, 1, "pendulum"),     # one live input in the second K tile (1 of 4 MFMAs issued); one output tile
    ([32, 196, 196, 17], 17, 15, "pendulum"),    # both input tiles full; last hidden tile holds 4 features; ONE live row in the 4-row output tile
    ([12, 193, 193, 10], 10, 2, "pendulum"),     # a single, partly filled input tile (3 of 4 MFMAs); last hidden tile holds 1 feature
    ([26, 208, 208, 20], 20, 6, "cheetah"),      # thirteen FULL hidden tiles (no half-tile paths), four live rows in the 4-row tile
    ([22, 200, 200, 18], 18, 4, "cheetah"),      # 6 live inputs in the second K tile (2 of 4 MFMAs), two live rows
    ([28, 200, 200, 21], 21, 7, "cheetah"),      # dim_S = 21: the second output tile stays on 16x16x4 (five rows)
])
def test_pipelined_tile_kernel_padding_paths(L, monkeypatch, pair, dims, S, U, reward):
    # Round 5 stopped issuing the all-zero MFMAs of the layer-0 tail (input slots permuted, operands gathered in slot order) and
    # moved a <= 4-row second output tile to v_mfma_f32_4x4x1_16b_f32 (k rows summed by lane swaps): every combination of
    # live MFMAs / live rows against the NumPy oracle, one-tile and two-tile form, which must also agree with each other bit for bit.
    monkeypatch.setenv("BBMPC_MLP_Q4", "0")
    monkeypatch.setenv("BBMPC_MLP_PAIR", pair)
    N, A, H = 70, 2, 6
    eng, ev, lo, hi = _problem(L, dims, ["tanh", "tanh", None], S, U, reward, True, A=A, H=H)
    rng = np.random.default_rng(sum(dims))
    states = (O.cheetah_start_states(A, S) if reward == "cheetah" else rng.normal(0, 0.3, (A, S)).astype(F))
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    eng.set_profiling(True)
    got = eng.evaluate(states, seq)
    assert eng.get_profile()[2] == "k_rollout_mlp_pair"
    want = ev(states, seq)
    assert np.all(np.isfinite(want))
    if reward == "cheetah":
        assert_cheetah_rewards(got, want, 1e-3, 1e-3 * H, margin=lambda: cheetah_threshold_margin(ev, states, seq))
    else:
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-3 * H)
    monkeypatch.setenv("BBMPC_MLP_PAIR", "1" if pair == "0" else "0")
    other, _, _, _ = _problem(L, dims, ["tanh", "tanh", None], S, U, reward, True, A=A, H=H)
    np.testing.assert_array_equal(other.evaluate(states, seq), got)


def test_profile_instantiation_names_the_template_that_ran(L):
    # bbmpc_profile_instantiation: what bench.py keys the committed PMC counters by (the strict-math and the default
    # instantiation of the persistent kernel share a plain name)
    from blackbox_mpc_amd.engine import Engine
    mk = lambda **kw: Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=30,
                             population_size=500, max_iterations=5, num_elite=50, **kw)
    st = O.pendulum_start_states(1)
    fast, strict = mk(), mk(quirks=L.STRICT_MATH)
    for e in (fast, strict):
        e.optimize(st)
        e.optimize(st)
        assert e.get_profile()[2] == "k_fused_pendulum"
        assert e.device == 0
    a, b = fast.profile_instantiation(), strict.profile_instantiation()
    assert a.startswith("k_fused_pendulum<2, true, true, ") and b.startswith("k_fused_pendulum<2, true, false, "), (a, b)
    # the quad kernel and the small-network kernel report theirs as rocprofv3 prints them (round 6: several instantiations of
    # one plain name sit in one profile of the shape sweep); kernels that do not record one report the plain name
    eng, ev, lo, hi = _problem(L, *CHEETAH, True, A=1, H=5)
    eng.evaluate(O.cheetah_start_states(1, 20), np.zeros((8, 1, 5, 6), F))
    assert eng.profile_instantiation() == "k_rollout_mlp_q4s<50, 7, 1, 1, 0, 1>" and eng.get_profile()[2] == "k_rollout_mlp_q4s"
    eng, ev, lo, hi = _problem(L, *PEND_MLP, True, A=1, H=5)
    eng.evaluate(O.pendulum_start_states(1), np.zeros((8, 1, 5, 1), F))
    assert eng.profile_instantiation() == "k_rollout_mlp_w4<4, true>" and eng.get_profile()[2] == "k_rollout_mlp_w4"
    eng, ev, lo, hi = _problem(L, [26, 500, 500, 500, 20], ["tanh", "tanh", "tanh", None], 20, 6, "cheetah", True, A=1, H=3)
    eng.evaluate(O.cheetah_start_states(1, 20), np.zeros((8, 1, 3, 6), F))
    assert eng.profile_instantiation() == eng.get_profile()[2] == "k_rollout_mlp"


Q4S_USER_REWARD = """
__device__ float bbmpc_user_reward(const float* cur, const float* act, const float* nxt, int S, int U) {
    float r = 0.0f;
    for (int s = 0; s < S; ++s) r = r + (nxt[s] - cur[s]) * (0.25f + 0.125f * (float)(s & 3));
    for (int u = 0; u < U; ++u) r = r - 0.0625f * (act[u] * act[u]);
    return r;
}
"""


def _q4s_user_reward_np(cur, act, nxt):
    S, U = cur.shape[1], act.shape[1]
    r = np.zeros(cur.shape[0], F)
    for s in range(S):
        r = (r + ((nxt[:, s] - cur[:, s]).astype(F) * F(0.25 + 0.125 * (s & 3))).astype(F)).astype(F)
    for u in range(U):
        r = (r - (F(0.0625) * (act[:, u] * act[:, u]).astype(F)).astype(F)).astype(F)
    return r


@pytest.mark.parametrize("acts,S,U,normalized,N,H", [
    (["tanh", "tanh", None], 20, 6, True, 100, 30),
    (["relu", "relu", None], 20, 6, True, 100, 30),            # compile-time relu instantiation
    (["sigmoid", "tanh", None], 20, 6, True, 37, 12),          # run-time activations
    (["tanh", "relu", "tanh"], 20, 6, False, 37, 12),          # an activation on the last layer, un-normalised
    (["tanh", "tanh", None], 18, 6, True, 100, 30),            # dim_S < 20: padded state groups (cheetah reward needs 18)
    (["relu", "relu", None], 19, 8, True, 64, 9),              # dim_U = 8: both action groups full
    (["tanh", "tanh", None], 20, 3, False, 41, 50),            # dim_U = 3: action groups mostly padding
    (["tanh", "tanh", None], 20, 8, False, 41, 40),            # 320 action elements per particle: two (particle, 4-element) pairs per thread
])
def test_quad_kernel_q4s_other_activations_and_state_widths(L, acts, S, U, normalized, N, H):
    # k_rollout_mlp_q4s is compiled for two hidden layers of 200 units; activations and dim_S <= 20 / dim_U <= 8 are free
    # (SURVEY 8 preamble: stock HalfCheetah-v2 has 17 states; deterministic_mlp.py:5-25 takes any activation list)
    A = 2
    dims = [S + U, 200, 200, S]
    eng, ev, lo, hi = _problem(L, dims, acts, S, U, "cheetah", normalized, A=A, H=H)
    rng = np.random.default_rng(S * 100 + U)
    states = rng.normal(0, 0.3, (A, S)).astype(F)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    eng.set_profiling(True)
    got = eng.evaluate(states, seq)
    assert eng.get_profile()[2] == "k_rollout_mlp_q4s"
    # the template instantiation as rocprofv3 names it (bench.py attaches committed counters by it): compile-time activations
    # for the tanh / relu networks, -1 (run time) otherwise; two action pairs per thread from 257 (particle, 4-element) pairs on
    code = {"tanh": 1, "relu": 2}
    ct = acts[2] is None and acts[0] == acts[1] and acts[0] in code
    a = [code[acts[0]], code[acts[1]], 0] if ct else [-1, -1, -1]
    assert eng.profile_instantiation() == "k_rollout_mlp_q4s<50, 7, %d, %d, %d, %d>" % (a[0], a[1], a[2], 1 if 4 * ((H * U + 3) // 4) <= 256 else 2)
    want = ev(states, seq)
    assert np.all(np.isfinite(want))
    assert_cheetah_rewards(got, want, 1e-3, 1e-3 * H, margin=lambda: cheetah_threshold_margin(ev, states, seq))


@pytest.mark.parametrize("S,U", [(17, 6), (9, 2)])
def test_quad_kernel_q4s_short_state_with_user_reward(L, S, U):
    # dim_S = 17 (stock HalfCheetah-v2) has no built-in reward (cost_func.py:18 indexes state[17]): a user function scores the
    # trajectory the quad kernel records -- rows of dim_S floats, written element by element when dim_S is no multiple of 4
    from blackbox_mpc_amd.engine import Engine
    A, N, H = 2, 100, 20
    dims, acts = [S + U, 200, 200, S], ["tanh", "tanh", None]
    ws, bs = O.make_mlp_params(dims, seed=5)
    rng = np.random.default_rng(S)
    bs = [rng.normal(0, 0.05, b.shape).astype(F) for b in bs]
    stats = _stats(S, U, 11)
    lo, hi = [-1.0] * U, [1.0] * U
    eng = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_USER, lo, hi, dim_s=S, num_agents=A, planning_horizon=H)
    eng.set_reward_source(Q4S_USER_REWARD)
    eng.set_mlp(ws, bs, [ACT[a] for a in acts], stats)
    ev = O.Evaluator(_q4s_user_reward_np, O.Handler(O.MLP(ws, bs, acts), False, True, stats))
    states = rng.normal(0, 0.3, (A, S)).astype(F)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    eng.set_profiling(True)
    got = eng.evaluate(states, seq)
    assert eng.get_profile()[2] == "k_rollout_mlp_q4s"
    np.testing.assert_allclose(got, ev(states, seq), rtol=1e-3, atol=1e-3 * H)
    # ... and the next state of a single step through the same network (the tail's kernel) agrees with the oracle
    a1 = seq[0, :, 0]
    np.testing.assert_allclose(eng.predict_next_state(states, a1), ev.predict_next_state(states, a1), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("acts,S,U,normalized,N,H", [
    (["tanh", "tanh", None], 20, 6, True, 100, 30),
    (["relu", "sigmoid", None], 18, 8, False, 37, 40),         # run-time activations, dim_S < 20, two action pairs per thread
])
def test_quad_kernel_q4s_256_hidden_units(L, acts, S, U, normalized, N, H):
    # the four-job form of k_rollout_mlp_q4s: 26-256-256-20, every wave four 16-feature jobs of layer 1 and no tail chain
    A = 2
    dims = [S + U, 256, 256, S]
    eng, ev, lo, hi = _problem(L, dims, acts, S, U, "cheetah", normalized, A=A, H=H)
    rng = np.random.default_rng(S * 10 + U)
    states = rng.normal(0, 0.3, (A, S)).astype(F)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    eng.set_profiling(True)
    got = eng.evaluate(states, seq)
    assert eng.get_profile()[2] == "k_rollout_mlp_q4s"
    want = ev(states, seq)
    assert np.all(np.isfinite(want))
    assert_cheetah_rewards(got, want, 1e-3, 1e-3 * H, margin=lambda: cheetah_threshold_margin(ev, states, seq))


def _random_shapes(seed, count, hidden_choices, n_hidden_choices):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        S = int(rng.integers(1, 21))
        U = int(rng.integers(1, 9))
        H = int(rng.choice([1, 2, 7, 16, 31, 33, 64]))
        while 4 * ((H * U + 3) // 4) > 512:                # the quad kernel keeps at most 512 (particle, 4-element) pairs per workgroup
            H //= 2
        N = int(rng.choice([1, 3, 4, 5, 17, 64, 130]))
        A = int(rng.integers(1, 4))
        nh = int(rng.choice(n_hidden_choices))
        hid = [int(rng.choice(hidden_choices)) for _ in range(nh)]
        acts = [str(rng.choice(["tanh", "relu", "sigmoid"])) for _ in range(nh)] + [None if rng.random() < 0.7 else "tanh"]
        out.append((S, U, H, N, A, hid, acts, bool(rng.random() < 0.7)))
    return out


@pytest.mark.parametrize("case", _random_shapes(601, 14, [200, 256], [2]), ids=lambda c: "S%d-U%d-H%d-N%d-A%d-%s" % (c[0], c[1], c[2], c[3], c[4], "x".join(map(str, c[5]))))
def test_quad_kernel_q4s_random_shapes(L, case):
    # k_rollout_mlp_q4s over drawn (dim_S, dim_U, horizon, population, agents, width, activations, normalisation): ragged
    # populations (the last quad partly empty), one-step horizons, one or two action pairs per thread, padded state groups
    from blackbox_mpc_amd.engine import Engine
    S, U, H, N, A, hid, acts, normalized = case
    if hid[0] != hid[1]:
        hid = [hid[0], hid[0]]
    dims = [S + U] + hid + [S]
    ws, bs = O.make_mlp_params(dims, seed=S * 31 + U)
    rng = np.random.default_rng(S * 1000 + U * 10 + H)
    bs = [rng.normal(0, 0.05, b.shape).astype(F) for b in bs]
    stats = _stats(S, U, 3) if normalized else None
    lo, hi = [-1.0] * U, [1.0] * U
    eng = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_USER, lo, hi, dim_s=S, num_agents=A, planning_horizon=H)
    eng.set_reward_source(Q4S_USER_REWARD)
    eng.set_mlp(ws, bs, [ACT[a] for a in acts], stats)
    ev = O.Evaluator(_q4s_user_reward_np, O.Handler(O.MLP(ws, bs, acts), False, normalized, stats))
    states = rng.normal(0, 0.3, (A, S)).astype(F)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    eng.set_profiling(True)
    got = eng.evaluate(states, seq)
    assert eng.get_profile()[2] == "k_rollout_mlp_q4s"
    np.testing.assert_allclose(got, ev(states, seq), rtol=1e-3, atol=1e-3 * H)


@pytest.mark.parametrize("case", _random_shapes(602, 14, [4, 8, 13, 16, 24, 32], [1, 2, 3]), ids=lambda c: "S%d-U%d-H%d-N%d-A%d-%s" % (c[0], c[1], c[2], c[3], c[4], "x".join(map(str, c[5]))))
def test_small_network_kernel_w4_random_shapes(L, case):
    # k_rollout_mlp_w4 over drawn shapes: 1-3 hidden layers of 4..32 units each (padded to 32 x 32 stages), any activations
    from blackbox_mpc_amd.engine import Engine
    S, U, H, N, A, hid, acts, normalized = case
    dims = [S + U] + hid + [S]
    ws, bs = O.make_mlp_params(dims, seed=S * 17 + U)
    rng = np.random.default_rng(S * 1000 + U * 10 + H + 5)
    bs = [rng.normal(0, 0.05, b.shape).astype(F) for b in bs]
    stats = _stats(S, U, 4) if normalized else None
    lo, hi = [-1.0] * U, [1.0] * U
    eng = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_USER, lo, hi, dim_s=S, num_agents=A, planning_horizon=H)
    eng.set_reward_source(Q4S_USER_REWARD)
    eng.set_mlp(ws, bs, [ACT[a] for a in acts], stats)
    ev = O.Evaluator(_q4s_user_reward_np, O.Handler(O.MLP(ws, bs, acts), False, normalized, stats))
    states = rng.normal(0, 0.3, (A, S)).astype(F)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    eng.set_profiling(True)
    got = eng.evaluate(states, seq)
    assert eng.get_profile()[2] == "k_rollout_mlp_w4"
    np.testing.assert_allclose(got, ev(states, seq), rtol=1e-3, atol=1e-3 * H)


@pytest.mark.parametrize("opt_name", ["RandomSearch", "PSO", "SPSA", "CMA-ES"])
def test_small_network_kernel_w4_under_every_candidate_source(L, monkeypatch, opt_name):
    # the optimizers hand the rollout kernels their candidates in different forms (uniform draws made in the kernel, a
    # candidate buffer, clip + penalty): the first iteration's rewards through k_rollout_mlp_w4 and through the general kernel
    # on the engine's own draws (counter-based: the same for both), and the control step's outputs
    dims, acts, S, U, reward = PEND_MLP
    N, A, H = 192, 2, 10
    opt = {"RandomSearch": L.OPT_RANDOM_SEARCH, "PSO": L.OPT_PSO, "SPSA": L.OPT_SPSA, "CMA-ES": L.OPT_CMAES}[opt_name]
    states = O.pendulum_start_states(A)
    out = []
    for small in (True, False):
        monkeypatch.setenv("BBMPC_MLP_WAVE", "1" if small else "0")
        monkeypatch.setenv("BBMPC_MLP_W4", "1" if small else "0")
        eng, ev, lo, hi = _problem(L, dims, acts, S, U, reward, True, A=A, H=H, opt=opt, N=N, iters=(0 if opt_name == "RandomSearch" else 1), k=16, seed=7)
        eng.set_trace(True)
        eng.set_profiling(True)
        a, n, r = eng.optimize(states)
        assert (eng.get_profile()[2] == "k_rollout_mlp_w4") == small
        out.append((a, n, r, eng.get_trace(0, L.TRACE_REWARDS)))
    monkeypatch.delenv("BBMPC_MLP_WAVE")
    monkeypatch.delenv("BBMPC_MLP_W4")
    np.testing.assert_allclose(out[0][3], out[1][3], rtol=1e-4, atol=1e-3)
    if opt_name != "SPSA":                          # (SPSA divides reward differences by 2 c_k: its step amplifies the kernels' last bits)
        np.testing.assert_allclose(out[0][0], out[1][0], rtol=0, atol=2e-3)
        np.testing.assert_allclose(out[0][1], out[1][1], rtol=0, atol=2e-3)
