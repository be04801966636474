"""Plain callables as reward_function / dynamics_function, as the reference's users pass them
(trajectory_evaluators/deterministic.py:13-18; called at :65-66 and :99-100): accepted when they work on PyTorch CUDA
tensors.  The engine evaluates step by step and calls back per planning step with its own HBM row batches aliased as
torch tensors (include/bbmpc.h bbmpc_set_*_callback, utils/device_functions.py); nothing runs on the CPU.

Checked: the same functions written as HIP source give the same results within 1e-5 (evaluator rewards within the
pendulum tolerance), through MPCPolicy.act with CEM; a torch.nn.Module as a learned, normalised dynamics model against
the engine's own MFMA path on the same weights; a custom inverse_transform_targets_func; a host-only callable is still
refused, and an exception raised inside a callback comes back as the cause of the BBMPCError."""
import numpy as np
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
F = np.float32
LO, HI = [-2.0], [2.0]

from tests.test_gpu_user_functions import INTENDED_PENDULUM_REWARD, USER_PENDULUM_MODEL      # noqa: E402  the HIP twins


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1
    return _lib


def torch_reward(cur, act, nxt):
    """utils/pendulum.py:10-35 in its DECLARED argument order, in torch (the twin of INTENDED_PENDULUM_REWARD)."""
    import math
    import torch
    th = torch.atan2(cur[:, 1], cur[:, 0])
    ang = torch.remainder(th + math.pi, 2.0 * math.pi) - math.pi
    return -(ang * ang + 0.1 * cur[:, 2] ** 2) - 0.001 * (act * act).sum(dim=1)


def torch_pendulum(x, train=False):
    """utils/pendulum.py:58-92 in torch: returns the state DELTA (the twin of USER_PENDULUM_MODEL)."""
    import math
    import torch
    th = torch.atan2(x[:, 1], x[:, 0])
    acc = -15.0 * torch.sin(th + math.pi) + 3.0 * x[:, 3]
    nthd = x[:, 2] + acc * 0.05
    nth = th + nthd * 0.05
    nthd = torch.clamp(nthd, -8.0, 8.0)
    return torch.stack([torch.cos(nth) - x[:, 0], torch.sin(nth) - x[:, 1], nthd - x[:, 2]], dim=1)


def _policy(reward, dynamics, seed=3, **kw):
    from blackbox_mpc_amd.policies import MPCPolicy
    from blackbox_mpc_amd.spaces import Box
    return MPCPolicy(reward_function=reward, env_action_space=Box(LO, HI), env_observation_space=Box([-1, -1, -8], [1, 1, 8]),
                     true_model=True, dynamics_function=dynamics, optimizer_name="CEM", num_agents=2, planning_horizon=12,
                     population_size=192, max_iterations=3, num_elite=24, seed=seed, **kw)


def test_torch_callables_drive_cem_like_their_hip_twins(L):
    from blackbox_mpc_amd.utils.device_functions import HipDynamicsFunction, HipRewardFunction
    pol_t = _policy(lambda cur, act, nxt: torch_reward(cur, act, nxt), torch_pendulum)
    pol_h = _policy(HipRewardFunction(INTENDED_PENDULUM_REWARD), HipDynamicsFunction(USER_PENDULUM_MODEL, dim_s=3, dim_u=1))
    obs = O.pendulum_start_states(2)
    for t in range(6):
        a_t, n_t, r_t = pol_t.act(obs, t)
        a_h, n_h, r_h = pol_h.act(obs, t)
        # same engine draws (seed), same arithmetic up to libm / torch rounding of sin, cos, atan2
        np.testing.assert_allclose(a_t, a_h, rtol=0, atol=1e-5)
        np.testing.assert_allclose(n_t, n_h, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r_t, r_h, rtol=1e-5, atol=1e-5)
        obs = n_h
    seq = np.random.default_rng(2).uniform(-2, 2, (64, 2, 12, 1)).astype(F)
    got_t = pol_t._trajectory_evaluator(obs, seq)
    got_h = pol_h._trajectory_evaluator(obs, seq)
    np.testing.assert_allclose(got_t, got_h, rtol=2e-4, atol=2e-3)
    # and against the oracle's restatement of the same functions
    intended = lambda c, a, n: O.pendulum_reward(c, a, n, as_executed=False)
    want = O.Evaluator(intended, O.Handler(O.pendulum_dynamics, True))(obs, seq)
    np.testing.assert_allclose(got_t, want, rtol=2e-4, atol=2e-3)


def test_torch_reward_with_the_builtin_model_and_other_optimizers(L):
    from blackbox_mpc_amd.policies import MPCPolicy
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.utils.device_functions import HipRewardFunction
    from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel
    for name, extra in (("PI2", {}), ("RandomSearch", {}), ("PSO", {})):
        kw = dict(env_action_space=Box(LO, HI), env_observation_space=Box([-1, -1, -8], [1, 1, 8]), true_model=True,
                  dynamics_function=PendulumTrueModel(), optimizer_name=name, num_agents=1, planning_horizon=10,
                  population_size=128, seed=5)
        if name != "RandomSearch":
            kw["max_iterations"] = 2
        pol_t = MPCPolicy(reward_function=torch_reward, **kw)
        pol_h = MPCPolicy(reward_function=HipRewardFunction(INTENDED_PENDULUM_REWARD), **kw)
        pol_t.reset(); pol_h.reset()
        obs = O.pendulum_start_states(1)
        for t in range(3):
            a_t, n_t, r_t = pol_t.act(obs, t)
            a_h, n_h, r_h = pol_h.act(obs, t)
            np.testing.assert_allclose(a_t, a_h, rtol=0, atol=2e-5, err_msg=name)
            np.testing.assert_allclose(r_t, r_h, rtol=1e-5, atol=1e-5, err_msg=name)
            obs = n_h


def test_torch_module_as_learned_normalised_dynamics(L):
    """A torch.nn.Module with the weights of a DeterministicMLP, behind a normalising handler: the step-wise torch path
    against the engine's fused MFMA path on the same weights and statistics (MLP tolerance of tests/test_gpu_mlp.py)."""
    import torch
    from blackbox_mpc_amd.dynamics_functions import DeterministicMLP
    from blackbox_mpc_amd.dynamics_handlers import SystemDynamicsHandler
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.trajectory_evaluators import DeterministicTrajectoryEvaluator
    from blackbox_mpc_amd.utils.cheetah import reward_function
    S, U, H, N, A = 20, 6, 12, 96, 2
    dims = [S + U, 200, 200, S]
    ws, bs = O.make_mlp_params(dims, seed=11)
    rng = np.random.default_rng(12)
    stats = [rng.normal(0, 0.2, S).astype(F), rng.uniform(0.5, 1.5, S).astype(F), rng.normal(0, 0.1, U).astype(F),
             rng.uniform(0.5, 1.5, U).astype(F), rng.normal(0, 0.01, S).astype(F), rng.uniform(0.05, 0.15, S).astype(F)]
    act_space, obs_space = Box([-1.0] * U, [1.0] * U), Box([-10.0] * S, [10.0] * S)
    net = DeterministicMLP(layers=dims, activation_functions=["tanh", "tanh", None], seed=1)
    net.set_weights(ws, bs)
    h_ref = SystemDynamicsHandler(act_space, obs_space, dynamics_function=net, true_model=False, is_normalized=True)
    h_ref.set_normalization_stats(*stats)
    mod = torch.nn.Sequential(torch.nn.Linear(dims[0], 200), torch.nn.Tanh(), torch.nn.Linear(200, 200), torch.nn.Tanh(),
                              torch.nn.Linear(200, S)).cuda()
    with torch.no_grad():
        for lin, w, b in zip([mod[0], mod[2], mod[4]], ws, bs):
            lin.weight.copy_(torch.from_numpy(w.T.copy()))
            lin.bias.copy_(torch.from_numpy(b))
    mod.requires_grad_(False)
    h_t = SystemDynamicsHandler(act_space, obs_space, dynamics_function=mod, true_model=False, is_normalized=True)
    h_t.set_normalization_stats(*stats)
    ev_ref = DeterministicTrajectoryEvaluator(reward_function, h_ref)
    ev_t = DeterministicTrajectoryEvaluator(reward_function, h_t)          # built-in reward over a torch model
    st = O.cheetah_start_states(A, S)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    np.testing.assert_allclose(ev_t(st, seq), ev_ref(st, seq), rtol=1e-3, atol=1e-3 * H)
    s1, a1 = rng.normal(0, 0.5, (8, S)).astype(F), rng.uniform(-1, 1, (8, U)).astype(F)
    np.testing.assert_allclose(ev_t.predict_next_state(s1, a1), ev_ref.predict_next_state(s1, a1), rtol=2e-5, atol=2e-4)
    # both a torch reward and a torch model
    cheetah_t = lambda cur, act, nxt: (-10.0 * (cur[:, 5] >= 0.2) - 10.0 * (cur[:, 6] >= 0.0) - 10.0 * (cur[:, 7] >= 0.0)
                                       + (nxt[:, 17] - cur[:, 17]) / 0.01 - 0.0 * (act * act).sum(dim=1))
    ev_tt = DeterministicTrajectoryEvaluator(cheetah_t, h_t)
    np.testing.assert_allclose(ev_tt(st, seq), ev_ref(st, seq), rtol=1e-3, atol=1e-3 * H)


def test_custom_inverse_target_transform_on_the_torch_path(L):
    """inverse_transform_targets_func(states, raw) (reference system_dynamics_handler.py:128-161): a model that
    predicts the absolute next state instead of the delta."""
    from blackbox_mpc_amd.dynamics_handlers import SystemDynamicsHandler
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.trajectory_evaluators import DeterministicTrajectoryEvaluator
    from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function
    act, obs = Box(LO, HI), Box([-1, -1, -8], [1, 1, 8])
    absolute = lambda x, train=False: torch_pendulum(x) + x[:, :3]          # predicts next state itself
    h_abs = SystemDynamicsHandler(act, obs, dynamics_function=absolute, true_model=True,
                                  inverse_transform_targets_func=lambda states, raw: raw)
    h_ref = SystemDynamicsHandler(act, obs, dynamics_function=PendulumTrueModel(), true_model=True)
    st = O.pendulum_start_states(2)
    seq = np.random.default_rng(4).uniform(-2, 2, (40, 2, 9, 1)).astype(F)
    got = DeterministicTrajectoryEvaluator(pendulum_reward_function, h_abs)(st, seq)
    want = DeterministicTrajectoryEvaluator(pendulum_reward_function, h_ref)(st, seq)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-3)
    with pytest.raises(NotImplementedError, match="inverse_transform_targets_func"):
        h_bad = SystemDynamicsHandler(act, obs, dynamics_function=PendulumTrueModel(), true_model=True,
                                      inverse_transform_targets_func=lambda s, r: r)
        DeterministicTrajectoryEvaluator(pendulum_reward_function, h_bad)(st, seq)


def test_host_only_callables_are_still_refused_and_callback_errors_surface(L):
    from blackbox_mpc_amd.dynamics_handlers import SystemDynamicsHandler
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.trajectory_evaluators import DeterministicTrajectoryEvaluator
    from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel
    act, obs = Box(LO, HI), Box([-1, -1, -8], [1, 1, 8])
    h = SystemDynamicsHandler(act, obs, dynamics_function=PendulumTrueModel(), true_model=True)
    st = O.pendulum_start_states(1)
    seq = np.zeros((8, 1, 4, 1), F)
    numpy_only = lambda cur, a, nxt: -np.asarray(cur)[:, 0]                 # np.asarray of a CUDA tensor raises
    with pytest.raises(NotImplementedError, match="HipRewardFunction"):
        DeterministicTrajectoryEvaluator(numpy_only, h)(st, seq)
    with pytest.raises(NotImplementedError, match="torch tensor on the GPU"):
        DeterministicTrajectoryEvaluator(lambda c, a, n: 0.0, h)(st, seq)
    # fails only on the real batch (8 rows), i.e. inside the engine's callback: the Python exception is the cause
    calls = {"n": 0}

    def flaky(cur, a, nxt):
        calls["n"] += 1
        if cur.shape[0] > 2:
            raise RuntimeError("boom in the reward")
        return -cur[:, 0]
    ev = DeterministicTrajectoryEvaluator(flaky, h)
    with pytest.raises(L.BBMPCError, match="reward callback") as info:
        ev(st, seq)
    assert isinstance(info.value.__cause__, RuntimeError) and "boom" in str(info.value.__cause__)
    # the handle is still usable afterwards
    np.testing.assert_allclose(DeterministicTrajectoryEvaluator(lambda c, a, n: -c[:, 0], h)(st, seq)[:, 0],
                               DeterministicTrajectoryEvaluator(lambda c, a, n: -c[:, 0], h)(st, seq)[:, 0])
