"""Dynamics-model training row on the GPU: HIP-graph-replayed training step against the NumPy
oracle, and the whole iterative loop (collect with MPC through the learned model -> refit -> re-upload)."""
import time

import numpy as np
import pytest

from oracle import oracle_np as O
from oracle import oracle_train as OT
from tests.test_train_cpu import _episodes, _handler

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1
    return _lib


@pytest.mark.parametrize("graph", ["1", "0"])
def test_gpu_training_matches_numpy_oracle(L, monkeypatch, graph):
    monkeypatch.setenv("BBMPC_TRAIN_GRAPH", graph)
    obs, acs, rews = _episodes(4, 40, 2, 2)
    h, fn = _handler(layers=(4, 64, 64, 3), acts=("tanh", "tanh", None), seed=3)
    w0, b0 = [w.copy() for w in fn.weights], [b.copy() for b in fn.biases]
    d_in, d_out = OT.assemble_dataset(obs, acs)
    rng = np.random.default_rng(4)
    mask = rng.random(d_in.shape[0]) > 0.25
    epochs, B = 6, 32
    perms = [rng.permutation(int(mask.sum())) for _ in range(epochs)]
    h.train(obs, acs, rews, batch_size=B, learning_rate=2e-3, epochs=epochs, split_mask=mask, permutations=perms)
    stats = OT.normalization_stats(d_in[mask], d_out[mask], 3)
    tin, tout = OT.normalize(d_in[mask], d_out[mask], stats, 3)
    vin, vout = OT.normalize(d_in[~mask], d_out[~mask], stats, 3)
    w, b, tl, vl = OT.train(w0, b0, ["tanh", "tanh", None], tin, tout, vin, vout, perms, batch_size=B, learning_rate=2e-3)
    for got, want in zip(fn.weights + fn.biases, w + b):
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-4)
    np.testing.assert_allclose(h.training_loss, tl, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(h.validation_loss, vl, rtol=2e-4, atol=1e-6)


def test_iterative_mpc_learns_the_pendulum_and_replans_through_the_new_model(L):
    from blackbox_mpc_amd import Box
    from blackbox_mpc_amd.dynamics_functions.deterministic_mlp import DeterministicMLP
    from blackbox_mpc_amd.dynamics_handlers.system_dynamics_handler import SystemDynamicsHandler
    from blackbox_mpc_amd.policies import RandomPolicy
    from blackbox_mpc_amd.trajectory_evaluators.deterministic import DeterministicTrajectoryEvaluator
    from blackbox_mpc_amd.utils.iterative_mpc import learn_dynamics_iteratively_w_mpc
    from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function
    from blackbox_mpc_amd.utils.rollouts import ModelEnvironment
    A = 4
    act_space, obs_space = Box(low=[-2.0], high=[2.0]), Box(low=[-1, -1, -8], high=[1, 1, 8])
    true_h = SystemDynamicsHandler(act_space, obs_space, dynamics_function=PendulumTrueModel(), true_model=True)
    true_ev = DeterministicTrajectoryEvaluator(pendulum_reward_function, true_h)
    env = ModelEnvironment(true_ev, O.pendulum_start_states(A))      # the "real system" = analytic pendulum on the GPU
    fn = DeterministicMLP([4, 64, 64, 3], ["tanh", "tanh", None], seed=0)
    handler, policy = learn_dynamics_iteratively_w_mpc(
        env, number_of_initial_rollouts=6, number_of_rollouts_for_refinement=2, number_of_refinement_steps=2,
        task_horizon=60, env_action_space=act_space, env_observation_space=obs_space,
        initial_policy=RandomPolicy(A, act_space, seed=0), planning_horizon=15,
        reward_function=pendulum_reward_function, optimizer_name="CEM", num_agents=A, dynamics_function=fn,
        epochs=40, batch_size=64, learning_rate=3e-3, train_args={"seed": 0},
        max_iterations=3, population_size=128, num_elite=16)
    assert handler._refining_model_iter == 3 and handler._training_iter == 3
    n_rows = handler._model_training_in.shape[0] + handler._model_validation_in.shape[0]
    assert n_rows == (6 + 2 * 2) * A * 60
    assert handler.validation_loss[-1] < 0.05                         # normalised-target MSE
    # the policy plans through the refitted model: its one-step prediction tracks the true system
    s = O.pendulum_start_states(A)
    a, pred, _ = policy.act(s, 0)
    true_next = true_ev.predict_next_state(s, a)
    assert np.max(np.abs(pred - true_next)) < 0.15
    learned_ev = policy._trajectory_evaluator
    rng = np.random.default_rng(1)
    sa = rng.uniform(-2, 2, (A, 1)).astype(F)
    # engine's learned-model step == host forward pass of the trained weights (float64) within the MLP tolerance
    x, _ = OT.normalize(np.concatenate([s, sa], 1), np.zeros((A, 3), F), handler.normalization_stats(), 3)
    raw = OT.forward(fn.weights, fn.biases, ["tanh", "tanh", None], x)[-1]
    ms, ss, ma, sa_, mt, st = handler.normalization_stats()
    want = s + mt + raw * (st + 1e-7)
    np.testing.assert_allclose(learned_ev.predict_next_state(s, sa), want, rtol=2e-5, atol=2e-5)


def test_training_step_rate_with_and_without_hip_graph(L, monkeypatch, capsys):
    rng = np.random.default_rng(0)
    n, S, U = 32768, 20, 6
    din, dout = rng.normal(size=(n, S + U)).astype(F), rng.normal(size=(n, S)).astype(F)
    from blackbox_mpc_amd.dynamics_functions._train_torch import DenseTrainer
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=1)
    rates = {}
    for graph in ("0", "1"):
        monkeypatch.setenv("BBMPC_TRAIN_GRAPH", graph)
        tr = DenseTrainer(ws, bs, [1, 1, 0], "cuda")
        tr.fit(din, dout, din[:256], dout[:256], 1, 128, generator_seed=0)
        t0 = time.perf_counter()
        tl, _ = tr.fit(din, dout, din[:256], dout[:256], 2, 128, generator_seed=1)
        rates[graph] = 2 * (n // 128) / (time.perf_counter() - t0)
        assert np.all(np.isfinite(tl))
    with capsys.disabled():
        print("\n[train] Adam steps/s (26-200-200-20, batch 128): eager %.0f, HIP graph %.0f" % (rates["0"], rates["1"]))
    assert rates["1"] > 0.7 * rates["0"]          # a measurement, not a race: only a collapse of the graph path fails it
