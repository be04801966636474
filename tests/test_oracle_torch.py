"""oracle/oracle_torch.py -- the torch-CPU op-by-op restatement that bench.py reports as `cpu_baseline` for the learned-model
legs -- against the NumPy oracle on the same injected draws.  torch's sgemm / libm are not correctly rounded
(oracle_np's are): fp32 tolerances."""
import numpy as np
import pytest

from oracle import oracle_np as O
from oracle import oracle_torch as T

F = np.float32


def _np_ev(env, dims=None):
    if env == "pendulum":
        return O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True)), None, None
    ws, bs = O.make_mlp_params(dims, seed=42)
    acts = ["tanh"] * (len(dims) - 2) + [None]
    z, o = np.zeros, np.ones
    stats = [z(20, F), o(20, F), z(6, F), o(6, F), z(20, F), np.full(20, 0.1, F)]      # the bench's statistics
    return O.Evaluator("cheetah", O.Handler(O.MLP(ws, bs, acts), False, True, stats)), (ws, bs, acts), stats


@pytest.mark.parametrize("env,opt,N,A,H,dims", [
    ("pendulum", "CEM", 64, 2, 8, None),
    ("pendulum", "PI2", 48, 3, 6, None),
    ("pendulum", "RandomSearch", 40, 1, 5, None),
    ("cheetah", "PI2", 32, 1, 6, [26, 200, 200, 20]),
    ("cheetah", "CEM", 32, 2, 5, [26, 200, 200, 20]),
    ("cheetah", "RandomSearch", 24, 1, 4, [26, 500, 500, 500, 20]),      # tutorials/mujoco/tutorial_two.py's network
])
def test_torch_restatement_follows_numpy_oracle(env, opt, N, A, H, dims):
    rng = np.random.default_rng(5)
    ev, mlp, stats = _np_ev(env, dims)
    U, S = (1, 3) if env == "pendulum" else (6, 20)
    lo, hi = ([-2.0], [2.0]) if env == "pendulum" else ([-1.0] * U, [1.0] * U)
    iters, k = 3, 8
    if opt == "CEM":
        ref = O.CEM(ev, lo, hi, horizon=H, max_iterations=iters, population=N, num_elite=k, num_agents=A)
    elif opt == "PI2":
        ref = O.PI2(ev, lo, hi, horizon=H, max_iterations=iters, population=N, num_agents=A)
    else:
        ref = O.RandomSearch(ev, lo, hi, horizon=H, population=N, num_agents=A)
    tor = T.make(opt, env, lo, hi, N, A, H, iters, k, mlp=mlp, stats=stats)
    state = O.pendulum_start_states(A) if env == "pendulum" else O.cheetah_start_states(A, S)
    for step in range(2):                                   # two control steps: PI2's warm start is exercised
        if opt == "RandomSearch":
            noise = {"uniform": rng.random((N, A, H, U)).astype(F)}
        else:
            noise = {"trunc": [O.truncated_normal_noise(rng, (N, A, H, U)) for _ in range(iters)]}
        a0, n0, r0 = ref.call(state, noise)
        a1, n1, r1 = tor.call(state, noise)
        np.testing.assert_allclose(a1.numpy(), a0, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(n1.numpy(), n0, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(r1.numpy(), r0, rtol=2e-4, atol=2e-3)
        state = n0


def test_own_draws_are_truncated_and_seeded():
    import torch
    g = torch.Generator().manual_seed(3)
    z = T.truncated_normal((4000,), g)
    assert float(z.abs().max()) < 2.0 and abs(float(z.mean())) < 0.1 and 0.8 < float(z.std()) < 0.95
    tor = T.make("CEM", "pendulum", [-2.0], [2.0], 64, 1, 6, 2, 8, seed=1)
    tor2 = T.make("CEM", "pendulum", [-2.0], [2.0], 64, 1, 6, 2, 8, seed=1)
    s = O.pendulum_start_states(1)
    assert np.array_equal(tor.call(s)[0].numpy(), tor2.call(s)[0].numpy())
