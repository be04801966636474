"""Synthetic inputs of the benchmark configurations (SURVEY.md 8d / BASELINE.md): there is no gym on the GPU box,
so start states, the learned model's weights and its normalisation statistics are generated here, seeded.

    Pendulum   : theta0 ~ U(-pi, pi), thetadot0 ~ U(-1, 1) from default_rng(1234 + agent) -> (cos, sin, thetadot)
                 (gym's Pendulum-v0 reset distribution)
    HalfCheetah: S = 20 (tutorials/mujoco/env_modified.py:22-27), U = 6, bounds +-1; MLP 26-200-200-20 tanh, tanh,
                 linear; kernels Glorot-uniform / zero bias (Keras Dense defaults, deterministic_mlp.py:21-24) from
                 default_rng(42) with the last layer scaled x0.1 so that 50-step rollouts stay finite; statistics
                 mu = 0, sigma = 1 for states / actions, targets mu = 0, sigma = 0.1; start state ~ N(0, 0.1^2) from
                 default_rng(7 + agent)

bench.py, the examples and the tests' engine side use these; the oracle keeps its own statement of the same recipe
(tests/test_synthetic.py holds the two together)."""
import numpy as np

F = np.float32
CHEETAH_DIMS = [26, 200, 200, 20]
CHEETAH_ACTIVATIONS = ["tanh", "tanh", None]


def pendulum_start_states(num_agents, agent_offset=0):
    out = np.zeros((num_agents, 3), F)
    for a in range(num_agents):
        rng = np.random.default_rng(1234 + agent_offset + a)
        th = rng.uniform(-np.pi, np.pi)
        thd = rng.uniform(-1.0, 1.0)
        out[a] = [np.cos(th), np.sin(th), thd]
    return out


def cheetah_start_states(num_agents, dim_s=20, agent_offset=0):
    out = np.zeros((num_agents, dim_s), F)
    for a in range(num_agents):
        rng = np.random.default_rng(7 + agent_offset + a)
        out[a] = (rng.standard_normal(dim_s) * 0.1).astype(F)
    return out


def make_mlp_params(dims=None, seed=42, last_scale=0.1):
    dims = CHEETAH_DIMS if dims is None else dims
    rng = np.random.default_rng(seed)
    ws, bs = [], []
    for i in range(len(dims) - 1):
        lim = np.sqrt(6.0 / (dims[i] + dims[i + 1]))
        w = rng.uniform(-lim, lim, size=(dims[i], dims[i + 1])).astype(F)
        if i == len(dims) - 2:
            w = (w * F(last_scale)).astype(F)
        ws.append(w)
        bs.append(np.zeros((dims[i + 1],), F))
    return ws, bs


def cheetah_stats(dim_s=20, dim_u=6):
    """(mean_states, std_states, mean_actions, std_actions, mean_targets, std_targets)"""
    z, o = np.zeros, np.ones
    return [z(dim_s, F), o(dim_s, F), z(dim_u, F), o(dim_u, F), z(dim_s, F), np.full(dim_s, 0.1, F)]
