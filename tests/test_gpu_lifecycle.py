"""Handle lifecycle: every device / pinned allocation of a handle goes back when it is destroyed, whatever it ran."""
import gc

import numpy as np
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1
    return _lib


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def _cycle(L, i):
    from blackbox_mpc_amd.engine import Engine
    kind = i % 5
    if kind == 0:       # persistent pendulum kernel + noise prefetch buffers
        eng = Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=2, planning_horizon=30,
                     population_size=500, max_iterations=5, num_elite=50, seed=i)
        s = O.pendulum_start_states(2)
        for t in range(3):
            _, s, _ = eng.optimize(s, t)
    elif kind == 1:     # CMA-ES work matrices
        eng = Engine(L.OPT_CMAES, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=40,
                     population_size=200, max_iterations=2, num_elite=20, seed=i)
        eng.optimize(O.pendulum_start_states(1), 0)
    elif kind == 2:     # learned dynamics: packed operands, trace buffers
        dims = [26, 200, 200, 20]
        ws, bs = O.make_mlp_params(dims, seed=i)
        eng = Engine(L.OPT_PI2, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * 6, [1.0] * 6, dim_s=20, num_agents=2, planning_horizon=20,
                     population_size=2000, max_iterations=2, seed=i)
        eng.set_mlp(ws, bs, [1, 1, 0], None)
        eng.set_trace(True)
        eng.optimize(O.cheetah_start_states(2, 20), 0)
    elif kind == 3:     # PSO swarm state, large population evaluator call
        eng = Engine(L.OPT_PSO, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=3, planning_horizon=25,
                     population_size=4000, max_iterations=2, seed=i)
        eng.reset()
        eng.optimize(O.pendulum_start_states(3), 0)
        seq = np.random.default_rng(i).uniform(-2, 2, (4000, 3, 25, 1)).astype(F)
        eng.evaluate(O.pendulum_start_states(3), seq)
    else:               # run-time compiled user functions (hiprtc modules)
        eng = Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_USER, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=10,
                     population_size=128, max_iterations=2, num_elite=16, seed=i)
        eng.set_reward_source("__device__ float bbmpc_user_reward(const float* c, const float* a, const float* n, int S, int U)"
                              " { return -(n[0] - 1.0f) * (n[0] - 1.0f) - 0.1f * a[0] * a[0]; }")
        eng.optimize(O.pendulum_start_states(1), 0)
    eng.close()


def test_handles_give_their_memory_back(L):
    for i in range(5):                       # first touch: code objects, hiprtc, allocator pools of the runtime itself
        _cycle(L, i)
    gc.collect()
    base = _free_bytes()
    for i in range(40):
        _cycle(L, i)
    gc.collect()
    after = _free_bytes()
    # 40 handles held between 1 MB and ~200 MB each while alive; a leak of any of their buffers would show as tens of MB
    assert base - after < 32 * 1024 * 1024, "device memory not returned: %.1f MB" % ((base - after) / 2 ** 20)


def test_close_is_idempotent_and_use_after_close_is_an_error(L):
    from blackbox_mpc_amd.engine import Engine
    eng = Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=5,
                 population_size=64, max_iterations=1, num_elite=8)
    eng.close()
    eng.close()
    with pytest.raises(Exception):
        eng.optimize(O.pendulum_start_states(1), 0)


def test_two_threads_two_handles_match_the_sequential_runs(L):
    # a handle is not re-entrant, but different handles may be driven from different threads (thread-per-GPU drivers):
    # ctypes releases the GIL during the calls, so the two control loops really overlap in the runtime
    import threading
    from blackbox_mpc_amd.engine import Engine

    def make(kind, seed):
        if kind == 0:
            return Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=2, planning_horizon=30,
                          population_size=500, max_iterations=5, num_elite=50, seed=seed)
        dims = [4, 32, 32, 32, 3]
        ws, bs = O.make_mlp_params(dims, seed=7)
        e = Engine(L.OPT_PI2, L.DYN_MLP, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=2, planning_horizon=15,
                   population_size=300, max_iterations=3, seed=seed)
        e.set_mlp(ws, bs, [1, 1, 1, 0], None)
        return e

    def loop(eng, steps, out):
        s = O.pendulum_start_states(2)
        acts = []
        for t in range(steps):
            a, s, _ = eng.optimize(s, t)
            acts.append(a.copy())
        out.append(np.stack(acts))

    steps = 60
    seq = []
    for kind in (0, 1):
        loop(make(kind, 3 + kind), steps, seq)
    par = [[], []]
    engines = [make(kind, 3 + kind) for kind in (0, 1)]
    threads = [threading.Thread(target=loop, args=(engines[k], steps, par[k])) for k in (0, 1)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
        assert not th.is_alive()
    np.testing.assert_array_equal(par[0][0], seq[0])
    np.testing.assert_array_equal(par[1][0], seq[1])
