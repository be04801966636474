"""f-4: population sharding for num_agents < n_gpus, PI2 and CEM.  PI2: the min / sum reductions of
pi2.py:80-87 split across ranks: per iteration every rank rolls out ITS particles of the shared population, produces
(min cost, sum of weights, weighted sums [H*U]) per agent, one collective hands every rank all partials and each merges
them in rank order.  RNG is keyed by the GLOBAL particle index, so a sharded run draws exactly the unsharded run's
samples and differs only in the order of the fp32 sums.

A GPU box here has one device, so the multi-rank arithmetic is exercised through the engine's loopback hook
(BBMPC_POPSHARD_LOOPBACK=G: one handle plays the G shards in turn, same kernels, same merge) and the collective itself
through a one-rank RCCL communicator (BBMPC_POPSHARD_FORCE).  Tolerance (its own, as SURVEY f-4 asks): mean / action
within 2e-5 of the unsharded engine, predicted next state within 2e-5 (pendulum) / 2e-4 (MLP)."""
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


def _mlp_engine(L, N, **kw):
    from blackbox_mpc_amd.engine import Engine
    S, U = 20, 6
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=42)
    stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]
    eng = Engine(L.OPT_PI2, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=1, planning_horizon=30,
                 population_size=N, max_iterations=5, lamda=1.0, seed=9, **kw)
    eng.set_mlp(ws, bs, [1, 1, 0], stats)
    return eng


@pytest.mark.parametrize("G", [2, 4, 8])
def test_sharded_population_equals_unsharded_mlp_pi2(L, monkeypatch, G):
    # north-star shape: HalfCheetah MLP, PI2, N = 1000 (G = 8: 125 particles per shard), H = 30, 5 iterations
    N = 1000
    full = _mlp_engine(L, N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(G))
    shard = _mlp_engine(L, N // G, population_offset=0, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    s_f = s_s = O.cheetah_start_states(1, 20)
    for t in range(3):                                   # warm start (shift-left) carried across control steps
        a_f, n_f, r_f = full.optimize(s_f, t)
        a_s, n_s, r_s = shard.optimize(s_s, t)
        np.testing.assert_allclose(a_s, a_f, rtol=0, atol=2e-5)
        np.testing.assert_allclose(n_s, n_f, rtol=0, atol=2e-4)
        np.testing.assert_allclose(shard.get_state("prev_mean"), full.get_state("prev_mean"), rtol=0, atol=2e-5)
        s_f, s_s = n_f, n_s


@pytest.mark.parametrize("G", [2, 8])
def test_sharded_population_equals_unsharded_mlp_cem(L, monkeypatch, G):
    # CEM (cem.py:97-125): local top-k + rows, one all-gather, global top-k (ties -> lower GLOBAL index) + refit.
    # Config-4 shape: N = 1000, k = 50, H = 30; at G = 8 a shard holds 125 particles.
    from blackbox_mpc_amd.engine import Engine
    N, k, S, U = 1000, 50, 20, 6
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=42)
    stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]

    def mk(n, **kw):
        e = Engine(L.OPT_CEM, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=1, planning_horizon=30,
                   population_size=n, max_iterations=5, num_elite=k, alpha=0.25, seed=4, **kw)
        e.set_mlp(ws, bs, [1, 1, 0], stats)
        return e
    full = mk(N)
    full.set_trace(True)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(G))
    shard = mk(N // G, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    shard.set_trace(True)
    s = O.cheetah_start_states(1, S)
    for t in range(2):
        a_f, n_f, _ = full.optimize(s, t)
        a_s, n_s, _ = shard.optimize(s, t)
        for it in range(5):                              # the global elite SET is the unsharded one (global indices)
            assert set(shard.get_trace(it, L.TRACE_ELITES)[0].tolist()) == set(full.get_trace(it, L.TRACE_ELITES)[0].tolist())
        np.testing.assert_allclose(shard.get_state("mean"), full.get_state("mean"), rtol=0, atol=2e-5)
        np.testing.assert_allclose(shard.get_state("var"), full.get_state("var"), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(a_s, a_f, rtol=0, atol=2e-5)
        np.testing.assert_allclose(n_s, n_f, rtol=0, atol=2e-4)
        s = n_f


def test_sharded_population_equals_unsharded_pendulum_pi2(L, monkeypatch):
    from blackbox_mpc_amd.engine import Engine
    monkeypatch.setenv("BBMPC_FUSED", "0")
    N, A, H, G = 768, 3, 20, 3
    mk = lambda n, **kw: Engine(L.OPT_PI2, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A,
                                planning_horizon=H, population_size=n, max_iterations=4, lamda=1.0, seed=5, **kw)
    full = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(G))
    shard = mk(N // G, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    s = O.pendulum_start_states(A)
    for t in range(3):
        a_f, n_f, _ = full.optimize(s, t)
        a_s, n_s, _ = shard.optimize(s, t)
        np.testing.assert_allclose(a_s, a_f, rtol=0, atol=2e-5)
        np.testing.assert_allclose(n_s, n_f, rtol=0, atol=2e-5)
        s = n_f


def test_exchange_through_a_one_rank_rccl_communicator(L, monkeypatch):
    # the real code path of a sharded rank -- partials, ncclAllGather on the launch stream, merge -- with one rank
    from blackbox_mpc_amd.engine import Engine
    N = 1000
    full = _mlp_engine(L, N)
    monkeypatch.setenv("BBMPC_POPSHARD_FORCE", "1")
    one = _mlp_engine(L, N, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_FORCE")
    one.comm_init(Engine.comm_unique_id(), 1, 0)
    assert one.comm_info()[0] == 1
    s = O.cheetah_start_states(1, 20)
    for t in range(2):
        a_f, n_f, _ = full.optimize(s, t)
        a_o, n_o, _ = one.optimize(s, t)
        np.testing.assert_allclose(a_o, a_f, rtol=0, atol=2e-5)
        np.testing.assert_allclose(n_o, n_f, rtol=0, atol=2e-4)
        s = n_f
    one.synchronize()
    one.comm_destroy()


def test_population_sharding_argument_checks(L):
    from blackbox_mpc_amd.engine import Engine
    kw = dict(dim_s=3, num_agents=1, planning_horizon=8, max_iterations=2)
    with pytest.raises(L.BBMPCError) as ei:              # PSO / SPSA / CMA-ES carry per-particle state: not built, said so
        Engine(L.OPT_PSO, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], population_size=64, population_global=128, **kw)
    assert ei.value.code == L.E_UNSUPPORTED
    with pytest.raises(L.BBMPCError):                    # the shard must lie inside the population
        Engine(L.OPT_PI2, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], population_size=64, population_offset=100,
               population_global=128, **kw)
    eng = Engine(L.OPT_PI2, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], population_size=64, population_offset=64,
                 population_global=128, **kw)
    with pytest.raises(L.BBMPCError) as ei:              # sharded, but nobody to exchange with
        eng.optimize(O.pendulum_start_states(1))
    assert ei.value.code == L.E_STATE
