"""f-4: population sharding for num_agents < n_gpus, every optimizer: RandomSearch, PI2, CEM, SPSA, PSO, CMA-ES.  PI2: the min / sum reductions of
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

from tests.parity_util import assert_cheetah_rewards

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


@pytest.mark.parametrize("G", [2, 5])
def test_sharded_spsa_against_the_oracle_and_the_unsharded_engine(L, monkeypatch, G):
    # SPSA (spsa.py:61-117): the gradient estimate is a mean over the perturbation pairs, so every shard sums its own pairs,
    # one all-gather, every rank adds the G row-sum vectors in rank order and divides by the GLOBAL population.  The
    # Rademacher draws are keyed by the global particle index: the unsharded engine's dumped draws feed the NumPy oracle.
    from blackbox_mpc_amd.engine import Engine
    monkeypatch.setenv("BBMPC_FUSED", "0")
    N, A, H, iters = 640, 2, 12, 4
    mk = lambda n, **kw: Engine(L.OPT_SPSA, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A,
                                planning_horizon=H, population_size=n, max_iterations=iters, seed=13, **kw)
    full = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(G))
    shard = mk(N // G, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    ev = O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))
    sp = O.SPSA(ev, [-2.0], [2.0], horizon=H, max_iterations=iters, population=N, num_agents=A)
    s = O.pendulum_start_states(A)
    for t in range(3):                                   # later control steps start from the shifted solution (spsa.py:114-115)
        delta = [full.dump_noise(L.NOISE_RADEMACHER, t, it, (N, A, H, 1)) for it in range(iters)]
        a_f, n_f, _ = full.optimize(s, t)
        a_s, n_s, _ = shard.optimize(s, t)
        a_o, n_o, _ = sp.call(s, {"rademacher": delta})
        np.testing.assert_allclose(a_s, a_f, rtol=0, atol=2e-5)          # sharded vs unsharded engine: order of the fp32 sums only
        np.testing.assert_allclose(shard.get_state("prev_mean"), full.get_state("prev_mean"), rtol=0, atol=2e-5)
        np.testing.assert_allclose(a_s, a_o, rtol=0, atol=1e-4)          # sharded engine vs the oracle (tolerances of test_gpu_spsa_pso.py)
        np.testing.assert_allclose(shard.get_state("prev_mean"), sp.params, rtol=0, atol=1e-4)
        np.testing.assert_allclose(n_s, n_o, rtol=1e-4, atol=1e-4)
        s = n_f


def test_sharded_spsa_mlp_equals_unsharded(L, monkeypatch):
    # learned-dynamics SPSA, population sharded 4 ways, over three control steps (warm start through k_tail_mlp)
    from blackbox_mpc_amd.engine import Engine
    S, U, N, H, iters, G = 20, 6, 512, 10, 3, 4
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=42)
    stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]

    def mk(n, **kw):
        e = Engine(L.OPT_SPSA, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=1, planning_horizon=H,
                   population_size=n, max_iterations=iters, seed=3, **kw)
        e.set_mlp(ws, bs, [1, 1, 0], stats)
        return e
    full = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(G))
    shard = mk(N // G, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    s = O.cheetah_start_states(1, S)
    for t in range(3):
        a_f, n_f, _ = full.optimize(s, t)
        a_s, n_s, _ = shard.optimize(s, t)
        np.testing.assert_allclose(a_s, a_f, rtol=0, atol=2e-5)
        np.testing.assert_allclose(n_s, n_f, rtol=0, atol=2e-4)
        np.testing.assert_allclose(shard.get_state("prev_mean"), full.get_state("prev_mean"), rtol=0, atol=2e-5)
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


@pytest.mark.parametrize("G", [2, 4])
def test_sharded_pso_is_the_unsharded_swarm_bit_for_bit(L, monkeypatch, G):
    # PSO (pso.py:70-141): the particles live across iterations; their only cross-particle operation is the argmax of the
    # personal bests (pso.py:94).  Sharded: local best -> all-gather -> first maximum by GLOBAL particle index, draws keyed
    # by the global particle -- so a sharded swarm is the unsharded one bit for bit, over control steps, re-seeds and a reset.
    from blackbox_mpc_amd.engine import Engine
    monkeypatch.setenv("BBMPC_FUSED", "0")
    N, A, H, iters = 256, 2, 11, 4
    mk = lambda n, **kw: Engine(L.OPT_PSO, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A,
                                planning_horizon=H, population_size=n, max_iterations=iters, seed=17, **kw)
    full = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(G))
    shard = mk(N // G, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    full.set_trace(True)
    shard.set_trace(True)
    s = O.pendulum_start_states(A)
    n = N // G
    for t in range(5):
        if t == 3:
            full.reset()
            shard.reset()
        a_f, n_f, r_f = full.optimize(s, t)
        a_s, n_s, r_s = shard.optimize(s, t)
        np.testing.assert_array_equal(a_s, a_f)
        np.testing.assert_array_equal(n_s, n_f)
        np.testing.assert_array_equal(r_s, r_f)
        for it in range(iters):                              # global best position and its GLOBAL particle index per iteration
            np.testing.assert_array_equal(shard.get_trace(it, L.TRACE_MEAN), full.get_trace(it, L.TRACE_MEAN))
            np.testing.assert_array_equal(shard.get_trace(it, L.TRACE_ELITES), full.get_trace(it, L.TRACE_ELITES))
        # shard 0's particles are the first N / G of the swarm (re-seeded around the same global best)
        np.testing.assert_array_equal(shard.get_state("pos", (n, A, H, 1)), full.get_state("pos", (N, A, H, 1))[:n])
        np.testing.assert_array_equal(shard.get_state("vel", (n, A, H, 1)), full.get_state("vel", (N, A, H, 1))[:n])
        s = n_f


def test_sharded_pso_mlp_equals_unsharded(L, monkeypatch):
    from blackbox_mpc_amd.engine import Engine
    S, U, N, H, iters, G = 20, 6, 384, 8, 3, 3
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=42)
    stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]

    def mk(n, **kw):
        e = Engine(L.OPT_PSO, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=2, planning_horizon=H,
                   population_size=n, max_iterations=iters, seed=8, **kw)
        e.set_mlp(ws, bs, [1, 1, 0], stats)
        return e
    full = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(G))
    shard = mk(N // G, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    s = O.cheetah_start_states(2, S)
    for t in range(3):
        a_f, n_f, _ = full.optimize(s, t)
        a_s, n_s, _ = shard.optimize(s, t)
        # a shard of 128 particles and the whole swarm of 384 are rolled out by different MFMA tilings (different order of the
        # fp32 sums inside a model step): the swarm's best is compared within the MLP tolerance, not bit for bit
        np.testing.assert_allclose(a_s, a_f, rtol=0, atol=1e-4)
        np.testing.assert_allclose(n_s, n_f, rtol=2e-5, atol=2e-4)
        s = n_f


@pytest.mark.parametrize("G,force", [(4, False), (1, True)])
def test_sharded_random_search_is_the_unsharded_one_bit_for_bit(L, monkeypatch, G, force):
    # RandomSearch (random_search.py:36-47): uniform draws keyed by the global particle, first maximum by GLOBAL index --
    # through the loopback hook (G shards) and through a one-rank RCCL communicator (the sharded rank's real code path)
    from blackbox_mpc_amd.engine import Engine
    monkeypatch.setenv("BBMPC_FUSED", "0")
    N, A, H = 512, 3, 15
    mk = lambda n, **kw: Engine(L.OPT_RANDOM_SEARCH, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A,
                                planning_horizon=H, population_size=n, seed=23, **kw)
    full = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_FORCE" if force else "BBMPC_POPSHARD_LOOPBACK", "1" if force else str(G))
    shard = mk(N // G, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_FORCE" if force else "BBMPC_POPSHARD_LOOPBACK")
    if force:
        shard.comm_init(Engine.comm_unique_id(), 1, 0)
    full.set_trace(True)
    shard.set_trace(True)
    s = O.pendulum_start_states(A)
    for t in range(3):
        a_f, n_f, r_f = full.optimize(s, t)
        a_s, n_s, r_s = shard.optimize(s, t)
        np.testing.assert_array_equal(a_s, a_f)
        np.testing.assert_array_equal(n_s, n_f)
        np.testing.assert_array_equal(shard.get_trace(0, L.TRACE_ELITES), full.get_trace(0, L.TRACE_ELITES))   # global index of the winner
        s = n_f
    if force:
        shard.synchronize()
        shard.comm_destroy()


@pytest.mark.parametrize("per_agent", [False, True])
@pytest.mark.parametrize("G,force", [(4, False), (1, True)])
def test_sharded_cmaes_is_the_unsharded_one_bit_for_bit(L, monkeypatch, G, force, per_agent):
    # CMA-ES (cma_es.py:129-213): every rank samples (draws keyed by the global particle) and rolls out ITS candidates, the
    # sorted local elites (summed reward, global index, candidate) are exchanged and merged into the swarm's k elites; mean,
    # paths, covariance and eigen-decomposition run replicated.  The merged elite list IS the unsharded one (same values, same
    # order), so everything downstream is bit-identical: coupled (the reference: one joint covariance over the agents) and
    # per-agent forms, through the loopback hook and through a one-rank RCCL communicator.
    from blackbox_mpc_amd.engine import Engine
    N, A, H, iters, k = 256, 2, 6, 3, 24
    quirks = L.CMAES_PER_AGENT if per_agent else 0
    mk = lambda n, **kw: Engine(L.OPT_CMAES, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A,
                                planning_horizon=H, population_size=n, max_iterations=iters, num_elite=k, seed=31, quirks=quirks, **kw)
    full = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_FORCE" if force else "BBMPC_POPSHARD_LOOPBACK", "1" if force else str(G))
    shard = mk(N // G, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_FORCE" if force else "BBMPC_POPSHARD_LOOPBACK")
    if force:
        shard.comm_init(Engine.comm_unique_id(), 1, 0)
    full.set_trace(True)
    shard.set_trace(True)
    s = O.pendulum_start_states(A)
    for t in range(3):
        a_f, n_f, r_f = full.optimize(s, t)
        a_s, n_s, r_s = shard.optimize(s, t)
        np.testing.assert_array_equal(a_s, a_f)
        np.testing.assert_array_equal(n_s, n_f)
        for it in range(iters):
            np.testing.assert_array_equal(shard.get_trace(it, L.TRACE_MEAN), full.get_trace(it, L.TRACE_MEAN))
            np.testing.assert_array_equal(shard.get_trace(it, L.TRACE_ELITES), full.get_trace(it, L.TRACE_ELITES))   # global indices, sorted
            np.testing.assert_array_equal(shard.get_trace(it, L.TRACE_CMA_C), full.get_trace(it, L.TRACE_CMA_C))
        s = n_f
    if force:
        shard.synchronize()
        shard.comm_destroy()


def test_sharded_cmaes_mlp_equals_unsharded(L, monkeypatch):
    from blackbox_mpc_amd.engine import Engine
    S, U, N, H, iters, k, G = 20, 6, 384, 5, 3, 32, 3
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=42)
    stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]

    def mk(n, **kw):
        e = Engine(L.OPT_CMAES, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=2, planning_horizon=H,
                   population_size=n, max_iterations=iters, num_elite=k, seed=12, quirks=L.CMAES_PER_AGENT, **kw)
        e.set_mlp(ws, bs, [1, 1, 0], stats)
        return e
    full = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(G))
    shard = mk(N // G, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    s = O.cheetah_start_states(2, S)
    a_f, n_f, _ = full.optimize(s, 0)
    a_s, n_s, _ = shard.optimize(s, 0)
    # a shard and the whole population are rolled out by different MFMA tilings: rewards differ in the last bits, so near-ties
    # at the k-th place may swap elites of (almost) equal weight -- the refitted mean is compared within the MLP tolerance
    np.testing.assert_allclose(a_s, a_f, rtol=0, atol=2e-3)
    np.testing.assert_allclose(n_s, n_f, rtol=1e-3, atol=2e-3)


def test_pso_exchange_through_a_one_rank_rccl_communicator(L, monkeypatch):
    from blackbox_mpc_amd.engine import Engine
    monkeypatch.setenv("BBMPC_FUSED", "0")
    N, A, H, iters = 192, 2, 7, 3
    mk = lambda **kw: Engine(L.OPT_PSO, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A,
                             planning_horizon=H, population_size=N, max_iterations=iters, seed=6, **kw)
    full = mk()
    monkeypatch.setenv("BBMPC_POPSHARD_FORCE", "1")
    one = mk(population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_FORCE")
    one.comm_init(Engine.comm_unique_id(), 1, 0)
    s = O.pendulum_start_states(A)
    for t in range(3):
        a_f, n_f, _ = full.optimize(s, t)
        a_o, n_o, _ = one.optimize(s, t)
        np.testing.assert_array_equal(a_o, a_f)
        np.testing.assert_array_equal(n_o, n_f)
        s = n_f
    one.synchronize()
    one.comm_destroy()


def test_spsa_exchange_through_a_one_rank_rccl_communicator(L, monkeypatch):
    # SPSA's sharded code path as a rank runs it -- row sums, ncclAllGather on the launch stream, merge -- with one rank
    from blackbox_mpc_amd.engine import Engine
    monkeypatch.setenv("BBMPC_FUSED", "0")
    N, A, H, iters = 256, 2, 9, 3
    mk = lambda **kw: Engine(L.OPT_SPSA, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A,
                             planning_horizon=H, population_size=N, max_iterations=iters, seed=2, **kw)
    full = mk()
    monkeypatch.setenv("BBMPC_POPSHARD_FORCE", "1")
    one = mk(population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_FORCE")
    one.comm_init(Engine.comm_unique_id(), 1, 0)
    s = O.pendulum_start_states(A)
    for t in range(2):
        a_f, n_f, _ = full.optimize(s, t)
        a_o, n_o, _ = one.optimize(s, t)
        np.testing.assert_allclose(a_o, a_f, rtol=0, atol=2e-5)
        np.testing.assert_allclose(n_o, n_f, rtol=0, atol=2e-5)
        s = n_f
    one.synchronize()
    one.comm_destroy()


def test_population_above_32768_is_played_as_shards_on_one_gpu(L):
    # The refit kernels keep an agent's rewards in one CU's LDS (32768 floats); a larger population runs as equal shards of
    # the same machinery on the one GPU (draws keyed by the global particle).  PI2 at N = 40000 and CEM at N = 65536 (two
    # shards of exactly 32768) against the NumPy oracle fed with the engine's own draws.
    from blackbox_mpc_amd.engine import Engine
    A, H, iters = 2, 6, 3
    ev = O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))
    s = O.pendulum_start_states(A)
    N = 40000
    eng = Engine(L.OPT_PI2, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A, planning_horizon=H,
                 population_size=N, max_iterations=iters, lamda=1.0, seed=3)
    pi2 = O.PI2(ev, [-2.0], [2.0], horizon=H, max_iterations=iters, population=N, num_agents=A, lamda=1.0)
    for t in range(2):
        noise = {"trunc": [eng.dump_noise(L.NOISE_TRUNC_NORMAL, t, it, (N, A, H, 1)) for it in range(iters)]}
        a_e, n_e, _ = eng.optimize(s, t)
        a_o = pi2._optimize(s, noise)
        np.testing.assert_allclose(a_e, a_o, rtol=0, atol=1e-4)
        s = n_e
    with pytest.raises(L.BBMPCError) as ei:              # the parity hooks are per shard
        eng.set_trace(True)
    assert ei.value.code == L.E_UNSUPPORTED
    N, k = 65536, 100
    eng = Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A, planning_horizon=H,
                 population_size=N, max_iterations=iters, num_elite=k, alpha=0.2, seed=4)
    cem = O.CEM(ev, [-2.0], [2.0], horizon=H, max_iterations=iters, population=N, num_elite=k, num_agents=A, alpha=0.2)
    s = O.pendulum_start_states(A)
    noise = {"trunc": [eng.dump_noise(L.NOISE_TRUNC_NORMAL, 0, it, (N, A, H, 1)) for it in range(iters)]}
    a_e, n_e, _ = eng.optimize(s, 0)
    a_o = cem._optimize(s, noise)
    np.testing.assert_allclose(a_e, a_o, rtol=0, atol=2e-4)
    with pytest.raises(L.BBMPCError):                    # 32771 is prime: no equal shards
        Engine(L.OPT_PI2, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=H,
               population_size=32771, max_iterations=1)


@pytest.mark.parametrize("opt_name", ["RandomSearch", "PSO", "SPSA", "CMA-ES"])
def test_population_above_32768_every_optimizer_runs(L, opt_name):
    # the remaining optimizers on the shard-played path at N = 40000: control steps (with the warm starts / re-seeds in
    # between) give finite, feasible actions and the same actions again from a second, identically seeded handle
    from blackbox_mpc_amd.engine import Engine
    opt = {"RandomSearch": L.OPT_RANDOM_SEARCH, "PSO": L.OPT_PSO, "SPSA": L.OPT_SPSA, "CMA-ES": L.OPT_CMAES}[opt_name]
    kw = dict(num_elite=64, quirks=L.CMAES_PER_AGENT) if opt_name == "CMA-ES" else {}
    mk = lambda: Engine(opt, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=2, planning_horizon=5,
                        population_size=40000, max_iterations=2, seed=77, **kw)
    e1, e2 = mk(), mk()
    s1 = s2 = O.pendulum_start_states(2)
    for t in range(3):
        a1, s1, r1 = e1.optimize(s1, t)
        a2, s2, r2 = e2.optimize(s2, t)
        assert np.all(np.isfinite(a1)) and np.all(np.abs(a1) <= 2.0) and np.all(np.isfinite(r1))
        np.testing.assert_array_equal(a1, a2)
        np.testing.assert_array_equal(s1, s2)


def test_population_sharding_argument_checks(L):
    from blackbox_mpc_amd.engine import Engine
    kw = dict(dim_s=3, num_agents=1, planning_horizon=8, max_iterations=2)
    with pytest.raises(L.BBMPCError) as ei:              # sharded CMA-ES: the elites must fit this rank's share
        Engine(L.OPT_CMAES, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], population_size=16, population_global=128, num_elite=32, **kw)
    assert ei.value.code == L.E_UNSUPPORTED
    with pytest.raises(L.BBMPCError) as ei:              # an evaluate-only handle has no population of its own to shard
        Engine(L.OPT_NONE, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], population_size=64, population_global=128, **kw)
    assert ei.value.code == L.E_UNSUPPORTED
    with pytest.raises(L.BBMPCError):                    # the shard must lie inside the population
        Engine(L.OPT_PI2, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], population_size=64, population_offset=100,
               population_global=128, **kw)
    eng = Engine(L.OPT_PI2, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], population_size=64, population_offset=64,
                 population_global=128, **kw)
    with pytest.raises(L.BBMPCError) as ei:              # sharded, but nobody to exchange with
        eng.optimize(O.pendulum_start_states(1))
    assert ei.value.code == L.E_STATE


# ---- the sharded refits held to the ORACLE (pi2.py:80-87, cem.py:97-112), not only to the unsharded engine ------------
# RNG is keyed by the global particle index, so an unsharded engine with the same seed draws the sharded run's samples:
# its dumped draws feed the NumPy optimizer (C library doing the 1000 x 30-step rollouts), its per-iteration rewards
# are checked against the oracle's within the MLP tolerance and carried over (lock-step, as in
# tests/test_gpu_fullsize.py::test_northstar_mlp_pi2_lockstep_full_size), and the SHARDED engine's refit -- partials,
# exchange, merge in rank order / global-index top-k -- is compared with the oracle's refit of the same inputs.
MLP_DIMS, MLP_ACTS = [26, 200, 200, 20], ["tanh", "tanh", None]


def _oracle_rollouts(N, k=1, iters=5):
    from oracle import oracle_c as OC
    S, U = 20, 6
    ws, bs = O.make_mlp_params(MLP_DIMS, seed=42)
    stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]
    return OC.COracle("mlp", "cheetah", [-1.0] * U, [1.0] * U, N, 1, 30, S, iters=iters, k=max(k, 1),
                      mlp=(ws, bs, MLP_ACTS), stats=stats)


@pytest.mark.parametrize("G", [2, 8])
def test_sharded_pi2_against_the_oracle_northstar_shape(L, monkeypatch, G):
    N, A, H, U, iters = 1000, 1, 30, 6, 5
    lo, hi = [-1.0] * U, [1.0] * U
    full = _mlp_engine(L, N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(G))
    shard = _mlp_engine(L, N // G, population_offset=0, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    full.set_trace(True)
    shard.set_trace(True)
    co = _oracle_rollouts(N)
    pi2 = O.PI2(co.as_evaluator(), lo, hi, horizon=H, max_iterations=iters, population=N, num_agents=A, lamda=1.0)
    RT, AT = 1e-3, 1e-3 * H
    s = O.cheetah_start_states(1, 20)
    for t in range(2):                                   # second control step: shift-left warm start (pi2.py:92-93)
        noise = {"trunc": [full.dump_noise(L.NOISE_TRUNC_NORMAL, t, it, (N, A, H, U)) for it in range(iters)]}
        a_f, n_f, _ = full.optimize(s, t)
        a_s, n_s, _ = shard.optimize(s, t)
        hip_r = [full.get_trace(it, L.TRACE_REWARDS) for it in range(iters)]
        # the rollouts are the same per particle: the shard last played (particles [N - N/G, N)) shows the unsharded bits
        # in iteration 0 of the first control step; afterwards the two runs' means differ in the last bits (order of the
        # fp32 sums), so do their samples and rewards -- by 1e-7 relative, far inside what is carried over below
        if t == 0:
            np.testing.assert_array_equal(shard.get_trace(0, L.TRACE_REWARDS), hip_r[0][N - N // G:])
        for it in range(iters):
            np.testing.assert_allclose(shard.get_trace(it, L.TRACE_REWARDS), hip_r[it][N - N // G:], rtol=2e-6, atol=2e-3)

        def lock(it, r_o):
            assert_cheetah_rewards(hip_r[it], r_o, RT, AT)
            return hip_r[it]
        act_o = pi2._optimize(s, noise, rewards_override=lock)
        for it in range(iters):
            np.testing.assert_allclose(shard.get_trace(it, L.TRACE_MEAN), pi2.trace[it]["mean"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(a_s, act_o, rtol=0, atol=2e-5)
        np.testing.assert_allclose(shard.get_state("prev_mean"), pi2.prev, rtol=0, atol=2e-5)
        np.testing.assert_allclose(n_s, co.predict_next_state(s, act_o), rtol=2e-5, atol=2e-4)
        s = n_f


@pytest.mark.parametrize("G", [2, 8])
def test_sharded_cem_against_the_oracle_config4_shape(L, monkeypatch, G):
    from blackbox_mpc_amd.engine import Engine
    N, A, H, U, iters, k, S = 1000, 1, 30, 6, 5, 50, 20
    lo, hi = [-1.0] * U, [1.0] * U
    ws, bs = O.make_mlp_params(MLP_DIMS, seed=42)
    stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]

    def mk(n, **kw):
        e = Engine(L.OPT_CEM, L.DYN_MLP, L.REW_CHEETAH, lo, hi, dim_s=S, num_agents=1, planning_horizon=H,
                   population_size=n, max_iterations=iters, num_elite=k, alpha=0.25, seed=4, **kw)
        e.set_mlp(ws, bs, [1, 1, 0], stats)
        e.set_trace(True)
        return e
    full = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(G))
    shard = mk(N // G, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    co = _oracle_rollouts(N, k=k)
    cem = O.CEM(co.as_evaluator(), lo, hi, horizon=H, max_iterations=iters, population=N, num_elite=k, num_agents=A, alpha=0.25)
    RT, AT = 1e-3, 1e-3 * H
    s = O.cheetah_start_states(1, S)
    for t in range(2):
        noise = {"trunc": [full.dump_noise(L.NOISE_TRUNC_NORMAL, t, it, (N, A, H, U)) for it in range(iters)]}
        a_f, n_f, _ = full.optimize(s, t)
        a_s, n_s, _ = shard.optimize(s, t)
        hip_r = [full.get_trace(it, L.TRACE_REWARDS) for it in range(iters)]

        def elites(it, r_o, own):
            # this oracle's rewards against the device's, then the elite set tf.nn.top_k takes from the DEVICE's rewards
            # (ties: lower index first), so that near-ties at the k-th place cannot split the two refits
            assert_cheetah_rewards(hip_r[it], r_o, RT, AT)
            return O.topk_desc(hip_r[it].T, k)
        act_o = cem._optimize(s, noise, forced_elites=elites)
        for it in range(iters):
            # the sharded elite set, by GLOBAL particle index, is that set
            want = O.topk_desc(hip_r[it][:, 0], k)
            assert set(shard.get_trace(it, L.TRACE_ELITES)[0].tolist()) == set(np.asarray(want).tolist())
            np.testing.assert_allclose(shard.get_trace(it, L.TRACE_MEAN), cem.trace[it]["mean"], rtol=0, atol=1e-4)
            np.testing.assert_allclose(shard.get_trace(it, L.TRACE_VAR), cem.trace[it]["var"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(a_s, act_o, rtol=0, atol=1e-4)
        np.testing.assert_allclose(n_s, co.predict_next_state(s, act_o), rtol=2e-5, atol=2e-4)
        s = n_f
