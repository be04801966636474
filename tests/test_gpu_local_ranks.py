"""Two -- and north_star's eight -- RANKS on one GPU: the engine's nranks > 1 paths -- record gather with its stream hand-offs, the per-iteration
exchanges and rank-ordered merges of a sharded population -- driven through the in-process communicator
(bbmpc_comm_init_local, csrc/comm.hpp), which serves the same call sites as RCCL's communicator does between processes.
Every rank is driven by its own host thread (the communicator's collectives hold a host rendezvous).
RCCL itself refuses two ranks on one device; tests/test_gpu_two_ranks.py is the two-process, two-GPU form of the same checks
and skips on a one-GPU box."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1
    return _lib


def _in_threads(fns):
    """Each rank on its own host thread (ctypes releases the GIL inside the library); re-raises the first failure."""
    out, err = [None] * len(fns), [None] * len(fns)

    def run(i):
        try:
            out[i] = fns[i]()
        except BaseException as ex:                       # noqa: BLE001
            err[i] = ex
    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(fns))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not any(t.is_alive() for t in ts), "a rank hung in its collective"
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("opt_name", ["CEM", "PI2"])
def test_agent_shards_gather_records_on_two_ranks(L, opt_name):
    import torch
    from blackbox_mpc_amd import parallel as P
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    opt = getattr(L, "OPT_" + opt_name)
    A_glob, N, H, iters, R = 4, 300, 20, 3, 2

    def pend(A, **kw):
        return Engine(opt, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A, planning_horizon=H,
                      population_size=N, max_iterations=iters, num_elite=30, seed=5, **kw)
    full = pend(A_glob)
    shards = [P.agent_shard(A_glob, R, r) for r in range(R)]
    ranks = [pend(cnt, agent_offset=off, num_agents_global=A_glob) for off, cnt in shards]
    P.attach_local_comm(ranks)
    for r, e in enumerate(ranks):
        assert e.comm_info()[:2] == (R, r)
    rec_w = 1 + 3 + 1
    gathered = [[torch.full((A_glob, rec_w), -7.0, device="cuda") for _ in range(2)] for _ in range(R)]
    state = SY.pendulum_start_states(A_glob)
    for t in range(4):
        a_f, n_f, r_f = full.optimize(state)
        want = np.concatenate([a_f, n_f, np.asarray(r_f).reshape(-1, 1)], axis=1)
        b = t & 1

        # a rank per host thread, as ranks are processes elsewhere: bbmpc_optimize_gather stages the rank's records and
        # bbmpc_gather_wait enqueues the collective and blocks until it is through -- which takes every rank's part
        def rank_step(r):
            off, cnt = shards[r]
            a_m, n_m, r_m = ranks[r].optimize_gather(state[off:off + cnt], gathered[r][b].data_ptr(), b, t)
            ranks[r].gather_wait(b, host_block=True)
            return a_m, n_m
        res = _in_threads([lambda r=r: rank_step(r) for r in range(R)])
        for r, (off, cnt) in enumerate(shards):
            a_m, n_m = res[r]
            assert np.array_equal(a_m, a_f[off:off + cnt]) and np.array_equal(n_m, n_f[off:off + cnt])
            g = gathered[r][b].cpu().numpy()
            assert np.array_equal(g.view(np.int32), want.astype(F).view(np.int32)), (t, r)      # every rank: every agent's record, bit for bit
        state = n_f
    for e in ranks:
        e.comm_destroy()


@pytest.mark.parametrize("opt_name", ["PI2", "CEM", "RandomSearch", "PSO", "SPSA", "CMA-ES"])
def test_population_shards_on_two_ranks_match_the_one_handle_loopback(L, monkeypatch, opt_name):
    # Two handles = two ranks, each rolling out half of ONE agent's population, exchanging partials every iteration on their launch
    # streams.  The single-handle loopback hook (BBMPC_POPSHARD_LOOPBACK=2: one handle plays both shards in turn and merges in rank
    # order) computes the same thing without a communicator: bit-identical.  And both stay within the sharding tolerance of the
    # unsharded optimizer (order of the fp32 sums: DESIGN.md section 6).
    from blackbox_mpc_amd import parallel as P
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    opt = {"PI2": L.OPT_PI2, "CEM": L.OPT_CEM, "RandomSearch": L.OPT_RANDOM_SEARCH, "PSO": L.OPT_PSO, "SPSA": L.OPT_SPSA,
           "CMA-ES": L.OPT_CMAES}[opt_name]
    S, U, H, N, R = 20, 6, 30, 1000, 2
    ws, bs = SY.make_mlp_params()
    stats = SY.cheetah_stats(S, U)

    def mk(n, **kw):
        e = Engine(opt, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=1, planning_horizon=H,
                   population_size=n, max_iterations=(0 if opt_name == "RandomSearch" else 5), num_elite=50, seed=9, **kw)
        e.set_mlp(ws, bs, [L.ACT_TANH, L.ACT_TANH, L.ACT_NONE], stats)
        return e
    one = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(R))
    loop = mk(N // R, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    ranks = []
    for r in range(R):
        off, cnt = P.population_shard(N, R, r)
        ranks.append(mk(cnt, population_offset=off, population_global=N))
    P.attach_local_comm(ranks)
    for e in (one, loop, *ranks):
        e.reset()
    st = SY.cheetah_start_states(1, S)
    s_one, s_loop, s_rk = st.copy(), st.copy(), st.copy()
    for t in range(3):
        a1, n1, _ = one.optimize(s_one, t)
        a2, n2, _ = loop.optimize(s_loop, t)
        res = _in_threads([lambda e=e: e.optimize(s_rk, t) for e in ranks])
        for a3, n3, _ in res:                                   # every rank ends the control step with the same action
            assert np.array_equal(a3, a2) and np.array_equal(n3, n2), (opt_name, t)
        tol = 0.0 if opt_name in ("RandomSearch", "PSO", "CMA-ES") else 2e-5      # argmax / sorted-elite exchanges: the unsharded bits
        assert float(np.abs(a1 - a2).max()) <= tol and float(np.abs(n1 - n2).max()) <= tol, (opt_name, t)
        s_one, s_loop, s_rk = n1, n2, res[0][1]
    for e in ranks:
        e.comm_destroy()


def _eight_agent_shards(L, make, A_glob, rec_w, state, steps, label):
    """north_star's split as stated -- 8 ranks, contiguous agent blocks, one all-gather of the [A_local, U+S+1] records per
    control step (optimizer_base.py:55-95 is per agent) -- on one GPU: every rank's gathered records are the
    unsharded engine's, bit for bit."""
    import torch
    from blackbox_mpc_amd import parallel as P
    R = 8
    full = make(A_glob)
    shards = [P.agent_shard(A_glob, R, r) for r in range(R)]
    ranks = [make(cnt, agent_offset=off, num_agents_global=A_glob) for off, cnt in shards]
    P.attach_local_comm(ranks)
    info = [e.comm_info()[:2] for e in ranks]
    assert info == [(R, r) for r in range(R)]
    print(f"[{label}] rccl_ranks={R} (in-process communicator), shard map (offset, agents): {shards}")
    gathered = [[torch.full((A_glob, rec_w), -7.0, device="cuda") for _ in range(2)] for _ in range(R)]
    for t in range(steps):
        a_f, n_f, r_f = full.optimize(state)
        want = np.concatenate([a_f, n_f, np.asarray(r_f).reshape(-1, 1)], axis=1).astype(F)
        b = t & 1

        def rank_step(r):
            off, cnt = shards[r]
            a_m, n_m, _ = ranks[r].optimize_gather(state[off:off + cnt], gathered[r][b].data_ptr(), b, t)
            ranks[r].gather_wait(b, host_block=True)
            return a_m, n_m
        res = _in_threads([lambda r=r: rank_step(r) for r in range(R)])
        for r, (off, cnt) in enumerate(shards):
            a_m, n_m = res[r]
            assert np.array_equal(a_m, a_f[off:off + cnt]) and np.array_equal(n_m, n_f[off:off + cnt]), (label, t, r)
            g = gathered[r][b].cpu().numpy()
            assert np.array_equal(g.view(np.int32), want.view(np.int32)), (label, t, r)
        state = n_f
    for e in ranks:
        e.comm_destroy()


def test_config3_eight_agent_shards_of_eight(L):
    # BASELINE configs[2]: Pendulum PI2, N=1000, H=30, 64 agents over 8 ranks
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY

    def make(A, **kw):
        return Engine(L.OPT_PI2, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A, planning_horizon=30,
                      population_size=1000, max_iterations=5, lamda=1.0, seed=3, **kw)
    _eight_agent_shards(L, make, 64, 1 + 3 + 1, SY.pendulum_start_states(64), 3, "config 3, 8 x 8 agents")


@pytest.mark.parametrize("opt_name", ["PSO", "CEM"])
def test_config5_eight_agent_shards_of_four(L, opt_name):
    # BASELINE configs[4], the PSO and CEM legs: learned 26-200-200-20 model, N=2000, H=50, 32 agents over 8 ranks
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    S, U = 20, 6
    ws, bs = SY.make_mlp_params()
    stats = SY.cheetah_stats(S, U)
    opt = {"PSO": L.OPT_PSO, "CEM": L.OPT_CEM}[opt_name]

    def make(A, **kw):
        e = Engine(opt, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=A, planning_horizon=50,
                   population_size=2000, max_iterations=5, num_elite=50, seed=21, **kw)
        e.set_mlp(ws, bs, [L.ACT_TANH, L.ACT_TANH, L.ACT_NONE], stats)
        return e
    _eight_agent_shards(L, make, 32, U + S + 1, SY.cheetah_start_states(32, S), 2, "config 5 %s, 8 x 4 agents" % opt_name)


@pytest.mark.parametrize("opt_name", ["PI2", "CEM", "RandomSearch", "PSO", "SPSA", "CMA-ES"])
def test_eight_population_shards_of_2000(L, monkeypatch, opt_name):
    # f-4 eight ways: ONE agent's population of 2000 (config 5's size) as eight shards of 250 on eight ranks, exchanging partials
    # every iteration; bit-identical to the one-handle loopback that plays the eight shards in turn, and within the sharding
    # tolerance of the unsharded optimizer (bit-identical for the argmax / sorted-elite exchanges).
    from blackbox_mpc_amd import parallel as P
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    opt = {"PI2": L.OPT_PI2, "CEM": L.OPT_CEM, "RandomSearch": L.OPT_RANDOM_SEARCH, "PSO": L.OPT_PSO, "SPSA": L.OPT_SPSA,
           "CMA-ES": L.OPT_CMAES}[opt_name]
    S, U, H, N, R = 20, 6, 30, 2000, 8
    ws, bs = SY.make_mlp_params()
    stats = SY.cheetah_stats(S, U)

    def mk(n, **kw):
        e = Engine(opt, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=1, planning_horizon=H,
                   population_size=n, max_iterations=(0 if opt_name == "RandomSearch" else 5), num_elite=50, seed=9, **kw)
        e.set_mlp(ws, bs, [L.ACT_TANH, L.ACT_TANH, L.ACT_NONE], stats)
        return e
    one = mk(N)
    monkeypatch.setenv("BBMPC_POPSHARD_LOOPBACK", str(R))
    loop = mk(N // R, population_global=N)
    monkeypatch.delenv("BBMPC_POPSHARD_LOOPBACK")
    shards = [P.population_shard(N, R, r) for r in range(R)]
    ranks = [mk(cnt, population_offset=off, population_global=N) for off, cnt in shards]
    P.attach_local_comm(ranks)
    assert [e.comm_info()[:2] for e in ranks] == [(R, r) for r in range(R)]
    print(f"[{opt_name}] rccl_ranks={R} (in-process communicator), population shards (offset, particles): {shards}")
    for e in (one, loop, *ranks):
        e.reset()
    st = SY.cheetah_start_states(1, S)
    s_one, s_loop, s_rk = st.copy(), st.copy(), st.copy()
    for t in range(2):
        a1, n1, _ = one.optimize(s_one, t)
        a2, n2, _ = loop.optimize(s_loop, t)
        res = _in_threads([lambda e=e: e.optimize(s_rk, t) for e in ranks])
        for a3, n3, _ in res:
            assert np.array_equal(a3, a2) and np.array_equal(n3, n2), (opt_name, t)
        tol = 0.0 if opt_name in ("RandomSearch", "PSO", "CMA-ES") else 2e-5
        assert float(np.abs(a1 - a2).max()) <= tol and float(np.abs(n1 - n2).max()) <= tol, (opt_name, t)
        s_one, s_loop, s_rk = n1, n2, res[0][1]
    for e in ranks:
        e.comm_destroy()


def test_local_group_refuses_a_join_after_a_rank_left_and_reports_the_byte_limit(L):
    from blackbox_mpc_amd.engine import Engine
    mk = lambda: Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=5,
                        population_size=64, max_iterations=2, num_elite=8)
    a, b, c = mk(), mk(), mk()
    a.comm_init_local(777, 3, 0)
    b.comm_init_local(777, 3, 1)
    b.comm_destroy()                                            # a rank leaves: the group is broken for good
    with pytest.raises(L.BBMPCError):
        c.comm_init_local(777, 3, 2)
    a.comm_destroy()
    c.comm_init_local(777, 3, 2)                                # the key is free again once every member has left
    c.comm_destroy()


def test_local_group_shape_is_checked(L):
    from blackbox_mpc_amd.engine import Engine
    mk = lambda: Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=5,
                        population_size=64, max_iterations=2, num_elite=8)
    a, b = mk(), mk()
    a.comm_init_local(12345, 2, 0)
    with pytest.raises(L.BBMPCError):
        b.comm_init_local(12345, 3, 1)                          # another group size under the same key
    with pytest.raises(L.BBMPCError):
        b.comm_init_local(12345, 2, 2)                          # rank out of range
    with pytest.raises(L.BBMPCError):
        a.comm_init_local(12345, 2, 1)                          # the handle already has a communicator
    with pytest.raises(L.BBMPCError):
        b.comm_init_local(12345, 2, 0)                          # rank 0 is taken
    b.comm_init_local(12345, 2, 1)
    assert a.comm_info()[:2] == (2, 0) and b.comm_info()[:2] == (2, 1)
    a.comm_destroy()
    b.comm_destroy()
