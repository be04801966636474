"""GPU parity tests for CMA-ES.

Per SURVEY.md H4 the eigen-factorisation's sign/order conventions are library-specific, so parity is
asserted (a) for iteration-0 samples (B = D = I), (b) for the deterministic update (m, sigma, p_sigma, p_C, C)
given identical samples, with the oracle continued from the engine's own (s, B) each iteration, and
(c) through invariants of the factorisation: B diag(D^2) B^T ~= C, D sorted descending, B orthonormal."""
import numpy as np
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
F = np.float32
LO, HI = [-2.0], [2.0]


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1
    return _lib


def _ev():
    return O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))


def _engine(L, A, H, N, iters, k, **kw):
    from blackbox_mpc_amd.engine import Engine
    return Engine(L.OPT_CMAES, L.DYN_PENDULUM, L.REW_PENDULUM, LO, HI, dim_s=3, num_agents=A, planning_horizon=H,
                  population_size=N, max_iterations=iters, num_elite=k, **kw)


def _state(eng, n, G=1):
    g = lambda name, shape: eng.get_state(name, shape)
    return dict(m=g("m", (G * n,)), sigma=g("sigma", (G * n,)), p_sigma=g("p_sigma", (G * n,)), p_C=g("p_C", (G * n,)),
                D=g("D", (G * n,)), C=g("C", (G, n, n)), B=g("B", (G, n, n)))


@pytest.mark.parametrize("N,A,H,k", [(64, 2, 5, 8), (200, 1, 30, 20)])
def test_cmaes_coupled_lockstep(L, N, A, H, k):
    n = A * H
    eng = _engine(L, A, H, N, 1, k)
    eng.set_trace(True)
    cma = O.CMAES(_ev(), LO, HI, horizon=H, max_iterations=1, population=N, num_elite=k, num_agents=A)
    rng = np.random.default_rng(N)
    states = O.pendulum_start_states(A)
    for step in range(4):
        z = rng.standard_normal((1, N, n)).astype(F)
        eng.inject_noise(L.NOISE_NORMAL, z.reshape(1, N, A, H, 1))
        act, nxt, rew = eng.optimize(states)
        st = _state(eng, n)
        hip_r = eng.get_trace(0, L.TRACE_REWARDS)
        hip_order = eng.get_trace(0, L.TRACE_ELITES)[0]

        def order(it, rsum, own):
            np.testing.assert_allclose(hip_r.sum(axis=1), rsum, rtol=2e-4, atol=2e-3 * A)
            np.testing.assert_array_equal(hip_order, O.topk_desc(hip_r.sum(axis=1, dtype=np.float32), k))
            if not np.array_equal(own[:k], hip_order):         # only near-ties may swap
                for a_, b_ in zip(own[:k], hip_order):
                    assert abs(rsum[a_] - rsum[b_]) <= 2e-3 * A + 2e-4 * abs(rsum[a_])
            return hip_order
        s_hip = (st["D"].astype(np.float64) ** 2).astype(F)
        cma._optimize(states, {"normal": [z[0]]}, eig=[(s_hip, st["B"][0])], forced_order=order)
        tr = cma.trace[0]
        np.testing.assert_allclose(eng.get_trace(0, L.TRACE_SAMPLES), tr["samples"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(hip_r, tr["rewards"], rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(st["m"], tr["m"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(st["sigma"], tr["sigma"], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(st["p_sigma"], tr["p_sigma"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(st["p_C"], tr["p_C"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(st["C"][0], tr["C"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(act, tr["m"].reshape(A, H, 1)[:, 0], rtol=0, atol=2e-5)
        # factorisation invariants (cma_es.py:195-198)
        B, D, C = st["B"][0].astype(np.float64), st["D"].astype(np.float64), st["C"][0].astype(np.float64)
        np.testing.assert_allclose(B @ np.diag(D ** 2) @ B.T, C, rtol=0, atol=2e-5 * max(1.0, np.abs(C).max()))
        np.testing.assert_allclose(B.T @ B, np.eye(n), rtol=0, atol=2e-5)
        assert np.all(D[:-1] >= D[1:] - 1e-6) and np.all(D > 0)
        np.testing.assert_array_equal(st["C"][0], st["C"][0].T)       # upper triangle mirrored (:188-190)
    # reset() restores only m and sigma (cma_es.py:215-227)
    c_before = eng.get_state("C", (1, n, n))
    eng.reset()
    np.testing.assert_array_equal(eng.get_state("C", (1, n, n)), c_before)
    np.testing.assert_allclose(eng.get_state("m", (n,)), 0.0, atol=0)
    np.testing.assert_allclose(eng.get_state("sigma", (n,)), 1.0, atol=0)


def test_cmaes_multi_iteration_and_engine_normals(L):
    N, A, H, k, iters = 256, 1, 12, 16, 4
    eng = _engine(L, A, H, N, iters, k, seed=21)
    z = eng.dump_noise(L.NOISE_NORMAL, 0, 0, (N, A, H, 1)).astype(np.float64).ravel()
    assert abs(z.mean()) < 0.06 and abs(z.std() - 1.0) < 0.05 and np.abs(z).max() < 6
    states = O.pendulum_start_states(A)
    r0 = None
    for step in range(3):
        act, nxt, rew = eng.optimize(states)
        assert np.all(np.isfinite(act)) and np.all(np.abs(act) <= 2.0 + 1e-6) is not None
        r0 = rew if r0 is None else r0
    n = A * H
    st = _state(eng, n)
    B, D, C = st["B"][0].astype(np.float64), st["D"].astype(np.float64), st["C"][0].astype(np.float64)
    np.testing.assert_allclose(B @ np.diag(D ** 2) @ B.T, C, rtol=0, atol=5e-5 * max(1.0, np.abs(C).max()))
    assert np.all(np.isfinite(st["m"])) and np.all(st["sigma"] > 0)


def test_cmaes_per_agent_mode_shards(L):
    # per-agent mode: each agent runs its own n = H*U instance; A=3 engine == three single-agent engines
    # (RNG keyed by global agent id), and for A=1 it coincides with the reference's coupled mode.
    N, A, H, k, iters = 96, 3, 6, 12, 2
    states = O.pendulum_start_states(A)
    full = _engine(L, A, H, N, iters, k, seed=4, quirks=L.CMAES_PER_AGENT)
    acts = [full.optimize(states)[0] for _ in range(2)]
    for a in range(A):
        one = _engine(L, 1, H, N, iters, k, seed=4, quirks=L.CMAES_PER_AGENT, agent_offset=a, num_agents_global=A)
        for s in range(2):
            np.testing.assert_array_equal(one.optimize(states[a:a + 1])[0], acts[s][a:a + 1])
    coupled = _engine(L, 1, H, N, iters, k, seed=4)
    np.testing.assert_array_equal(coupled.optimize(states[:1])[0], acts[0][:1])
    from blackbox_mpc_amd import _lib
    with pytest.raises(_lib.BBMPCError):          # coupled mode cannot be sharded
        _engine(L, 1, H, N, iters, k, agent_offset=0, num_agents_global=2)


def test_cmaes_with_learned_dynamics_runs(L):
    from blackbox_mpc_amd.engine import Engine
    S, U, N, A, H, k, iters = 20, 6, 128, 2, 10, 16, 2
    ws, bs = O.make_mlp_params([26, 200, 200, 20])
    z, o = np.zeros, np.ones
    stats = [z(S, F), o(S, F), z(U, F), o(U, F), z(S, F), np.full(S, 0.1, F)]
    eng = Engine(L.OPT_CMAES, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=A,
                 planning_horizon=H, population_size=N, max_iterations=iters, num_elite=k, quirks=L.CMAES_PER_AGENT)
    eng.set_mlp(ws, bs, [1, 1, 0], stats)
    states = O.cheetah_start_states(A, S)
    for _ in range(2):
        act, nxt, rew = eng.optimize(states)
        assert act.shape == (A, U) and np.all(np.isfinite(act)) and np.all(np.isfinite(nxt)) and np.all(np.isfinite(rew))


@pytest.mark.parametrize("H", [96, 132, 148])
def test_block_jacobi_path_uneven_blocks_and_sharding(L, monkeypatch, H):
    # n = 96 takes the LDS-resident single-workgroup kernel with eight elements per lane (64 < n <= 128);
    # n = H*U >= 128 takes the block-Jacobi decomposition (8 column blocks over 4 workgroups per instance); sizes
    # that do not divide into equal blocks (132 = 7*17 + 13, 148 = 7*19 + 15), two instances at once (per-agent
    # mode), its invariants, shard-vs-full bit equality.
    N, A, k, iters = 96, 2, 12, 2
    n = H
    states = O.pendulum_start_states(A)

    def run():
        eng = _engine(L, A, H, N, iters, k, seed=9, quirks=L.CMAES_PER_AGENT)
        acts = [eng.optimize(states)[0] for _ in range(2)]
        return eng, acts
    eng, acts = run()
    st = _state(eng, n, G=A)
    for g in range(A):
        B = st["B"][g].astype(np.float64)
        D = st["D"][g * n:(g + 1) * n].astype(np.float64)
        C = st["C"][g].astype(np.float64)
        np.testing.assert_allclose(B @ np.diag(D ** 2) @ B.T, C, rtol=0, atol=5e-5 * max(1.0, np.abs(C).max()))
        np.testing.assert_allclose(B.T @ B, np.eye(n), rtol=0, atol=5e-5)
        assert np.all(D[:-1] >= D[1:] - 1e-6) and np.all(D > 0)
    # sharding: each agent alone gives the same actions bit for bit
    for a in range(A):
        one = _engine(L, 1, H, N, iters, k, seed=9, quirks=L.CMAES_PER_AGENT, agent_offset=a, num_agents_global=A)
        for s in range(2):
            np.testing.assert_array_equal(one.optimize(states[a:a + 1])[0], acts[s][a:a + 1])
    # (the single-workgroup kernel, BBMPC_CMA_SVD_ROUNDS=1, picks a different basis inside eigenvalue clusters, and
    # the sampling y = z @ (B @ D) depends on that basis (quirk Q5), so the two kernels are compared through the
    # invariants only)
    monkeypatch.setenv("BBMPC_CMA_SVD_ROUNDS", "1")
    eng2, _ = run()
    st2 = _state(eng2, n, G=A)
    for g in range(A):
        B = st2["B"][g].astype(np.float64)
        D = st2["D"][g * n:(g + 1) * n].astype(np.float64)
        C = st2["C"][g].astype(np.float64)
        np.testing.assert_allclose(B @ np.diag(D ** 2) @ B.T, C, rtol=0, atol=5e-5 * max(1.0, np.abs(C).max()))


@pytest.mark.parametrize("A,per_agent,H", [(1, False, 30), (3, True, 12), (2, True, 50)])
def test_fused_control_step_is_bit_identical_to_the_per_iteration_kernels(L, monkeypatch, A, per_agent, H):
    # opt-in (BBMPC_CMA_FUSED=1): small search dimensions (n = H*U <= 64) run the whole control step -- every
    # iteration's sampling, rollouts, top-k, evolution paths, covariance, warm start, Jacobi -- in ONE launch
    # (kernels_fused_cma.hpp); it calls the same device functions / operation orders as the eleven per-iteration
    # kernels, so results must match bit for bit, state included
    N, k, iters = 500, 50, 5
    q = L.CMAES_PER_AGENT if per_agent else 0
    ref = _engine(L, A, H, N, iters, k, seed=9, quirks=q)
    monkeypatch.setenv("BBMPC_CMA_FUSED", "1")
    fus = _engine(L, A, H, N, iters, k, seed=9, quirks=q)
    monkeypatch.delenv("BBMPC_CMA_FUSED")
    n, G = (H if per_agent else A * H), (A if per_agent else 1)
    s_r = s_f = O.pendulum_start_states(A)
    ref.set_profiling(True)
    fus.set_profiling(True)
    for t in range(5):
        if t == 3:
            ref.reset()
            fus.reset()
        a_r, n_r, r_r = ref.optimize(s_r, t, add_exploration_noise=(t == 2))
        a_f, n_f, r_f = fus.optimize(s_f, t, add_exploration_noise=(t == 2))
        np.testing.assert_array_equal(a_f, a_r)
        np.testing.assert_array_equal(n_f, n_r)
        np.testing.assert_array_equal(r_f, r_r)
        st_r, st_f = _state(ref, n, G), _state(fus, n, G)
        for name in st_r:
            np.testing.assert_array_equal(st_f[name], st_r[name], err_msg=name)
        s_r, s_f = n_r, n_f
    fused_applies = n <= 64 and G == A
    assert fus.get_profile()[2] == ("k_fused_cma_pendulum" if fused_applies else "k_rollout_pendulum")
    # per-iteration path: at n <= 32 with one agent per instance the rollouts ride on the sampling launch (kernels_eigh_small.hpp)
    assert ref.get_profile()[2] == ("k_cma_sample_roll_small" if (n <= 32 and G == A) else "k_rollout_pendulum")


def _config5_cma_engine(L, A=2, N=400, k=40, H=50):
    from blackbox_mpc_amd.engine import Engine
    S, U = 20, 6                                           # n = H * U = 300: BASELINE config 5's per-agent search dimension
    eng = Engine(L.OPT_CMAES, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=A, planning_horizon=H,
                 population_size=N, max_iterations=5, num_elite=k, seed=4, quirks=L.CMAES_PER_AGENT)
    ws, bs = O.make_mlp_params([S + U, 200, 200, S], seed=42)
    stats = [np.zeros(S, np.float32), np.ones(S, np.float32), np.zeros(U, np.float32), np.ones(U, np.float32),
             np.zeros(S, np.float32), np.full(S, 0.1, np.float32)]
    eng.set_mlp(ws, bs, [L.ACT_TANH, L.ACT_TANH, L.ACT_NONE], stats)
    eng.set_trace(True)
    return eng, O.cheetah_start_states(A, S)


def _check_factorisation(eng, L, A, n, iters, atol):
    worst = 0.0
    for it in range(iters):
        B = eng.get_trace(it, L.TRACE_CMA_B).astype(np.float64)
        C = eng.get_trace(it, L.TRACE_CMA_C).astype(np.float64)
        D = eng.get_trace(it, L.TRACE_CMA_D).astype(np.float64)
        for g in range(A):
            assert np.all(np.diff(D[g]) <= 0), "D must be sorted descending (tf.linalg.svd order)"
            worst = max(worst, np.abs(B[g] @ np.diag(D[g] ** 2) @ B[g].T - C[g]).max(), np.abs(B[g].T @ B[g] - np.eye(n)).max())
            np.testing.assert_allclose(D[g] ** 2, np.linalg.eigvalsh(C[g])[::-1], rtol=0, atol=atol)
    return worst


def test_direct_eigensolver_over_a_closed_loop(L):
    # s, U, _ = tf.linalg.svd(C) (cma_es.py:195) at n = 300 by the direct solver (csrc/kernels_eigh.hpp): the rank-deficient
    # covariances of the first iterations (C = alpha I + low rank: the tridiagonal splits) AND the full-rank ones later
    # in the episode.  Every decomposition must be accepted by the solver's own checks (statistics word 15 == 0: the
    # block Jacobi did not run) and hold the invariants to 5e-6 -- the Jacobi's threshold alone is 9e-6.
    A, n, iters = 2, 300, 5
    eng, state = _config5_cma_engine(L, A)
    for step in range(6):
        act, state, rew = eng.optimize(state)
        for it in range(iters):
            st = eng.get_trace(it, L.TRACE_CMA_SVD_STATS)
            assert np.all(st[:, 15] == 0) and np.all(st[:, :15] == 0), (step, it, st)
        assert _check_factorisation(eng, L, A, n, iters, 5e-6) <= 5e-6


@pytest.mark.parametrize("H", [22, 30, 52])
def test_direct_eigensolver_other_sizes(L, H):
    # n = 132 (just above the one-sided Jacobi's range), 180 and 312 (the direct solver takes n % 4 == 0 up to 320; other
    # sizes stay with the block Jacobi): the same checks as at n = 300 -- row classes that are partly or wholly padding,
    # 45 / 78 multisection groups
    A, n, iters = 2, 6 * H, 5
    eng, state = _config5_cma_engine(L, A, H=H)
    for step in range(3):
        act, state, rew = eng.optimize(state)
        for it in range(iters):
            st = eng.get_trace(it, L.TRACE_CMA_SVD_STATS)
            assert np.all(st[:, 15] == 0) and np.all(st[:, :15] == 0), (step, it, st)
        assert _check_factorisation(eng, L, A, n, iters, 5e-6) <= 5e-6


def test_direct_eigensolver_at_the_padded_length(L):
    # n = 320 = the padded length of every vector on the direct solver's path (pendulum, one agent, H = 320): nothing may
    # rely on a zero behind the last entry
    A, H, N, k, iters = 1, 320, 256, 32, 3
    eng = _engine(L, A, H, N, iters, k, seed=5)
    eng.set_trace(True)
    states = O.pendulum_start_states(A)
    for step in range(3):
        act, states, rew = eng.optimize(states)
        for it in range(iters):
            st = eng.get_trace(it, L.TRACE_CMA_SVD_STATS)
            assert np.all(st[:, 15] == 0) and np.all(st[:, :15] == 0), (step, it, st)
        assert _check_factorisation(eng, L, 1, H, iters, 5e-6) <= 5e-6


@pytest.mark.parametrize("scale", [1.0e-5, 3.0e4])
def test_direct_eigensolver_at_other_scales(L, scale):
    # the solver's three-term sequences run eight steps between rescalings, so k_eigh_tri_solve scales T by a power of two
    # into [1, 2) first (kernels_eigh.hpp): a covariance far from unit scale must decompose like any other -- the direct
    # solver (word 15 == 0), the same invariants
    # relative to |C|.  The small scale is a different animal: the update adds its O(c1 + cmu) rank-40 terms to 1e-5 C, i.e.
    # 260 eigenvalues within 1e-5 of each other under a norm of 1e-2 -- the direct solver's vectors fail its own checks
    # there (as they did with the quotient form) and the block Jacobi takes over; what is asserted is the factorisation.
    A, n, iters = 2, 300, 5
    eng, state = _config5_cma_engine(L, A)
    for step in range(2):
        act, state, rew = eng.optimize(state)
    eng.set_state("C", eng.get_state("C", (A, n, n)) * np.float32(scale))
    act, state, rew = eng.optimize(state)
    st = eng.get_trace(0, L.TRACE_CMA_SVD_STATS)
    if scale > 1.0:
        assert np.all(st[:, 15] == 0), st
    tol = 5e-6 if scale > 1.0 else 5e-5
    B = eng.get_trace(0, L.TRACE_CMA_B).astype(np.float64)
    C = eng.get_trace(0, L.TRACE_CMA_C).astype(np.float64)
    D = eng.get_trace(0, L.TRACE_CMA_D).astype(np.float64)
    for g in range(A):
        cn = np.abs(np.linalg.eigvalsh(C[g])).max()
        assert (cn > 1024.0) if scale > 1.0 else (cn < 0.1), cn
        assert np.abs(B[g] @ np.diag(D[g] ** 2) @ B[g].T - C[g]).max() <= tol * max(cn, 1.0)
        assert np.abs(B[g].T @ B[g] - np.eye(n)).max() <= tol


def test_direct_eigensolver_failure_hands_over_to_the_jacobi(L, monkeypatch):
    # BBMPC_CMA_EIGH_FAIL: the direct solver reports failure for every instance -> B, D must come from the block Jacobi
    # (rotations counted, word 15 == 1) and still factorise C; the same control steps as the direct path within the
    # eigenvector-basis freedom, i.e. the same eigenvalues
    monkeypatch.setenv("BBMPC_CMA_EIGH_FAIL", "1")
    A, n, iters = 2, 300, 5
    eng, state = _config5_cma_engine(L, A)
    for step in range(2):
        act, state, rew = eng.optimize(state)
        for it in range(iters):
            st = eng.get_trace(it, L.TRACE_CMA_SVD_STATS)
            assert np.all(st[:, 15] == 1) and np.all(st[:, 0] > 0), (step, it, st)
        assert _check_factorisation(eng, L, A, n, iters, 5e-5) <= 5e-5


def _small_cma_invariants(eng, G, n, atol):
    worst = 0.0
    for g in range(G):
        B = eng.get_state("B", (G, n, n))[g].astype(np.float64)
        C = eng.get_state("C", (G, n, n))[g].astype(np.float64)
        D = eng.get_state("D", (G * n,))[g * n:(g + 1) * n].astype(np.float64)
        assert np.all(np.isfinite(B)) and np.all(D > 0)
        assert np.all(np.diff(D) <= 0), "D must be sorted descending (tf.linalg.svd order)"
        worst = max(worst, np.abs(B @ np.diag(D ** 2) @ B.T - C).max(), np.abs(B.T @ B - np.eye(n)).max())
        np.testing.assert_allclose(D ** 2, np.linalg.eigvalsh(C)[::-1], rtol=0, atol=atol)
    return worst


@pytest.mark.parametrize("H,A,per_agent", [(2, 1, False), (3, 1, False), (7, 1, False), (30, 1, False), (31, 1, False),
                                           (32, 1, False), (16, 2, False), (10, 3, True)])
def test_small_direct_eigensolver_sizes(L, H, A, per_agent):
    # n = A*H <= 32 (per-agent mode: n = H, one instance per agent): the one-workgroup direct solver of
    # csrc/kernels_eigh_small.hpp.  Closed loop, so both the split tridiagonals of the first iterations (C = I, then
    # I + low rank) and the full ones later are decomposed; invariants to 5e-6 (the Jacobi's own threshold is 9e-6).
    N, k, iters = 128, 16, 3
    eng = _engine(L, A, H, N, iters, k, seed=5, **({"quirks": L.CMAES_PER_AGENT} if per_agent else {}))
    G, n = (A, H) if per_agent else (1, A * H)
    states = O.pendulum_start_states(A)
    for step in range(5):
        act, states, rew = eng.optimize(states)
        assert np.all(np.isfinite(act))
        assert _small_cma_invariants(eng, G, n, 5e-6) <= 5e-6


def test_small_direct_eigensolver_failure_hands_over_to_the_jacobi(L, monkeypatch):
    # BBMPC_CMA_EIGH_FAIL: the solver refuses every instance -> the same kernel runs warm start + Jacobi + finish as before
    monkeypatch.setenv("BBMPC_CMA_EIGH_FAIL", "1")
    N, A, H, k, iters = 128, 1, 30, 16, 3
    eng = _engine(L, A, H, N, iters, k, seed=5)
    states = O.pendulum_start_states(A)
    for step in range(3):
        act, states, rew = eng.optimize(states)
        assert _small_cma_invariants(eng, 1, A * H, 5e-5) <= 5e-5


@pytest.mark.parametrize("H,A,per_agent", [(30, 1, False), (8, 3, False), (10, 3, True)])
def test_three_launches_per_iteration_are_bit_identical_to_eleven(L, monkeypatch, H, A, per_agent):
    # n <= 32: sample | roll out | update (k_cma_sample_small, the rollout, k_cma_update_small; BBMPC_CMA_SMALL3, default
    # on) against one launch per phase -- the same device functions / operation order, so a closed loop agrees bit for bit
    N, k, iters = 200, 20, 3
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("BBMPC_CMA_SMALL3", mode)
        eng = _engine(L, A, H, N, iters, k, seed=11, **({"quirks": L.CMAES_PER_AGENT} if per_agent else {}))
        G, n = (A, H) if per_agent else (1, A * H)
        states = O.pendulum_start_states(A)
        rec = []
        for step in range(4):
            act, states, rew = eng.optimize(states)
            st = _state(eng, n, G)
            rec.append((act.copy(), states.copy(), rew.copy()) + tuple(st[name].copy() for name in sorted(st)))
        out[mode] = rec
    for r0, r1 in zip(out["0"], out["1"]):
        for x0, x1 in zip(r0, r1):
            np.testing.assert_array_equal(x0, x1)


@pytest.mark.parametrize("fail", ["0", "1"])
@pytest.mark.parametrize("H,N,k", [(50, 400, 40), (5, 200, 20)])          # n = 300 (kernels_eigh.hpp) and n = 30 (kernels_eigh_small.hpp)
def test_indefinite_covariance_is_ranked_by_singular_value(L, monkeypatch, fail, H, N, k):
    # u, B, _ = tf.linalg.svd(C) (cma_es.py:195-197) orders by the SINGULAR value |lambda|; a slightly indefinite C (fp32 drift,
    # set_state("C")) has a negative eigenvalue whose magnitude is not the smallest.  The direct solvers must rank it where the
    # SVD -- and the Jacobi fall-back (BBMPC_CMA_EIGH_FAIL=1) -- put it: D descending, D^2 = |eigenvalues|, the column of the
    # negative eigenvalue in the same place on both paths.
    monkeypatch.setenv("BBMPC_CMA_EIGH_FAIL", fail)
    A = 2
    eng, state = _config5_cma_engine(L, A, N=N, k=k, H=H)
    n = H * 6
    rng = np.random.default_rng(17)
    Cs = np.zeros((A, n, n), np.float32)
    for g in range(A):
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.linspace(0.05, 0.9, n)
        lam[n // 3] = -0.6                                   # the odd one out: negative, in the upper half by magnitude
        Cs[g] = ((Q * lam) @ Q.T).astype(np.float32)
        Cs[g] = np.triu(Cs[g]) + np.triu(Cs[g], 1).T
    eng.set_state("C", Cs)
    act, state, rew = eng.optimize(state)
    B = eng.get_trace(0, L.TRACE_CMA_B).astype(np.float64)
    C = eng.get_trace(0, L.TRACE_CMA_C).astype(np.float64)
    D = eng.get_trace(0, L.TRACE_CMA_D).astype(np.float64)
    for g in range(A):
        w, V = np.linalg.eigh(C[g])
        assert w.min() < -0.1, "the update must leave the covariance indefinite for this test to mean anything"
        sv = np.sort(np.abs(w))[::-1]
        assert np.all(np.diff(D[g]) <= 1e-6), "D must be sorted descending (tf.linalg.svd order)"
        np.testing.assert_allclose(D[g] ** 2, sv, rtol=0, atol=5e-5)
        assert np.abs(B[g].T @ B[g] - np.eye(n)).max() <= 5e-5
        # the negative eigenvalue's vector sits at the rank of its magnitude
        rank = int(np.argmin(np.abs(sv - abs(w.min()))))
        assert 0 < rank < n - 1, "the negative eigenvalue must not be the smallest singular value"
        vneg = V[:, int(np.argmin(w))]
        if min(sv[rank - 1] - sv[rank], sv[rank] - sv[rank + 1]) > 5e-3:          # (its vector is only defined when the value is isolated)
            assert abs(abs(B[g][:, rank] @ vneg) - 1.0) <= 1e-3, (rank, B[g][:, rank] @ vneg)
        # |C b_i| = s_i for every column (left singular vectors of a symmetric matrix)
        np.testing.assert_allclose(np.linalg.norm(C[g] @ B[g], axis=0), D[g] ** 2, rtol=0, atol=2e-4)
