"""The agent-sharded path's record all-gather through the engine's own RCCL communicator (include/bbmpc.h
"multi-GPU", csrc/comm.hpp), exercised as a one-rank group on one GPU: every hand-off form (sequence number
published by the persistent kernel, event on the launch stream; flags or events for completion) must deliver exactly
the records the control step wrote, and must not change them.  The >1-rank data path is RCCL's own; the rank logic
around it is covered on CPU with gloo (tests/test_parallel_cpu.py)."""
import numpy as np
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    assert _lib.device_count() >= 1
    return _lib


def _pendulum_engine(L, opt, A=2, N=200, H=12, iters=3, **kw):
    from blackbox_mpc_amd.engine import Engine
    return Engine(opt, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A, planning_horizon=H,
                  population_size=N, max_iterations=iters, num_elite=20, seed=3, **kw)


def _mlp_engine(L, A=2, N=64, H=6, iters=2):
    from blackbox_mpc_amd.engine import Engine
    U, S = 6, 20
    eng = Engine(L.OPT_CEM, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=A, planning_horizon=H,
                 population_size=N, max_iterations=iters, num_elite=8, seed=3)
    ws, bs = O.make_mlp_params([S + U, 32, 32, S], seed=1)
    eng.set_mlp(ws, bs, [L.ACT_TANH, L.ACT_TANH, L.ACT_NONE], None)
    return eng


def _closed_loop(eng, start, steps, gather):
    """`steps` closed-loop control steps on the device; returns the per-step records (and gathered copies)."""
    import torch
    dev = torch.device("cuda", 0)
    A, S = start.shape
    rec_w = eng.U + S + 1
    state = torch.from_numpy(start).to(dev)
    nxt = torch.empty_like(state)
    records = [torch.zeros((A, rec_w), device=dev) for _ in range(2)]
    gathered = [torch.full((A, rec_w), -7.0, device=dev) for _ in range(2)]
    out, out_g = [], []
    for t in range(steps):
        b = t & 1
        if gather:
            eng.gather_wait(b)
            eng.optimize_gather_dev(state.data_ptr(), records[b].data_ptr(), gathered[b].data_ptr(), b,
                                    d_next_state=nxt.data_ptr())
            eng.gather_wait(b, host_block=True)
            out_g.append(gathered[b].cpu().numpy().copy())
        else:
            eng.optimize_dev(state.data_ptr(), records[b].data_ptr(), d_next_state=nxt.data_ptr())
        eng.synchronize()
        out.append(records[b].cpu().numpy().copy())
        state, nxt = nxt, state
    return out, out_g


@pytest.mark.parametrize("sync", ["flags", "event"])
@pytest.mark.parametrize("kind", ["cem", "pi2", "spsa", "rs", "pso", "cmaes", "mlp"])
def test_one_rank_gather_delivers_the_records(L, monkeypatch, kind, sync):
    # cem / pi2 / spsa / rs / pso: one persistent kernel per control step (it publishes the hand-off itself in flags mode);
    # cmaes / mlp: many launches per control step (event hand-off)
    from blackbox_mpc_amd.engine import Engine
    monkeypatch.setenv("BBMPC_COMM_SYNC", sync)

    def make():
        if kind == "mlp":
            return _mlp_engine(L), O.cheetah_start_states(2, 20)
        opt = {"cem": L.OPT_CEM, "pi2": L.OPT_PI2, "spsa": L.OPT_SPSA, "rs": L.OPT_RANDOM_SEARCH, "pso": L.OPT_PSO,
               "cmaes": L.OPT_CMAES}[kind]
        return _pendulum_engine(L, opt), O.pendulum_start_states(2)

    ref_eng, start = make()
    ref_eng.reset()
    ref, _ = _closed_loop(ref_eng, start, 6, gather=False)

    eng, start = make()
    eng.reset()
    eng.comm_init(Engine.comm_unique_id(), 1, 0)
    got, got_g = _closed_loop(eng, start, 6, gather=True)
    for t in range(6):
        np.testing.assert_array_equal(got[t].view(np.int32), ref[t].view(np.int32))      # the gather changes nothing
        np.testing.assert_array_equal(got_g[t].view(np.int32), got[t].view(np.int32))    # and delivers the records
    eng.comm_destroy()
    # the handle keeps working without a communicator
    more, _ = _closed_loop(eng, start, 1, gather=False)
    assert np.isfinite(more[0]).all()


def test_gather_overlaps_following_control_steps(L):
    # bench.py's pattern: no host wait between control steps, two slots in flight
    import torch
    from blackbox_mpc_amd.engine import Engine
    dev = torch.device("cuda", 0)
    eng = _pendulum_engine(L, L.OPT_CEM, A=3, N=500, H=30, iters=5)
    eng.reset()
    eng.comm_init(Engine.comm_unique_id(), 1, 0)
    state = torch.from_numpy(O.pendulum_start_states(3)).to(dev)
    nxt = torch.empty_like(state)
    records = [torch.zeros((3, 5), device=dev) for _ in range(2)]
    gathered = [torch.zeros((3, 5), device=dev) for _ in range(2)]
    for t in range(200):
        b = t & 1
        eng.gather_wait(b)
        eng.optimize_gather_dev(state.data_ptr(), records[b].data_ptr(), gathered[b].data_ptr(), b,
                                d_next_state=nxt.data_ptr())
        state, nxt = nxt, state
    eng.synchronize()
    for b in range(2):
        assert torch.equal(gathered[b].view(torch.int32), records[b].view(torch.int32))
    assert not torch.equal(gathered[0], gathered[1])


def test_comm_call_sequence_errors(L):
    import torch
    from blackbox_mpc_amd.engine import Engine
    dev = torch.device("cuda", 0)
    eng = _pendulum_engine(L, L.OPT_CEM)
    buf = torch.zeros((2, 5), device=dev)
    out = torch.zeros((2, 5), device=dev)
    st = torch.from_numpy(O.pendulum_start_states(2)).to(dev)
    with pytest.raises(L.BBMPCError) as ei:                      # no communicator yet
        eng.gather_records_dev(buf.data_ptr(), out.data_ptr(), 10, 0)
    assert ei.value.code == L.E_STATE
    eng.gather_wait(0)                                           # nothing pending: a no-op
    with pytest.raises(L.BBMPCError):
        eng.comm_init(Engine.comm_unique_id(), 2, 2)             # rank out of range
    with pytest.raises(ValueError):
        eng.comm_init(b"short", 1, 0)
    eng.comm_init(Engine.comm_unique_id(), 1, 0)
    with pytest.raises(L.BBMPCError) as ei:                      # one communicator per handle
        eng.comm_init(Engine.comm_unique_id(), 1, 0)
    assert ei.value.code == L.E_STATE
    with pytest.raises(L.BBMPCError):
        eng.gather_records_dev(buf.data_ptr(), out.data_ptr(), 10, 2)      # slot out of range
    eng.optimize_gather_dev(st.data_ptr(), buf.data_ptr(), out.data_ptr(), 1)
    with pytest.raises(L.BBMPCError) as ei:                      # slot 1 is still pending
        eng.optimize_gather_dev(st.data_ptr(), buf.data_ptr(), out.data_ptr(), 1)
    assert ei.value.code == L.E_STATE
    eng.gather_wait(1, host_block=True)
    assert torch.equal(out, buf)
    eng.gather_records_dev(buf.data_ptr(), out.data_ptr(), 10, 1)          # the two-call form
    eng.gather_wait(1, host_block=True)
    eng.comm_destroy()
    eng.comm_destroy()                                           # idempotent


@pytest.mark.parametrize("which", ["default", "side"])
def test_set_torch_stream_orders_torch_work_with_the_kernels(L, which):
    # PyTorch's default stream has the NULL handle, which bbmpc_set_stream reads as "the handle's own stream";
    # set_torch_stream maps it to hipStreamLegacy.  Torch work issued right before / after a ~50 us control step, with
    # no synchronisation, must be ordered with it.
    import torch
    dev = torch.device("cuda", 0)
    eng = _pendulum_engine(L, L.OPT_CEM, A=1, N=500, H=30, iters=5)
    eng.reset()
    s = torch.cuda.default_stream(dev) if which == "default" else torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        eng.set_torch_stream(s)
        state = torch.from_numpy(O.pendulum_start_states(1)).to(dev)
        nxt = torch.empty_like(state)
        rec = torch.zeros((1, 5), device=dev)
        snaps = []
        for t in range(30):
            rec.fill_(-100.0)                                    # ordered before the kernel's record store
            eng.optimize_dev(state.data_ptr(), rec.data_ptr(), d_next_state=nxt.data_ptr())
            snaps.append(rec.clone())                            # ordered after it
            state, nxt = nxt, state
        torch.cuda.synchronize()
    eng.set_stream(None)
    for snap in snaps:
        assert (snap != -100.0).all() and torch.isfinite(snap).all()


def test_sequence_number_wrap(L, monkeypatch):
    # the hand-off's sequence numbers are reset (both streams drained, flags zeroed) before they reach 2^31
    import torch
    from blackbox_mpc_amd.engine import Engine
    monkeypatch.setenv("BBMPC_COMM_SEQ_START", str(0x7FFFFFFF - 5))
    dev = torch.device("cuda", 0)

    def run(gather):
        eng = _pendulum_engine(L, L.OPT_CEM, A=2, N=200, H=12, iters=3)
        eng.reset()
        if gather:
            eng.comm_init(Engine.comm_unique_id(), 1, 0)
        return _closed_loop(eng, O.pendulum_start_states(2), 14, gather=gather)

    ref, _ = run(False)
    got, got_g = run(True)
    for t in range(14):
        np.testing.assert_array_equal(got[t].view(np.int32), ref[t].view(np.int32))
        np.testing.assert_array_equal(got_g[t].view(np.int32), got[t].view(np.int32))


def test_handles_on_two_devices_in_one_process():
    # every entry point runs under a device guard -- handles on different GPUs in one process, and the
    # caller's current device is left as it was.  Needs two GPUs (skipped on the one-GPU test box).
    import torch
    from blackbox_mpc_amd import _lib as L
    from blackbox_mpc_amd.engine import Engine
    from oracle import oracle_np as O
    if L.device_count() < 2:
        pytest.skip("needs two GPUs")
    torch.cuda.set_device(0)
    mk = lambda dev: Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=2,
                            planning_horizon=10, population_size=128, max_iterations=2, num_elite=16, seed=3, device=dev)
    e0, e1 = mk(0), mk(1)
    assert torch.cuda.current_device() == 0
    s = O.pendulum_start_states(2)
    a0 = e0.optimize(s)[0]
    a1 = e1.optimize(s)[0]                    # lazy allocations / launches of the second handle under device 1
    assert torch.cuda.current_device() == 0
    np.testing.assert_array_equal(a0, a1)     # same seed, same inputs: the device does not matter
    np.testing.assert_array_equal(e0.optimize(s)[0], e1.optimize(s)[0])


@pytest.mark.parametrize("sync", ["flags", "event"])
@pytest.mark.parametrize("kind,A", [("cem", 1), ("pi2", 3), ("mlp", 2)])
def test_host_call_with_gather_matches_the_plain_host_call(L, monkeypatch, kind, A, sync):
    # bbmpc_optimize_gather = bbmpc_optimize for the rank's agents (for one agent: served by the resident kernel) + the
    # records' way back to HBM and into the collective on the communication stream only
    import torch
    from blackbox_mpc_amd.engine import Engine
    monkeypatch.setenv("BBMPC_COMM_SYNC", sync)

    def make():
        if kind == "mlp":
            return _mlp_engine(L, A=A)
        return _pendulum_engine(L, L.OPT_CEM if kind == "cem" else L.OPT_PI2, A=A)
    ref, eng = make(), make()
    eng.comm_init(Engine.comm_unique_id(), 1, 0)
    dev = torch.device("cuda", 0)
    S = ref.S
    rec_w = ref.U + S + 1
    gathered = [torch.full((A, rec_w), -7.0, device=dev) for _ in range(2)]
    s_ref = s = (O.cheetah_start_states(A, S) if kind == "mlp" else O.pendulum_start_states(A))
    for t in range(24):                              # crosses noise-prefetch chunks; slots alternate
        b = t & 1
        a_r, n_r, r_r = ref.optimize(s_ref, t)
        eng.gather_wait(b)
        a_g, n_g, r_g = eng.optimize_gather(s, gathered[b].data_ptr(), b, t)
        np.testing.assert_array_equal(a_g, a_r)
        np.testing.assert_array_equal(n_g, n_r)
        np.testing.assert_array_equal(r_g, r_r)
        if t % 5 == 0:
            eng.gather_wait(b, host_block=True)
            want = np.concatenate([a_g, n_g, r_g.reshape(-1, 1)], axis=1)
            np.testing.assert_array_equal(gathered[b].cpu().numpy(), want)
        s_ref, s = n_r, n_g
    for b in range(2):
        eng.gather_wait(b, host_block=True)
