"""Thin object wrapper over the C ABI handle (include/bbmpc.h).  NumPy in, NumPy
out; device-pointer variants take integer addresses (e.g. torch `data_ptr()`)."""
import ctypes

import numpy as np

from . import _lib as L


class Engine:
    def __init__(self, optimizer, dynamics, reward, action_low, action_high, dim_s, num_agents,
                 planning_horizon, population_size=0, max_iterations=0, num_elite=0, seed=0, quirks=0,
                 agent_offset=0, num_agents_global=None, device=-1, alpha=0.25, lamda=1.0,
                 pso_c1=0.3, pso_c2=0.5, pso_w=0.2, pso_v0_fraction=0.01,
                 spsa_alpha=0.602, spsa_gamma=0.101, spsa_a=0.01, spsa_c=0.3,
                 cma_alpha_cov=2.0, cma_h_sigma=1.0, population_offset=0, population_global=0):
        self._h = ctypes.c_void_p()
        self._lo = L.f32c(np.asarray(action_low).reshape(-1))
        self._hi = L.f32c(np.asarray(action_high).reshape(-1))
        if self._lo.shape != self._hi.shape:
            raise ValueError("action_low/action_high shapes differ")
        c = L.Config()
        c.abi_version = L.ABI_VERSION
        c.optimizer, c.dynamics, c.reward = int(optimizer), int(dynamics), int(reward)
        c.population_size, c.num_agents = int(population_size), int(num_agents)
        c.planning_horizon, c.dim_u, c.dim_s = int(planning_horizon), int(self._lo.shape[0]), int(dim_s)
        c.max_iterations = int(max_iterations or 0)
        c.num_elite = int(num_elite)
        c.agent_offset = int(agent_offset)
        c.num_agents_global = int(num_agents_global if num_agents_global is not None else num_agents)
        c.device = int(device)
        c.quirks = int(quirks)
        c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        c.alpha, c.lamda = float(alpha), float(lamda)
        c.pso_c1, c.pso_c2, c.pso_w, c.pso_v0_fraction = float(pso_c1), float(pso_c2), float(pso_w), float(pso_v0_fraction)
        c.spsa_alpha, c.spsa_gamma, c.spsa_a, c.spsa_c = float(spsa_alpha), float(spsa_gamma), float(spsa_a), float(spsa_c)
        c.cma_alpha_cov, c.cma_h_sigma = float(cma_alpha_cov), float(cma_h_sigma)
        c.population_offset, c.population_global = int(population_offset), int(population_global or 0)
        c.action_low = self._lo.ctypes.data_as(L.c_float_p)
        c.action_high = self._hi.ctypes.data_as(L.c_float_p)
        self.cfg = c
        self.N, self.A, self.H = c.population_size, c.num_agents, c.planning_horizon
        self.U, self.S = c.dim_u, c.dim_s
        self.iters = 1 if optimizer == L.OPT_RANDOM_SEARCH else c.max_iterations
        self.k = c.num_elite
        L.check(L.lib.bbmpc_create(ctypes.byref(c), ctypes.byref(self._h)))
        dev = ctypes.c_int32(-1)
        L.check(L.lib.bbmpc_handle_device(self._h, ctypes.byref(dev)))             # where the handle lives, fixed at creation: the
        self._device = int(dev.value)                                               # library's answer, not a guess from torch's state

    @property
    def device(self):
        """The GPU this handle lives on: bbmpc_config.device, or (-1) the process's current device AT CREATION -- the
        torch callbacks alias the engine's HBM pointers and stream on this device whatever is current later."""
        return self._device

    # -- lifecycle -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            L.lib.bbmpc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        """Launch on the given hipStream_t handle; 0 / None = the handle's own stream (include/bbmpc.h)."""
        L.check(L.lib.bbmpc_set_stream(self._h, ctypes.c_void_p(stream_ptr or 0)))

    def set_torch_stream(self, stream):
        """Launch on a torch.cuda.Stream so that torch work on that stream (events, collectives, copies) is ordered
        with the engine's kernels.  PyTorch's default stream has the NULL handle, which bbmpc_set_stream reads as
        "the handle's own stream": it goes through bbmpc_set_stream_default."""
        ptr = int(stream.cuda_stream)
        if ptr != 0:
            self.set_stream(ptr)
        else:
            L.check(L.lib.bbmpc_set_stream_default(self._h))

    def set_mlp(self, weights, biases, activations, stats=None):
        n = len(weights)
        ws = [L.f32c(w) for w in weights]
        bs = [L.f32c(b) for b in biases]
        dims = [ws[0].shape[0]] + [w.shape[1] for w in ws]
        for i, (w, b) in enumerate(zip(ws, bs)):
            if w.ndim != 2 or w.shape[0] != dims[i] or b.shape != (w.shape[1],):
                raise ValueError("layer %d: kernel must be [in,out] and bias [out]" % i)
        dims_a = (ctypes.c_int32 * (n + 1))(*dims)
        acts_a = (ctypes.c_int32 * n)(*[int(a) for a in activations])
        wp = (ctypes.c_void_p * n)(*[w.ctypes.data for w in ws])
        bp = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
        if stats is not None:
            st = [L.f32c(np.asarray(s).reshape(-1)) for s in stats]
            sp = (ctypes.c_void_p * 6)(*[s.ctypes.data for s in st])
            L.check(L.lib.bbmpc_set_mlp(self._h, n, dims_a, acts_a, wp, bp, 1, sp))
        else:
            L.check(L.lib.bbmpc_set_mlp(self._h, n, dims_a, acts_a, wp, bp, 0, None))

    def set_reward_source(self, hip_source):
        """HIP source defining `__device__ float bbmpc_user_reward(cur, act, nxt, S, U)` (include/bbmpc.h)."""
        L.check(L.lib.bbmpc_set_reward_source(self._h, hip_source.encode()))

    def set_dynamics_source(self, hip_source):
        """HIP source defining `__device__ void bbmpc_user_dynamics(x, delta, S, U)` (include/bbmpc.h)."""
        L.check(L.lib.bbmpc_set_dynamics_source(self._h, hip_source.encode()))

    def set_reward_callback(self, fn):
        """fn(d_cur, d_actions, d_next, batch, d_out, hip_stream) -> status, all device pointers (include/bbmpc.h
        bbmpc_rows_callback); `fn` is an _lib.ROWS_CALLBACK the caller keeps alive; None clears it."""
        self._rew_cb = fn
        L.check(L.lib.bbmpc_set_reward_callback(self._h, fn if fn is not None else L.ROWS_CALLBACK(), None))

    def set_dynamics_callback(self, fn):
        self._dyn_cb = fn
        L.check(L.lib.bbmpc_set_dynamics_callback(self._h, fn if fn is not None else L.ROWS_CALLBACK(), None))

    def mlp_forward(self, x):
        """DeterministicMLP.__call__ on the device: raw Dense stack on processed inputs [B, S+U] -> [B, S]."""
        x = L.f32c(x)
        if x.ndim != 2 or x.shape[1] != self.S + self.U:
            raise ValueError("x must be [B, dim_S + dim_U] = [B, %d], got %s" % (self.S + self.U, x.shape))
        out = np.empty((x.shape[0], self.S), np.float32)
        if x.shape[0]:
            L.check(L.lib.bbmpc_mlp_forward(self._h, L.ptr(x), x.shape[0], L.ptr(out)))
        return out

    def _process(self, fn, first, second, width_second, width_out, stats):
        first, second = L.f32c(first), L.f32c(second)
        b = first.shape[0] if first.ndim == 2 else -1
        if first.shape != (b, self.S) or second.shape != (b, width_second):
            raise ValueError("expected [B,%d] and [B,%d], got %s and %s" % (self.S, width_second, first.shape, second.shape))
        out = np.empty((b, width_out), np.float32)
        if b:
            if stats is not None:
                st = [L.f32c(np.asarray(v).reshape(-1)) for v in stats]
                sp = (ctypes.c_void_p * 6)(*[v.ctypes.data for v in st])
            else:
                sp = None
            L.check(fn(self._h, L.ptr(first), L.ptr(second), b, sp, L.ptr(out)))
        return out

    def process_input(self, states, actions, stats=None):
        return self._process(L.lib.bbmpc_process_input, states, actions, self.U, self.S + self.U, stats)

    def process_output(self, states, raw_output, stats=None):
        return self._process(L.lib.bbmpc_process_output, states, raw_output, self.S, self.S, stats)

    def reset(self):
        L.check(L.lib.bbmpc_reset(self._h))

    def synchronize(self):
        L.check(L.lib.bbmpc_synchronize(self._h))

    # -- multi-GPU: one all-gather of the per-agent records per control step (include/bbmpc.h) ------
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from rank 0 that every rank passes to comm_init (ship them over any side channel)."""
        buf = ctypes.create_string_buffer(L.COMM_ID_BYTES)
        L.check(L.lib.bbmpc_comm_unique_id(buf, L.COMM_ID_BYTES))
        return buf.raw

    def comm_init(self, unique_id, nranks, rank):
        if len(unique_id) != L.COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % L.COMM_ID_BYTES)
        L.check(L.lib.bbmpc_comm_init(self._h, ctypes.c_char_p(bytes(unique_id)), int(nranks), int(rank)))

    def comm_init_local(self, group_key, nranks, rank):
        """Rank `rank` of an in-process communicator shared by the handles that pass the same key (one device, no RCCL):
        bbmpc_comm_init_local.  See parallel.attach_local_comm."""
        L.check(L.lib.bbmpc_comm_init_local(self._h, ctypes.c_uint64(int(group_key)), int(nranks), int(rank)))

    def gather_records_dev(self, d_records, d_gathered, count_per_rank, slot):
        L.check(L.lib.bbmpc_gather_records_dev(self._h, ctypes.c_void_p(d_records), ctypes.c_void_p(d_gathered),
                                               int(count_per_rank), int(slot)))

    def optimize_gather_dev(self, d_state, d_records, d_gathered, slot, t=0, add_exploration_noise=False,
                            d_next_state=0):
        L.check(L.lib.bbmpc_optimize_gather_dev(self._h, ctypes.c_void_p(d_state), int(t),
                                                int(bool(add_exploration_noise)), ctypes.c_void_p(d_records),
                                                ctypes.c_void_p(d_next_state or 0), ctypes.c_void_p(d_gathered),
                                                int(slot)))

    def graph_stats(self):
        """Host-in / host-out calls served by replaying a captured hipGraph -- bbmpc_graph_stats."""
        a = ctypes.c_int64(0)
        L.check(L.lib.bbmpc_graph_stats(self._h, ctypes.byref(a)))
        return a.value

    def call_stats(self):
        """(calls served by the resident kernel of the previous call, calls that launched) -- bbmpc_call_stats."""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        L.check(L.lib.bbmpc_call_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def comm_info(self):
        """(nranks, rank, sync_mode) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        n, r, m = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        L.check(L.lib.bbmpc_comm_info(self._h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(m)))
        return n.value, r.value, m.value

    def optimize_gather(self, state, d_gathered, slot, t=0, add_exploration_noise=False):
        """`optimize` for this rank's agents (NumPy in / out) + the device all-gather of their records into
        d_gathered (device address of [num_agents_global, U+S+1] floats), overlapped with the next control step."""
        io = self._io_buffers()
        st, action, nxt, rew, p_st, p_act, p_nxt, p_rew = io
        state = np.asarray(state)
        if state.shape != (self.A, self.S):
            raise ValueError("state must be [num_agents, dim_S] = [%d, %d], got %s" % (self.A, self.S, state.shape))
        np.copyto(st, state, casting="unsafe")
        L.check(L.lib.bbmpc_optimize_gather(self._h, p_st, int(t), 1 if add_exploration_noise else 0, p_act, p_nxt,
                                            p_rew, ctypes.c_void_p(d_gathered), int(slot)))
        return action.copy(), nxt.copy(), rew.copy()

    def gather_wait(self, slot, host_block=False):
        L.check(L.lib.bbmpc_gather_wait(self._h, int(slot), int(bool(host_block))))

    def comm_destroy(self):
        L.check(L.lib.bbmpc_comm_destroy(self._h))

    # -- hot path --------------------------------------------------------------------------
    def optimize(self, state, t=0, add_exploration_noise=False):
        # persistent I/O buffers with cached ctypes pointers: building four `ndarray.ctypes` views per call costs more
        # host time (~15 us) than the H2D/D2H traffic of a control step
        st, action, nxt, rew, p_st, p_act, p_nxt, p_rew = self._io_buffers()
        state = np.asarray(state)
        if state.shape != (self.A, self.S):
            raise ValueError("state must be [num_agents, dim_S] = [%d, %d], got %s" % (self.A, self.S, state.shape))
        np.copyto(st, state, casting="unsafe")
        code = L.lib.bbmpc_optimize(self._h, p_st, int(t), 1 if add_exploration_noise else 0, p_act, p_nxt, p_rew)
        if code != 0:
            L.check(code)
        return action.copy(), nxt.copy(), rew.copy()

    def _io_buffers(self):
        io = self.__dict__.get("_io")
        if io is None:
            bufs = (np.empty((self.A, self.S), np.float32), np.empty((self.A, self.U), np.float32),
                    np.empty((self.A, self.S), np.float32), np.empty((self.A,), np.float32))
            io = self._io = bufs + tuple(L.ptr(b) for b in bufs)
        return io

    def rollout_episode(self, start_state, num_steps, add_exploration_noise=False):
        """T closed-loop control steps on the device (model = environment); returns
        (actions [T,A,U], next_states [T,A,S], rewards [T,A])."""
        s = L.f32c(start_state)
        if s.shape != (self.A, self.S):
            raise ValueError("start_state must be [%d, %d]" % (self.A, self.S))
        rec = np.empty((int(num_steps), self.A, self.U + self.S + 1), np.float32)
        L.check(L.lib.bbmpc_rollout_episode(self._h, L.ptr(s), int(num_steps), int(bool(add_exploration_noise)),
                                            L.ptr(rec)))
        return rec[..., :self.U], rec[..., self.U:self.U + self.S], rec[..., self.U + self.S]

    def optimize_dev(self, d_state, d_record, t=0, add_exploration_noise=False, d_next_state=0):
        L.check(L.lib.bbmpc_optimize_dev(self._h, ctypes.c_void_p(d_state), int(t), int(bool(add_exploration_noise)),
                                         ctypes.c_void_p(d_record), ctypes.c_void_p(d_next_state or 0)))

    def evaluate(self, state, action_sequences):
        state = L.f32c(state)
        seq = L.f32c(action_sequences)
        if state.shape != (self.A, self.S):
            raise ValueError("current_states must be [%d, %d], got %s" % (self.A, self.S, state.shape))
        if seq.ndim != 4 or seq.shape[1:] != (self.A, self.H, self.U):
            raise ValueError("action_sequences must be [n, %d, %d, %d], got %s" % (self.A, self.H, self.U, seq.shape))
        n = seq.shape[0]
        out = np.empty((n, self.A), np.float32)
        if n == 0:
            return out
        L.check(L.lib.bbmpc_evaluate(self._h, L.ptr(state), L.ptr(seq), n, L.ptr(out)))
        return out

    def evaluate_dev(self, d_state, d_seq, n_pop, d_rewards):
        L.check(L.lib.bbmpc_evaluate_dev(self._h, ctypes.c_void_p(d_state), ctypes.c_void_p(d_seq), int(n_pop),
                                         ctypes.c_void_p(d_rewards)))

    def predict_next_state(self, states, actions):
        states, actions = L.f32c(states), L.f32c(actions)
        b = states.shape[0]
        if states.shape != (b, self.S) or actions.shape != (b, self.U):
            raise ValueError("states [B,%d] / actions [B,%d] expected" % (self.S, self.U))
        out = np.empty((b, self.S), np.float32)
        if b:
            L.check(L.lib.bbmpc_predict_next_state(self._h, L.ptr(states), L.ptr(actions), b, L.ptr(out)))
        return out

    def evaluate_next_reward(self, states, next_states, actions):
        states, next_states, actions = L.f32c(states), L.f32c(next_states), L.f32c(actions)
        b = states.shape[0] if states.ndim == 2 else -1
        # the C side copies b*S / b*S / b*U floats from these buffers: a wrong shape must not become an out-of-bounds read
        if states.shape != (b, self.S) or next_states.shape != (b, self.S) or actions.shape != (b, self.U):
            raise ValueError("states / next_states [B,%d] and actions [B,%d] expected, got %s, %s, %s"
                             % (self.S, self.U, states.shape, next_states.shape, actions.shape))
        out = np.empty((b,), np.float32)
        if b:
            L.check(L.lib.bbmpc_evaluate_next_reward(self._h, L.ptr(states), L.ptr(next_states), L.ptr(actions), b,
                                                     L.ptr(out)))
        return out

    def step_dev(self, d_states, d_actions, action_stride, batch, d_next_states, d_rewards=0):
        L.check(L.lib.bbmpc_step_dev(self._h, ctypes.c_void_p(d_states), ctypes.c_void_p(d_actions),
                                     int(action_stride), int(batch), ctypes.c_void_p(d_next_states),
                                     ctypes.c_void_p(d_rewards or 0)))

    # -- parity hooks ----------------------------------------------------------------------
    def inject_noise(self, kind, data):
        if data is None:
            L.check(L.lib.bbmpc_inject_noise(self._h, int(kind), None, 0))
            return
        d = L.f32c(data)
        L.check(L.lib.bbmpc_inject_noise(self._h, int(kind), L.ptr(d), d.size))

    def dump_noise(self, kind, control_step, iteration, shape):
        out = np.empty(shape, np.float32)
        L.check(L.lib.bbmpc_dump_noise(self._h, int(kind), int(control_step), int(iteration), L.ptr(out), out.size))
        return out

    def set_trace(self, enabled=True):
        L.check(L.lib.bbmpc_set_trace(self._h, int(bool(enabled))))

    def get_trace(self, iteration, item):
        if item == L.TRACE_REWARDS:
            out = np.empty(((2 if self.cfg.optimizer == L.OPT_SPSA else 1) * self.N, self.A), np.float32)
        elif item in (L.TRACE_MEAN, L.TRACE_VAR):
            out = np.empty((self.A, self.H, self.U), np.float32)
        elif item == L.TRACE_ELITES:
            if self.cfg.optimizer == L.OPT_CEM:
                out = np.empty((self.A, self.k), np.int32)
            elif self.cfg.optimizer == L.OPT_CMAES:
                groups = self.A if (self.cfg.quirks & L.CMAES_PER_AGENT) else 1
                out = np.empty((groups, self.k), np.int32)
            else:
                out = np.empty((self.A,), np.int32)
        elif item == L.TRACE_SAMPLES:
            out = np.empty((self.N, self.A, self.H, self.U), np.float32)
        elif item in (L.TRACE_CMA_B, L.TRACE_CMA_C, L.TRACE_CMA_D):
            per_agent = bool(self.cfg.quirks & L.CMAES_PER_AGENT)
            groups = self.A if per_agent else 1
            n = self.H * self.U * (1 if per_agent else self.A)
            out = np.empty((groups, n) if item == L.TRACE_CMA_D else (groups, n, n), np.float32)
        elif item == L.TRACE_CMA_SVD_STATS:
            out = np.empty((self.A if (self.cfg.quirks & L.CMAES_PER_AGENT) else 1, 16), np.int32)
        else:
            raise ValueError(item)
        L.check(L.lib.bbmpc_get_trace(self._h, int(iteration), int(item), L.ptr(out), out.nbytes))
        return out

    def get_state(self, name, shape=None):
        out = np.empty(shape if shape is not None else (self.A, self.H, self.U), np.float32)
        L.check(L.lib.bbmpc_get_state(self._h, name.encode(), L.ptr(out), out.size))
        return out

    def set_state(self, name, data):
        d = L.f32c(data)
        L.check(L.lib.bbmpc_set_state(self._h, name.encode(), L.ptr(d), d.size))

    # -- measurement -----------------------------------------------------------------------
    def set_profiling(self, enabled=True, every=1):
        """HIP events around every `every`-th launch of the dominant kernel (see include/bbmpc.h)."""
        L.check(L.lib.bbmpc_set_profiling(self._h, (max(int(every), 1) if enabled else 0)))

    def get_profile(self):
        ms, n, name = ctypes.c_double(), ctypes.c_int64(), ctypes.c_char_p()
        L.check(L.lib.bbmpc_get_profile(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(name)))
        return ms.value, n.value, (name.value or b"").decode()

    def profile_instantiation(self):
        """The dominant kernel with its template arguments, as rocprofv3 names it minus "void " (include/bbmpc.h)."""
        name = ctypes.c_char_p()
        L.check(L.lib.bbmpc_profile_instantiation(self._h, ctypes.byref(name)))
        return (name.value or b"").decode()
