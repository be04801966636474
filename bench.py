#!/usr/bin/env python
"""Headline benchmark: MPC control steps / second (and candidate trajectories / second).

A "step" is ONE control step of the hot path (all optimizer iterations: sample -> rollout -> reduce -> refit, plus the
action / next-state tail) for every agent this process owns, closed loop: the engine's own model plays the environment
(SURVEY.md 8d), so warm starts are exercised.

`value` is the metric as SURVEY.md 8(d) / BASELINE.md define it: 1 / MEDIAN wall time of one `MPCPolicy.act(obs, t)`
-- NumPy observation in, all optimizer iterations on the GPU, NumPy action / next observation / reward out, the same
bracket as the reference's utils/rollouts.py:92-101 -- times the agents the job owns, with p10 / p90.  The
device-resident rate of the same closed loop (state never leaves HBM, launches never synchronised) is reported next
to it as `device_resident_control_steps_per_sec`, and the dominant kernel's roofline is measured on that loop with HIP
events on the launch stream.

Default workload = BASELINE.json configs[1]: Pendulum-v0 true dynamics, CEM, num_agents=1 per GPU, N=500, H=30,
5 iterations, k=50.  At one GPU the same JSON line carries a `secondary` block for the north star's second target --
HalfCheetah learned MLP (26-200-200-20), PI2, N=1000, H=30 -- with its own MFMA roofline and CPU baseline.

With --gpus N every rank owns its own agents (weak scaling, independent agents, RNG keyed by global agent id); the
only exchange the path has is one RCCL all-gather of the packed (action | next_state | reward) records per control
step, issued by the engine on its own communicator and stream and overlapped with the next control step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs; A = agents per GPU
    "cfg1": dict(env="pendulum", opt="RandomSearch", N=200, A=1, H=20, iters=1, k=0),
    "cfg2": dict(env="pendulum", opt="CEM", N=500, A=1, H=30, iters=5, k=50),
    "cfg3": dict(env="pendulum", opt="PI2", N=1000, A=8, H=30, iters=5, k=0),      # 64 agents over 8 GPUs
    "cfg3full": dict(env="pendulum", opt="PI2", N=1000, A=64, H=30, iters=5, k=0),  # all 64 agents on one GPU
    "cfg4": dict(env="cheetah", opt="CEM", N=1000, A=1, H=30, iters=5, k=50),
    # north_star target 2: HalfCheetah learned MLP, PI2, N=1000, H=30
    "cfg4pi2": dict(env="cheetah", opt="PI2", N=1000, A=1, H=30, iters=5, k=0),
    "cfg5cem": dict(env="cheetah", opt="CEM", N=2000, A=4, H=50, iters=5, k=50),     # config-5 shape per GPU, CEM
    "cfg5full": dict(env="cheetah", opt="CEM", N=2000, A=32, H=50, iters=5, k=50),   # all 32 agents on one GPU
    # the other optimizers at config 2's size (Pendulum, N=500, H=30, 5 iterations)
    "cfg2pso": dict(env="pendulum", opt="PSO", N=500, A=1, H=30, iters=5, k=0),
    "cfg2spsa": dict(env="pendulum", opt="SPSA", N=500, A=1, H=30, iters=5, k=0),
    "cfg2cma": dict(env="pendulum", opt="CMA-ES", N=500, A=1, H=30, iters=5, k=50),
    "cfg2pi2": dict(env="pendulum", opt="PI2", N=500, A=1, H=30, iters=5, k=0),
    # BASELINE config 5 proper (its two optimizers), one GPU's share of the 32 agents
    "cfg5pso": dict(env="cheetah", opt="PSO", N=2000, A=4, H=50, iters=5, k=0),
    "cfg5cma": dict(env="cheetah", opt="CMA-ES", N=2000, A=4, H=50, iters=5, k=50),  # per-agent CMA-ES (n = 300 each)
    # the one learned-model configuration the reference itself pins (tutorials/mujoco/tutorial_two.py:23-33,52-53):
    # DeterministicMLP 26-500-500-500-20 tanh x3 + linear, RandomSearch, population 4048, planning horizon 15, one agent
    "cfg_tut2": dict(env="cheetah", opt="RandomSearch", N=4048, A=1, H=15, iters=1, k=0, dims=[26, 500, 500, 500, 20]),
}
DEFAULT_STEPS = {"cfg1": 2000, "cfg2": 2000, "cfg3": 2000, "cfg3full": 300, "cfg4": 400, "cfg4pi2": 400, "cfg5cem": 60,
                 "cfg5pso": 60, "cfg5full": 10, "cfg5cma": 30, "cfg_tut2": 100}
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3
VALU_ISSUE_NS_MEASURED = 1.07       # one VALU instruction per SIMD, measured on one CU (tools/microbench/pk_fp32.hip)
VALU_ISSUE_CYCLES_GUIDE = 2         # v_fma_f32, wave64 on a SIMD-32 (MI355X_MICROARCH.md)
CLOCK_GHZ = 2.4
MLP_DIMS = [26, 200, 200, 20]


def mlp_dims(c):
    return list(c.get("dims") or MLP_DIMS)


def mlp_acts(c):
    return ["tanh"] * (len(mlp_dims(c)) - 2) + [None]
SECONDARY = "cfg4pi2"


def _pct(xs, q):
    return float(np.percentile(np.asarray(xs, np.float64), q))


# ------------------------------------------------------------------------------------------------------------------
# the workload: a rank's agents behind the reference's own API (MPCPolicy), plus the device-resident loop
# ------------------------------------------------------------------------------------------------------------------
class Workload:
    """One rank's share of a configuration on the GPU.  `act_step` = what the metric brackets (MPCPolicy.act);
    `dev_step` = the same control step with the state resident in HBM."""

    def __init__(self, name, rank, world, local, dev, use_dist, backend, gather_mode, agents=None):
        import torch
        import torch.distributed as dist
        from blackbox_mpc_amd import _lib as L
        from blackbox_mpc_amd.dynamics_functions import DeterministicMLP
        from blackbox_mpc_amd.dynamics_handlers import SystemDynamicsHandler
        from blackbox_mpc_amd.policies import MPCPolicy
        from blackbox_mpc_amd.spaces import Box
        from blackbox_mpc_amd.utils import synthetic as SY
        self.torch, self.dist, self.L = torch, dist, L
        self.name, self.rank, self.world, self.dev, self.use_dist, self.backend = name, rank, world, dev, use_dist, backend
        c = self.c = dict(CONFIGS[name])
        if agents is not None:
            c["A"] = int(agents)                 # agents this rank owns (config 3 as BASELINE states it: 64 in total)
        N, A, H, iters, k = c["N"], c["A"], c["H"], c["iters"], c["k"]
        self.mlp = c["env"] == "cheetah"
        quirks = L.CMAES_PER_AGENT if c["opt"] == "CMA-ES" else 0   # the shardable CMA-ES mode
        quirks |= int(os.environ.get("BENCH_EXTRA_QUIRKS", "0"))       # (value_strict_math: the same calls with BBMPC_STRICT_MATH)
        opt_args = dict(planning_horizon=H, population_size=N, seed=0, quirks=quirks, agent_offset=rank * A,
                        num_agents_global=world * A, device=local)
        if c["opt"] != "RandomSearch":
            opt_args["max_iterations"] = iters
        if k:
            opt_args["num_elite"] = k
        if self.mlp:
            from blackbox_mpc_amd.utils.cheetah import reward_function
            self.U, self.S = 6, 20
            act_space, obs_space = Box([-1.0] * self.U, [1.0] * self.U), Box([-10.0] * self.S, [10.0] * self.S)
            net = DeterministicMLP(layers=mlp_dims(c), activation_functions=mlp_acts(c), seed=1)
            net.set_weights(*SY.make_mlp_params(mlp_dims(c), seed=42))      # Glorot-uniform / zero bias, last layer x0.1
            handler = SystemDynamicsHandler(act_space, obs_space, dynamics_function=net, true_model=False,
                                            is_normalized=True)
            handler.set_normalization_stats(*SY.cheetah_stats(self.S, self.U))
            self.policy = MPCPolicy(reward_function=reward_function, env_action_space=act_space,
                                    env_observation_space=obs_space, dynamics_handler=handler,
                                    optimizer_name=c["opt"], num_agents=A, **opt_args)
            self.start = SY.cheetah_start_states(A, self.S, agent_offset=rank * A)
        else:
            from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function
            self.U, self.S = 1, 3
            self.policy = MPCPolicy(reward_function=pendulum_reward_function, env_action_space=Box([-2.0], [2.0]),
                                    env_observation_space=Box([-1, -1, -8], [1, 1, 8]), true_model=True,
                                    dynamics_function=PendulumTrueModel(), optimizer_name=c["opt"], num_agents=A,
                                    **opt_args)
            self.start = SY.pendulum_start_states(A, agent_offset=rank * A)
        self.eng = self.policy._optimizer._require_engine()
        self.policy.reset()              # episode start, as utils/rollouts.py:_sample does (PSO draws its swarm here)
        self.rec = self.U + self.S + 1
        self.A = A
        # ---- the one exchange: all-gather of the records, on the engine's own RCCL communicator + stream -------------
        self.gather_mode = "none"
        self.fallback_reason = None
        self.native = False
        self.comm_info = None
        if use_dist:
            self.gather_mode = gather_mode
            if backend == "nccl" and gather_mode == "native":
                from blackbox_mpc_amd.parallel import attach_record_comm
                ok = torch.ones(1, device=dev)
                why = ""
                try:
                    attach_record_comm(self.eng, device=dev)
                    self.comm_info = self.eng.comm_info()
                    if self.comm_info[0] != world:
                        # a communicator of the wrong size is not something to fall back from: the number would not be an
                        # N-GPU number.  Loud, on every rank.
                        raise SystemExit("bench.py[rank %d]: the engine's RCCL communicator has ncclCommCount = %d, the job has %d ranks"
                                         % (rank, self.comm_info[0], world))
                except Exception as ex:                       # e.g. librccl not loadable
                    why = "%s: %s" % (type(ex).__name__, ex)
                    print("bench[rank %d]: engine-owned record gather unavailable (%s)" % (rank, why), file=sys.stderr)
                    ok.zero_()
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if ok.item() == 0:
                    # every rank agrees to use torch.distributed's collective instead -- reported, never silent
                    self.gather_mode, self.fallback_reason = "async", (why or "another rank could not create its communicator")
                    try:
                        self.eng.comm_destroy()
                    except Exception:
                        pass
                else:
                    self.native = True
            elif backend != "nccl":
                self.gather_mode = "host(%s)" % backend
        # The engine launches on its own (non-blocking) stream.  Only the torch.distributed gather modes put torch work
        # into the loop (events, collectives): they run the loop on a torch side stream that the engine is told to use.
        self.torch_loop = (use_dist and not self.native) or bool(os.environ.get("BBMPC_BENCH_TORCH_STREAM"))
        if self.torch_loop:
            self.stream = torch.cuda.Stream(device=dev)
            torch.cuda.set_stream(self.stream)
            self.eng.set_torch_stream(self.stream)
        else:
            self.stream = None
            self.eng.set_stream(None)
        self.state = torch.from_numpy(self.start).to(dev)
        self.nxt = torch.empty_like(self.state)
        # records / gathered results are double buffered: the all-gather of control step t runs on its own stream
        # while step t+1 computes (the local optimizer never needs the other ranks' records)
        self.records = [torch.zeros((A, self.rec), device=dev, dtype=torch.float32) for _ in range(2)]
        self.gathered = [torch.zeros((world * A, self.rec), device=dev, dtype=torch.float32) for _ in range(2)] if use_dist else None
        self.comm_stream = torch.cuda.Stream(device=dev) if (use_dist and self.torch_loop) else None
        self.comm_done = [None, None]
        self.ready_ev = [torch.cuda.Event(), torch.cuda.Event()] if self.comm_stream is not None else None
        self.done_ev = [torch.cuda.Event(), torch.cuda.Event()] if self.comm_stream is not None else None
        self.works = [None, None]
        self.tick = 0
        self.obs = self.start.copy()
        self.last_host_record = [None, None]

    # -- what `value` is measured on: MPCPolicy.act, NumPy in / NumPy out ---------------------------------------------
    def act_step(self, t):
        b = self.tick & 1
        self.tick += 1
        if self.native:
            # this rank's act + the all-gather of its records enqueued behind it on the communication stream
            self.eng.gather_wait(b)
            a, n, r = self.eng.optimize_gather(self.obs, self.gathered[b].data_ptr(), b, t)
        else:
            a, n, r = self.policy.act(self.obs, t)
            if self.use_dist:
                self._torch_gather_host(b, a, n, r)
        self.obs = n                      # closed loop: the model is the environment
        self.last_host_record[b] = np.concatenate([a, n, np.asarray(r, np.float32).reshape(-1, 1)], axis=1)

    def _torch_gather_host(self, b, a, n, r):
        torch, dist = self.torch, self.dist
        rec = torch.from_numpy(np.concatenate([a, n, np.asarray(r, np.float32).reshape(-1, 1)], axis=1).astype(np.float32))
        if self.backend == "nccl":
            if self.works[b] is not None:
                self.works[b].wait()
            self.records[b].copy_(rec, non_blocking=True)
            self.works[b] = dist.all_gather_into_tensor(self.gathered[b], self.records[b], async_op=True)
        else:
            out = torch.empty((self.world * self.A, self.rec), dtype=torch.float32)
            dist.all_gather_into_tensor(out, rec)
            self.gathered[b].copy_(out)
            self.records[b].copy_(rec)

    # -- the same control step with the state resident in HBM ---------------------------------------------------------
    def dev_step(self):
        torch, dist, eng = self.torch, self.dist, self.eng
        b = self.tick & 1
        self.tick += 1
        record = self.records[b]
        if self.native:
            eng.gather_wait(b)
            eng.optimize_gather_dev(self.state.data_ptr(), record.data_ptr(), self.gathered[b].data_ptr(), b,
                                    d_next_state=self.nxt.data_ptr())
            self.state, self.nxt = self.nxt, self.state
            return
        if self.works[b] is not None:
            self.works[b].wait()                     # stream-level: the gather that last read this buffer has finished
            self.works[b] = None
        if self.comm_done[b] is not None:
            self.stream.wait_event(self.comm_done[b])
        eng.optimize_dev(self.state.data_ptr(), record.data_ptr(), d_next_state=self.nxt.data_ptr())
        if self.use_dist:
            if self.backend == "nccl" and self.gather_mode == "async":
                self.works[b] = dist.all_gather_into_tensor(self.gathered[b], record, async_op=True)
            else:
                self.ready_ev[b].record(self.stream)
                self.comm_stream.wait_event(self.ready_ev[b])
                with torch.cuda.stream(self.comm_stream):
                    if self.backend == "nccl":
                        dist.all_gather_into_tensor(self.gathered[b], record)
                    else:
                        out = torch.empty((self.world * self.A, self.rec), dtype=torch.float32)
                        dist.all_gather_into_tensor(out, record.cpu())
                        self.gathered[b].copy_(out)
                    self.done_ev[b].record(self.comm_stream)
                    self.comm_done[b] = self.done_ev[b]
        self.state, self.nxt = self.nxt, self.state

    def fence(self):
        for b in range(2):
            if self.works[b] is not None:
                self.works[b].wait()
                self.works[b] = None
        if self.native:
            for b in range(2):
                self.eng.gather_wait(b, host_block=True)
        self.eng.synchronize()                       # launch stream + the engine's communication stream
        self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def set_profiling(self, on, every=1):
        self.eng.set_profiling(on, every=every)

    def get_profile(self):
        return self.eng.get_profile()

    def profile_instantiation(self):
        return self.eng.profile_instantiation()

    def check_gather(self, host_records):
        """The gathered rows of this rank's own agents must be the records it produced (bit for bit), and the row
        blocks of the other ranks must be populated.  Returns the number of rows checked."""
        if not self.use_dist:
            return 0
        torch = self.torch
        rows = 0
        for b in range(2):
            mine = self.gathered[b][self.rank * self.A:(self.rank + 1) * self.A]
            ref = torch.from_numpy(host_records[b]).to(mine.device) if host_records and host_records[b] is not None else self.records[b]
            if not torch.equal(mine.view(torch.int32), ref.contiguous().view(torch.int32)):
                raise RuntimeError("all-gather returned different records for the local agents: %r vs %r" % (mine, ref))
            if not bool(torch.isfinite(self.gathered[b]).all()):
                raise RuntimeError("gathered records contain non-finite rows")
            rows += int(self.gathered[b].shape[0])
        return rows

    def call_stats(self):
        return self.eng.call_stats()

    def close(self):
        if self.native:
            self.eng.synchronize()
            self.eng.comm_destroy()


class StubWorkload:
    """BBMPC_BENCH_STUB=1: the same interface without a GPU, so that bench.py's rank logic (environment variables,
    process group, barriers, max-over-ranks timing, the gathered-rows check, the JSON line) runs under the gloo
    backend in the CPU test suite (tests/test_bench_ranks_cpu.py).  It computes nothing worth timing and its numbers
    mean nothing; it is refused outside BBMPC_BENCH_BACKEND=gloo."""

    def __init__(self, name, rank, world, local, dev, use_dist, backend, gather_mode, agents=None):
        import torch
        import torch.distributed as dist
        assert backend == "gloo", "the stub workload only exists for the gloo rank-logic test"
        self.torch, self.dist = torch, dist
        c = self.c = dict(CONFIGS[name])
        if agents is not None:
            c["A"] = int(agents)
        self.name, self.rank, self.world, self.use_dist, self.backend = name, rank, world, use_dist, backend
        self.mlp = c["env"] == "cheetah"
        self.U, self.S = (6, 20) if self.mlp else (1, 3)
        self.A, self.rec = c["A"], self.U + self.S + 1
        self.gather_mode, self.fallback_reason, self.native, self.comm_info = "host(gloo)", None, False, None
        self.records = [torch.zeros((self.A, self.rec)) for _ in range(2)]
        self.gathered = [torch.zeros((world * self.A, self.rec)) for _ in range(2)] if use_dist else None
        self.tick = 0
        self.last_host_record = [None, None]
        self.obs = np.full((self.A, self.S), 0.01 * (rank + 1), np.float32)

    def _step(self):
        b = self.tick & 1
        self.tick += 1
        ga = self.rank * self.A + np.arange(self.A, dtype=np.float32)
        rec = np.zeros((self.A, self.rec), np.float32)
        rec[:, 0] = ga + 0.001 * self.tick                          # "action" keyed by global agent id and step
        rec[:, self.U:self.U + self.S] = self.obs * 0.5
        rec[:, -1] = -ga
        self.obs = rec[:, self.U:self.U + self.S].copy()
        self.records[b] = self.torch.from_numpy(rec)
        self.last_host_record[b] = rec
        if self.use_dist:
            self.dist.all_gather_into_tensor(self.gathered[b], self.records[b])

    def act_step(self, t):
        self._step()

    def dev_step(self):
        self._step()

    def fence(self):
        if self.use_dist:
            self.dist.barrier()

    def set_profiling(self, on, every=1):
        pass

    def get_profile(self):
        return 0.0, 0, "stub"

    def profile_instantiation(self):
        return "stub"

    check_gather = Workload.check_gather

    def call_stats(self):
        return 0, self.tick

    def close(self):
        pass


# ------------------------------------------------------------------------------------------------------------------
def measure(W, steps, warmup, dist, use_dist, red_dev, act_only=False):
    """Times `steps` control steps twice: through MPCPolicy.act (the metric) and device-resident (the roofline).
    Both regions are bracketed by barrier + synchronize on both sides; elapsed = MAX over ranks.
    act_only: just the MPCPolicy.act region (the launch-per-call repeat of the headline)."""
    torch = W.torch

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- (1) the metric: wall time of MPCPolicy.act, host in / host out ------------------------------------------
    for t in range(warmup):
        W.act_step(t)
    W.fence()
    # exactly `steps` calls inside the barrier-to-barrier bracket (ms_per_step); the percentiles want >= 30 samples, so a
    # shorter run appends the missing calls AFTER the bracket and only their per-call wall times join the sample
    n_samples = max(steps, 30)
    walls = np.empty(n_samples, np.float64)
    t0 = time.perf_counter()
    for i in range(steps):
        s0 = time.perf_counter()
        W.act_step(warmup + i)
        walls[i] = time.perf_counter() - s0
    W.fence()
    t1 = time.perf_counter()
    for i in range(steps, n_samples):
        s0 = time.perf_counter()
        W.act_step(warmup + i)
        walls[i] = time.perf_counter() - s0
    W.fence()
    host_records = list(W.last_host_record)
    rows_act = W.check_gather(host_records)
    act_elapsed = reduce_max(t1 - t0)
    med, p10, p90 = (reduce_max(_pct(walls, q)) for q in (50, 10, 90))
    if act_only:
        return dict(walls=walls, act_elapsed=act_elapsed, median=med, p10=p10, p90=p90, n_samples=n_samples,
                    gather_rows_checked=rows_act)

    # ---- (2) device-resident closed loop + HIP events around the dominant kernel ----------------------------------
    for _ in range(max(2, min(warmup, 20))):
        W.dev_step()
    W.fence()
    # an event pair costs a few microseconds of stream time (a sixth of a config-2 control step): every launch is
    # bracketed when the run is short, every 8th otherwise; the un-instrumented repeat below gives the rate
    every = 1 if steps <= 64 else 8
    W.set_profiling(True, every=every)
    W.get_profile()
    t2 = time.perf_counter()
    for _ in range(steps):
        W.dev_step()
    W.fence()
    t3 = time.perf_counter()
    roll_ms, roll_n, kname = W.get_profile()
    kinst = W.profile_instantiation()          # the template instantiation those events bracketed
    W.set_profiling(False)
    W.fence()
    t4 = time.perf_counter()
    for _ in range(steps):
        W.dev_step()
    W.fence()
    t5 = time.perf_counter()
    rows_dev = W.check_gather(None)
    dev_elapsed = reduce_max(t5 - t4)
    return dict(walls=walls, act_elapsed=act_elapsed, median=med, p10=p10, p90=p90, dev_elapsed=dev_elapsed,
                dev_elapsed_instrumented=reduce_max(t3 - t2), roll_ms=roll_ms, roll_n=roll_n, kname=kname, kinst=kinst,
                prof_every=every, gather_rows_checked=rows_act + rows_dev, n_samples=n_samples)


SHAPE_KEYS = ("env", "opt", "N", "A", "H", "iters")


def profile_config_for(c):
    """Name of the bench configuration whose SHAPE equals c's (the key the committed profiles are stored under), or None."""
    for nm, cc in CONFIGS.items():
        if all(cc.get(k) == c.get(k) for k in SHAPE_KEYS):
            return nm
    return None


def profile_shape_matches(profile_json, prof_name, c):
    """The profile file records the shape it was taken at ("_shapes", tools/summarize_profiles.py); older files are taken
    at CONFIGS[prof_name]."""
    shp = profile_json.get("_shapes", {}).get(prof_name) or {k: CONFIGS[prof_name].get(k) for k in SHAPE_KEYS}
    return all(shp.get(k) == c.get(k) for k in SHAPE_KEYS)


def _plain_name(kn):
    """rocprofv3's kernel name without the leading "void " and without template arguments."""
    kn = kn.strip()
    if kn.startswith("void "):
        kn = kn[5:]
    return kn.split("<", 1)[0].strip()


def profile_entry(entries, kname, inst):
    """(profile key, value) of the kernel a measurement belongs to, or None.  `inst` is what the engine reports
    (bbmpc_profile_instantiation): with template arguments it must equal the profile's kernel name exactly (the profile may
    hold several instantiations of one kernel -- strict-math, resident -- whose counters differ by 2-8x); a plain name is
    accepted only when the profile holds exactly ONE instantiation of that kernel."""
    norm = lambda x: x.strip()[5:].strip() if x.strip().startswith("void ") else x.strip()
    if "<" in inst:
        hit = [(kn, v) for kn, v in entries.items() if norm(kn) == inst]
    else:
        hit = [(kn, v) for kn, v in entries.items() if _plain_name(kn) == _plain_name(kname)]
    return hit[0] if len(hit) == 1 else None


def roofline(W, m, name, world):
    c = W.c
    N, A, H, iters = c["N"], c["A"], c["H"], c["iters"]
    U, S, mlp, kname = W.U, W.S, W.mlp, m["kname"]
    bytes_per_traj = 4.0 * (H * U + 1) + 4.0 * S / N        # SURVEY 8d / BASELINE.md
    avg_ms = m["roll_ms"] / max(m["roll_n"], 1)
    fused = kname.startswith("k_fused")
    # trajectories one launch of the dominant kernel processes: the persistent kernel runs every iteration of every
    # local agent in one launch, the per-iteration kernels one iteration (SPSA: two launches of N per iteration)
    launch_traj = N * A * (iters if fused else 1)
    if mlp:
        dims = mlp_dims(c)
        flops_per_traj = H * 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
        achieved = launch_traj * flops_per_traj / (avg_ms * 1e-3) / 1e12 if m["roll_n"] else None
        roof = {"bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (achieved / MFMA_F32_PEAK_TFLOPS) if achieved else None, "traffic": None,
                "algorithmic_flops_per_launch": launch_traj * flops_per_traj,
                "note": "fp32-in/fp32-acc MFMA (%s); peak = dense fp32 matrix rate"
                        % ("v_mfma_f32_4x4x1_16b_f32" if "_q4" in kname else "v_mfma_f32_16x16x4_f32")}
    else:
        achieved = launch_traj * bytes_per_traj / (avg_ms * 1e-3) / 1e9 if m["roll_n"] else None
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": None,
                "algorithmic_bytes_per_launch": launch_traj * bytes_per_traj,
                "note": "the fused kernel keeps the H-step recurrence in registers/LDS: the path is VALU-issue "
                        "bound, the HBM fraction is nominal (DESIGN.md)"}
    roof.update({"kernel": kname, "avg_launch_us": avg_ms * 1e3, "launches": m["roll_n"],
                 "launches_note": "HIP-event pairs on the launch stream around %s launch of the kernel in the "
                                  "device-resident timed region" % ("every" if m["prof_every"] == 1 else "every %dth" % m["prof_every"])})
    # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    # collected separately, (2*FETCH + WRITE)*1024 -- tools/profile_round.sh, profiles/*_hbm_traffic.json).  Counters are
    # attached only when the PROFILED shape is the timed one -- the lookup goes by (environment, optimizer, N, A, H,
    # iterations), not by the configuration's name (run_block times "cfg3" with other agent counts) -- and only from the
    # profile entry of the template INSTANTIATION the engine says it timed (bbmpc_profile_instantiation): the strict-math and
    # the resident instantiations of the persistent kernel sit in the same profile files under the same plain name.
    import glob
    inst = m.get("kinst") or kname
    roof["kernel_instantiation"] = inst
    prof_name = profile_config_for(c)
    if prof_name is None:
        roof["counters_note"] = "no committed profile of this shape (agents=%d): traffic / valu_issue not attached" % A
    try:
        tr = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))[-1]))
        ent = profile_entry(tr.get(prof_name, {}), kname, inst) if prof_name else None
        if ent is not None and world == 1 and profile_shape_matches(tr, prof_name, c):
            roof["traffic"] = ent[1]
            roof["traffic_source"] = ("profiles/ (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, separate passes; configuration %s, "
                                      "kernel %s)" % (prof_name, ent[0]))
    except Exception:
        pass
    # A bound that means something for the persistent pendulum kernels (their HBM fraction is nominal): VALU issue.
    # Wave-instructions per launch come from the committed rocprofv3 PMC pass (SQ_INSTS_VALU); the kernel occupies
    # one CU per agent.  Two peaks: the rate one CU was MEASURED to issue at (one VALU instruction per SIMD per 1.07 ns,
    # tools/microbench/pk_fp32.hip) and the guide's figure (v_fma_f32: 2 cycles per wave64 instruction per SIMD at 2.4 GHz).
    try:
        if not mlp and m["roll_n"] and world == 1 and prof_name:
            sqc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")))[-1]))
            ent = profile_entry(sqc.get(prof_name, {}), kname, inst)
            if ent is not None and profile_shape_matches(sqc, prof_name, c):
                insts = ent[1]["SQ_INSTS_VALU"]
                peak = A * 4 / VALU_ISSUE_NS_MEASURED * 1e9
                peak_guide = A * 4 * CLOCK_GHZ * 1e9 / VALU_ISSUE_CYCLES_GUIDE
                ach = insts / (avg_ms * 1e-3)
                roof["valu_issue"] = {"achieved": ach, "peak": peak, "unit": "wave-instructions/s", "frac": ach / peak,
                                      "peak_guide": peak_guide, "frac_guide": ach / peak_guide,
                                      "insts_per_launch": insts, "cus": A, "kernel": ent[0],
                                      "peaks": "peak = 4 SIMDs x 1 instruction / %.2f ns measured on one CU; peak_guide = 4 SIMDs x %.1f GHz "
                                               "/ %d cycles per wave64 v_fma_f32 (MI355X_MICROARCH.md)"
                                               % (VALU_ISSUE_NS_MEASURED, CLOCK_GHZ, VALU_ISSUE_CYCLES_GUIDE),
                                      "source": "profiles/*_sq_counters.json (rocprofv3 --pmc SQ_INSTS_VALU; configuration %s)" % prof_name}
    except Exception:
        pass
    return roof


def describe(name, W, world):
    c = W.c
    return ("BASELINE %s: %s, %s, num_agents=%d/GPU, N=%d, H=%d, %d iters%s, closed loop (the model is the environment)"
            % (name, ("HalfCheetah(mod) S=20 U=6 learned MLP %s dynamics" % "-".join(map(str, mlp_dims(c)))) if W.mlp else "Pendulum-v0 true dynamics",
               c["opt"], c["A"], c["N"], c["H"], c["iters"], (", k=%d" % c["k"]) if c["k"] else ""))


def multi_gpu_block(W, m, world):
    """Evidence that the exchange ran: communicator size, hand-off mode, gathered rows checked."""
    return {
        "ranks": world,
        "rccl_ranks": W.comm_info[0] if W.comm_info else None,             # ncclCommCount of the engine's communicator
        "rccl_rank0": W.comm_info[1] if W.comm_info else None,
        "gather_mode": ("engine-owned RCCL communicator + stream (%s hand-off)"
                        % ("signal-memory sequence number" if W.comm_info and W.comm_info[2] == 1 else "event"))
                       if W.native else "torch.distributed " + W.gather_mode,
        "fallback_reason": W.fallback_reason,
        "gathered_rows_checked": m["gather_rows_checked"],
        "gathered_rows_check": "rows of the local agents bit-equal to the records produced, all rows finite",
    }


def result_block(name, W, m, world, steps, warmup):
    c = W.c
    total_agents = world * c["A"]
    value = total_agents / m["median"]
    dev_rate = steps * total_agents / m["dev_elapsed"]
    traj = c["N"] * c["iters"] * (2 if c["opt"] == "SPSA" else 1)
    return {
        "metric": "MPC control-steps/sec (agent-control-steps; %s, %s N=%d H=%d)"
                  % (("HalfCheetah learned MLP %s" % "-".join(map(str, mlp_dims(c)))) if W.mlp else "Pendulum true model", c["opt"], c["N"], c["H"]),
        "value": value,
        "unit": "control-steps/s",
        "value_definition": ("agents / median wall time of MPCPolicy.act(obs, t): NumPy observation in, all optimizer "
                             "iterations on the GPU, NumPy action / next observation / reward out (rollouts.py:92-101 bracket)")
                            if not getattr(W, "native", False) else
                            ("all ranks' agents / MAX over ranks of the median wall time of Engine.optimize_gather(obs, t) -- "
                             "the call MPCPolicy.act makes (NumPy observation in, all optimizer iterations, NumPy action / next "
                             "observation / reward out) with the rank's records handed to the engine's RCCL all-gather; the "
                             "collective of call t is enqueued by call t+1 and overlaps it, so the per-call median does not "
                             "contain the collective's latency: ms_per_step (barrier to barrier, every gather completed) is "
                             "the figure that does"),
        "median_ms": m["median"] * 1e3, "p10_ms": m["p10"] * 1e3, "p90_ms": m["p90"] * 1e3,
        "percentile_samples": m["n_samples"],
        "steps": steps, "warmup": warmup,
        "ms_per_step": m["act_elapsed"] / steps * 1e3,
        "candidate_trajectories_per_sec": value * traj,
        "dyn_steps_per_sec": value * traj * c["H"],
        "device_resident_control_steps_per_sec": dev_rate,
        "device_resident_ms_per_step": m["dev_elapsed"] / steps * 1e3,
        "device_resident_ms_per_step_with_events": m["dev_elapsed_instrumented"] / steps * 1e3,
        "roofline": roofline(W, m, name, world),
    }


def _self_launch(n):
    """`python bench.py --gpus N` by itself: re-run this script under torch.distributed.run with N ranks on this node
    (rendezvous on 127.0.0.1, a free port) and return its exit code; rank 0's JSON line goes to our stdout."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, BBMPC_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed control steps (default: per config, ~0.1-2 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the MLP (north-star target 2) block at --gpus 1")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = DEFAULT_STEPS.get(args.config, 200)
    if args.warmup is None:
        args.warmup = max(5, args.steps // 20)

    # --gpus N started without a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves, one per GPU,
    # the way the driver's launch line does, and hand its exit code back.  Under a launcher (torch.distributed.run sets
    # WORLD_SIZE) this process IS one of the ranks.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); start it as `python bench.py "
                         "--gpus N` or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # BBMPC_BENCH_BACKEND=gloo + BBMPC_BENCH_ONE_DEVICE=1: rank-logic smoke test on a 1-GPU box (all ranks share
    # device 0, records gathered through host memory); + BBMPC_BENCH_STUB=1: the same without any GPU (CPU test suite);
    # the real runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("BBMPC_BENCH_BACKEND", "nccl")
    stub = bool(os.environ.get("BBMPC_BENCH_STUB"))
    if stub and backend != "gloo":
        raise SystemExit("BBMPC_BENCH_STUB is only valid with BBMPC_BENCH_BACKEND=gloo (rank-logic test)")
    if os.environ.get("BBMPC_BENCH_ONE_DEVICE"):
        local = 0
    dev = None
    if not stub:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    # BBMPC_BENCH_FORCE_DIST=1: run the per-step all-gather path in a one-rank group (what it costs on a 1-GPU box)
    use_dist = world > 1 or bool(os.environ.get("BBMPC_BENCH_FORCE_DIST"))
    # how the per-step record all-gather is issued under the nccl backend: "native" = the engine's own RCCL
    # communicator and stream; "async" / "events" = torch.distributed, for comparison
    gather_mode = os.environ.get("BBMPC_BENCH_GATHER", "native")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if not stub:
        from blackbox_mpc_amd import _build
        if local == 0 or os.environ.get("BBMPC_BENCH_ONE_DEVICE"):
            if rank == 0 or not os.environ.get("BBMPC_BENCH_ONE_DEVICE"):
                _build.build()                   # one builder per node; the others wait for it
    if use_dist:
        dist.barrier()
    red_dev = dev if (backend == "nccl" and not stub) else "cpu"
    cls = StubWorkload if stub else Workload

    launch = "bench.py --gpus N started its own ranks (torch.distributed.run)" if os.environ.get("BBMPC_BENCH_SELF_LAUNCHED") \
        else ("external launcher (torch.distributed.run)" if "WORLD_SIZE" in os.environ else "single process")

    def run_block(name, steps, warmup, agents=None, scaling="weak", launch_per_call=True):
        """One configuration on all ranks: the MPCPolicy.act region, the device-resident region, and (launch_per_call)
        the act region once more on a second handle created with BBMPC_LINGER_US=0.  Returns rank 0's block."""
        W = cls(name, rank, world, local, dev, use_dist, backend, gather_mode, agents=agents)
        m = measure(W, steps, warmup, dist, use_dist, red_dev)
        blk = None
        if rank == 0:
            blk = result_block(name, W, m, world, steps, warmup)
            blk.update({
                "n_gpus": world, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
                "data": "synthetic", "launch": launch,
                "config": {"workload": describe(name, W, world),
                           "parallelism": ("agents sharded %d/GPU, no data-path collective; 1 all-gather of [A,%d] records "
                                           "per control step" % (W.A, W.rec)) if use_dist else "single GPU"},
            })
            if use_dist:
                blk["multi_gpu"] = multi_gpu_block(W, m, world)
        served, launched = W.call_stats()              # bbmpc_call_stats: how this handle's act() calls were served
        resident = served > launched
        W.close()
        del W
        # the host path both ways: `value` is measured with the engine's default -- the control-step
        # kernel of a host-in / host-out call stays resident for BBMPC_LINGER_US and serves the next call from a mailbox
        # when the caller comes back in time, which this loop does; value_launch_per_call is the same >= 30 calls on a
        # handle created with BBMPC_LINGER_US=0 (one launch per call, whatever the caller's cadence)
        if launch_per_call:
            saved = os.environ.get("BBMPC_LINGER_US")
            os.environ["BBMPC_LINGER_US"] = "0"
            try:
                Wl = cls(name, rank, world, local, dev, use_dist, backend, gather_mode, agents=agents)
                ml = measure(Wl, min(steps, 200), min(warmup, 10), dist, use_dist, red_dev, act_only=True)
                Wl.close()
                del Wl
            finally:
                if saved is None:
                    os.environ.pop("BBMPC_LINGER_US", None)
                else:
                    os.environ["BBMPC_LINGER_US"] = saved
            if rank == 0:
                c = blk_cfg = CONFIGS[name]
                A_tot = world * (agents if agents is not None else c["A"])
                blk["resident"] = resident
                blk["calls_served_by_resident_kernel"], blk["calls_launched"] = served, launched
                blk["resident_note"] = ("value: the control-step kernel of a call stays resident and takes the next call from a "
                                        "request mailbox (BBMPC_LINGER_US=%s, bbmpc_call_stats above); value_launch_per_call: "
                                        "the same calls with BBMPC_LINGER_US=0" % (saved or "200")) if resident else \
                                       "this configuration launches per call either way (no resident kernel on this path)"
                blk["value_launch_per_call"] = A_tot / ml["median"]
                blk["launch_per_call_median_ms"] = ml["median"] * 1e3
                blk["launch_per_call_p10_ms"], blk["launch_per_call_p90_ms"] = ml["p10"] * 1e3, ml["p90"] * 1e3
            # the same calls with the pendulum model evaluated op for op as the reference writes it (BBMPC_STRICT_MATH:
            # atan2 / sin / cos every step instead of the carried angle): what the default's
            # reformulation buys, in the same units as `value`
            if CONFIGS[name]["env"] == "pendulum" and not stub:
                from blackbox_mpc_amd import _lib as _L
                os.environ["BENCH_EXTRA_QUIRKS"] = str(_L.STRICT_MATH)
                try:
                    Ws = cls(name, rank, world, local, dev, use_dist, backend, gather_mode, agents=agents)
                    ms = measure(Ws, min(steps, 200), min(warmup, 10), dist, use_dist, red_dev, act_only=True)
                    Ws.close()
                    del Ws
                finally:
                    os.environ.pop("BENCH_EXTRA_QUIRKS", None)
                if rank == 0:
                    A_tot = world * (agents if agents is not None else CONFIGS[name]["A"])
                    blk["value_strict_math"] = A_tot / ms["median"]
                    blk["strict_math_median_ms"] = ms["median"] * 1e3
        return blk

    out = run_block(args.config, args.steps, args.warmup)

    if args.config == "cfg2" and not args.no_secondary:
        if world == 1 and not use_dist and not stub:
            # ---- north star target 2 in the same line: HalfCheetah learned MLP, PI2, N=1000, H=30 (MFMA path) ---------
            s_steps = max(200, min(DEFAULT_STEPS[SECONDARY], args.steps))
            # (60 warm-up calls, at least 200 timed ones = 0.1 s in all: the steady-state control step is captured as a
            # hipGraph at the fifth identical call, and coming from the resident pendulum kernel the part's clock takes
            # some tens of milliseconds of this kernel to settle -- with 20 + 30 calls the block read 0.330 ms where a run
            # of its own reads 0.317; the block reports its own steps / warmup)
            sec = run_block(SECONDARY, s_steps, 60, launch_per_call=False)
            if not args.no_cpu_baseline:
                sec["cpu_baseline"] = cpu_baseline(CONFIGS[SECONDARY], budget_s=6.0)
            out["secondary"] = sec
            # ---- the one learned-model configuration the reference itself pins (tutorials/mujoco/tutorial_two.py:23-33,52-53):
            # MLP 26-500-500-500-20, RandomSearch, population 4048, horizon 15 -- the generic MFMA rollout kernel
            tut = run_block("cfg_tut2", max(100, min(200, args.steps)), 20, launch_per_call=False)
            if not args.no_cpu_baseline:
                tut["cpu_baseline"] = cpu_baseline(CONFIGS["cfg_tut2"], budget_s=4.0)
            out["tutorial_two"] = tut
        # ---- BASELINE config 3 as it is stated: 64 agents IN TOTAL (strong scaling: 64 / N per GPU), Pendulum PI2
        # N=1000 H=30.  One GPU takes all 64 in about the time it takes 8 (the persistent kernel is one workgroup per
        # agent on a 256-CU part), so this curve is flat by construction -- it is reported so that nobody has to guess.
        if 64 % world == 0:
            c3_steps = max(100, min(300, args.steps))
            c3 = run_block("cfg3", c3_steps, 5, agents=64 // world, scaling="strong", launch_per_call=False)
            if rank == 0:
                c3["note"] = "BASELINE configs[2]: 64 agents in total, %d per GPU on %d GPU(s)" % (64 // world, world)
                out["config3" if world == 1 else "secondary"] = c3

    if rank == 0:
        c = CONFIGS[args.config]
        if not args.no_cpu_baseline and world == 1 and not stub and c["opt"] in ("RandomSearch", "CEM", "PI2"):
            out["cpu_baseline"] = cpu_baseline(c)
        # RCCL prints its version banner through C stdio, which is block-buffered on a pipe and would otherwise
        # come out after this line at exit: push it out first so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(c, budget_s=8.0):
    """`_cpu_baseline_impl` behind a net: the CPU leg runs after every GPU measurement of the line, so a problem in it (a torch-CPU
    fault, the checker's build) is recorded in its block instead of costing the line."""
    try:
        return _cpu_baseline_impl(c, budget_s)
    except Exception as ex:                                # noqa: BLE001
        return {"error": "%s: %s" % (type(ex).__name__, ex), "unit": "control-steps/s", "kind": "port", "value": None}


def _cpu_baseline_impl(c, budget_s=8.0):
    """The CPU restatements of the same hot path timed on this host, closed loop, noise drawn inside the timed region as
    the reference's graph does.  No TF-CPU number can exist (TensorFlow absent, the reference publishes none), so:

    * learned-model configurations: `value` = oracle/oracle_torch.py, the op graph of deterministic.py:62-73 /
      deterministic_mlp.py:49-50 one torch-CPU op per TF op, fp32, the framework's multi-threaded sgemm at the best
      thread count of a short ladder (the closest stand-in for TF-CPU's Eigen contraction + executor); `checker_value` = the plain-C
      checker oracle/oracle_c.c (a per-row double-precision matvec: a parity tool, not a fast CPU implementation);
    * pendulum configurations: `value` = oracle/oracle_c.c with OpenMP over candidate trajectories (the axis the reference's
      TF-CPU executor parallelises), best team size of a short ladder; `torch_value` = the op-by-op torch restatement,
      which at 500 x 3 floats per op is bound by the framework's per-op overhead, like an eager TF run.

    This is the ONLY place bench.py touches oracle/.  The whole leg is capped at about 2 x budget_s of wall time."""
    os.environ.setdefault("OMP_WAIT_POLICY", "ACTIVE")
    import torch
    from oracle import oracle_c as OC
    from oracle import oracle_torch as OT
    from blackbox_mpc_amd.utils import synthetic as SY
    N, A, H, iters, k = c["N"], c["A"], c["H"], c["iters"], c["k"]
    mlp = c["env"] == "cheetah"
    if mlp:
        S, U = 20, 6
        ws, bs = SY.make_mlp_params(mlp_dims(c), seed=42)
        acts, stats = mlp_acts(c), SY.cheetah_stats(S, U)
        lo, hi = [-1.0] * U, [1.0] * U
        start = SY.cheetah_start_states(A, S)
        co = OC.COracle("mlp", "cheetah", lo, hi, N, A, H, S, iters=iters, k=max(k, 1), mlp=(ws, bs, acts), stats=stats)
        to = OT.make(c["opt"], "cheetah", lo, hi, N, A, H, iters, k, mlp=(ws, bs, acts), stats=stats)
    else:
        S, U = 3, 1
        lo, hi = [-2.0], [2.0]
        start = SY.pendulum_start_states(A)
        co = OC.COracle("pendulum", "pendulum", lo, hi, N, A, H, S, iters=iters, k=max(k, 1))
        to = OT.make(c["opt"], "pendulum", lo, hi, N, A, H, iters, k)

    def run_c(budget, max_n):
        co.reset()
        state, n, t_used = start, 0, 0.0
        while t_used < budget and n < max_n:
            t0 = time.perf_counter()
            _, nxt, _ = co.optimize(c["opt"], state, noise=None, seed=n)
            t_used += time.perf_counter() - t0
            state = nxt
            n += 1
        return n, n * A / t_used

    def run_torch(budget, max_n):
        to.reset()
        state, n, t_used = torch.from_numpy(start), 0, 0.0
        with torch.no_grad():
            while t_used < budget and n < max_n:
                t0 = time.perf_counter()
                _, nxt, _ = to.call(state)
                t_used += time.perf_counter() - t0
                state = nxt
                n += 1
        return n, n * A / t_used

    max_t = OC.num_threads()
    host_threads = os.cpu_count() or max_t
    saved_torch_threads = torch.get_num_threads()
    res = {"unit": "control-steps/s", "kind": "port",
           "note": "no TF-CPU number exists for the reference (TensorFlow absent, reference publishes none)"}
    try:
        if mlp:
            # ---- torch-CPU op by op on every host thread (a short ladder: a small sgemm does not always want them all)
            probe = {}
            # (every thread of a 256-thread host: 0.01 steps/s; on a small host the ladder is its halves and quarters)
            for t in sorted({max(1, min(host_threads, x)) for x in (8, 16, 32, host_threads // 2, host_threads // 4) if x <= 32}):
                torch.set_num_threads(t)
                run_torch(0.2, 1)
                probe[t] = run_torch(0.6, 50)[1]
            cores = max(probe, key=probe.get)
            torch.set_num_threads(cores)
            n_t, v_t = run_torch(budget_s, 40000)
            OC.set_num_threads(min(max_t, 32))
            n_c, v_c = run_c(2.0, 400)
            res.update({"value": v_t, "cores": cores,
                        "sample": "%d closed-loop control steps of the same workload with oracle/oracle_torch.py: torch-CPU fp32, one op "
                                  "per TF op of deterministic.py:62-73 / deterministic_mlp.py:49-50, multi-threaded sgemm, best of a thread "
                                  "ladder = %d of %d host threads, draws generated in the timed region" % (n_t, cores, host_threads),
                        "thread_ladder": {str(t): round(v, 2) for t, v in probe.items()},
                        "checker_value": v_c,
                        "checker_note": "oracle/oracle_c.c (the parity checker: per-row double-precision matvec, OpenMP over "
                                        "trajectories, %d threads, %d steps) -- not a tuned CPU implementation" % (min(max_t, 32), n_c)})
        else:
            run_c(0.2, 3)                                   # warm the thread pool / page in
            ladder = sorted({t for t in (1, 4, 8, 16, 32, max_t) if t <= max_t})
            probe = {}
            for t in ladder:
                OC.set_num_threads(t)
                probe[t] = run_c(0.3, 400)[1]
            cores = max(probe, key=probe.get)
            OC.set_num_threads(cores)
            n_all, v_all = run_c(budget_s, 40000)
            torch.set_num_threads(min(host_threads, 8))
            run_torch(0.2, 1)
            n_t, v_t = run_torch(2.0, 200)
            res.update({"value": v_all, "cores": cores,
                        "sample": "%d closed-loop control steps of the same workload with the C restatement oracle/oracle_c.c "
                                  "(OpenMP over trajectories, best of a thread-count ladder = %d of %d threads, noise drawn in the "
                                  "timed region)" % (n_all, cores, max_t),
                        "thread_ladder": {str(t): round(v, 2) for t, v in probe.items()},
                        "single_core_value": probe.get(1),
                        "torch_value": v_t,
                        "torch_note": "oracle/oracle_torch.py, one torch-CPU op per TF op (%d steps): per-op overhead bound at this "
                                      "size, like an eager TF run" % n_t})
    finally:
        torch.set_num_threads(saved_torch_threads)
        OC.set_num_threads(max_t)
    return res


if __name__ == "__main__":
    main()
