"""Wall time of one control step at config 2 through the three host layers (run on a GPU box): MPCPolicy.act, Engine.optimize
(the ctypes wrapper) and the raw C-ABI call."""
import os, time, numpy as np, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackbox_mpc_amd import Box
from blackbox_mpc_amd.policies.mpc_policy import MPCPolicy
from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function
from oracle import oracle_np as O
pol = MPCPolicy(reward_function=pendulum_reward_function, env_action_space=Box(low=[-2.0], high=[2.0]),
                env_observation_space=Box(low=[-1, -1, -8], high=[1, 1, 8]), dynamics_function=PendulumTrueModel(),
                true_model=True, optimizer_name="CEM", num_agents=1, planning_horizon=30, max_iterations=5,
                population_size=500, num_elite=50)
s = O.pendulum_start_states(1)
for t in range(50):
    a, s, r = pol.act(s, t)
t0 = time.perf_counter()
K = 2000
for t in range(K):
    a, s, r = pol.act(s, t)
dt = time.perf_counter() - t0
print("MPCPolicy.act host-in/host-out: %.1f us per call, %.0f control steps/s" % (dt / K * 1e6, K / dt))
eng = pol._optimizer._require_engine()
s = O.pendulum_start_states(1)
t0 = time.perf_counter()
for t in range(K):
    a, s, r = eng.optimize(s)
dt = time.perf_counter() - t0
print("Engine.optimize (ctypes wrapper): %.1f us per call" % (dt / K * 1e6))
import ctypes
from blackbox_mpc_amd import _lib as L
st = np.ascontiguousarray(s, np.float32); act = np.empty((1, 1), np.float32); nx = np.empty((1, 3), np.float32); rw = np.empty((1,), np.float32)
ps, pa, pn, pr = [L.ptr(v) for v in (st, act, nx, rw)]
t0 = time.perf_counter()
for t in range(K):
    L.lib.bbmpc_optimize(eng._h, ps, 0, 0, pa, pn, pr)
    st[:] = nx
dt = time.perf_counter() - t0
print("raw bbmpc_optimize: %.1f us per call" % (dt / K * 1e6))
