"""Wall time of consecutive CMA-ES control steps (config-2 size) in a closed loop, 25-step means: shows the data
dependence of the Jacobi sweep count.  BBMPC_CMA_FUSED=1 for the one-launch form."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackbox_mpc_amd import _build
_build.build()
from blackbox_mpc_amd import _lib as L
from blackbox_mpc_amd.policies import MPCPolicy
from blackbox_mpc_amd.spaces import Box
from blackbox_mpc_amd.utils import synthetic as SY
from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function
pol = MPCPolicy(reward_function=pendulum_reward_function, env_action_space=Box([-2.0], [2.0]),
                env_observation_space=Box([-1, -1, -8], [1, 1, 8]), true_model=True, dynamics_function=PendulumTrueModel(),
                optimizer_name="CMA-ES", num_agents=1, planning_horizon=30, population_size=500, max_iterations=5, num_elite=50)
eng = pol._optimizer._require_engine()
obs = SY.pendulum_start_states(1)
st, action, nxt, rew, p_st, p_act, p_nxt, p_rew = eng._io_buffers()
st[:] = obs
ts = []
for i in range(400):
    t0 = time.perf_counter()
    L.lib.bbmpc_optimize(eng._h, p_st, 0, 0, p_act, p_nxt, p_rew)
    ts.append(time.perf_counter() - t0)
    st[:] = nxt
ts = np.array(ts) * 1e6
print("fused" if os.environ.get("BBMPC_CMA_FUSED") else "per-iteration", " ".join("%.0f" % ts[i:i + 25].mean() for i in range(0, 400, 25)))
