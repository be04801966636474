#!/usr/bin/env python
"""Where the wall time of one MPCPolicy.act goes (config 2): raw C-ABI call vs Engine.optimize vs MPCPolicy.act.

usage: act_overhead.py [CEM|CMA-ES|PI2|...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def med(f, n=600, w=50):
    for _ in range(w):
        f()
    ts = np.empty(n)
    for i in range(n):
        t0 = time.perf_counter()
        f()
        ts[i] = time.perf_counter() - t0
    return np.median(ts) * 1e6, np.percentile(ts, 10) * 1e6, np.percentile(ts, 90) * 1e6


def main():
    opt = sys.argv[1] if len(sys.argv) > 1 else "CEM"
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib as L
    from blackbox_mpc_amd.policies import MPCPolicy
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.utils import synthetic as SY
    from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function
    pol = MPCPolicy(reward_function=pendulum_reward_function, env_action_space=Box([-2.0], [2.0]),
                    env_observation_space=Box([-1, -1, -8], [1, 1, 8]), true_model=True, dynamics_function=PendulumTrueModel(),
                    optimizer_name=opt, num_agents=1, planning_horizon=30, population_size=500, max_iterations=5, num_elite=50)
    eng = pol._optimizer._require_engine()
    obs = SY.pendulum_start_states(1)
    st, action, nxt, rew, p_st, p_act, p_nxt, p_rew = eng._io_buffers()
    st[:] = obs
    h = eng._h
    fn = L.lib.bbmpc_optimize
    state = {"o": obs}

    def raw():
        fn(h, p_st, 0, 0, p_act, p_nxt, p_rew)
        st[:] = nxt

    def e_opt():
        a, n, r = eng.optimize(state["o"])
        state["o"] = n

    def act():
        a, n, r = pol.act(state["o"], 0)
        state["o"] = n
    for name, f in (("raw bbmpc_optimize (ctypes, prebuilt arguments)", raw), ("Engine.optimize", e_opt), ("MPCPolicy.act", act)):
        print("%-50s median %.2f us  p10 %.2f  p90 %.2f" % ((name,) + med(f)))


if __name__ == "__main__":
    main()
