import time, sys, numpy as np
sys.path.insert(0, '.')
from blackbox_mpc_amd import _lib as L
from blackbox_mpc_amd.engine import Engine
from oracle import oracle_np as O
t=time.time()
eng = Engine(L.OPT_CMAES, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0],[2.0], dim_s=3, num_agents=2, planning_horizon=5, population_size=64, max_iterations=1, num_elite=8)
print('create', time.time()-t)
s = O.pendulum_start_states(2)
for i in range(3):
    t=time.time(); a,n,r = eng.optimize(s); print('optimize', i, time.time()-t, a.ravel())
n=10
for name in ['m','sigma','p_sigma','p_C','D']:
    print(name, eng.get_state(name,(n,)))
print(O.cmaes_constants(64,8,10))
