"""Error of the optional bf16-input MLP modes against the C oracle (BASELINE config 5 per-GPU share)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_c as OC
from oracle import oracle_np as O
from blackbox_mpc_amd import _lib as L
from blackbox_mpc_amd.engine import Engine
F = np.float32
N, A, H, S, U = 2000, 4, 50, 20, 6
ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=42)
stats = [np.zeros(S, F), np.ones(S, F), np.zeros(U, F), np.ones(U, F), np.zeros(S, F), np.full(S, 0.1, F)]
co = OC.COracle("mlp", "cheetah", [-1.0] * U, [1.0] * U, N, A, H, S, mlp=(ws, bs, ["tanh", "tanh", None]), stats=stats)
rng = np.random.default_rng(55)
states = O.cheetah_start_states(A, S)
seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
want = co.evaluate(states, seq)
for mode in ("0", "3", "1"):
    os.environ["BBMPC_MLP_BF16"] = mode
    eng = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=A, planning_horizon=H)
    eng.set_mlp(ws, bs, [1, 1, 0], stats)
    got = eng.evaluate(states, seq)
    err = np.abs(got - want)
    print("BBMPC_MLP_BF16=%s: max |err| %.3e  median %.3e  p99 %.3e   (reward scale: std %.2f)" % (mode, err.max(), np.median(err), np.quantile(err, 0.99), want.std()))
