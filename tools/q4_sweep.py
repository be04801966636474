"""Device-resident PI2 control-step time of the learned-model rollout on the 16-particle-tile kernels against the 4-particle
quad kernels over the population size (run on a GPU box): where the engine's size heuristic should switch."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import popshard_table as P
from blackbox_mpc_amd import _lib as L
for N in (1024, 1500, 2000, 2500, 3000, 4000, 6000):
    out = []
    for q4 in ("0", "1"):
        os.environ["BBMPC_MLP_Q4"] = q4
        out.append(P.rate(P.plain(L, N), 100))
    print("N=%d  16-particle tiles %.1f us/step   quads %.1f us/step" % (N, out[0], out[1]), flush=True)
