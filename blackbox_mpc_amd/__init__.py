"""blackbox_mpc_amd -- MI355X-native sampling-MPC rollout engine behind the
blackbox_mpc (ossamaAhmed/blackbox_mpc v0.3) MPCPolicy / Optimizer /
TrajectoryEvaluator API.  All compute runs in hand-written HIP kernels
(libbbmpc.so, C ABI in include/bbmpc.h); there is no CPU fallback."""
from .spaces import Box  # noqa: F401

__version__ = "0.1.0"
