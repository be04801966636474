from .mpc_policy import MPCPolicy  # noqa: F401
from .model_based_base_policy import ModelBasedBasePolicy  # noqa: F401
