from .mpc_policy import MPCPolicy  # noqa: F401
from .model_based_base_policy import ModelBasedBasePolicy  # noqa: F401
from .model_free_base_policy import ModelFreeBasePolicy  # noqa: F401
from .random_policy import RandomPolicy  # noqa: F401
