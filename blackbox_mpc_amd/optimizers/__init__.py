from .random_search import RandomSearchOptimizer  # noqa: F401
from .spsa import SPSAOptimizer  # noqa: F401
from .cma_es import CMAESOptimizer  # noqa: F401
from .cem import CEMOptimizer  # noqa: F401
from .pi2 import PI2Optimizer  # noqa: F401
from .pso import PSOOptimizer  # noqa: F401
from .optimizer_base import OptimizerBase  # noqa: F401
