from .deterministic import DeterministicTrajectoryEvaluator  # noqa: F401
from .evaluator_base import EvaluatorBase  # noqa: F401
