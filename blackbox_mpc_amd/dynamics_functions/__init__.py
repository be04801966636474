from .deterministic_mlp import DeterministicMLP  # noqa: F401
