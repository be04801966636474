from .system_dynamics_handler import SystemDynamicsHandler  # noqa: F401
