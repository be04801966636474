"""Minimal stand-in for gym.spaces.Box: the hot path only reads `.shape[0]`, `.low`, `.high`
(reference optimizers/optimizer_base.py:31-36, dynamics_handlers/system_dynamics_handler.py:61-62)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        if shape is not None:
            low = np.broadcast_to(low, shape).copy()
            high = np.broadcast_to(high, shape).copy()
        if low.shape != high.shape or low.ndim != 1:
            raise ValueError("Box needs 1-D low/high of equal shape")
        self.low, self.high, self.shape, self.dtype = low, high, low.shape, dtype

    def __repr__(self):
        return "Box(%s, %s)" % (self.low, self.high)
