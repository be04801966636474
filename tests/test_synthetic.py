"""The benchmark's synthetic inputs (blackbox_mpc_amd/utils/synthetic.py) and the oracle's own statement
of the same recipe must agree bit for bit: parity tests build the engine side from one and the oracle side from the
other."""
import numpy as np

from blackbox_mpc_amd.utils import synthetic as SY
from oracle import oracle_np as O


def test_generators_agree_with_the_oracle():
    for A, off in ((1, 0), (4, 8), (64, 0)):
        np.testing.assert_array_equal(SY.pendulum_start_states(A, off), O.pendulum_start_states(A, off))
        np.testing.assert_array_equal(SY.cheetah_start_states(A, 20, off), O.cheetah_start_states(A, 20, off))
    for dims, seed in (([26, 200, 200, 20], 42), ([4, 32, 32, 3], 7)):
        (w1, b1), (w2, b2) = SY.make_mlp_params(dims, seed), O.make_mlp_params(dims, seed)
        for a, b in zip(w1 + b1, w2 + b2):
            np.testing.assert_array_equal(a, b)
    st = SY.cheetah_stats(20, 6)
    assert [v.shape[0] for v in st] == [20, 20, 6, 6, 20, 20] and float(st[5][0]) == np.float32(0.1)
    s = SY.pendulum_start_states(3)
    np.testing.assert_allclose(s[:, 0] ** 2 + s[:, 1] ** 2, 1.0, atol=1e-6)
    assert np.all(np.abs(s[:, 2]) <= 1.0)
