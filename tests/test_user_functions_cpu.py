"""User-supplied reward / dynamics device functions, CPU side: the run-time compile
works without a GPU (hiprtc cross-compiles for gfx950), compiler errors come back with the log, and a plain Python
callable is refused with a message that says what to do instead -- never evaluated on the host."""
import numpy as np
import pytest

from blackbox_mpc_amd import _lib as L
from blackbox_mpc_amd.utils.device_functions import HipDynamicsFunction, HipRewardFunction, check_source

GOOD_REWARD = """
__device__ float bbmpc_user_reward(const float* cur, const float* act, const float* nxt, int S, int U) {
    float r = 0.0f;
    for (int i = 0; i < S; ++i) r -= cur[i] * cur[i];
    for (int i = 0; i < U; ++i) r -= 0.1f * act[i] * act[i];
    return r + nxt[0];
}
"""
GOOD_DYNAMICS = """
__device__ void bbmpc_user_dynamics(const float* x, float* delta, int S, int U) {
    for (int i = 0; i < S; ++i) delta[i] = 0.05f * x[S + (i % U)] - 0.01f * x[i];
}
"""


def test_sources_compile_without_a_gpu(built_lib):
    check_source(L.USER_KIND_REWARD, GOOD_REWARD, 3, 1)
    check_source(L.USER_KIND_REWARD, GOOD_REWARD, 20, 6)
    check_source(L.USER_KIND_DYNAMICS, GOOD_DYNAMICS, 4, 2)


def test_fused_rollout_kernel_compiles_with_the_engine_headers(built_lib):
    # analytic dynamics: the engine builds ONE lane-per-trajectory kernel with the user function(s) inlined next to its own
    # PendulumTrueModel / rewards (models.hpp handed to hiprtc as text) -- every combination must compile
    chk = lambda d, r, ds, rs, S, U: L.check(L.lib.bbmpc_check_user_rollout(d, r, ds.encode() if ds else None,
                                                                            rs.encode() if rs else None, S, U))
    chk(L.DYN_USER, L.REW_USER, GOOD_DYNAMICS, GOOD_REWARD, 4, 2)
    chk(L.DYN_PENDULUM, L.REW_USER, None, GOOD_REWARD, 3, 1)
    chk(L.DYN_USER, L.REW_PENDULUM, GOOD_DYNAMICS, None, 3, 1)
    chk(L.DYN_USER, L.REW_CHEETAH, GOOD_DYNAMICS, None, 20, 6)
    with pytest.raises(L.BBMPCError):                        # the learned MLP's rollouts live on the matrix cores
        chk(L.DYN_MLP, L.REW_USER, None, GOOD_REWARD, 20, 6)
    with pytest.raises(L.BBMPCError):
        chk(L.DYN_USER, L.REW_USER, None, GOOD_REWARD, 3, 1)


def test_compile_errors_carry_the_compiler_log(built_lib):
    with pytest.raises(L.BBMPCError) as ei:
        check_source(L.USER_KIND_REWARD, "__device__ float bbmpc_user_reward(const float* c) { return undeclared_name; }", 3, 1)
    msg = str(ei.value)
    assert ei.value.code == L.E_INVALID and "undeclared_name" in msg and "failed to compile" in msg
    with pytest.raises(L.BBMPCError):           # the expected entry point is missing altogether
        check_source(L.USER_KIND_DYNAMICS, "__device__ int unrelated() { return 1; }", 3, 1)
    with pytest.raises(L.BBMPCError):
        check_source(7, GOOD_REWARD, 3, 1)


def test_host_callables_are_refused_not_evaluated_on_the_cpu(built_lib):
    from blackbox_mpc_amd.dynamics_handlers import SystemDynamicsHandler
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.trajectory_evaluators.deterministic import plugin_kinds
    from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function
    act, obs = Box([-2.0], [2.0]), Box([-1, -1, -8], [1, 1, 8])
    h = SystemDynamicsHandler(act, obs, dynamics_function=PendulumTrueModel(), true_model=True)
    with pytest.raises(NotImplementedError, match="HipRewardFunction"):
        plugin_kinds(lambda c, a, n: -np.sum(c * c, axis=1), h)
    h2 = SystemDynamicsHandler(act, obs, dynamics_function=lambda x, train: x[:, :3] * 0, true_model=True)
    with pytest.raises(NotImplementedError, match="HipDynamicsFunction"):
        plugin_kinds(pendulum_reward_function, h2)
    # device-code objects map to the user kinds; anything with a `hip_source` attribute counts
    assert plugin_kinds(HipRewardFunction(GOOD_REWARD), h) == (L.DYN_PENDULUM, L.REW_USER)
    h3 = SystemDynamicsHandler(act, obs, dynamics_function=HipDynamicsFunction(GOOD_DYNAMICS), true_model=True)
    assert plugin_kinds(pendulum_reward_function, h3) == (L.DYN_USER, L.REW_PENDULUM)

    class Tagged:
        hip_source = GOOD_REWARD
    assert plugin_kinds(Tagged(), h)[1] == L.REW_USER
    with pytest.raises(Exception, match="true_model"):
        plugin_kinds(pendulum_reward_function, SystemDynamicsHandler(act, obs, dynamics_function=HipDynamicsFunction(GOOD_DYNAMICS),
                                                                     true_model=False))
    with pytest.raises(ValueError):
        HipRewardFunction("   ")
