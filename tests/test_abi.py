"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads without a GPU, exports every
symbol include/bbmpc.h declares, and refuses to compute without a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bbmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)           # comments mention names that are not entry points
    return sorted(set(re.findall(r"\b(bbmpc_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported(built_lib):
    lib = ctypes.CDLL(built_lib)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libbbmpc.so does not export %s" % name
    from blackbox_mpc_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared


def test_config_struct_matches_header(built_lib):
    # field order/types of the ctypes mirror follow the header's struct
    text = open(os.path.join(ROOT, "include", "bbmpc.h")).read()
    body = text[text.index("typedef struct bbmpc_config {"):text.index("} bbmpc_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct bbmpc_config {", "").strip()
        if not decl:
            continue
        parts = [p.strip() for p in decl.split(",")]
        first = parts[0].split()[-1].lstrip("*")
        names.append(first)
        names += [p.lstrip("*") for p in parts[1:]]
    from blackbox_mpc_amd._lib import Config
    assert [n for n, _ in Config._fields_] == names


def test_abi_version_and_no_device_is_loud(built_lib):
    from blackbox_mpc_amd import _lib as L
    assert L.lib.bbmpc_abi_version() == L.ABI_VERSION
    if L.device_count() > 0:
        pytest.skip("a GPU is present")
    from blackbox_mpc_amd.engine import Engine
    with pytest.raises(L.BBMPCError) as ei:
        Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=5,
               population_size=16, max_iterations=2, num_elite=4)
    assert ei.value.code == L.E_NO_DEVICE


def test_invalid_configs_rejected_before_touching_a_device(built_lib):
    from blackbox_mpc_amd import _lib as L
    from blackbox_mpc_amd.engine import Engine
    with pytest.raises(L.BBMPCError) as ei:      # pendulum model needs S=3,U=1
        Engine(L.OPT_NONE, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0, -2.0], [2.0, 2.0], dim_s=3, num_agents=1,
               planning_horizon=5)
    assert ei.value.code == L.E_INVALID
    with pytest.raises(L.BBMPCError) as ei:      # num_elite > population
        Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=5,
               population_size=8, max_iterations=1, num_elite=9)
    assert ei.value.code == L.E_INVALID
    with pytest.raises(L.BBMPCError):
        Engine(L.OPT_NONE, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * 6, [1.0] * 6, dim_s=17, num_agents=1, planning_horizon=5)


def test_host_api_mirrors_reference_errors(built_lib):
    from blackbox_mpc_amd.optimizers import OptimizerBase
    from blackbox_mpc_amd.policies import MPCPolicy
    from blackbox_mpc_amd.spaces import Box
    from blackbox_mpc_amd.trajectory_evaluators import EvaluatorBase
    from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function
    act, obs = Box([-2.0], [2.0]), Box([-1, -1, -8], [1, 1, 8])
    with pytest.raises(Exception, match="Please Specify Num Of Agents"):
        MPCPolicy(reward_function=pendulum_reward_function, env_action_space=act, env_observation_space=obs,
                  true_model=True, dynamics_function=PendulumTrueModel(), optimizer_name="CEM")
    with pytest.raises(AttributeError):          # unknown optimizer name -> None optimizer (mpc_policy.py:120)
        MPCPolicy(reward_function=pendulum_reward_function, env_action_space=act, env_observation_space=obs,
                  true_model=True, dynamics_function=PendulumTrueModel(), optimizer_name="nope", num_agents=1)
    base = OptimizerBase("x", 5, 2, 1, act, obs)
    with pytest.raises(Exception, match="not implemented"):
        base(np.zeros((1, 3), np.float32), 0, False)
    with pytest.raises(Exception, match="not implemented"):
        base.reset()
    with pytest.raises(Exception, match="not implemented"):
        EvaluatorBase(None, None)(None, None, 0)
    with pytest.raises(NotImplementedError):     # arbitrary python callables have no device functor
        from blackbox_mpc_amd.dynamics_handlers import SystemDynamicsHandler
        from blackbox_mpc_amd.trajectory_evaluators import DeterministicTrajectoryEvaluator
        h = SystemDynamicsHandler(act, obs, dynamics_function=PendulumTrueModel(), true_model=True)
        DeterministicTrajectoryEvaluator(lambda a, b, c: 0, h)(np.zeros((1, 3)), np.zeros((2, 1, 3, 1)), 0)


def test_philox_restatement_known_answer():
    # Philox4x32-10 KAT from the Random123 distribution (counter=0,key=0 / all-ones / pi digits)
    from tests.philox_np import philox4x32_10
    r = philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(x) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(x) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
