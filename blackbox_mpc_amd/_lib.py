"""ctypes binding of include/bbmpc.h.  There is no CPU fallback: if the HIP
library cannot be loaded, importing this module raises."""
import ctypes
import threading
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BBMPC_LIB") or os.path.join(HERE, "libbbmpc.so")   # BBMPC_LIB: a debug build (tools/)

ABI_VERSION = 2

# enums (bbmpc.h)
OPT_NONE, OPT_RANDOM_SEARCH, OPT_CEM, OPT_PI2, OPT_PSO, OPT_CMAES, OPT_SPSA = range(7)
DYN_PENDULUM, DYN_MLP, DYN_USER = 1, 2, 3
REW_PENDULUM, REW_CHEETAH, REW_USER = 1, 2, 3
USER_KIND_REWARD, USER_KIND_DYNAMICS = 1, 2
ACT_NONE, ACT_TANH, ACT_RELU, ACT_SIGMOID = range(4)
FIX_Q1_REWARD_ARG_ORDER = 1 << 0
FIX_Q2_CEM_WARM_START = 1 << 1
FIX_Q7_EXPL_NOISE_ZERO_MEAN = 1 << 2
CMAES_PER_AGENT = 1 << 8
STRICT_MATH = 1 << 9
NOISE_TRUNC_NORMAL, NOISE_UNIFORM, NOISE_RADEMACHER, NOISE_NORMAL, NOISE_PSO_SCALARS = 1, 2, 3, 4, 5
NOISE_PSO_RESEED_TRUNC, NOISE_PSO_RESEED_UNIFORM, NOISE_PSO_RESET_POS, NOISE_PSO_RESET_VEL = 6, 7, 8, 9
NOISE_EXPLORATION = 10
TRACE_REWARDS, TRACE_MEAN, TRACE_VAR, TRACE_ELITES, TRACE_SAMPLES = 1, 2, 3, 4, 5
TRACE_CMA_B, TRACE_CMA_C, TRACE_CMA_D, TRACE_CMA_SVD_STATS = 6, 7, 8, 9

E_INVALID, E_NO_DEVICE, E_HIP, E_STATE, E_UNSUPPORTED = -1, -2, -3, -4, -5

c_float_p = ctypes.POINTER(ctypes.c_float)


class Config(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("optimizer", ctypes.c_int32), ("dynamics", ctypes.c_int32),
        ("reward", ctypes.c_int32), ("population_size", ctypes.c_int32), ("num_agents", ctypes.c_int32),
        ("planning_horizon", ctypes.c_int32), ("dim_u", ctypes.c_int32), ("dim_s", ctypes.c_int32),
        ("max_iterations", ctypes.c_int32), ("num_elite", ctypes.c_int32), ("agent_offset", ctypes.c_int32),
        ("num_agents_global", ctypes.c_int32), ("device", ctypes.c_int32), ("quirks", ctypes.c_uint32),
        ("reserved0", ctypes.c_uint32), ("seed", ctypes.c_uint64),
        ("alpha", ctypes.c_float), ("lamda", ctypes.c_float),
        ("pso_c1", ctypes.c_float), ("pso_c2", ctypes.c_float), ("pso_w", ctypes.c_float),
        ("pso_v0_fraction", ctypes.c_float),
        ("spsa_alpha", ctypes.c_float), ("spsa_gamma", ctypes.c_float), ("spsa_a", ctypes.c_float),
        ("spsa_c", ctypes.c_float),
        ("cma_alpha_cov", ctypes.c_float), ("cma_h_sigma", ctypes.c_float),
        ("action_low", c_float_p), ("action_high", c_float_p),
        ("population_offset", ctypes.c_int32), ("population_global", ctypes.c_int32),
    ]


class BBMPCError(Exception):
    def __init__(self, code, msg):
        super().__init__("bbmpc error %d: %s" % (code, msg))
        self.code = code


# every symbol include/bbmpc.h declares (tests check the library exports all of them)
SYMBOLS = [
    "bbmpc_abi_version", "bbmpc_last_error", "bbmpc_device_count", "bbmpc_create", "bbmpc_destroy",
    "bbmpc_set_stream", "bbmpc_set_mlp", "bbmpc_reset", "bbmpc_optimize", "bbmpc_optimize_dev", "bbmpc_evaluate",
    "bbmpc_evaluate_dev", "bbmpc_predict_next_state", "bbmpc_evaluate_next_reward", "bbmpc_step_dev",
    "bbmpc_inject_noise", "bbmpc_dump_noise", "bbmpc_set_trace", "bbmpc_get_trace", "bbmpc_get_state",
    "bbmpc_set_state", "bbmpc_set_profiling", "bbmpc_get_profile", "bbmpc_profile_instantiation", "bbmpc_synchronize", "bbmpc_rollout_episode",
    "bbmpc_comm_unique_id", "bbmpc_comm_init", "bbmpc_comm_init_local", "bbmpc_gather_records_dev", "bbmpc_gather_wait", "bbmpc_comm_destroy",
    "bbmpc_optimize_gather_dev", "bbmpc_set_stream_default", "bbmpc_optimize_gather", "bbmpc_comm_info", "bbmpc_call_stats",
    "bbmpc_graph_stats", "bbmpc_handle_device",
    "bbmpc_set_reward_source", "bbmpc_set_dynamics_source", "bbmpc_check_user_source", "bbmpc_mlp_forward",
    "bbmpc_set_reward_callback", "bbmpc_set_dynamics_callback",
    "bbmpc_process_input", "bbmpc_process_output", "bbmpc_check_user_rollout",
]
COMM_ID_BYTES = 128
# bbmpc_rows_callback (include/bbmpc.h): user, d_cur, d_actions, d_next, batch, d_out, hip_stream -> status
ROWS_CALLBACK = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                 ctypes.c_void_p, ctypes.c_void_p)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "blackbox_mpc_amd: %s is missing. Build it with `python -m blackbox_mpc_amd._build` "
            "(needs hipcc). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm carries its own libamdhip64 and must be the first to load it --
    # libbbmpc.so then binds to that same runtime (shared device pointers / streams with torch tensors).  Loaded the
    # other way round, torch comes up with a second runtime and reports no devices.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    if lib.bbmpc_abi_version() != ABI_VERSION:
        raise ImportError("blackbox_mpc_amd: libbbmpc.so ABI %d != expected %d; rebuild"
                          % (lib.bbmpc_abi_version(), ABI_VERSION))
    lib.bbmpc_last_error.restype = ctypes.c_char_p
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.bbmpc_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    lib.bbmpc_destroy.argtypes = [vp]
    lib.bbmpc_set_stream.argtypes = [vp, vp]
    lib.bbmpc_set_stream_default.argtypes = [vp]
    lib.bbmpc_set_mlp.argtypes = [vp, i32, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(vp),
                                  ctypes.POINTER(vp), i32, ctypes.POINTER(vp)]
    lib.bbmpc_reset.argtypes = [vp]
    lib.bbmpc_optimize.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    lib.bbmpc_optimize_dev.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.bbmpc_evaluate.argtypes = [vp, vp, vp, i32, vp]
    lib.bbmpc_evaluate_dev.argtypes = [vp, vp, vp, i32, vp]
    lib.bbmpc_predict_next_state.argtypes = [vp, vp, vp, i32, vp]
    lib.bbmpc_evaluate_next_reward.argtypes = [vp, vp, vp, vp, i32, vp]
    lib.bbmpc_step_dev.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.bbmpc_inject_noise.argtypes = [vp, i32, vp, i64]
    lib.bbmpc_dump_noise.argtypes = [vp, i32, i32, i32, vp, i64]
    lib.bbmpc_set_trace.argtypes = [vp, i32]
    lib.bbmpc_get_trace.argtypes = [vp, i32, i32, vp, i64]
    lib.bbmpc_get_state.argtypes = [vp, ctypes.c_char_p, vp, i64]
    lib.bbmpc_set_state.argtypes = [vp, ctypes.c_char_p, vp, i64]
    lib.bbmpc_set_profiling.argtypes = [vp, i32]
    lib.bbmpc_get_profile.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64),
                                      ctypes.POINTER(ctypes.c_char_p)]
    lib.bbmpc_profile_instantiation.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p)]
    lib.bbmpc_synchronize.argtypes = [vp]
    lib.bbmpc_rollout_episode.argtypes = [vp, vp, i32, i32, vp]
    lib.bbmpc_comm_unique_id.argtypes = [vp, i64]
    lib.bbmpc_comm_init.argtypes = [vp, vp, i32, i32]
    lib.bbmpc_comm_init_local.argtypes = [vp, ctypes.c_uint64, i32, i32]
    lib.bbmpc_gather_records_dev.argtypes = [vp, vp, vp, i64, i32]
    lib.bbmpc_gather_wait.argtypes = [vp, i32, i32]
    lib.bbmpc_optimize_gather_dev.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32]
    lib.bbmpc_comm_destroy.argtypes = [vp]
    lib.bbmpc_optimize_gather.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, i32]
    lib.bbmpc_comm_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    lib.bbmpc_call_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    lib.bbmpc_graph_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int64)]
    lib.bbmpc_handle_device.argtypes = [vp, ctypes.POINTER(i32)]
    lib.bbmpc_set_reward_source.argtypes = [vp, ctypes.c_char_p]
    lib.bbmpc_set_dynamics_source.argtypes = [vp, ctypes.c_char_p]
    lib.bbmpc_set_reward_callback.argtypes = [vp, ROWS_CALLBACK, vp]
    lib.bbmpc_set_dynamics_callback.argtypes = [vp, ROWS_CALLBACK, vp]
    lib.bbmpc_check_user_source.argtypes = [i32, ctypes.c_char_p, i32, i32]
    lib.bbmpc_mlp_forward.argtypes = [vp, vp, i32, vp]
    lib.bbmpc_check_user_rollout.argtypes = [i32, i32, ctypes.c_char_p, ctypes.c_char_p, i32, i32]
    lib.bbmpc_process_input.argtypes = [vp, vp, vp, i32, ctypes.POINTER(vp), vp]
    lib.bbmpc_process_output.argtypes = [vp, vp, vp, i32, ctypes.POINTER(vp), vp]
    return lib


lib = _load()


# an exception raised inside a Python callback the engine called (utils/device_functions.py): the C ABI can only carry a
# status code, so the callback parks it here and check() re-raises it as the cause
_callback_state = threading.local()                       # per thread, like bbmpc_last_error()


def park_callback_error(ex):
    _callback_state.error = ex


def check(code):
    if code != 0:
        err = BBMPCError(code, lib.bbmpc_last_error().decode("utf-8", "replace"))
        cause, _callback_state.error = getattr(_callback_state, "error", None), None
        if cause is not None:
            raise err from cause
        raise err


def device_count():
    return lib.bbmpc_device_count()


def f32c(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
