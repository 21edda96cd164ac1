"""ctypes binding of include/w2a.h (libw2a.so). No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

ROW_FLOATS = 32

OK = 0
ST_BAD_EPISODE, ST_BAD_ACTION, ST_STEP_AFTER_DONE, ST_STALE_GRAPH = 1, 2, 4, 8
ACT_I32, ACT_I64, ACT_U8 = 0, 1, 2
STEP_AUTORESET, STEP_NO_OBS, STEP_CLASSIC, STEP_REWARD_GIVEN, STEP_WIDE, STEP_SKIP_FINISHED, STEP_UNPACKED, STEP_NEXT_STEP, STEP_NO_CAPTURE = 1, 2, 8, 16, 32, 64, 128, 256, 512
S64_MIN_ENVS = 131072  # w2a_step picks the 64-envs-per-wave kernel from this batch size on (csrc/w2a_step64.hip.h)
ABI_VERSION = 18
FIX_BITS = {"alert_2wks": 1, "lag": 2, "penalty": 4, "obs": 8, "augment": 16}  # + "budget" (sticky = 0)
BUDGET_FIXED, BUDGET_LESS_THAN, BUDGET_CENTERED = 0, 1, 2

# every symbol include/w2a.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "w2a_abi_version", "w2a_last_error", "w2a_state_bytes", "w2a_create", "w2a_destroy", "w2a_reset",
    "w2a_reset_device_rng", "w2a_set_autoreset", "w2a_step", "w2a_get_state", "w2a_read_status",
    "w2a_sort_workspace_bytes", "w2a_sort_episodes", "w2a_reset_device_rng_sorted", "w2a_observe", "w2a_rollout", "w2a_rollout_order_workspace_bytes", "w2a_rollout_order_attach", "w2a_rollout_order", "w2a_rollout_posterior_mean", "w2a_policy_actions", "w2a_set_semantics",
    "w2a_group_workspace_bytes", "w2a_group_by_column", "w2a_posterior_mean_reward", "w2a_set_posterior_kernel", "w2a_invalidate", "w2a_query", "w2a_rollout_mfma_workspace_bytes", "w2a_rollout_mfma_prepare",
]
Q_LOCKSTEP_DAY, Q_PACKED_ELIGIBLE, Q_PACKED_CURRENT, Q_CANONICAL_CURRENT, Q_LAST_ROLLOUT_KERNEL, Q_LAST_STEP_KERNEL, Q_LOCKSTEP = 0, 1, 2, 3, 4, 5, 6
PM_KERNELS = {"vector": 0, "matrix": 1, "matrix_i8": 2}  # W2A_PM_VECTOR, W2A_PM_MATRIX_F64, W2A_PM_MATRIX_I8
POLICY_KINDS = {"never": 0, "always": 1, "bernoulli": 2, "threshold": 3, "table": 4}


class Policy(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p", C.c_float), ("obs_col", C.c_int32), ("threshold", C.c_float),
                ("obs_lag", C.c_int32), ("require_budget", C.c_int32), ("table", C.c_void_p),
                ("table_R", C.c_int32), ("seed", C.c_uint64)]


class Tables(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("n_days", C.c_void_p), ("B0", C.c_void_p), ("W", C.c_void_p),
        ("fips_to_weather", C.c_void_p), ("sim_cnt", C.c_void_p),
        ("T", C.c_int32), ("S_w", C.c_int32), ("Y", C.c_int32), ("S", C.c_int32), ("n_samples", C.c_int32),
        ("n_obs", C.c_int32), ("obs_slot", C.c_int32 * ROW_FLOATS), ("slot_heat_qi", C.c_int32),
        ("sim_ptr", C.c_void_p), ("sim_idx", C.c_void_p), ("slot_alerts_2wks", C.c_int32),
        ("gate_bits", C.c_void_p), ("gate_words", C.c_int32),
    ]


STATE_FIELDS = ["t", "used", "streak", "hist14", "last_actual", "at_budget", "budget", "n_days", "county_w",
                "year_i", "coef_col", "sample", "sticky_budget", "episode_no", "finished", "episode_return"]


class StateView(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in STATE_FIELDS]


class W2AError(RuntimeError):
    pass


_lib = None


def lib_path() -> str:
    # W2A_LIB: load another build of the library instead of the in-tree one (tools/mutation_fuzz.py: deliberately broken
    # builds that the call-sequence fuzz has to catch). Never set in normal use.
    return os.environ.get("W2A_LIB") or _build.LIB


def load(build_if_missing: bool = True):
    """Load libw2a.so (building it with hipcc first when the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if build_if_missing and not os.environ.get("W2A_LIB") and _build.needs_build():
        try:
            _build.build_lib()
        except Exception as e:  # noqa: BLE001
            # never fall back to an older library when its sources have changed: the ABI version check below would
            # accept a stale .so whose kernels differ from the sources in the tree
            raise W2AError(f"libw2a.so is out of date (or missing) and could not be rebuilt: {e}") from e
    if not os.path.exists(path):
        raise W2AError(f"{path} not found: run `python -m weather2alert_amd.build` (there is no CPU fallback)")
    lib = C.CDLL(path)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    lib.w2a_abi_version.restype = C.c_int
    lib.w2a_abi_version.argtypes = []
    lib.w2a_last_error.restype = C.c_char_p
    lib.w2a_last_error.argtypes = []
    lib.w2a_state_bytes.restype = C.c_size_t
    lib.w2a_state_bytes.argtypes = [i64]
    lib.w2a_create.restype = C.c_int
    lib.w2a_create.argtypes = [C.POINTER(Tables), i64, i64, vp, C.c_size_t, vp, C.POINTER(vp)]
    lib.w2a_destroy.restype = None
    lib.w2a_destroy.argtypes = [vp]
    lib.w2a_reset.restype = C.c_int
    lib.w2a_reset.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.w2a_reset_device_rng.restype = C.c_int
    lib.w2a_reset_device_rng.argtypes = [vp, u64, i32, C.c_int, i32, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    lib.w2a_set_autoreset.restype = C.c_int
    lib.w2a_set_autoreset.argtypes = [vp, u64, i32, C.c_int, i32, C.c_int, C.c_int]
    lib.w2a_step.restype = C.c_int
    lib.w2a_step.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp]
    lib.w2a_get_state.restype = C.c_int
    lib.w2a_get_state.argtypes = [vp, C.POINTER(StateView), vp]
    lib.w2a_read_status.restype = C.c_int
    lib.w2a_read_status.argtypes = [vp, C.POINTER(i32), vp]
    lib.w2a_sort_workspace_bytes.restype = C.c_size_t
    lib.w2a_sort_workspace_bytes.argtypes = [i64]
    lib.w2a_sort_episodes.restype = C.c_int
    lib.w2a_sort_episodes.argtypes = [vp, vp, C.c_size_t, vp]
    lib.w2a_reset_device_rng_sorted.restype = C.c_int
    lib.w2a_reset_device_rng_sorted.argtypes = [vp, u64, i32, C.c_int, i32, C.c_int, C.c_int, C.c_int, vp, vp, C.c_size_t, vp]
    lib.w2a_group_workspace_bytes.restype = C.c_size_t
    lib.w2a_group_workspace_bytes.argtypes = [i64, C.c_int32, C.c_int32]
    lib.w2a_group_by_column.restype = C.c_int
    lib.w2a_group_by_column.argtypes = [vp, vp, C.c_size_t, vp]
    lib.w2a_posterior_mean_reward.restype = C.c_int
    lib.w2a_posterior_mean_reward.argtypes = [vp, vp, C.c_int, vp, vp]
    lib.w2a_set_posterior_kernel.restype = C.c_int
    lib.w2a_set_posterior_kernel.argtypes = [vp, C.c_int]
    lib.w2a_rollout_mfma_workspace_bytes.restype = C.c_size_t
    lib.w2a_rollout_mfma_workspace_bytes.argtypes = [i64, i64, i32, i32]
    lib.w2a_rollout_mfma_prepare.restype = C.c_int
    lib.w2a_rollout_mfma_prepare.argtypes = [vp, vp, C.c_size_t, vp]
    lib.w2a_query.restype = C.c_int
    lib.w2a_query.argtypes = [vp, C.c_int]
    lib.w2a_invalidate.restype = C.c_int
    lib.w2a_invalidate.argtypes = [vp, vp]
    lib.w2a_observe.restype = C.c_int
    lib.w2a_observe.argtypes = [vp, vp, vp]
    lib.w2a_set_semantics.restype = C.c_int
    lib.w2a_set_semantics.argtypes = [vp, C.c_uint32]
    lib.w2a_rollout.restype = C.c_int
    lib.w2a_rollout.argtypes = [vp, C.POINTER(Policy), i32, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    lib.w2a_rollout_order_workspace_bytes.restype = C.c_size_t
    lib.w2a_rollout_order_workspace_bytes.argtypes = [i64, i64]
    lib.w2a_rollout_order.restype = C.c_int
    lib.w2a_rollout_order.argtypes = [vp, vp, C.c_size_t, vp]
    lib.w2a_rollout_order_attach.restype = C.c_int
    lib.w2a_rollout_order_attach.argtypes = [vp, vp, C.c_size_t]
    lib.w2a_rollout_posterior_mean.restype = C.c_int
    lib.w2a_rollout_posterior_mean.argtypes = [vp, C.POINTER(Policy), i32, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    lib.w2a_policy_actions.restype = C.c_int
    lib.w2a_policy_actions.argtypes = [vp, C.POINTER(Policy), vp, vp, vp, vp, vp, i32, vp]
    if lib.w2a_abi_version() != ABI_VERSION:
        raise W2AError(f"libw2a.so ABI {lib.w2a_abi_version()} != {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != OK:
        msg = load().w2a_last_error().decode("utf-8", "replace")
        raise W2AError(f"{what} failed ({rc}): {msg}")
