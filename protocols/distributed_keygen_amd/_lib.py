"""ctypes binding of libmxpaillier.so (include/mxpaillier.h).  No CPU fallback exists: if the
HIP library is missing, importing this module fails loudly."""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_int64, c_void_p, POINTER
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libmxpaillier.so"

MX_OK = 0
ERRORS = {-1: "MX_ERR_ARG", -2: "MX_ERR_SIZE", -3: "MX_ERR_MODULUS", -4: "MX_ERR_WORKSPACE", -5: "MX_ERR_HIP"}

class NsquarePlan(ctypes.Structure):
    """mx_nsquare_plan (include/mxpaillier.h)."""

    _fields_ = [
        ("d_plan", c_void_p), ("plan_bytes", c_int64), ("limbs_n", ctypes.c_int32), ("n_bits", ctypes.c_int32),
        ("exp_bits", ctypes.c_int32), ("window", ctypes.c_int32), ("ntape", ctypes.c_int32),
        ("n_sqr", ctypes.c_int32), ("n_mul", ctypes.c_int32), ("geometries", ctypes.c_int32),
        ("n_slot_reads", ctypes.c_int32), ("n_slot_writes", ctypes.c_int32),
    ]


class CombinePlan(ctypes.Structure):
    """mx_combine_plan (include/mxpaillier.h)."""

    _fields_ = [
        ("d_plan", c_void_p), ("plan_bytes", c_int64), ("limbs", ctypes.c_int32), ("limbs2", ctypes.c_int32),
        ("n_bits", ctypes.c_int32), ("n2_bits", ctypes.c_int32),
    ]


MX_PLAN_FIXED_WINDOW = 1      # include/mxpaillier.h

_P4 = [POINTER(c_int)] * 4

# name -> (restype, argtypes); every symbol include/mxpaillier.h declares
SYMBOLS = {
    "mx_version": (c_int, []),
    "mx_error_string": (c_char_p, [c_int]),
    "mx_last_hip_error": (c_char_p, []),
    "mx_powmod_workspace_bytes": (c_int64, [c_int, c_int, c_int64, c_int64]),
    "mx_powmod_shared": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_powmod_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_powmod_nsquare_workspace_bytes": (c_int64, [c_int, c_int, c_int64]),
    "mx_powmod_nsquare": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_powmod_shared_lpl": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "mx_powmod_multi_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "mx_nsquare_plan_bytes": (c_int64, [c_int, c_int]),
    "mx_powmod_nsquare_prepare": (c_int, [POINTER(NsquarePlan), c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "mx_powmod_nsquare_prepare_ex": (c_int, [POINTER(NsquarePlan), c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "mx_powmod_nsquare_run_workspace_bytes": (c_int64, [POINTER(NsquarePlan), c_int64]),
    "mx_powmod_nsquare_run": (c_int, [POINTER(NsquarePlan), c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "mx_nsquare_launch_shape": (c_int, [c_int, c_int64, c_int, c_int, *_P4, POINTER(c_int)]),
    "mx_nsquare_launch_instance": (c_int, [c_int, c_int64, c_int, c_int, *[POINTER(c_int)] * 5]),
    "mx_nsquare_latency_form": (c_int, [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int64)]),
    "mx_nsquare_launch_split": (c_int, [c_int, c_int64, POINTER(c_int64), *[POINTER(c_int)] * 4]),
    "mx_nsquare_launch_timesliced": (c_int, [c_int, c_int64, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "mx_nsquare_pieces_shape": (c_int, [c_int, c_int64, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "mx_combine_plan_bytes": (c_int64, [c_int, c_int]),
    "mx_combine_prepare": (c_int, [POINTER(CombinePlan), c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "mx_combine_run": (c_int, [POINTER(CombinePlan), c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "mx_biprime_verdict_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_jacobi_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p]),
    "mx_jacobi_dev_range": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mx_nsquare_geometry_for": (c_int, [c_int, c_int64, c_int, *_P4]),
    "mx_powmod_geometry_for": (c_int, [c_int, c_int64, c_int64, c_int, *_P4]),
    "mx_powmod_launch_form": (c_int, [c_int, c_int64, c_int64, c_int, POINTER(c_int), POINTER(c_int)]),
    "mx_modinv_workspace_bytes": (c_int64, [c_int]),
    "mx_modinv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_field_workspace_bytes": (c_int64, [c_int, c_int]),
    "mx_fma_mod": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_lincomb_mod": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_sieve_workspace_bytes": (c_int64, [c_int, c_int]),
    "mx_sieve": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_combine_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int64]),
    "mx_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_verdict_workspace_bytes": (c_int64, [c_int, c_int, c_int64, c_int64]),
    "mx_biprime_verdict": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_mulmod_workspace_bytes": (c_int64, [c_int]),
    "mx_mulmod_shared": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_jacobi_workspace_bytes": (c_int64, [c_int, c_int64]),
    "mx_jacobi": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "mx_select_first": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "mx_selftest_lanes": (c_int, [c_void_p]),
    "mx_debug_knob": (c_int, [c_int, c_int]),
    "mx_spin": (c_int, [c_int64, c_void_p]),
    "mx_clock_probe": (c_int, [c_int64, c_void_p, c_void_p]),
    "mx_stream_create_cu_slice": (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    "mx_stream_destroy": (c_int, [c_void_p]),
    "mx_geometry": (c_int, [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "mx_profile": (c_int, [c_int]),
    "mx_profile_collect": (c_int, [POINTER(ctypes.c_double), POINTER(c_int)]),
    "mx_nsquare_geometry": (c_int, [c_int, c_int64, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
}


class MxError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, code: int, where: str, detail: str = "") -> None:
        self.code = code
        super().__init__(f"{where}: {ERRORS.get(code, code)}{(' — ' + detail) if detail else ''}")


def load() -> ctypes.CDLL:
    global LIB_PATH
    override = os.environ.get("MX_LIBRARY")      # developer knob: A/B runs of differently built kernels
    if override:
        LIB_PATH = Path(override)
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing — build it with `python -m protocols.distributed_keygen_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    # The HIP runtime must be the one PyTorch ships (torch/lib/libamdhip64.so, same SONAME as the
    # system ROCm's): whichever is loaded first serves the whole process, and device memory and
    # streams come from torch.  Loading this library before torch would bind both to /opt/rocm's
    # runtime, which does not see the GPU on the test machines.
    import torch  # noqa: F401

    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(code: int, where: str) -> int:
    if code < 0:
        detail = ""
        if code == -5:
            detail = lib().mx_last_hip_error().decode()
        raise MxError(code, where, detail)
    return code
