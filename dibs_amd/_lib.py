"""Loader for libdibs_hip.so (the HIP kernels + C ABI of include/dibs_hip.h).

There is NO CPU fallback: if the library is missing or no gfx950 device is present, every entry point
raises.  ``build()`` compiles the library in-tree with hipcc (cross-compiles without a GPU)."""
import ctypes as C
import os
import subprocess

from ._abi import DibsConfig

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("DIBS_HIP_LIB") or os.path.join(_CSRC, "_build", "libdibs_hip.so")  # env: explicit library path
_lib = None

EXPORTS = [
    "dibs_last_error", "dibs_abi_version", "dibs_engine_create", "dibs_engine_destroy", "dibs_engine_set_data",
    "dibs_engine_init_particles", "dibs_engine_set_state", "dibs_engine_get_state", "dibs_engine_run",
    "dibs_engine_step_local", "dibs_engine_step_update", "dibs_engine_gather_elems_per_rank", "dibs_engine_sync",
    "dibs_engine_read_buffer", "dibs_engine_buffer_bytes", "dibs_engine_theta_size", "dibs_engine_set_profiling",
    "dibs_engine_get_timers", "dibs_engine_reset_timers", "dibs_engine_get_counters", "dibs_score_graphs",
    "dibs_engine_plane_elems_per_rank", "dibs_engine_export_values", "dibs_engine_step_local_grads", "dibs_engine_step_update_planes",
    "dibs_engine_kmat_values", "dibs_engine_eval_gradients", "dibs_comm_unique_id", "dibs_engine_comm_init",
    "dibs_engine_comm_destroy", "dibs_engine_run_sharded", "dibs_engine_gather_particles", "dibs_engine_ipc_export",
    "dibs_engine_comm_init_ipc", "dibs_engine_flag_fallbacks", "dibs_engine_debug_drop_next_flag",
]


class DibsHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libdibs_hip.so for gfx950 (hipcc --offload-arch=gfx950)."""
    r = subprocess.run(["make", "-j8", "-C", _CSRC], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout + r.stderr)
    if r.returncode:
        raise DibsHipError("building libdibs_hip.so failed")
    return LIB_PATH


def load():
    """dlopen the library and set the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DibsHipError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    # One HIP runtime per process: torch ships its own libamdhip64 / libhsa-runtime64 (same SONAME as /opt/rocm's).  Loaded
    # first, it is the copy libdibs_hip.so binds to as well; the other order leaves two runtimes in the process and torch
    # then finds "no ROCm-capable device".  (torch is the host layer's plumbing for streams and torch.distributed.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.dibs_last_error.restype = C.c_char_p
    lib.dibs_abi_version.restype = i32
    lib.dibs_engine_create.argtypes = [C.POINTER(DibsConfig), vp, C.POINTER(vp)]
    lib.dibs_engine_destroy.argtypes = [vp]
    lib.dibs_engine_set_data.argtypes = [vp, vp, vp, vp]
    lib.dibs_engine_init_particles.argtypes = [vp, vp]
    lib.dibs_engine_set_state.argtypes = [vp] + [vp] * 6
    lib.dibs_engine_get_state.argtypes = [vp] + [vp] * 6
    lib.dibs_engine_run.argtypes = [vp, i32, i32]
    lib.dibs_engine_step_local.argtypes = [vp, i32, vp]
    lib.dibs_engine_step_update.argtypes = [vp, i32, vp]
    lib.dibs_engine_gather_elems_per_rank.argtypes = [vp]
    lib.dibs_engine_gather_elems_per_rank.restype = i64
    lib.dibs_engine_plane_elems_per_rank.argtypes = [vp]
    lib.dibs_engine_plane_elems_per_rank.restype = i64
    lib.dibs_engine_export_values.argtypes = [vp, vp]
    lib.dibs_engine_step_local_grads.argtypes = [vp, i32, vp]
    lib.dibs_engine_kmat_values.argtypes = [vp, vp, vp]
    lib.dibs_engine_step_update_planes.argtypes = [vp, i32, vp, vp]
    lib.dibs_engine_sync.argtypes = [vp]
    lib.dibs_engine_read_buffer.argtypes = [vp, i32, vp, i64]
    lib.dibs_engine_buffer_bytes.argtypes = [vp, i32]
    lib.dibs_engine_buffer_bytes.restype = i64
    lib.dibs_engine_theta_size.argtypes = [vp]
    lib.dibs_engine_theta_size.restype = i64
    lib.dibs_engine_set_profiling.argtypes = [vp, i32]
    lib.dibs_engine_get_timers.argtypes = [vp, vp, vp, i32]
    lib.dibs_engine_reset_timers.argtypes = [vp]
    lib.dibs_engine_get_counters.argtypes = [vp, vp, i32]
    lib.dibs_engine_eval_gradients.argtypes = [vp, i32] + [vp] * 7
    lib.dibs_comm_unique_id.argtypes = [vp]
    lib.dibs_engine_comm_init.argtypes = [vp, vp, i32]
    lib.dibs_engine_comm_destroy.argtypes = [vp]
    lib.dibs_engine_ipc_export.argtypes = [vp, vp]
    lib.dibs_engine_comm_init_ipc.argtypes = [vp, vp]
    lib.dibs_engine_flag_fallbacks.argtypes = [vp]
    lib.dibs_engine_debug_drop_next_flag.argtypes = [vp]
    lib.dibs_engine_run_sharded.argtypes = [vp, i32, i32, i32]
    lib.dibs_engine_gather_particles.argtypes = [vp, vp, vp]
    lib.dibs_score_graphs.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp]
    for name in EXPORTS:
        if name not in ("dibs_last_error", "dibs_abi_version", "dibs_engine_gather_elems_per_rank", "dibs_engine_plane_elems_per_rank",
                        "dibs_engine_buffer_bytes", "dibs_engine_theta_size"):
            getattr(lib, name).restype = i32
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise DibsHipError(load().dibs_last_error().decode())
