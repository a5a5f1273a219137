"""ctypes binding of the C-ABI library (include/gvi_hip.h).  No fallback: if the HIP extension is
missing or does not load, importing the compute API raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# GVI_LIB_PATH: an A/B build of the same sources (tools/build_variant.py); the default is the in-tree library
LIB_PATH = os.environ.get("GVI_LIB_PATH") or os.path.join(HERE, "libgvi_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int8_p = C.POINTER(C.c_int8)
c_void_pp = C.POINTER(C.c_void_p)
# gvi_allgather_fn: int (*)(void* user, const void* send, void* recv, int64_t count, void* hip_stream)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

# name -> argtypes; every function returns int status except the two string getters.
# This table is also what tests/test_abi.py checks against include/gvi_hip.h.
SIGNATURES = {
    "gvi_ctx_create": [C.c_int, C.c_int, c_void_pp],
    "gvi_ctx_destroy": [C.c_void_p],
    "gvi_ctx_set_stream": [C.c_void_p, C.c_void_p],
    "gvi_ctx_sync": [C.c_void_p],
    "gvi_spgh_count": [C.c_int, C.c_int, C.POINTER(C.c_int64)],
    "gvi_spgh_nodes": [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p],
    "gvi_chain_set": [C.c_void_p, C.c_int, C.c_int],
    "gvi_factors_add": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                        C.c_void_p, C.POINTER(C.c_int)],
    "gvi_factors_add_table": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                              C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)],
    "gvi_debug_cost_log": [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double)],
    "gvi_factors_set_table": [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p],
    "gvi_factors_set_sdf2d": [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p],
    "gvi_table_file_list": [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "gvi_table_file_read": [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "gvi_table_file_write": [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p],
    "gvi_ngd_trial_publish": [C.c_void_p],
    "gvi_ngd_trial_wait": [C.c_void_p, C.c_void_p],
    "gvi_ngd_spec_gradients_local": [C.c_void_p],
    "gvi_ngd_spec_gradients_finish": [C.c_void_p],
    "gvi_ngd_accept_spec": [C.c_void_p],
    "gvi_ngd_set_update_rule": [C.c_void_p, C.c_int],
    "gvi_prox_gradients": [C.c_void_p, C.c_double],
    "gvi_prox_trial": [C.c_void_p, C.c_double, C.c_void_p],
    "gvi_prox_step": [C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "gvi_factors_set_closed_form": [C.c_void_p, C.c_int, C.c_int],
    "gvi_factors_set_sdf3d": [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "gvi_factors_set_arm": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                            C.c_void_p, C.c_void_p],
    "gvi_factors_set_temperature": [C.c_void_p, C.c_int, C.c_void_p],
    "gvi_factors_info": [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                         C.POINTER(C.c_int64)],
    "gvi_moments": [C.c_void_p, C.c_int] + [C.c_void_p] * 5,
    "gvi_moments_dev": [C.c_void_p, C.c_int] + [C.c_void_p] * 5,
    "gvi_raw_moments": [C.c_void_p, C.c_int] + [C.c_void_p] * 5,
    "gvi_costs": [C.c_void_p, C.c_int] + [C.c_void_p] * 3,
    "gvi_costs_dev": [C.c_void_p, C.c_int] + [C.c_void_p] * 3,
    "gvi_expand": [C.c_void_p, C.c_int] + [C.c_void_p] * 3,
    "gvi_moments_from_psi": [C.c_void_p, C.c_int] + [C.c_void_p] * 6,
    "gvi_bt_assemble": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "gvi_bt_solve": [C.c_void_p] + [C.c_void_p] * 4,
    "gvi_bt_logdet": [C.c_void_p] + [C.c_void_p] * 3,
    "gvi_bt_marginals": [C.c_void_p] + [C.c_void_p] * 4,
    "gvi_gather_marginals": [C.c_void_p, C.c_int] + [C.c_void_p] * 5,
    "gvi_ngd_init": [C.c_void_p] + [C.c_void_p] * 3,
    "gvi_ngd_cost": [C.c_void_p, c_double_p],
    "gvi_ngd_factor_costs": [C.c_void_p, C.c_int, C.c_void_p],
    "gvi_ngd_gradients": [C.c_void_p],
    "gvi_ngd_trial": [C.c_void_p, C.c_double, c_double_p],
    "gvi_ngd_accept": [C.c_void_p],
    "gvi_ngd_step": [C.c_void_p, C.c_double, C.c_int, c_double_p, C.POINTER(C.c_int), c_double_p, C.POINTER(C.c_int)],
    "gvi_ngd_run": [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)],
    "gvi_ngd_set_mode": [C.c_void_p, C.c_int, C.c_int],
    "gvi_ngd_counters": [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int],
    "gvi_ngd_gradients_local": [C.c_void_p],
    "gvi_ngd_gradients_finish": [C.c_void_p],
    "gvi_ngd_cost_local": [C.c_void_p],
    "gvi_ngd_cost_finish": [C.c_void_p, c_double_p],
    "gvi_ngd_trial_local": [C.c_void_p, C.c_double],
    "gvi_ngd_trial_finish": [C.c_void_p, c_double_p],
    "gvi_ngd_exchange": [C.c_void_p, C.c_int, c_void_pp, C.POINTER(C.c_int64)],
    "gvi_dist_unique_id": [C.c_void_p],
    "gvi_dist_init_rccl": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "gvi_dist_init_callback": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "gvi_dist_info": [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "gvi_ngd_get_state": [C.c_void_p] + [C.c_void_p] * 5,
    "gvi_ngd_get_gradients": [C.c_void_p] + [C.c_void_p] * 6,
    "gvi_profile_enable": [C.c_void_p, C.c_int],
    "gvi_profile_last": [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)],
    "gvi_profile_geometry": [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)],
    "gvi_profile_stages": [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)],
    "gvi_set_variant": [C.c_void_p, C.c_int],
    "gvi_set_option": [C.c_void_p, C.c_char_p, C.c_int],
}
STRING_GETTERS = {"gvi_version": [], "gvi_last_error": [C.c_void_p]}

_lib = None


def _torch_runtime_first():
    """PyTorch's ROCm wheels bundle their own libamdhip64 / libhsa-runtime64; the library links libamdhip64.so.7 by
    SONAME.  If torch is imported FIRST the dynamic loader resolves that SONAME to torch's already-loaded copy and the
    process holds one HIP runtime.  If the library is loaded first it brings in the system ROCm runtime, a later
    `import torch` adds a second one, and torch.cuda then reports "No HIP GPUs are available" (measured on the MI355X
    pool, tools/torch_after_probe.py).  The Python binding exists beside torch (device memory, streams,
    torch.distributed), so it imports torch before the dlopen whenever torch is installed; GVI_NO_TORCH_PRELOAD=1 skips
    that for torch-free processes that want the 1.5 s back.  C / C++ hosts of the C ABI are not affected."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("GVI_NO_TORCH_PRELOAD") == "1":
        return
    if importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass


def load():
    """Load gaussianvi_amd/libgvi_hip.so (built by gaussianvi_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m gaussianvi_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _torch_runtime_first()
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = C.c_int
    for name, args in STRING_GETTERS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_char_p
    _lib = lib
    return lib
