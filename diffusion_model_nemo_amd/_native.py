"""ctypes binding of libdmn_hip.so (C-ABI declared in include/dmn.h).

There is no CPU fallback: if the library is missing or cannot be loaded the first use raises.
"""
import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdmn_hip.so")
# Tuning aid: DMN_LIB_VARIANT=<tag> loads lib/libdmn_hip_<tag>.so instead (another build of the same sources, for A/B runs on
# one box). There is still no fallback: a variant that is missing raises like the default library would.
if os.environ.get("DMN_LIB_VARIANT"):
    LIB_PATH = os.path.join(_HERE, "lib", f"libdmn_hip_{os.environ['DMN_LIB_VARIANT']}.so")

DMN_MAX_MULTS = 8
TABLE_COLS = 12
F32, BF16 = 0, 1


class DmnUnetCfg(C.Structure):
    _fields_ = [("dim", C.c_int), ("n_mults", C.c_int), ("dim_mults", C.c_int * DMN_MAX_MULTS), ("channels", C.c_int),
                ("out_dim", C.c_int), ("groups", C.c_int), ("heads", C.c_int), ("dim_head", C.c_int),
                ("tail_norm", C.c_int), ("with_time_emb", C.c_int), ("dtype", C.c_int), ("num_classes", C.c_int)]


class DmnTensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]


_vp, _i, _f, _u64, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_int64, C.c_size_t

# name -> (restype, argtypes); the list every test checks against include/dmn.h
SIGNATURES = {
    "dmn_last_error": (C.c_char_p, []),
    "dmn_version": (C.c_char_p, []),
    "dmn_unet_create": (_i, [C.POINTER(DmnUnetCfg), C.POINTER(_vp)]),
    "dmn_unet_destroy": (None, [_vp]),
    "dmn_unet_load_weights": (_i, [_vp, C.POINTER(DmnTensorDesc), _i]),
    "dmn_unet_set_class_labels": (_i, [_vp, _vp]),
    "dmn_unet_set_plan_batch": (_i, [_vp, _i]),
    "dmn_unet_handoff_giveups": (_i, [_vp, C.POINTER(C.c_uint)]),
    "dmn_test_probe_read": (_i, [_vp, _i, _vp, _i]),
    "dmn_test_probe_site": (_i, [_vp, _i, _vp]),
    "dmn_unet_chains_concurrent": (_i, [_vp, _vp]),
    "dmn_test_step_trace": (_i, [_vp, _i, _vp]),
    "dmn_bench_launch_floor": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "dmn_unet_workspace_bytes": (_i, [_vp, _i, _i, _i, C.POINTER(_sz)]),
    "dmn_unet_forward": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "dmn_ddpm_step": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _u64, _i64, _vp]),
    "dmn_ddpm_sample_loop": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _u64, _i64, _vp, _vp, _sz, _vp, _i]),
    "dmn_sde_predictor_step": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _u64, _i64, _vp]),
    "dmn_sde_corrector_step": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _u64, _i64, _vp, _vp]),
    "dmn_sde_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "dmn_sde_corrector_step_static": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _u64, _i64, _vp]),
    "dmn_pc_sample_loop": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _f, _u64, _i64, _vp, _vp, _vp,
                                _sz, _vp, _i]),
    "dmn_bpd_loop": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _u64, _i64, _vp, _i, _vp, _f, _f, _vp, _sz, _vp, _i]),
    "dmn_sample_workspace_bytes": (_i, [_vp, _i, _i, _i, C.POINTER(_sz)]),
    "dmn_loop_last_ms": (_i, [_vp, C.POINTER(_f)]),
    "dmn_bench_dominant_kernel": (_i, [_vp, _i, _i, _i, _i, _vp, _sz, _vp, C.POINTER(_f), C.POINTER(C.c_double)]),
    "dmn_unet_probe_dominant": (_i, [_vp, _i]),
    "dmn_unet_probe_result": (_i, [_vp, C.POINTER(_f), C.POINTER(_f), C.POINTER(_i), C.POINTER(C.c_double)]),
    "dmn_bench_stream": (_i, [_i, _vp, _vp, _sz, _i, _vp, C.POINTER(_f)]),
    "dmn_streams_share_queue": (_i, [_vp, _vp, C.POINTER(_i)]),
    "dmn_bench_mfma": (_i, [_i, _i, _i, _i, _vp, _vp, C.POINTER(_f), C.POINTER(C.c_double)]),
    "dmn_train_create": (_i, [C.POINTER(DmnUnetCfg), _i, C.POINTER(_vp)]),
    "dmn_train_destroy": (None, [_vp]),
    "dmn_train_set_class_labels": (_i, [_vp, _vp]),
    "dmn_train_num_tensors": (_i, [_vp]),
    "dmn_train_flat_size": (_i64, [_vp]),
    "dmn_train_tensor_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i64 * 4)]),
    "dmn_train_workspace_bytes": (_i, [_vp, _i, _i, _i, C.POINTER(_sz)]),
    "dmn_train_loss_and_grads": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "dmn_adamw_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _f, _f, _vp, _vp]),
    "dmn_fill_normal": (_i, [_vp, _i, _i, _u64, C.c_uint32, _i64, _f, _vp]),
    "dmn_unet_set_sample_chains": (_i, [_vp, _i, _i]),
    "dmn_ddpm_loop_chains": (_i, [_vp, _i]),
    "dmn_test_attn_train": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dmn_test_attention": (_i, [_i, _vp, _vp, _i, _i, _i, _vp]),
    "dmn_test_unet_trace": (_i, [_vp, _vp, _i]),
    "dmn_test_unet_trace_count": (_i, [_vp]),
    "dmn_test_unet_trace_stop": (_i, [_vp, _i]),
    "dmn_test_unet_trace_poison": (_i, [_vp, C.c_uint32]),
    "dmn_test_unet_trace_name": (C.c_char_p, [_vp, _i]),
    "dmn_test_set_conv_debug": (_i, [_vp]),
    "dmn_test_linattn_block": (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i,
                                    C.POINTER(C.c_uint), _vp]),
    "dmn_test_linattn": (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i), _vp, _sz, _vp]),
    "dmn_test_conv_block_tail": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp,
                                      _i, C.POINTER(_i), _i, _i, _i, _i, C.POINTER(C.c_uint), _vp]),
    "dmn_test_conv2d": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                             _vp, _vp, _i, _i, _i, _vp, _i, C.POINTER(_i), _i, _i, _vp]),
}

_lib = None


class DmnError(RuntimeError):
    pass


def lib():
    """Loads the shared library on first use; raises if it is not built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DmnError(f"{LIB_PATH} is missing: build it with `python -m diffusion_model_nemo_amd.build` "
                           "(the HIP kernels are the only implementation of this path)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int):
    if rc != 0:
        raise DmnError(f"libdmn_hip error {rc}: {lib().dmn_last_error().decode()}")


def loop_last_ms(handle) -> float:
    ms = C.c_float()
    check(lib().dmn_loop_last_ms(handle, C.byref(ms)))
    return float(ms.value)


def ptr(t) -> Optional[int]:
    """Raw device (or host) address of a torch tensor / numpy array, or None."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return t.data_ptr()


def current_stream_handle(device=None) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def make_tensor_descs(params: Dict[str, np.ndarray]):
    """params: name -> contiguous float32 ndarray. Returns (ctypes array, keepalive list)."""
    n = len(params)
    arr = (DmnTensorDesc * n)()
    keep = []
    for i, (name, a) in enumerate(params.items()):
        a = np.ascontiguousarray(a, dtype=np.float32)
        keep.append(a)
        bname = name.encode()
        keep.append(bname)
        arr[i].name = bname
        arr[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].ndim = a.ndim
        for d in range(a.ndim):
            arr[i].shape[d] = a.shape[d]
    return arr, keep
