"""ctypes binding of libvlsat_hip.so (C ABI in include/vlsat.h).

The library is loaded from this directory (built in-tree by ``build.py``).  There is no
fallback: if it is missing, ``load()`` raises -- the product path never computes on the CPU.
torch must be imported first so that the HIP runtime already mapped by torch
(``libamdhip64.so.7``) is the one this library binds to (one runtime, one set of streams).
"""
from __future__ import annotations

import ctypes as C
import os
import re

import torch  # noqa: F401  (maps libamdhip64 before our library is loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvlsat_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "vlsat.h")


class VlsatDims(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_heads", C.c_int32), ("dim_atten", C.c_int32),
                ("gcn_aggr", C.c_int32), ("dim_point", C.c_int32), ("n_obj_class", C.c_int32),
                ("n_rel_class", C.c_int32), ("obj_logit_scale", C.c_float), ("use_gcn_edge", C.c_int32),
                ("multi_rel_outputs", C.c_int32), ("feature_transform", C.c_int32)]


class VlsatError(RuntimeError):
    pass


_lib = None
_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

_SIGNATURES = {
    "vlsat_last_error": (C.c_char_p, []),
    "vlsat_version": (C.c_char_p, []),
    "vlsat_create": (C.c_int, [C.POINTER(VlsatDims), C.POINTER(_vp)]),
    "vlsat_destroy": (None, [_vp]),
    "vlsat_load_weight": (C.c_int, [_vp, C.c_char_p, _vp, _sz]),
    "vlsat_finalize_weights": (C.c_int, [_vp]),
    "vlsat_plan_create": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, C.POINTER(_vp)]),
    "vlsat_plan_destroy": (None, [_vp]),
    "vlsat_plan_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_sz), C.POINTER(_i32)]),
    "vlsat_plan_check_graph": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "vlsat_forward": (C.c_int, [_vp] * 9 + [_vp]),
    "vlsat_forward_train": (C.c_int, [_vp] * 12 + [_vp]),
    "vlsat_set_gemm_precision": (C.c_int, [_vp, _i32]),
    "vlsat_set_edge_attention_scope": (C.c_int, [_vp, _i32]),
    "vlsat_profile_enable": (C.c_int, [_vp, _i32]),
    "vlsat_profile_num_classes": (C.c_int, []),
    "vlsat_profile_class_name": (C.c_char_p, [_i32]),
    "vlsat_profile_read": (C.c_int, [_vp, _i32, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double)]),
    "vlsat_k_gemm": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _f32,
                               _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vlsat_k_split_bf16": (C.c_int, [_vp, _sz, _vp, _vp, _vp]),
    "vlsat_k_gemm_planes": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f32,
                                      _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "vlsat_k_gemm_bf16": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f32,
                                    _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vlsat_k_pointnet": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "vlsat_k_flash_attn": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _f32, _vp]),
    "vlsat_k_flash_attn_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _f32, _i32, _i32, _vp]),
    "vlsat_k_layernorm": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "vlsat_k_edge_gate": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                    _i32, _i32, _vp]),
    "vlsat_k_aggregate": (C.c_int, [_vp, _i32, _vp, _i64, _i32, _i32, _vp, _i32, _i32, _vp]),
    "vlsat_k_node_attn": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _f32, _i32, _vp]),
    "vlsat_k_dist_bias": (C.c_int, [_vp, _i32, _vp, _i32, _i32] + [_vp] * 10 + [_vp, _vp]),
    "vlsat_prepare_objects": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "vlsat_fc_edges": (C.c_int, [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp]),
    "vlsat_sample_objects_scratch": (_sz, [_i64, _i32]),
    "vlsat_sample_objects": (C.c_int, [_vp, _i64, _vp, _i32, _i32, C.c_uint64, _vp, _i32, _vp, _vp, _vp, _vp]),
    "vlsat_k_softmax_rows": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "vlsat_eval_ranks": (C.c_int, [_vp] * 6 + [_i32] * 7 + [_f32] + [_vp] * 5 + [_vp]),
    "vlsat_eval_ranks_scratch_floats": (C.c_int64, [_i32] * 3),
    "vlsat_eval_counts": (C.c_int, [_vp] * 10 + [_i32] * 4 + [_vp, _vp]),
    "vlsat_process_val_counts": (C.c_int, [_vp] * 8 + [_i32, _vp, _vp]),
    "vlsat_scene_checksums": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "vlsat_comm_unique_id": (C.c_int, [_vp]),
    "vlsat_comm_init": (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "vlsat_metrics_allreduce": (C.c_int, [_vp, _vp, _i32, _vp]),
    "vlsat_comm_destroy": (None, [_vp]),
    "vlsat_debug_gemm_clock_probe": (C.c_int, [_vp]),
    "vlsat_debug_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "vlsat_debug_stop_after": (C.c_int, [_vp, _i32]),
    "vlsat_debug_read": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "vlsat_debug_buffer": (C.c_int, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_i32),
                                     C.POINTER(_i32)]),
}


def identity(lib_path: str = "") -> dict:
    """What a measurement was taken on: SHA-256 of the shared library that is (or would be) loaded and of the sources it is
    built from (csrc/*.hip, csrc/*.h, include/vlsat.h, build flags) -- the GPU box has no .git, so the source digest is what a
    profile summary and a later bench run can compare; the commit is added when the summary is published (tools/)."""
    import hashlib
    path = lib_path or LIB_PATH
    out = {"lib_sha256": None, "lib_bytes": None, "source_sha256": None}
    if os.path.exists(path):
        out["lib_sha256"] = hashlib.sha256(open(path, "rb").read()).hexdigest()
        out["lib_bytes"] = os.path.getsize(path)
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    for f in sorted(os.listdir(csrc)) + [HEADER_PATH, os.path.join(_HERE, "build.py")]:
        fp = f if os.path.isabs(f) else os.path.join(csrc, f)
        if fp.endswith((".hip", ".h", ".py")):
            h.update(os.path.basename(fp).encode() + b"\0" + open(fp, "rb").read())
    out["source_sha256"] = h.hexdigest()
    return out


def declared_symbols() -> list:
    """Every function include/vlsat.h declares (parsed from the header text)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vlsat_[a-z0-9_]+)\s*\(", txt)))


def load():
    """Load the shared library; raises VlsatError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VlsatError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc, gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        msg = load().vlsat_last_error().decode(errors="replace")
        err = VlsatError(f"libvlsat_hip error {code}: {msg}")
        err.code = code
        raise err


def ptr(t) -> int:
    """Raw device/host address of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream
