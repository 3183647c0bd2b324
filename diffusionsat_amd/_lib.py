"""ctypes binding of the C-ABI in ``include/dsat.h``.

The HIP extension is the product: if ``libdsat.so`` is missing or fails to load this
raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSAT_LIB") or os.path.join(HERE, "libdsat.so")  # DSAT_LIB: A/B builds of the same ABI

N_MLP = 5
MLP_NAMES = ("variables_query", "lit_mlp", "clause_mlp", "update_gate", "variables_output")


class Layout(C.Structure):
    _fields_ = [("feature_maps", C.c_int32), ("query_maps", C.c_int32), ("logit_maps", C.c_int32),
                ("n_layers", C.c_int32 * N_MLP), ("hidden", C.c_int32 * N_MLP)]


class Tap(C.Structure):
    """``dsat_tap`` of include/dsat.h (debug tap of one message-passing round)."""
    _fields_ = [("round", C.c_int32)] + [(k, C.c_void_p) for k in (
        "query", "literals", "clauses_loss4", "clause_messages", "variables_loss_all", "new_clause_value",
        "variables_grad", "variables_loss", "new_variables", "variables", "clause_state")]


class DsatError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("dsat status %d: %s" % (status, message))
        self.status = status


_vp, _i32, _i64, _u64, _u32, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32, C.c_float
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/dsat.h declaration by declaration
SIGNATURES = {
    "dsat_last_error": (C.c_char_p, []),
    "dsat_version": (_i32, []),
    "dsat_default_layout": (C.c_int, [C.POINTER(Layout)]),
    "dsat_layout_blob_floats": (_i64, [C.POINTER(Layout)]),
    "dsat_layout_layer_dims": (C.c_int, [C.POINTER(Layout), _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "dsat_graph_create": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _pp]),
    "dsat_graph_create_coo": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _pp]),
    "dsat_graph_create_coo_in": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _pp]),
    "dsat_mlp_create": (C.c_int, [_i32, _vp, _vp, _i64, _pp]),
    "dsat_mlp_workspace_floats": (_i64, [_vp, _i32]),
    "dsat_mlp_forward": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "dsat_mlp_destroy": (None, [_vp]),
    "dsat_pair_norm_workspace_floats": (_i64, [_i32, _i32, _i32]),
    "dsat_pair_norm": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _f32, _i32, _vp, _vp, _vp]),
    "dsat_graph_destroy": (None, [_vp]),
    "dsat_graph_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i32)]),
    "dsat_arena_create": (C.c_int, [_i64, _i64, _pp]),
    "dsat_arena_reset": (C.c_int, [_vp]),
    "dsat_arena_usage": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "dsat_arena_destroy": (None, [_vp]),
    "dsat_graph_create_in": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _pp]),
    "dsat_session_create_in": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _pp]),
    "dsat_weights_load": (C.c_int, [_vp, _i64, C.POINTER(Layout), _pp]),
    "dsat_weights_destroy": (None, [_vp]),
    "dsat_session_create": (C.c_int, [_vp, _vp, _i32, _pp]),
    "dsat_session_destroy": (None, [_vp]),
    "dsat_session_set_graph_replay": (C.c_int, [_vp, _i32]),
    "dsat_session_set_supervised": (C.c_int, [_vp, _i32]),
    "dsat_diffusion_step": (C.c_int, [_vp, _f32, _vp, _vp, _vp, _u64, _u32, _i32, _i32,
                                       _vp, _vp, _vp, _vp, _vp, _vp]),
    "dsat_diffusion_step_self": (C.c_int, [_vp, _f32, _vp, _vp, _vp, _vp, _u64, _u32, _i32, _i32,
                                            _vp, _vp, _vp, _vp, _vp, _vp]),
    "dsat_host_solve": (C.c_int, [_i32, _i32, _vp, _vp, C.c_int64, _vp, _vp]),
    "dsat_host_gen_sr": (C.c_int, [_u64, _i32, _i32, C.c_double, C.c_double, C.c_double, C.POINTER(_i32),
                                   C.POINTER(_i32), _vp, _i32, _vp, _i32]),
    "dsat_sample": (C.c_int, [_vp, _i32, _i32, _u64, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                              C.POINTER(_f32), _vp]),
    "dsat_sample_self": (C.c_int, [_vp, _i32, _i32, _u64, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                              C.POINTER(_f32), _vp]),
    "dsat_op_mlp": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp]),
    "dsat_op_mlp_ex": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    "dsat_bench_mlp": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_double), _vp]),
    "dsat_op_clause_gather": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "dsat_op_literal_gather": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dsat_op_pair_norm": (C.c_int, [_vp, _i32, _vp, _vp, _vp]),
    "dsat_op_is_batch_sat": (C.c_int, [_vp, _vp, C.POINTER(_i32), _vp]),
    "dsat_op_is_graph_sat": (C.c_int, [_vp, _vp, _vp, _vp]),
    "dsat_session_packed_bytes": (_i64, [_vp, _vp]),
    "dsat_session_split_mask": (_i32, [_vp]),
    "dsat_session_set_throughput": (C.c_int, [_vp, _i32]),
    "dsat_op_pack_assignment": (C.c_int, [_vp, _vp, _vp, _vp]),
    "dsat_op_rng": (C.c_int, [_vp, _u64, _u32, _i32, _vp, _vp, _vp, _vp]),
    "dsat_session_set_profile": (C.c_int, [_vp, _i32]),
    "dsat_session_set_retire": (C.c_int, [_vp, _i32]),
    "dsat_session_set_tap": (C.c_int, [_vp, C.POINTER(Tap)]),
    "dsat_session_read_stamps": (C.c_int, [_vp, C.POINTER(_i64), _i64]),
    "dsat_session_profile": (C.c_int, [_vp, _i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                        C.POINTER(_i64)]),
}

_lib = None


def load():
    """Load libdsat.so (once) and attach the prototypes.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "diffusionsat_amd: HIP extension %s is missing. Build it with "
            "`python -m diffusionsat_amd.build` (needs hipcc); there is no CPU fallback." % LIB_PATH)
    # PyTorch first: it ships its own HIP runtime, and a process must end up with ONE -- if libdsat.so pulled in
    # /opt/rocm's copy before torch loaded its own, HIP calls from the extension later found no device
    # ("hipErrorNoDevice" in a build()-then-smoke() process).  Loaded after torch, the extension binds to the
    # runtime torch uses (same soname), which is also what makes torch's streams and pointers valid in it.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        msg = load().dsat_last_error()
        raise DsatError(status, msg.decode("utf-8", "replace") if msg else "")


def default_layout() -> Layout:
    lay = Layout()
    check(load().dsat_default_layout(C.byref(lay)))
    return lay
