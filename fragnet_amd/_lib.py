"""ctypes binding of libfragnet_hip.so (C-ABI in include/fragnet_hip.h).

There is no fallback: if the library is missing or a call fails, this raises.  The shared object is
built in-tree by ``__graft_entry__.build()`` / ``python -m fragnet_amd.build`` into fragnet_amd/lib/.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfragnet_hip.so")

ABI_VERSION = 12
FN_D = 128
FN_MAX_TASKS = 16
FN_MAX_EDGE_K = 8
FN_MAX_PART = 4096
ROLE_PLAIN, ROLE_DST, ROLE_SRC = 0, 1, 2
FN_EINVAL, FN_EUNSUPPORTED, FN_ETOOMANY = -1, -2, -3

i32, i64, u64, f32, vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p
ip = C.POINTER(C.c_int)


class CsrTask(C.Structure):
    _fields_ = [("key", vp), ("other_key", vp), ("n_real", i64), ("n_loops", i64), ("n_seg", i64),
                ("item_base", i64), ("seg_base", i64), ("role", i32), ("partner", i32)]


FN_MAX_SPACES = 8


class MolLayout(C.Structure):
    _fields_ = [("offsets", vp), ("n_spaces", i32), ("pad_", i32), ("n_mols", i64),
                ("node_space", i32 * FN_MAX_TASKS), ("item_space", i32 * FN_MAX_TASKS), ("counts_dev", vp),
                ("cap", i64 * FN_MAX_SPACES), ("pad_mod", i64 * FN_MAX_SPACES), ("max_per_mol", i64 * FN_MAX_SPACES),
                ("pad_hint", i64 * FN_MAX_SPACES)]


class EdgeTerm(C.Structure):
    _fields_ = [("mode", i32), ("K", i32), ("d_e", i32), ("mid_off", i32),
                ("s_sorted", vp), ("x_sorted", vp), ("embW", vp), ("embb", vp), ("x_src", vp)]


class ActEpilogue(C.Structure):
    _fields_ = [("y", vp), ("p", f32), ("relu", i32), ("seed", u64), ("offset", u64), ("offset_dev", vp)]


class GatPlan(C.Structure):
    _fields_ = [("rowptr_d", vp), ("eid_d", vp), ("src_d", vp), ("rowptr_s", vp), ("dst_s", vp), ("dpos_s", vp),
                ("inv_d", vp), ("spos_d", vp), ("pos_base_d", i32), ("pos_base_s", i32), ("n", i64), ("m", i64), ("m_real", i64)]


class SegPlan(C.Structure):
    _fields_ = [("rowptr", vp), ("perm", vp), ("index", vp), ("n_seg", i64), ("n_items", i64), ("pos_base", i32), ("pad_", i32)]


LAYER_FIELDS = ("proj_b_w", "proj_b_b", "proj_a_w", "proj_a_b", "proj_fb_w", "proj_fb_b", "emb_b_w", "emb_b_b",
                "emb_fb_w", "emb_fb_b", "a_b", "a", "f", "f_a_b")
FN_MAX_LAYERS = 8


class LayerWeights(C.Structure):
    _fields_ = [(name, vp) for name in LAYER_FIELDS]


class Tower(C.Structure):
    _fields_ = [("x", vp), ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("w3", vp), ("b3", vp), ("h1", vp), ("h2", vp), ("out", vp),
                ("M", i64), ("g_out", vp), ("g_x", vp), ("g_w1", vp), ("g_b1", vp), ("g_w2", vp), ("g_b2", vp), ("g_w3", vp), ("g_b3", vp)]


FN_MAX_TOWERS = 4


class StageField(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("n_real", i64), ("cap", i64), ("width", i32), ("kind", i32),
                ("pad_hi", i64), ("pad_mod", i64)]


FN_MAX_STAGE_FIELDS = 40
STAGE_ROWS, STAGE_IDS, STAGE_COLS, STAGE_MASK, STAGE_COUNT, STAGE_BUMP, STAGE_ZERO, STAGE_OFFSETS = 0, 1, 2, 3, 4, 5, 6, 7
PLAN_PREZEROED = 1


class MseTask(C.Structure):
    _fields_ = [("out", vp), ("y", vp), ("w", vp), ("g_out", vp), ("B", i64), ("T", i32), ("scale_idx", i32), ("coef", f32), ("pad_", f32)]


class CollateField(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("rows", i64), ("src_rows", i64), ("width_words", i32), ("space", i32), ("kind", i32),
                ("rebase_space", i32), ("src_global", i32), ("max_seg_rows", i32)]


FN_MAX_COLLATE_FIELDS = 24
COLLATE_ROWS, COLLATE_BATCH, COLLATE_IDS = 0, 1, 2


class AdamSlice(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", i64), ("lr_dev", vp), ("step_dev", vp),
                ("beta1", f32), ("beta2", f32), ("eps", f32), ("weight_decay", f32), ("launched", i32), ("pad_", i32)]


class SmallDw(C.Structure):
    _fields_ = [("g", vp), ("x", vp), ("dW", vp), ("db", vp), ("loss_part", vp), ("loss", vp), ("n_part", i64), ("M", i64), ("K", i64), ("C", i64)]


LOSS_MSE, LOSS_BCE = 0, 1
SMALL_LINEAR_LOSS_MAX_K = 1024


class Encoder(C.Structure):
    _fields_ = [("n_layers", i32), ("heads", i32), ("k_atom0", i32), ("k_bond0", i32), ("k_fbond0", i32), ("k_fattr", i32),
                ("training", i32), ("variant", i32), ("drop_p", f32), ("pad2_", f32), ("seed", u64), ("offset", u64), ("offset_dev", vp),
                ("N", i64), ("E", i64), ("F", i64), ("EF", i64),
                ("bond", GatPlan), ("atom", GatPlan), ("fbond", GatPlan), ("frag", GatPlan), ("a2f", SegPlan),
                ("x_atoms", vp), ("bond_nodes", vp), ("fbond_nodes", vp), ("cos_sorted", vp), ("fattr_sorted", vp),
                ("cos_raw", vp), ("fattr_raw", vp),
                ("w", LayerWeights * FN_MAX_LAYERS), ("ws", vp), ("ws_floats", i64),
                ("mol_atoms", SegPlan), ("mol_frags", SegPlan), ("n_mols", i64), ("counts_dev", vp), ("status", vp),
                ("mol_contiguous", i32), ("no_backward", i32), ("pooled", vp), ("g_pooled", vp), ("adam_rider", vp)]


# name -> argtypes; every function returns int (0 ok / <0 argument error / >0 hipError_t) unless noted.
SIGNATURES = {
    "fn_abi_version": [],
    "fn_last_error": [],
    "fn_set_tuning": [C.c_int, C.c_int],
    "fn_debug_set_stamps": [vp, i64],
    "fn_debug_set_profile_events": [vp],
    "fn_plan_layout": [C.POINTER(CsrTask), C.c_int, C.POINTER(i64), C.POINTER(i64)],
    "fn_plan_build": [C.POINTER(CsrTask), C.c_int, vp, vp, vp, vp, vp, vp, i32, vp],
    "fn_plan_build_mol": [C.POINTER(CsrTask), C.c_int, C.POINTER(MolLayout), vp, vp, vp, vp, vp, vp, i32, vp],
    "fn_node_scalars_f32": [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, i64, C.c_int, vp],
    "fn_gat_fwd_f32": [vp, vp, vp, vp, C.c_int, C.POINTER(EdgeTerm), C.POINTER(GatPlan), f32, vp, vp, vp, vp, vp, C.c_int,
                       C.POINTER(ActEpilogue), C.c_int, vp],
    "fn_gat_bwd_one_f32": [vp, vp, vp, vp, vp, C.POINTER(EdgeTerm), vp, C.c_int, C.c_int, C.c_int, C.POINTER(GatPlan), f32,
                           vp, vp, vp, vp, ip, vp, ip, C.c_int, vp, C.c_int, vp],
    "fn_gat_gsd_f32": [vp, C.POINTER(GatPlan), vp, vp, vp, C.c_int, vp],
    "fn_gat_cu_f32": [vp, vp, vp, vp, f32, vp, vp, i64, C.c_int, vp],
    "fn_gat_bwd_dst_f32": [vp, vp, vp, C.POINTER(EdgeTerm), C.POINTER(GatPlan), f32, vp, vp, vp, vp, vp, ip, C.c_int, vp],
    "fn_gat_bwd_src_f32": [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(GatPlan), vp, vp, ip, C.c_int, vp],
    "fn_tower_fwd_f32": [C.POINTER(Tower), C.c_int, vp],
    "fn_tower_bwd_ws": [C.POINTER(Tower), C.c_int],
    "fn_tower_bwd_f32": [C.POINTER(Tower), C.c_int, vp, vp],
    "fn_gat_bwd_finalize_f32": [vp, C.c_int, vp, C.c_int, C.POINTER(EdgeTerm), vp, C.c_int, C.c_int, C.c_int, vp, vp, vp,
                                C.c_int, vp],
    "fn_attn_by_src_f32": [vp, C.POINTER(GatPlan), vp, C.c_int, vp],
    "fn_row_dots_sorted_f32": [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(GatPlan), vp, vp],
    "fn_row_dots_sorted_bwd_f32": [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(GatPlan), vp, vp, ip, vp],
    "fn_sort_edge_attr_f32": [vp, C.c_int, C.POINTER(GatPlan), vp, vp],
    "fn_sort_edge_attr_src_f32": [vp, C.c_int, C.POINTER(GatPlan), vp, vp],
    "fn_colsum_f32": [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp],
    "fn_transpose_w_f32": [vp, C.c_int, vp, vp],
    "fn_linear128_f32": [vp, C.c_int, vp, vp, vp, i64, C.POINTER(ActEpilogue), vp],
    "fn_linear128_wgrad_ws": [i64, C.c_int],
    "fn_linear128_wgrad_f32": [vp, vp, C.c_int, i64, vp, vp, vp, vp],
    "fn_encoder_fused_tail": [C.POINTER(Encoder)],
    "fn_encoder_ws_floats": [C.POINTER(Encoder)],
    "fn_encoder_bwd_ws_floats": [C.POINTER(Encoder)],
    "fn_encoder_rng_blocks": [C.POINTER(Encoder)],
    "fn_encoder_forward": [C.POINTER(Encoder), vp, vp, vp, vp, vp],
    "fn_encoder_backward": [C.POINTER(Encoder), vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(LayerWeights), vp, i64, vp],
    "fn_segment_sum_f32": [vp, i64, vp, vp, i32, vp, i64, i64, i64, vp],
    "fn_gather_rows_f32": [vp, vp, vp, i64, i64, vp],
    "fn_segment_softmax_f32": [vp, vp, vp, i32, vp, i64, i64, vp],
    "fn_segment_softmax_bwd_f32": [vp, vp, vp, vp, i32, vp, i64, i64, vp],
    "fn_dropout_act_f32": [vp, vp, i64, f32, u64, u64, vp, C.c_int, vp],
    "fn_dropout_act_bwd_f32": [vp, vp, vp, i64, f32, u64, u64, vp, C.c_int, vp],
    "fn_adam_f32": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, vp],
    "fn_adam_dev_f32": [vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, vp, vp],
    "fn_edge_concat_f32": [vp, vp, vp, vp, i64, vp],
    "fn_stage_padded": [C.POINTER(StageField), C.c_int, vp],
    "fn_pool_cat_f32": [vp, vp, C.POINTER(SegPlan), C.POINTER(SegPlan), vp, vp],
    "fn_pool_cat_bwd_f32": [vp, vp, vp, vp, vp, i64, i64, vp],
    "fn_masked_mse_f32": [vp, vp, vp, i64, C.c_int, vp, vp, vp],
    "fn_bond_graph_ws": [i64, i64],
    "fn_bond_graph_count": [vp, vp, i64, i64, i64, C.c_int, vp, vp, vp],
    "fn_bond_graph_fill": [vp, vp, i64, i64, i64, C.c_int, vp, vp, i64, vp],
    "fn_masked_bce_f32": [vp, vp, vp, i64, C.c_int, vp, vp, vp],
    "fn_masked_mse_multi_ws": [C.c_int],
    "fn_masked_mse_multi_f32": [C.POINTER(MseTask), C.c_int, vp, vp, vp, vp],
    "fn_gate_colsum_ws": [i64, i64],
    "fn_gate_colsum_f32": [vp, vp, vp, vp, i64, i64, f32, vp, vp],
    "fn_small_linear_f32": [vp, vp, vp, vp, i64, i64, i64, i64, vp],
    "fn_dense_fwd_f32": [vp, vp, vp, vp, i64, i64, i64, C.POINTER(ActEpilogue), vp],
    "fn_dense_bwd_f32": [vp, vp, vp, vp, f32, vp, vp, i64, i64, i64, i64, vp],
    "fn_small_linear_bwd_ws": [i64, i64, i64],
    "fn_collate_store": [C.POINTER(CollateField), C.c_int, vp, vp, C.c_int, i64, vp, vp],
    "fn_small_linear_loss_ws": [i64],
    "fn_small_linear_loss_f32": [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, f32, vp, i64, i64, i64, i64, vp],
    "fn_dense_bwd_tail_f32": [vp, vp, vp, vp, f32, vp, vp, i64, i64, i64, i64, C.POINTER(SmallDw), vp],
    "fn_small_linear_bwd_f32": [vp, vp, vp, vp, vp, vp, i64, i64, i64, f32, vp, vp],
}

_lib = None


class FragnetHipError(RuntimeError):
    pass


def load():
    """Loads the shared library once; raises FragnetHipError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FragnetHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). fragnet_amd has no CPU/PyTorch fallback for its kernels.")
    # a library older than its sources is refused, not used: the digest of the .hip / .inc / .h files is stored next to it
    from . import build
    if os.path.isdir(os.path.join(build.HERE, "csrc")) and build.stale():
        raise FragnetHipError(
            f"{LIB_PATH} does not match the sources it was built from (content digest differs): rebuild it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or fragnet_amd.build.build_lib()).")
    # torch ships its own libamdhip64; it must be the HIP runtime this library binds to (same streams,
    # same allocations), so torch is imported before the dlopen.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        if name == "fn_last_error":
            fn.restype = C.c_char_p
        elif name.endswith("_ws") or name.endswith("_ws_floats"):
            fn.restype = i64
        elif name == "fn_encoder_rng_blocks":
            fn.restype = u64
        else:
            fn.restype = C.c_int
    if lib.fn_abi_version() != ABI_VERSION:
        raise FragnetHipError(f"ABI version mismatch: library {lib.fn_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().fn_last_error()
        raise FragnetHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def call(name: str, *args):
    check(getattr(load(), name)(*args), name)
