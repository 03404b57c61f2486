"""
ctypes binding of ``libgraphinvent_amd.so`` (the C ABI declared in ``include/graphinvent_amd.h``).

The shared object is built in-tree by ``graphinvent_amd/csrc/Makefile`` (``hipcc
--offload-arch=gfx950``) — see ``__graft_entry__.build()``.  There is deliberately NO fallback: if
the library is missing, fails to load, or lacks a symbol, importing the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

# torch must be imported BEFORE the shared object is dlopen'ed: torch ships its own
# libamdhip64.so, and the kernels here are enqueued on torch's streams, so both must resolve to the
# same HIP runtime instance (loading ours first binds /opt/rocm's copy and every launch then
# fails with hipErrorNoDevice).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgraphinvent_amd.so")
CSRC = os.path.join(_HERE, "csrc")

GI_MAX_GROUPS = 8
GI_MAX_NODES = 128
EPI_BIAS, EPI_SELU, EPI_DSELU, EPI_ACCUM, GEMM_SPLITK = 1, 2, 4, 8, 16
EPI_MULACT = 64
GEMM_BF3A, GEMM_BF3B_F32 = 256, 512       # with GEMM_BF3: A is an image too / B is the plain fp32 matrix
AMAX_WORDS = 2048     # floats of an amax cell (include/graphinvent_amd.h GI_AMAX_WORDS)
GEMM_T128 = 2048      # with GEMM_BF3 | GEMM_X2, weight-gradient layout: the 128 x 128-tile kernel (gi_gemm_b3v.hip)
GEMM_X2 = 1024        # with GEMM_BF3 and fp32 operands: two scaled fp16 planes per operand, three products (needs a_amax / b_amax)
GEMM_BF3 = 128        # GI_GEMM_BF3: B is a gi_bf3_pack image; the launch runs as bf16x3 splits on the bf16 MFMA pipe
KIND_GGNN, KIND_ATTGGNN = 0, 1
BWD_ALL, BWD_READOUT, BWD_PASSES = 0, 1, 2
BWD_PREPACKED, BWD_NO_X2 = 0x100, 0x200        # OR-ed into the phase (GI_BWD_PREPACKED, GI_BWD_NO_X2)
RUN_PREPACK_BWD, RUN_NO_X2 = 1, 2              # gi_ggnn_forward_ex flags (GI_RUN_*)
X2_GUARD_WORDS = 4                             # GI_X2_GUARD_WORDS
COUNTS = 24          # GI_COUNTS
ABI_VERSION = 18
#: bumped by code that rewrites model weights through raw pointers (optim.FusedAdam.step,
#: dp.DataParallel.broadcast_parameters): invalidates gnn.mpnn's pass-0 row cache
WEIGHTS_EPOCH = [0]     # GI_ABI_VERSION
DTYPE_F32, DTYPE_I8 = 0, 1

vp = C.c_void_p
ci = C.c_int
cll = C.c_longlong


class CompactLayout(C.Structure):
    _fields_ = [(n, ci) for n in ("total_ints", "counts", "type_off", "cidx", "node_mask",
                                  "slot_of", "seg_off", "src_off", "type_off0", "scratch", "dims")]


class GemmParams(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("bias", vp), ("act", vp),
                ("a_idx", vp), ("b_idx", vp), ("grp_off", vp),
                ("M", ci), ("N", ci), ("K", ci),
                ("lda", ci), ("ldb", ci), ("ldc", ci), ("ldact", ci),
                ("flags", ci), ("a_major", ci), ("b_major", ci), ("tm", ci), ("tn", ci),
                ("ngroups", ci), ("nsplit", ci), ("max_group_rows", ci), ("ones_col", ci),
                ("c_split_stride", cll),
                ("Bg", vp * GI_MAX_GROUPS), ("biasg", vp * GI_MAX_GROUPS),
                ("Cg", vp * GI_MAX_GROUPS), ("gsplit", ci * GI_MAX_GROUPS),
                ("m_dev", vp), ("k_dev", vp), ("a_amax", vp), ("b_amax", vp), ("c_amax", vp),
                ("x2_guard", vp), ("x2_guard_host", vp)]


CHAIN_MAXL, CHAIN_MAXW = 8, 256       # GI_CHAIN_MAXL, GI_CHAIN_MAXW


class ChainLayer(C.Structure):
    _fields_ = [("W", vp * GI_MAX_GROUPS), ("bias", vp * GI_MAX_GROUPS), ("out", vp), ("ldo", ci),
                ("act", vp), ("ldact", ci), ("K", ci), ("N", ci), ("out_amax", vp)]


class ChainParams(C.Structure):
    _fields_ = [("layer", ChainLayer * CHAIN_MAXL), ("nlayers", ci), ("X", vp), ("ldx", ci),
                ("x_idx", vp), ("grp_off", vp), ("ngroups", ci), ("group_rows", ci * GI_MAX_GROUPS),
                ("rows", ci), ("backward", ci), ("image", vp), ("image_stride", cll), ("skip_flag", vp),
                ("tile_rows_dev", vp), ("x2_wamax", vp), ("x_amax", vp), ("x2_rows32", ci)]


class ReduceDesc(C.Structure):
    _fields_ = [("slabs", vp), ("dW", vp), ("db", vp), ("slab_stride", cll),
                ("n_slabs", ci), ("N", ci), ("K", ci), ("ld", ci)]


class AbsmaxDesc(C.Structure):            # gi_absmax_desc
    _fields_ = [("x", vp), ("rows", ci), ("cols", ci), ("ld", ci), ("out", vp)]


class Bf3PackDesc(C.Structure):
    """gi_bf3_pack_desc"""
    _fields_ = [("W", vp), ("rows", ci), ("cols", ci), ("ld", ci), ("transpose", ci), ("image", vp), ("as_f32", ci)]


class Graph(C.Structure):
    """gi_graph: the compacted graph as the fused model calls take it."""
    _fields_ = [("S", ci), ("E", ci), ("U", ci), ("gfix", vp), ("u_src", vp), ("in_perm", vp),
                ("mu_off", vp), ("mu_dst", vp), ("mu_slot", vp), ("out_perm", vp),
                ("Ut", C.POINTER(ci)), ("D0", ci), ("ldc0", ci), ("d_src", vp), ("cmat", vp),
                ("e2d", vp), ("cls_off", vp), ("cls_edges", vp), ("bounded", ci), ("p0_cache", vp),
                ("x2_guard", vp), ("x2_guard_host", vp), ("wcache", vp), ("wcache_valid", ci)]


class GgnnDims(C.Structure):
    _fields_ = [(n, ci) for n in ("B", "N", "Fn", "Fe", "H", "M", "G", "A", "C", "passes",
                                  "enn_depth", "enn_hidden", "att_depth", "att_hidden",
                                  "emb_depth", "emb_hidden", "mlp1_depth", "mlp1_hidden",
                                  "mlp2_depth", "mlp2_hidden")] + [("big_positive", C.c_float)] + \
               [(n, ci) for n in ("kind", "eatt_depth", "eatt_hidden", "dropout")] + \
               [(n, C.c_float) for n in ("drop_enn", "drop_eatt", "drop_att", "drop_emb", "drop_mlp1",
                                         "drop_mlp2")] + [("drop_seed", C.c_ulonglong)]


class DropoutParams(C.Structure):                       # gi_dropout_params
    _fields_ = [("seed", C.c_ulonglong), ("id", C.c_uint), ("thresh", C.c_uint),
                ("a", C.c_float), ("b_keep", C.c_float), ("b_drop", C.c_float)]


# name -> (restype, argtypes); every symbol include/graphinvent_amd.h declares
SIGNATURES = {
    "gi_abi_version": (ci, []),
    "gi_compact_layout": (ci, [ci, ci, ci, C.POINTER(CompactLayout)]),
    "gi_compact_count": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, vp]),
    "gi_compact_count_ex": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, ci, vp]),
    "gi_dropout_setup": (ci, [C.c_double, C.c_ulonglong, C.c_uint, C.POINTER(DropoutParams)]),
    "gi_alpha_dropout_fwd": (ci, [vp, ci, ci, ci, cll, C.POINTER(DropoutParams), vp]),
    "gi_dropout_mask": (ci, [C.POINTER(DropoutParams), ci, ci, vp, ci, vp]),
    "gi_seg_sum_dselu_f": (ci, [vp, ci, vp, vp, ci, ci, vp, ci, cll, vp]),
    "gi_selu_bwd_rows_f": (ci, [vp, ci, vp, vp, ci, vp, ci, ci, ci, cll, vp]),
    "gi_compress_slots_f": (ci, [vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, ci, cll, vp]),
    "gi_compact_fill": (ci, [vp, ci, ci, ci, ci, ci, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci,
                             ci, ci, vp, vp, ci, vp, vp]),
    "gi_compact_class_csr": (ci, [vp, ci, ci, vp, vp, vp]),
    "gi_compact_bound": (ci, [vp, ci, ci, ci, ci, ci, vp, vp]),
    "gi_absmax": (ci, [vp, ci, vp]),
    "gi_x2_weight_guard": (ci, [vp, ci, vp, vp, vp]),
    "gi_host_flag_create": (ci, [C.POINTER(vp), C.POINTER(vp)]),
    "gi_host_flag_destroy": (ci, [vp]),
    "gi_b3p_enable": (ci, [ci]),
    "gi_x2_enable": (ci, [ci]),
    "gi_b3v_enable": (ci, [ci]),
    "gi_class_sum_dselu": (ci, [vp, vp, ci, vp, vp, ci, ci, vp, vp, ci, vp]),
    "gi_gemm": (ci, [C.POINTER(GemmParams), vp]),
    "gi_gemm_batch": (ci, [C.POINTER(GemmParams), ci, vp]),
    "gi_gemm_config": (ci, [ci, ci]),
    "gi_mlp_chain": (ci, [C.POINTER(ChainParams), ci, vp]),
    "gi_mlp_chain_config": (ci, [ci, ci, ci, vp]),
    "gi_mlp_chain_pack": (ci, [C.POINTER(ChainParams), ci, vp]),
    "gi_mlp_chain_image_floats": (cll, [C.POINTER(ChainParams)]),
    "gi_seg_sum": (ci, [vp, ci, vp, vp, ci, ci, vp, ci, ci, vp]),
    "gi_seg_softmax_fwd": (ci, [vp, vp, ci, vp, vp, ci, ci, vp, ci, vp]),
    "gi_seg_softmax_bwd": (ci, [vp, vp, ci, vp, vp, ci, ci, vp, ci, vp, vp, ci, vp]),
    "gi_seg_sum_dselu": (ci, [vp, ci, vp, vp, ci, ci, vp, ci, vp]),
    "gi_slab_sum_dselu": (ci, [vp, ci, cll, ci, ci, ci, vp, ci, vp]),
    "gi_slab_epilogue": (ci, [vp, ci, cll, ci, ci, ci, ci, vp, vp, ci, vp, ci, vp]),
    "gi_selu_bwd_rows": (ci, [vp, ci, vp, vp, ci, vp, ci, ci, ci, vp]),
    "gi_gru_gates_fwd": (ci, [vp, vp, ci, vp, vp, ci, vp, ci, ci, ci, vp]),
    "gi_gru_gates_bwd": (ci, [vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, ci, vp]),
    "gi_gather_readout_fwd": (ci, [vp, vp, ci, vp, vp, ci, ci, ci, C.c_float,
                                   vp, ci, vp, ci, vp, ci, vp]),
    "gi_gather_readout_bwd": (ci, [vp, vp, ci, vp, vp, ci, ci, ci, ci, C.c_float,
                                   vp, ci, vp, ci, vp, ci, vp, vp]),
    "gi_gather_readout_bwd_f": (ci, [vp, vp, ci, vp, vp, ci, ci, ci, ci, C.c_float,
                                     vp, ci, vp, ci, vp, ci, vp, cll, vp]),
    "gi_expand_slots": (ci, [vp, ci, vp, ci, ci, ci, vp, ci, vp]),
    "gi_compress_slots": (ci, [vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, ci, vp]),
    "gi_colsum": (ci, [vp, ci, ci, ci, vp, vp, vp]),
    "gi_colsum_multi": (ci, [vp, ci, vp]),
    "gi_reduce_slabs": (ci, [C.POINTER(ReduceDesc), ci, vp]),
    "gi_adam_step": (ci, [vp, vp, vp, vp, cll, C.c_double, C.c_double, C.c_double, C.c_double,
                          C.c_double, ci, vp]),
    "gi_kl_loss": (ci, [vp, ci, vp, ci, ci, ci, ci, vp, vp, ci, vp, vp]),
    "gi_scale_by_scalar": (ci, [vp, cll, vp, vp]),
    "gi_prof_enable": (ci, [ci]),
    "gi_prof_collect": (ci, [vp, vp, vp, vp]),
    "gi_prof_pipes": (ci, [vp, vp, vp]),
    "gi_sample_actions": (ci, [vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp]),
    "gi_side_stream_create": (ci, [C.POINTER(vp)]),
    "gi_side_stream_destroy": (ci, [vp]),
    "gi_ggnn_num_params": (ci, [C.POINTER(GgnnDims)]),
    "gi_ggnn_workspace_floats": (cll, [C.POINTER(GgnnDims), ci, ci, ci, ci]),
    "gi_p0_cache_words": (cll, [C.POINTER(GgnnDims)]),
    "gi_bf3_enable": (ci, [ci]),
    "gi_bf3_image_elems": (cll, [ci, ci]),
    "gi_bf3_pack": (ci, [C.POINTER(Bf3PackDesc), ci, vp]),
    "gi_ggnn_slab_floats": (cll, [C.POINTER(GgnnDims), ci, ci, C.POINTER(ci)]),
    "gi_ggnn_hx0_offset": (cll, [C.POINTER(GgnnDims), ci, ci, ci, ci]),
    "gi_ggnn_ldhx": (ci, [C.POINTER(GgnnDims)]),
    "gi_ggnn_ws_query": (ci, [C.POINTER(GgnnDims), ci, ci, ci, ci, C.c_char_p, ci, ci,
                              C.POINTER(cll), C.POINTER(ci)]),
    "gi_ggnn_forward": (ci, [C.POINTER(GgnnDims), C.POINTER(vp), C.POINTER(Graph), vp, vp, ci, vp]),
    "gi_ggnn_forward_ex": (ci, [C.POINTER(GgnnDims), C.POINTER(vp), C.POINTER(Graph), vp, vp, ci, vp, vp, ci]),
    "gi_ggnn_backward": (ci, [C.POINTER(GgnnDims), C.POINTER(vp), C.POINTER(Graph), vp, vp, vp, ci,
                              vp, ci, C.POINTER(vp), vp, vp]),
    "gi_ggnn_backward_phase": (ci, [C.POINTER(GgnnDims), C.POINTER(vp), C.POINTER(Graph), vp, vp, vp,
                                    ci, vp, ci, C.POINTER(vp), vp, vp, ci]),
    "gi_ggnn_first_readout_param": (ci, [C.POINTER(GgnnDims)]),
    "gi_ggnn_wcache_floats": (cll, [C.POINTER(GgnnDims)]),
    "gi_gru_forward": (ci, [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, ci, ci, ci, vp]),
    "gi_fuse_flags": (ci, []),
    "gi_gru_gates_bwd_ex": (ci, [vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, ci, vp, vp, ci, vp, vp,
                                 vp]),
    "gi_selu_bwd_cols3_f": (ci, [vp, ci, vp, ci, cll, ci, ci, vp, ci, ci, vp, ci, ci, vp, ci, vp]),
    "gi_expand_slots2": (ci, [vp, ci, ci, vp, ci, vp, ci, ci, vp, ci, vp, ci, ci, vp]),
    "gi_compress_slots2_f": (ci, [vp, ci, ci, vp, ci, vp, ci, vp, ci, ci, vp, ci, vp, ci, vp, ci, ci, ci,
                                  cll, vp]),
}
FUSE_GATES_V4, FUSE_DH_SCATTER, FUSE_TIER2_DSELU, FUSE_SLOTS = 1, 2, 4, 8  # GI_FUSE_*

_lib = None


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into the in-tree shared object."""
    res = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libgraphinvent_amd.so failed (hipcc --offload-arch=gfx950)")
    return LIB_PATH


def load() -> C.CDLL:
    """Load the HIP library; raises (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the MI355X HIP extension is required (there is no CPU or "
            "eager fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {CSRC}`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)             # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.gi_abi_version() != ABI_VERSION:
        raise RuntimeError("libgraphinvent_amd.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    if rc < 0:
        kind = {-1: "GI_EINVAL (bad argument)", -2: "GI_ELIMIT (compiled-in limit exceeded)"}.get(
            rc, f"error {rc}")
        raise RuntimeError(f"graphinvent_amd {what}: {kind}")
    raise RuntimeError(f"graphinvent_amd {what}: HIP error {rc}")
