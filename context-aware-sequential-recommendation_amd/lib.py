"""ctypes binding of libcastrec.so (the C ABI of include/castrec.h).

No fallback: if the library has not been built (``python -m castrec_amd.build``) importing this
module raises.  The structures below mirror include/castrec.h field by field."""
import ctypes as C
import os

# torch first: it ships its own HIP runtime (libamdhip64).  Loading libcastrec.so before it would pull the
# system ROCm copy into the process as a SECOND runtime, on which kernel launches fail with
# "no ROCm-capable device is detected".  With torch loaded, the soname is already resolved and shared.
import torch  # noqa: F401

from . import PKG_DIR

# CASTREC_TIMELINE=1 selects the instrumented build (python -m castrec_amd.build --timeline); debugging tools only
LIB_PATH = os.path.join(PKG_DIR, "libcastrec_tl.so" if os.environ.get("CASTREC_TIMELINE") == "1" else "libcastrec.so")
if os.environ.get("CASTREC_LIB"):           # diagnostics: another build of the same sources (tools/diag_repro2.py compares builds)
    LIB_PATH = os.path.abspath(os.environ["CASTREC_LIB"])
if not os.path.exists(LIB_PATH):
    raise ImportError("castrec_amd: native library %s not found -- build it with "
                      "`python -m castrec_amd.build` (needs hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
_lib = C.CDLL(LIB_PATH)

c_f = C.c_float
c_i = C.c_int
c_u32 = C.c_uint32
c_p = C.c_void_p

CR_MAX_BATCH = 4
CR_STATE_FLOATS = 16
PREC_F32, PREC_BF16X3, PREC_BF16 = 0, 1, 2
ELT_COPY, ELT_ADD, ELT_DROPOUT, ELT_RELU_BWD, ELT_ROWMASK, ELT_GRADPREP = 0, 1, 2, 3, 4, 5


class Rng(C.Structure):
    _fields_ = [("rate", c_f), ("site", c_u32), ("seed", c_u32), ("step", c_p), ("row_offset", c_u32)]


class EmbedDesc(C.Structure):
    _fields_ = [("ids", c_p), ("table", c_p), ("M", c_i), ("T", c_i), ("D", c_i), ("V", c_i),
                ("zero_pad", c_i), ("scale", c_f), ("pos_table", c_p), ("addend", c_p), ("ld_add", c_i),
                ("drop", Rng), ("mask_ids", c_p), ("out", c_p), ("ld_out", c_i), ("col_off", c_i)]


class EmbedBwdDesc(C.Structure):
    _fields_ = [("f", EmbedDesc), ("table_grad", c_p), ("pos_grad", c_p), ("d_addend", c_p),
                ("slab_stride", c_i), ("n_slabs", c_i), ("out2", c_p)]


class LnDesc(C.Structure):
    _fields_ = [("x", c_p), ("ldx", c_i), ("gamma", c_p), ("beta", c_p), ("y", c_p), ("ldy", c_i),
                ("M", c_i), ("D", c_i), ("eps", c_f), ("x_nonzero", c_p), ("y_nonzero", c_p)]


class LnBwdDesc(C.Structure):
    _fields_ = [("x", c_p), ("ldx", c_i), ("gamma", c_p), ("dy", c_p), ("lddy", c_i), ("dx", c_p), ("lddx", c_i),
                ("accumulate", c_i), ("dgamma", c_p), ("dbeta", c_p), ("slab_stride", c_i), ("n_slabs", c_i),
                ("M", c_i), ("D", c_i), ("eps", c_f)]


class GemmDesc(C.Structure):
    _fields_ = [("A", c_p), ("lda", c_i), ("B", c_p), ("ldb", c_i), ("bias", c_p), ("C", c_p), ("ldc", c_i),
                ("M", c_i), ("N", c_i), ("K", c_i), ("trans_b", c_i), ("relu", c_i), ("drop", Rng),
                ("residual", c_p), ("ldr", c_i), ("mask_ids", c_p), ("accumulate", c_i), ("precision", c_i)]


class WgradDesc(C.Structure):
    _fields_ = [("A", c_p), ("lda", c_i), ("G", c_p), ("ldg", c_i), ("dW", c_p), ("ldw", c_i), ("db", c_p),
                ("M", c_i), ("N", c_i), ("K", c_i), ("precision", c_i)]


class EltDesc(C.Structure):
    _fields_ = [("op", c_i), ("x", c_p), ("ldx", c_i), ("aux", c_p), ("ldaux", c_i), ("y", c_p), ("ldy", c_i),
                ("M", c_i), ("N", c_i), ("drop", Rng), ("mask_ids", c_p), ("accumulate", c_i)]


class AttnDesc(C.Structure):
    _fields_ = [("Q", c_p), ("K", c_p), ("V", c_p), ("ld", c_i), ("k_valid", c_p), ("q_valid", c_p),
                ("residual", c_p), ("ldr", c_i), ("dead_ids", c_p), ("out", c_p), ("ldo", c_i),
                ("attn_weights", c_p), ("B", c_i), ("T", c_i), ("H", c_i), ("d", c_i), ("drop", Rng),
                ("batch_global", c_i), ("row_stats", c_p), ("precision", c_i)]


class AttnBwdDesc(C.Structure):
    _fields_ = [("f", AttnDesc), ("dout", c_p), ("lddo", c_i), ("dQ", c_p), ("dK", c_p), ("dV", c_p),
                ("ldg", c_i), ("stats", c_p), ("delta", c_p), ("dQ_part", c_p)]


class BlockDesc(C.Structure):
    _fields_ = [("M", c_i), ("D", c_i), ("ln1_g", c_p), ("ln1_b", c_p), ("wqkv", c_p), ("bqkv", c_p), ("ln2_g", c_p),
                ("ln2_b", c_p), ("w1", c_p), ("b1", c_p), ("w2", c_p), ("b2", c_p), ("x", c_p), ("q_in", c_p), ("qkv", c_p),
                ("k_valid", c_p), ("q_valid", c_p), ("o", c_p), ("f_in", c_p), ("hid", c_p), ("y", c_p), ("mask_ids", c_p),
                ("drop_ffn1", Rng), ("drop_ffn2", Rng)]


class BlockBwdDesc(C.Structure):
    _fields_ = [("f", BlockDesc), ("dy", c_p), ("d_o", c_p), ("dqkv", c_p), ("dx", c_p), ("dx_accumulate", c_i),
                ("g_ln1_g", c_p), ("g_ln1_b", c_p), ("g_wqkv", c_p), ("g_bqkv", c_p), ("g_ln2_g", c_p), ("g_ln2_b", c_p),
                ("g_w1", c_p), ("g_b1", c_p), ("g_w2", c_p), ("g_b2", c_p), ("slab_stride", c_i), ("n_slabs", c_i),
                ("attn_delta", c_p), ("dq_part", c_p)]


class BlockBwd1Ext(C.Structure):
    _fields_ = [("dy2", c_p), ("dx2", c_p), ("lnf_dy2", c_p), ("d_addend2", c_p)]


class BlockTailDesc(C.Structure):
    _fields_ = [("kind", c_i), ("next", C.POINTER(BlockDesc)), ("lnf_gamma", c_p), ("lnf_beta", c_p), ("out", c_p),
                ("ld_out", c_i), ("col_out", c_i)]


class StackDesc(C.Structure):
    _fields_ = [("n_blocks", c_i), ("blocks", C.POINTER(BlockDesc)), ("attn", C.POINTER(AttnDesc)), ("lnf_gamma", c_p),
                ("lnf_beta", c_p), ("out", c_p), ("ld_out", c_i), ("col_out", c_i), ("embed", C.POINTER(EmbedDesc))]


class HeadDesc(C.Structure):
    _fields_ = [("seq_emb", c_p), ("ld", c_i), ("table", c_p), ("pos", c_p), ("neg", c_p),
                ("M", c_i), ("D", c_i), ("V", c_i), ("state", c_p), ("d_seq_emb", c_p), ("ldd", c_i),
                ("table_grad", c_p), ("pos_logits", c_p), ("neg_logits", c_p), ("coef_out", c_p)]


class IndexLayout(C.Structure):
    _fields_ = [("M", c_i), ("V", c_i), ("T_pos", c_i), ("ng", c_i), ("ent", c_i), ("cap_blocks", c_i), ("cap_occ", c_i), ("bitmap_words", c_i),
                ("off_recs", C.c_int64), ("total_words", C.c_int64)]


class TgradDesc(C.Structure):
    _fields_ = [("index", c_p), ("ring", c_p), ("ring_slots", c_i), ("slot_words", C.c_int64), ("index_off", C.c_int64), ("step", c_p),
                ("lay", IndexLayout), ("rows", c_p), ("rows2", c_p), ("ld_rows", c_i), ("scale", c_f), ("seq_emb", c_p), ("ld_emb", c_i),
                ("coef", c_p), ("D", c_i), ("part_rows", c_p), ("tickets", c_p)]


class AdamDesc(C.Structure):
    _fields_ = [("p", c_p), ("m", c_p), ("v", c_p), ("table_grad", c_p), ("dense_slabs", c_p),
                ("n_table", C.c_int64), ("n_dense", c_i), ("n_slabs", c_i), ("lr", c_f), ("beta1", c_f),
                ("beta2", c_f), ("eps", c_f), ("state", c_p), ("stats", c_p), ("step_snapshot", c_p), ("l2", c_f), ("n_l2", C.c_int64),
                ("lazy_ids", c_p), ("n_lazy_ids", c_i), ("lazy_rows", c_i), ("lazy_D", c_i), ("lazy_flags", c_p), ("slab_counts", c_p),
                ("ids_ring", c_p), ("ids_ring_slots", c_i), ("ids_slot_elems", C.c_int64), ("ids_dst", c_p),
                ("ids_copy_elems", C.c_int64), ("tg", C.POINTER(TgradDesc))]


def _sig(name, restype, argtypes):
    f = getattr(_lib, name)
    f.restype = restype
    f.argtypes = argtypes
    return f


# every symbol include/castrec.h declares (tests/test_abi.py checks this list against the header)
_sig("cr_version", c_i, [])
_sig("cr_last_error", C.c_char_p, [])
_sig("cr_step_begin", c_i, [c_p, c_p])
_sig("cr_ids_ring_next", c_i, [c_p, c_i, C.c_int64, c_p, c_p, c_p])
_sig("cr_embed_fwd", c_i, [C.POINTER(EmbedDesc), c_p])
_sig("cr_embed_bwd", c_i, [C.POINTER(EmbedBwdDesc), c_p])
_sig("cr_layernorm_fwd", c_i, [C.POINTER(LnDesc), c_p])
_sig("cr_layernorm_bwd", c_i, [C.POINTER(LnBwdDesc), c_p])
_sig("cr_gemm_rows", c_i, [C.POINTER(GemmDesc), c_i, c_p])
_sig("cr_gemm_wgrad", c_i, [C.POINTER(WgradDesc), c_i, c_i, c_i, c_p])
_sig("cr_eltwise", c_i, [C.POINTER(EltDesc), c_p])
_sig("cr_attn_fwd", c_i, [C.POINTER(AttnDesc), c_p])
_sig("cr_attn_bwd", c_i, [C.POINTER(AttnBwdDesc), c_p])
_sig("cr_block_ln_qkv_fwd", c_i, [C.POINTER(BlockDesc), c_p])
_sig("cr_block_ln_qkv_fwd_gather", c_i, [C.POINTER(BlockDesc), C.POINTER(EmbedDesc), c_p])
_sig("cr_block_ln_ffn_fwd", c_i, [C.POINTER(BlockDesc), c_p])
_sig("cr_block_ln_ffn_fwd_tail", c_i, [C.POINTER(BlockDesc), C.POINTER(BlockTailDesc), c_p])
_sig("cr_stack_fwd_supported", c_i, [C.POINTER(StackDesc)])
_sig("cr_stack_fwd", c_i, [C.POINTER(StackDesc), c_p])
_sig("cr_block_ln_ffn_bwd", c_i, [C.POINTER(BlockBwdDesc), c_p])
_sig("cr_block_ln_qkv_bwd", c_i, [C.POINTER(BlockBwdDesc), c_p])
_sig("cr_stack_bwd_supported", c_i, [C.POINTER(BlockBwdDesc), c_i, c_i, c_i])
_sig("cr_stack_ffn_bwd", c_i, [C.POINTER(BlockBwdDesc), c_i, c_i, c_i, c_p])
_sig("cr_stack_ffn_bwd_ln", c_i, [C.POINTER(BlockBwdDesc), C.POINTER(LnBwdDesc), c_i, c_i, c_i, c_p])
_sig("cr_stack_ffn_bwd_heads", c_i, [C.POINTER(BlockBwdDesc), C.POINTER(LnBwdDesc), c_i, c_i, c_i, c_i, c_p])
_sig("cr_stack_qkv_bwd", c_i, [C.POINTER(BlockBwdDesc), c_i, c_i, c_i, c_p])
_sig("cr_stack_qkv_bwd_scatter", c_i, [C.POINTER(BlockBwdDesc), C.POINTER(EmbedBwdDesc), c_i, c_i, c_i, c_p])
_sig("cr_stack_block_bwd_supported", c_i, [C.POINTER(BlockBwdDesc), C.POINTER(AttnDesc), c_i, c_i, c_i])
_sig("cr_stack_block_bwd_deal", c_i, [c_i, c_i, C.POINTER(C.c_uint32)])
_sig("cr_stack_block_bwd", c_i, [C.POINTER(BlockBwdDesc), C.POINTER(AttnDesc), C.POINTER(BlockBwd1Ext), C.POINTER(LnBwdDesc), C.POINTER(EmbedBwdDesc),
                                 c_i, c_i, c_i, c_p])
_sig("cr_block_ln_qkv_bwd_scatter", c_i, [C.POINTER(BlockBwdDesc), C.POINTER(EmbedBwdDesc), c_p])
_sig("cr_wide_supported", c_i, [C.POINTER(BlockDesc), c_i])
_sig("cr_wide_ln_qkv_fwd", c_i, [C.POINTER(BlockDesc), c_i, c_p])
_sig("cr_wide_ln_ffn_fwd", c_i, [C.POINTER(BlockDesc), c_i, c_p])
_sig("cr_wide_ln_ffn_fwd_tail", c_i, [C.POINTER(BlockDesc), C.POINTER(BlockTailDesc), c_i, c_p])
_sig("cr_wide_ln_ffn_bwd", c_i, [C.POINTER(BlockBwdDesc), c_p, c_p, c_i, c_i, c_p])
_sig("cr_wide_ln_qkv_bwd", c_i, [C.POINTER(BlockBwdDesc), c_i, c_p])
_sig("cr_head_fwd_bwd", c_i, [C.POINTER(HeadDesc), c_p])
_sig("cr_head_fwd_bwd_ln", c_i, [C.POINTER(HeadDesc), C.POINTER(LnBwdDesc), c_p])
_sig("cr_stack_fwd_head_supported", c_i, [C.POINTER(StackDesc), C.POINTER(HeadDesc), C.POINTER(LnBwdDesc)])
_sig("cr_stack_fwd_head", c_i, [C.POINTER(StackDesc), C.POINTER(HeadDesc), C.POINTER(LnBwdDesc), c_p])
_sig("cr_test_logits", c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p])
_sig("cr_adam_step", c_i, [C.POINTER(AdamDesc), c_p])
_sig("cr_reduce_slabs", c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p])
_sig("cr_l2_penalty", c_i, [c_p, C.c_int64, c_f, c_p, c_p])
_sig("cr_rows_pack", c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p])
_sig("cr_rows_add", c_i, [c_p, c_p, c_i, c_i, c_i, c_p])
_sig("cr_graph_begin", c_i, [c_p])
_sig("cr_graph_end", c_i, [c_p, C.POINTER(c_p)])
_sig("cr_graph_launch", c_i, [c_p, c_p])
_sig("cr_graph_destroy", c_i, [c_p])
_sig("cr_sampler_create", c_p, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.c_double, C.c_double, c_u32, c_i])
_sig("cr_sampler_next", c_i, [c_p] + [c_p] * 8)
_sig("cr_sampler_destroy", None, [c_p])
_sig("cr_tgrad_geometry", c_i, [c_i, C.POINTER(c_i), C.POINTER(c_i)])
_sig("cr_batch_index_layout", c_i, [c_i, c_i, c_i, c_i, c_i, C.POINTER(IndexLayout)])
_sig("cr_index_builder_create", c_p, [c_i, c_i, c_i, c_i, c_i])
_sig("cr_index_build", c_i, [c_p, c_p, c_p, c_p, c_p])
_sig("cr_index_builder_destroy", None, [c_p])
_sig("cr_table_grad", c_i, [C.POINTER(TgradDesc), c_p, c_p])

EXPORTS = ["cr_version", "cr_last_error", "cr_step_begin", "cr_ids_ring_next", "cr_embed_fwd", "cr_embed_bwd", "cr_layernorm_fwd",
           "cr_layernorm_bwd", "cr_gemm_rows", "cr_gemm_wgrad", "cr_eltwise", "cr_attn_fwd", "cr_attn_bwd",
           "cr_block_ln_qkv_fwd", "cr_block_ln_qkv_fwd_gather", "cr_block_ln_ffn_fwd", "cr_block_ln_ffn_fwd_tail", "cr_stack_fwd_supported", "cr_stack_fwd", "cr_block_ln_ffn_bwd", "cr_block_ln_qkv_bwd", "cr_stack_bwd_supported", "cr_stack_ffn_bwd", "cr_stack_ffn_bwd_ln", "cr_stack_ffn_bwd_heads", "cr_stack_qkv_bwd", "cr_stack_qkv_bwd_scatter", "cr_stack_block_bwd_supported", "cr_stack_block_bwd", "cr_stack_block_bwd_deal", "cr_rows_pack", "cr_rows_add", "cr_block_ln_qkv_bwd_scatter",
           "cr_wide_supported", "cr_wide_ln_qkv_fwd", "cr_wide_ln_ffn_fwd", "cr_wide_ln_ffn_fwd_tail", "cr_wide_ln_ffn_bwd", "cr_wide_ln_qkv_bwd",
           "cr_head_fwd_bwd", "cr_head_fwd_bwd_ln", "cr_stack_fwd_head_supported", "cr_stack_fwd_head", "cr_test_logits", "cr_adam_step", "cr_reduce_slabs", "cr_l2_penalty", "cr_graph_begin", "cr_graph_end", "cr_graph_launch",
           "cr_graph_destroy", "cr_sampler_create", "cr_sampler_next", "cr_sampler_destroy",
           "cr_tgrad_geometry", "cr_batch_index_layout", "cr_index_builder_create", "cr_index_build", "cr_index_builder_destroy", "cr_table_grad"]

lib = _lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("castrec %s failed (%d): %s" % (what, rc, _lib.cr_last_error().decode()))


def call(name, *args):
    check(getattr(_lib, name)(*args), name)
