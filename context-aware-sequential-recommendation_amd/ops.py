"""Thin op wrappers: torch CUDA tensors in, C-ABI descriptors out.  PyTorch is used only for device
memory and the current HIP stream; every computation is a kernel of libcastrec.so."""
import ctypes as C

import torch

from . import lib as L


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32(t, name):
    if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
        raise TypeError("%s must be a float32 CUDA tensor" % name)
    return t


def _i32(t, name):
    if t is not None and (t.dtype != torch.int32 or not t.is_cuda):
        raise TypeError("%s must be an int32 CUDA tensor" % name)
    return t


class Drop:
    """Dropout context of one step: seed, device step counter, data-parallel row offset."""

    def __init__(self, rate, seed, state, row_offset=0):
        self.rate, self.seed, self.state, self.row_offset = float(rate), int(seed) & 0xFFFFFFFF, state, int(row_offset)

    def rng(self, site, enabled=True):
        rate = self.rate if enabled else 0.0
        return L.Rng(rate, site, self.seed, self.state.data_ptr() + 16, self.row_offset)   # &state[4]


NO_DROP = L.Rng(0.0, 0, 0, None, 0)


def step_begin(state):
    L.call("cr_step_begin", _p(state), _stream())


def embed_fwd(ids, table, T, out, ld_out, col_off=0, zero_pad=True, scale=1.0, pos_table=None, addend=None,
              ld_add=0, rng=NO_DROP, mask_ids=None):
    V, D = table.shape
    d = L.EmbedDesc(_p(_i32(ids, "ids")), _p(_f32(table, "table")), ids.numel(), T, D, V, int(zero_pad), float(scale),
                    _p(pos_table), _p(addend), ld_add, rng, _p(mask_ids), _p(_f32(out, "out")), ld_out, col_off)
    L.call("cr_embed_fwd", C.byref(d), _stream())
    return d


def embed_bwd(fdesc, dout, table_grad=None, pos_grad=None, d_addend=None, slab_stride=0, n_slabs=0):
    f = L.EmbedDesc.from_buffer_copy(fdesc)
    f.out = _p(dout)
    d = L.EmbedBwdDesc(f, _p(table_grad), _p(pos_grad), _p(d_addend), slab_stride, n_slabs)
    L.call("cr_embed_bwd", C.byref(d), _stream())


def layernorm_fwd(x, ldx, gamma, beta, y, ldy, M, D, x_nonzero=None, y_nonzero=None, eps=1e-8):
    d = L.LnDesc(_p(x), ldx, _p(gamma), _p(beta), _p(y), ldy, M, D, eps, _p(x_nonzero), _p(y_nonzero))
    L.call("cr_layernorm_fwd", C.byref(d), _stream())


def layernorm_bwd(x, ldx, gamma, dy, lddy, dx, lddx, dgamma, dbeta, slab_stride, n_slabs, M, D, accumulate=False, eps=1e-8):
    d = L.LnBwdDesc(_p(x), ldx, _p(gamma), _p(dy), lddy, _p(dx), lddx, int(accumulate), _p(dgamma), _p(dbeta),
                    slab_stride, n_slabs, M, D, eps)
    L.call("cr_layernorm_bwd", C.byref(d), _stream())


def gemm_desc(A, lda, B, ldb, Cm, ldc, M, N, K, bias=None, trans_b=False, relu=False, rng=NO_DROP, residual=None,
              ldr=0, mask_ids=None, accumulate=False, precision=0):
    return L.GemmDesc(_p(A), lda, _p(B), ldb, _p(bias), _p(Cm), ldc, M, N, K, int(trans_b), int(relu), rng,
                      _p(residual), ldr, _p(mask_ids), int(accumulate), int(precision))


def gemm_rows(descs):
    arr = (L.GemmDesc * len(descs))(*descs)
    L.call("cr_gemm_rows", arr, len(descs), _stream())


def wgrad_desc(A, lda, G, ldg, dW, db, M, N, K, ldw=None, precision=0):
    return L.WgradDesc(_p(A), lda, _p(G), ldg, _p(dW), N if ldw is None else ldw, _p(db), M, N, K, int(precision))


def gemm_wgrad(descs, slab_stride, n_slabs):
    arr = (L.WgradDesc * len(descs))(*descs)
    L.call("cr_gemm_wgrad", arr, len(descs), slab_stride, n_slabs, _stream())


def eltwise(op, x, ldx, y, ldy, M, N, aux=None, ldaux=0, rng=NO_DROP, mask_ids=None, accumulate=False):
    d = L.EltDesc(op, _p(x), ldx, _p(aux), ldaux, _p(y), ldy, M, N, rng, _p(mask_ids), int(accumulate))
    L.call("cr_eltwise", C.byref(d), _stream())


def attn_desc(Q, K, V, ld, k_valid, q_valid, residual, ldr, out, ldo, B, T, H, d, rng=NO_DROP, batch_global=None,
              dead_ids=None, attn_weights=None, row_stats=None, precision=0):
    return L.AttnDesc(_p(Q), _p(K), _p(V), ld, _p(k_valid), _p(q_valid), _p(residual), ldr, _p(dead_ids), _p(out), ldo,
                      _p(attn_weights), B, T, H, d, rng, B if batch_global is None else batch_global, _p(row_stats),
                      int(precision))


def attn_fwd(desc):
    L.call("cr_attn_fwd", C.byref(desc), _stream())


def attn_bwd(fdesc, dout, lddo, dQ, dK, dV, ldg, stats, delta=None, dQ_part=None):
    d = L.AttnBwdDesc(L.AttnDesc.from_buffer_copy(fdesc), _p(dout), lddo, _p(dQ), _p(dK), _p(dV), ldg, _p(stats),
                      _p(delta), _p(dQ_part))
    L.call("cr_attn_bwd", C.byref(d), _stream())


def head_fwd_bwd(seq_emb, ld, table, pos, neg, M, D, state, d_seq_emb=None, ldd=0, table_grad=None,
                 pos_logits=None, neg_logits=None):
    d = L.HeadDesc(_p(seq_emb), ld, _p(table), _p(pos), _p(neg), M, D, table.shape[0], _p(state), _p(d_seq_emb), ldd,
                   _p(table_grad), _p(pos_logits), _p(neg_logits))
    L.call("cr_head_fwd_bwd", C.byref(d), _stream())


def test_logits(seq_emb, ld, table, cand, B, T, D, logits):
    L.call("cr_test_logits", _p(seq_emb), ld, _p(table), _p(_i32(cand, "cand")), B, T, D, table.shape[0],
           cand.shape[1], _p(logits), _stream())


def adam_step(p, m, v, table_grad, dense_slabs, n_table, n_dense, n_slabs, lr, state, beta1=0.9, beta2=0.98, eps=1e-8,
              stats=None, step_snapshot=None, lazy_ids=None, lazy_rows=0, lazy_D=0, lazy_flags=None):
    d = L.AdamDesc(_p(p), _p(m), _p(v), _p(table_grad), _p(dense_slabs), n_table, n_dense, n_slabs, lr, beta1, beta2,
                   eps, _p(state), _p(stats) if stats is not None else None,
                   _p(step_snapshot) if step_snapshot is not None else None, 0.0, 0,
                   _p(lazy_ids), 0 if lazy_ids is None else lazy_ids.numel(), lazy_rows, lazy_D, _p(lazy_flags), None)
    L.call("cr_adam_step", C.byref(d), _stream())


class Graph:
    """HIP graph of one captured step (cr_graph_*)."""

    def __init__(self):
        self.exec = C.c_void_p()

    def begin(self):
        L.call("cr_graph_begin", _stream())

    def end(self):
        L.call("cr_graph_end", _stream(), C.byref(self.exec))

    def launch(self):
        L.call("cr_graph_launch", self.exec, _stream())

    def __del__(self):
        try:
            if self.exec:
                L.lib.cr_graph_destroy(self.exec)
        except Exception:
            pass
