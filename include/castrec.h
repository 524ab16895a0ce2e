/* castrec.h -- C ABI of libcastrec.so, the MI355X (gfx950) native hot path for
 * SASRec / CAST training (drop-in for the hot path of
 * Spijkervet/Context-Aware-Sequential-Recommendation).
 *
 * The reference has no FFI: its "operator interface" is the set of Python free
 * functions in modules.py plus the graph code in models/<model>.py that TensorFlow
 * executes.  Each entry point below replaces one of those (file:line cited); the
 * Python mirror of the reference interface (castrec_amd/models, castrec_amd/sampler)
 * binds them through ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - plain C: pointers, ints, floats.  No torch / HIP types in signatures
 *    (`stream` is a hipStream_t passed as void*; NULL = default stream).
 *  - every device buffer is allocated and owned by the caller; the library
 *    allocates nothing on the device and keeps no pointer after return.
 *  - all kernels are asynchronous on `stream`; no entry point synchronises.
 *  - activations are fp32 row-major matrices [M, ld] with M = B*T rows
 *    (row b*T + t), `ld` = leading dimension in floats.
 *  - return value: 0 on success, <0 on error (CR_ERR_*); cr_last_error() gives a
 *    thread-local message.  The Python side raises RuntimeError.
 *  - dropout uses a counter-based generator (cr_rng): keep(e) is a pure function
 *    of (seed, *step, site, element index) so backward regenerates the forward
 *    mask and results are independent of how a batch is sharded over GPUs.
 */
#ifndef CASTREC_H
#define CASTREC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CR_OK 0
#define CR_ERR_INVALID (-1)     /* bad argument / shape */
#define CR_ERR_UNSUPPORTED (-2) /* shape outside what the kernels implement */
#define CR_ERR_HIP (-3)         /* HIP runtime error (launch failure ...) */

#define CR_MAX_BATCH 4          /* problems per batched GEMM launch */
#define CR_STATE_FLOATS 16      /* device state block, see cr_step_begin */

int cr_version(void);
const char* cr_last_error(void);
/* number of device slabs a dense-gradient producer writes (= grid of the reducing
 * kernels); the caller sizes the slab buffer as n_slabs * n_dense floats. */

/* ---- dropout generator -------------------------------------------------------- */
typedef struct {
    float rate;            /* 0 = no dropout (is_training False) */
    uint32_t site;         /* distinct id per dropout call site in the graph */
    uint32_t seed;
    const uint32_t* step;  /* device pointer: step counter (state[4] as uint32) */
    uint32_t row_offset;   /* global row index of local row 0 (data-parallel shards) */
} cr_rng;

/* ---- per-step device state ---------------------------------------------------- */
/* state (device, CR_STATE_FLOATS floats):
 *   [0] loss_sum  [1] auc_sum  [2] n_target  (accumulated by cr_head_fwd_bwd)
 *   [3] reserved  [4] step counter (uint32 bits)  [5] loss  [6] auc (written by cr_adam_step)
 *   [8] [9] [10] copies of [0] [1] [2] and [11] a copy of [4], taken by the LAST workgroup of cr_head_fwd_bwd(_ln)
 *   to finish (ticket in [12]): a consistent snapshot that later kernels may read while [0..4] move on.
 * Two ways to drive a step:
 *   - cr_step_begin first (zeroes [0..3], increments [4]), cr_adam_step with step_snapshot == NULL; or
 *   - no cr_step_begin: [4] holds the number of the CURRENT step (1 for the first), cr_adam_step gets
 *     step_snapshot = state + 11 (and stats = state + 8 or the all-reduced copy) and, when done, zeroes [0..3] and
 *     sets [4] to the next step -- one kernel boundary per step less. */
int cr_step_begin(float* state, void* stream);

/* The id batch of the NEXT step out of a ring of batches resident in HBM: dst[0 .. slot_elems) <- ring slot
 * ((*step + 1) mod n_slots), `step` = the snapshot of the running step's number (state + 11).  The reference feeds every
 * step through sess.run's feed_dict (/root/reference/main.py:190-196); here a step's graph ends with this launch beside
 * cr_adam_step, so that consecutive steps need no host-side copy between them.  (16-byte moves when slot_elems is a multiple
 * of 4 and ring, dst are 16-byte aligned.) */
int cr_ids_ring_next(const int32_t* ring, int n_slots, int64_t slot_elems, int32_t* dst, const uint32_t* step, void* stream);

/* ---- embedding gather (modules.py:83-164 `embedding`, sasrec.py:27-62, cast_1.py:86-91) */
typedef struct {
    const int32_t* ids;     /* [M] */
    const float* table;     /* [V, D] */
    int M, T, D, V;
    int zero_pad;           /* modules.py:154-156: id 0 reads as a zero row */
    float scale;            /* modules.py:159-160: sqrt(D) or 1 */
    const float* pos_table; /* optional [T, D]: learned dec_pos (sasrec.py:40-50) or the static
                               sinusoid (modules.py:27-37); row t = m % T is added */
    const float* addend;    /* optional [M, ld_add]: e.g. the context sequence (cast_1.py:87) */
    int ld_add;
    cr_rng drop;            /* sasrec.py:59-61 */
    const int32_t* mask_ids;/* optional [M]: out *= (mask_ids[m] != 0)  (sasrec.py:62) */
    float* out;             /* [M, ld_out], written at columns [col_off, col_off + D) */
    int ld_out, col_off;
} cr_embed_desc;
int cr_embed_fwd(const cr_embed_desc* d, void* stream);

/* backward of cr_embed_fwd: g = dout (*mask)(*dropout); table_grad[id] += scale*g (atomics,
 * row 0 skipped when zero_pad); pos_grad[t] = sum_b g (written, not accumulated);
 * d_addend = g (written).  Unused outputs may be NULL. */
typedef struct {
    cr_embed_desc f;        /* same description as the forward call (out = dout here) */
    float* table_grad;      /* [V, D] accumulated (n_slabs == 0), or slab 0 of a small table */
    float* pos_grad;        /* [T, D] or NULL (large-table mode only) */
    float* d_addend;        /* [M, ld_add] or NULL */
    int slab_stride, n_slabs; /* n_slabs > 0: small-table mode (V*D*4 <= 48 KiB, context tables with
                               8..201 rows): each of n_slabs workgroups reduces its rows in LDS and
                               WRITES slab s (no hot-row global atomics) */
    const float* out2;      /* optional: second addend of dout, laid out like f.out (the gradient then is f.out + out2:
                               cr_stack_block_bwd returns a block input's gradient as two partials) */
} cr_embed_bwd_desc;
int cr_embed_bwd(const cr_embed_bwd_desc* d, void* stream);

/* ---- layer normalisation (modules.py:53-80 `normalize`) ------------------------- */
typedef struct {
    const float* x; int ldx;
    const float* gamma; const float* beta;
    float* y; int ldy;
    int M, D;
    float eps;              /* 1e-8, inside the sqrt (modules.py:77) */
    float* x_nonzero;       /* optional [M]: (sum_c x[m,c] != 0)  -> key mask   (modules.py:222) */
    float* y_nonzero;       /* optional [M]: (sum_c y[m,c] != 0)  -> query mask (modules.py:248-249) */
} cr_ln_desc;
int cr_layernorm_fwd(const cr_ln_desc* d, void* stream);

typedef struct {
    const float* x; int ldx;     /* forward input */
    const float* gamma;
    const float* dy; int lddy;
    float* dx; int lddx;
    int accumulate;              /* dx += ... instead of dx = ... */
    float* dgamma; float* dbeta; /* slab 0; slab s at + s*slab_stride; every slab is written */
    int slab_stride, n_slabs;
    int M, D;
    float eps;
} cr_ln_bwd_desc;
int cr_layernorm_bwd(const cr_ln_bwd_desc* d, void* stream);

/* Arithmetic of the matrix products (cr_attn_desc.precision, cr_gemm_desc.precision, cr_wgrad_desc.precision).  The reference is fp32 end to end
 * (modules.py:203-262); BASELINE.json configs[1] names bf16.
 *   CR_PREC_F32     v_mfma_f32_16x16x4_f32: exact fp32 fma chains (T <= 256, head dim <= 64; other shapes: general kernels)
 *   CR_PREC_BF16X3  v_mfma_f32_16x16x32_bf16 on operands split into bf16 hi + lo, three products per term
 *                   (hi*hi + hi*lo + lo*hi): ~1e-5 relative, inside the 1e-3 fp32 bound; forward T <= 256,
 *                   backward T <= 1024 (needs row_stats), head dim <= 64
 *   CR_PREC_BF16    the same kernels on the hi halves only: plain bf16 operands, fp32 accumulation */
#define CR_PREC_F32 0
#define CR_PREC_BF16X3 1
#define CR_PREC_BF16 2

/* ---- row GEMM with fused epilogue -------------------------------------------------
 * C[M,N] = epilogue(A[M,K] @ op(B) + bias)
 *   tf.layers.dense (modules.py:203-205,333-334), conv1d k=1 (modules.py:300-310),
 *   and their data gradients (trans_b = 1).
 * epilogue order: +bias -> relu -> dropout -> +residual -> *row mask -> (accumulate) */
typedef struct {
    const float* A; int lda;
    const float* B; int ldb;   /* trans_b=0: B is [K,N]; trans_b=1: B is [N,K] (C = A @ B^T) */
    const float* bias;         /* [N] or NULL */
    float* C; int ldc;
    int M, N, K;
    int trans_b;
    int relu;
    cr_rng drop;               /* element index = m*N + n */
    const float* residual; int ldr;
    const int32_t* mask_ids;   /* [M] or NULL */
    int accumulate;            /* C += value */
    int precision;             /* CR_PREC_*: fp32 MFMA (exact), or bf16 MFMA on split / plain operands (K, N >= 8) */
} cr_gemm_desc;
int cr_gemm_rows(const cr_gemm_desc* d, int n_problems, void* stream);

/* weight gradient: dW[K,N] = A[M,K]^T @ G[M,N], db[N] = colsum(G); reduction over M is split
 * over n_slabs workgroups, each writing its own slab (no atomics, bitwise reproducible). */
typedef struct {
    const float* A; int lda;
    const float* G; int ldg;
    float* dW; int ldw;        /* slab 0, row-major [K,N] with row pitch ldw (>= N): column blocks of a
                                  fused [D,3D] QKV weight are addressed this way */
    float* db;                 /* slab 0 or NULL */
    int M, N, K;
    int precision;             /* CR_PREC_* as in cr_gemm_desc */
} cr_wgrad_desc;
int cr_gemm_wgrad(const cr_wgrad_desc* d, int n_problems, int slab_stride, int n_slabs, void* stream);

/* ---- element-wise helpers (dropout on concats, ReLU/dropout gradients, adds, copies) --- */
#define CR_ELT_COPY 0        /* y = x                                   (concat / split)        */
#define CR_ELT_ADD 1         /* y = x + aux                                                      */
#define CR_ELT_DROPOUT 2     /* y = x * keep * 1/(1-rate) (* row mask)  (fwd and bwd are equal)  */
#define CR_ELT_RELU_BWD 3    /* y = x * (aux > 0)                       (modules.py:300,333-334) */
#define CR_ELT_ROWMASK 4     /* y = x * (mask_ids[m] != 0)                                       */
#define CR_ELT_GRADPREP 5    /* gradient wrt a layer's pre-activation from the gradient of its output:
                                aux != NULL: y = (aux > 0) ? x/(1-rate) : 0   (ReLU [+dropout] gate read from the
                                             stored output: kept & positive  <=>  output > 0)
                                aux == NULL: y = x * keep/(1-rate)            (regenerated dropout mask)
                                then * row mask */
typedef struct {
    int op;
    const float* x; int ldx;
    const float* aux; int ldaux;
    float* y; int ldy;
    int M, N;
    cr_rng drop;             /* CR_ELT_DROPOUT: element index = m*N + n */
    const int32_t* mask_ids; /* optional */
    int accumulate;          /* y += value */
} cr_elt_desc;
int cr_eltwise(const cr_elt_desc* d, void* stream);


/* ---- causal multi-head self-attention core (modules.py:208-269) ---------------------
 * Per head j (columns [j*d, (j+1)*d) of Q/K/V, d = D/H):
 *   S = Q K^T / sqrt(d); masked (key invalid or key > query) -> -2^32+1; softmax over all T keys
 *   (a row with no valid key is uniform 1/T over ALL keys, modules.py:227-244);
 *   A = softmax * q_valid; dropout; out = A V + residual.
 * q_valid / k_valid are the data-dependent masks sign(|sum_c .|) computed by cr_layernorm_fwd. */
typedef struct {
    const float* Q; const float* K; const float* V; int ld;   /* [M, ld], M = B*T */
    const float* k_valid;      /* [M] */
    const float* q_valid;      /* [M] */
    const float* residual; int ldr;  /* queries (modules.py:269) */
    const int32_t* dead_ids;   /* optional [M]: rows with id 0 are known-dead downstream
                                  (every block ends with `*= mask`, sasrec.py:83): computed as A = 0 */
    float* out; int ldo;
    float* attn_weights;       /* optional [H*B, T, T] (modules.py:259), head j of sample n at row j*B+n */
    int B, T, H, d;
    cr_rng drop;               /* element index = ((j*Bglobal + n)*T + q)*T + k, see batch_global */
    int batch_global;          /* Bglobal (>= B) for shard-invariant dropout indices */
    float* row_stats;          /* optional [H*B*T*4]: the forward saves {row max (base-2 units), 1/sum, flag, 0} per
                                  query row (flag 0 normal, 1 uniform row, 2 dead row) for the single-pass backward */
    int precision;             /* CR_PREC_*: arithmetic of the score / A V / dS K products (accumulation is fp32 in all) */
} cr_attn_desc;
int cr_attn_fwd(const cr_attn_desc* d, void* stream);

typedef struct {
    cr_attn_desc f;            /* forward description (out / attn_weights unused) */
    const float* dout; int lddo;   /* gradient of `out` (the residual branch is handled by the caller) */
    float* dQ; float* dK; float* dV; int ldg;
    float* stats;              /* workspace [H*B*T*4] floats (two-pass backward) */
    /* Single-pass backward (5 MFMA products instead of 7, one launch): taken when f.row_stats (saved by
     * cr_attn_fwd with the same description), `delta` and `dQ_part` are all given and the shape fits.
     *   delta[m]  = sum_c dout[m][c] * (out[m][c] - residual[m][c])      (cr_block_ln_ffn_bwd can emit it)
     *   dQ is returned as TWO partial sums, dQ and dQ_part ([M, ldg] like dQ); the caller adds them
     *   (cr_block_ln_qkv_bwd does).  Heads H = 1 only. */
    const float* delta;        /* optional [M] */
    float* dQ_part;            /* optional [M, ldg] */
} cr_attn_bwd_desc;
int cr_attn_bwd(const cr_attn_bwd_desc* d, void* stream);

/* ---- fused row-phase kernels of one transformer block, hidden size D <= 64 ---------------------
 * The block of sasrec.py:65-83 is, apart from the attention core, row-local: a 64-row tile of the
 * activations stays in LDS across LayerNorm -> projections (and back).  These four entry points replace
 * chains of cr_layernorm / cr_gemm_rows / cr_eltwise / cr_gemm_wgrad launches; same results up to fp32 rounding.
 *   cr_block_ln_qkv_fwd : q_in = LN1(x) (+ key/query masks); Q = q_in Wq + bq; K = x Wk + bk; V = x Wv + bv
 *   cr_block_ln_ffn_fwd : f_in = LN2(o); hid = drop(relu(f_in W1 + b1)); y = (drop(hid W2 + b2) + f_in) * mask
 *   cr_block_ln_ffn_bwd : dy -> d_o, and slabs of dW2 db2 dW1 db1 dgamma2 dbeta2
 *   cr_block_ln_qkv_bwd : (dQ|dK|dV, d_o) -> dx (= or +=), and slabs of dWqkv dbqkv dgamma1 dbeta1        */
typedef struct {
    int M, D;
    const float* ln1_g; const float* ln1_b;
    const float* wqkv; const float* bqkv;      /* [D,3D], [3D] */
    const float* ln2_g; const float* ln2_b;
    const float* w1; const float* b1; const float* w2; const float* b2;
    const float* x;                            /* block input [M,D] */
    float* q_in; float* qkv;                   /* [M,D]; [3,M,D]: Q rows, then K rows, then V rows */
    float* k_valid; float* q_valid;            /* [M] */
    float* o;                                  /* attention output (+ residual) [M,D] */
    float* f_in; float* hid; float* y;         /* [M,D] */
    const int32_t* mask_ids;                   /* [M] */
    cr_rng drop_ffn1, drop_ffn2;
} cr_block_desc;
int cr_block_ln_qkv_fwd(const cr_block_desc* d, void* stream);
int cr_block_ln_ffn_fwd(const cr_block_desc* d, void* stream);

/* cr_block_ln_qkv_fwd of a stack's FIRST block with the stack input composed in the same kernel: `e` is the
 * cr_embed_fwd call that would have produced d->x (sasrec.py:27-62 / cast_1.py:30-38,86-91); it must describe
 * exactly that matrix (e->out == d->x, ld_out == D, col_off == 0, same M and D).  x is still written (the backward
 * and the residual read it); results equal cr_embed_fwd followed by cr_block_ln_qkv_fwd. */
int cr_block_ln_qkv_fwd_gather(const cr_block_desc* d, const cr_embed_desc* e, void* stream);

/* cr_block_ln_ffn_fwd with a tail stage applied to the output rows while they are still on chip:
 *   kind 1: the NEXT block's cr_block_ln_qkv_fwd (next->x must be this block's y) -- one launch per block boundary less;
 *   kind 2: the stack's final LayerNorm (sasrec.py:85): out[:, col_out : col_out + D] = LN(y; lnf_gamma, lnf_beta).
 * y is written in both cases (the backward needs it); results equal the separate calls up to fp32 rounding
 * (the row mean of the final LayerNorm is sum * (1/D) here and sum / D in cr_layernorm_fwd). */
typedef struct {
    int kind;                                  /* 0 none, 1 next block's LN1 + QKV, 2 final LayerNorm */
    const cr_block_desc* next;                 /* kind 1 */
    const float* lnf_gamma; const float* lnf_beta; float* out; int ld_out, col_out;   /* kind 2 */
} cr_block_tail_desc;
int cr_block_ln_ffn_fwd_tail(const cr_block_desc* d, const cr_block_tail_desc* t, void* stream);

/* ---- a whole stack of blocks (sasrec.py:65-85) forward in one launch, one workgroup per sequence -------------
 * Equals, per block i, cr_block_ln_qkv_fwd(blocks[i]); cr_attn_fwd(attn[i]); cr_block_ln_ffn_fwd(blocks[i]) and then the
 * final LayerNorm of kind-2 tails, up to the rounding of the bf16 split products (the projections and the feed-forward
 * run on the bf16 matrix pipe here as well): every buffer of the descriptions is written as those calls write it, so
 * the backward entry points are unchanged.  Requirements (cr_stack_fwd_supported): 1..CR_STACK_MAX_BLOCKS blocks of one
 * shape, H = 1 with D = d in 8..64 (or H = 2, d = 32, D = 64), T <= 208 (CR_PREC_BF16X3, or two heads) or 256 (CR_PREC_BF16, one head), attn[i] wired
 * to blocks[i]'s buffers
 * (Q/K/V = qkv parts, residual = q_in, out = o, masks), no attention weights, blocks[i+1].x == blocks[i].y. */
#define CR_STACK_MAX_BLOCKS 4
typedef struct {
    int n_blocks;
    const cr_block_desc* blocks;               /* [n_blocks] */
    const cr_attn_desc* attn;                  /* [n_blocks] */
    const float* lnf_gamma; const float* lnf_beta; float* out; int ld_out, col_out;   /* optional (out != NULL): final LayerNorm */
    const cr_embed_desc* embed;                /* optional: the cr_embed_fwd call that composes blocks[0].x (embed->out == blocks[0].x,
                                                  dense); x is then composed -- and written -- by this launch instead */
} cr_stack_desc;
int cr_stack_fwd_supported(const cr_stack_desc* d);   /* 1 / 0 */
int cr_stack_fwd(const cr_stack_desc* d, void* stream);

typedef struct {
    cr_block_desc f;
    const float* dy;                           /* gradient of y [M,D] */
    float* d_o;                                /* out of ffn_bwd, in of qkv_bwd: gradient of o [M,D] */
    const float* dqkv;                         /* [3,M,D] from cr_attn_bwd (dQ rows, dK rows, dV rows) */
    float* dx; int dx_accumulate;              /* gradient of x */
    float* g_ln1_g; float* g_ln1_b; float* g_wqkv; float* g_bqkv;      /* slab-0 pointers */
    float* g_ln2_g; float* g_ln2_b; float* g_w1; float* g_b1; float* g_w2; float* g_b2;
    int slab_stride, n_slabs;
    float* attn_delta;                         /* optional [M], written by ffn_bwd: sum_c d_o[m][c] * (o[m][c] - q_in[m][c]),
                                                  the softmax-backward row term the single-pass cr_attn_bwd needs */
    const float* dq_part;                      /* optional [M,D], read by qkv_bwd: second partial of dQ (added to dqkv's dQ rows) */
} cr_block_bwd_desc;
int cr_block_ln_ffn_bwd(const cr_block_bwd_desc* d, void* stream);
int cr_block_ln_qkv_bwd(const cr_block_bwd_desc* d, void* stream);

/* The same two backward steps on the bf16 matrix pipe, one workgroup per sequence, rows in registers (cr_stack_bwd.hip):
 * same description, inputs, outputs and slab layout as cr_block_ln_ffn_bwd / cr_block_ln_qkv_bwd (dq_part must be NULL);
 * results equal theirs up to the rounding of the split products.  B, T: the sequences behind the M = B * T rows;
 * precision: CR_PREC_BF16X3 or CR_PREC_BF16.  Shapes: 8 <= D <= 64, T <= 224 (cr_stack_bwd_supported). */
int cr_stack_bwd_supported(const cr_block_bwd_desc* d, int B, int T, int precision);   /* 1 / 0 */
int cr_stack_ffn_bwd(const cr_block_bwd_desc* d, int B, int T, int precision, void* stream);
/* ... of a stack's LAST block, with the backward of the stack's final LayerNorm (sasrec.py:85) applied to the gradient rows
 * on the way in: `n` is the cr_layernorm_bwd call that would have produced d->dy (n->x == d->f.y; d->dy and n->dx are not
 * touched; n->accumulate == 0; same slabs as d) */
int cr_stack_ffn_bwd_ln(const cr_block_bwd_desc* d, const cr_ln_bwd_desc* n, int B, int T, int precision, void* stream);
/* ... either of the two (n may be NULL) with d->attn_delta per head, [heads, M]: one head, or two heads of 32 columns at D = 64
 * (config C3) -- what cr_attn_bwd's one-launch form takes as cr_attn_bwd_desc.delta */
int cr_stack_ffn_bwd_heads(const cr_block_bwd_desc* d, const cr_ln_bwd_desc* n, int B, int T, int heads, int precision, void* stream);
int cr_stack_qkv_bwd(const cr_block_bwd_desc* d, int B, int T, int precision, void* stream);
/* ... with the embedding gather's backward applied instead of storing dx: arguments as cr_block_ln_qkv_bwd_scatter */
int cr_stack_qkv_bwd_scatter(const cr_block_bwd_desc* d, const cr_embed_bwd_desc* sc, int B, int T, int precision, void* stream);

/* ---- a whole block backward in ONE launch (round 3; csrc/cr_stack_bwd1.hip) --------------------------------------
 * = cr_stack_ffn_bwd[_ln] -> cr_attn_bwd -> cr_stack_qkv_bwd[_scatter] of one block (sasrec.py:65-83 and its autodiff), per
 * sequence, on a PAIR of workgroups (query side / key side) that never wait for each other.  Same inputs and slab layout as
 * those calls, with these differences:
 *   - the gradient of the block input leaves as TWO partials, d->dx (query side: LayerNorm-1 branch) and x->dx2 (key side:
 *     K / V projections); the caller's gradient is their SUM.  Likewise the input gradient may arrive as a sum: d->dy + x->dy2
 *     (or, with the final LayerNorm `lnf` as in cr_stack_ffn_bwd_ln: lnf->dy + x->lnf_dy2);
 *   - with `sc` (the embedding backward of the block input, as cr_stack_qkv_bwd_scatter): each side applies it to its partial:
 *     table / positional gradients by float atomics, the addend's gradient as sc->d_addend + x->d_addend2 (their sum); a
 *     small-table recipe (sc->n_slabs > 0, cr_embed_bwd's small-table mode; here: V <= 256) is formed on chip instead -- one matrix
 *     product per side, one-hot(ids)^T x the side's rows as bf16 hi + lo (2^-17 relative per element) -- and leaves as slabs:
 *     pair p writes slabs p and min(B, n_slabs) + p of sc->table_grad (sc->n_slabs >= 2 min(B, n_slabs) required);
 *   - d->d_o, d->dqkv ([3, M, D]) are workspaces here; d->attn_delta is not used (delta stays on chip);
 *   - ONE slab per sequence pair: workgroup pair p writes slab p (p < min(B, n_slabs)) and adds its later sequences to it;
 *     slabs >= min(B, n_slabs) are not written by this call.
 * `ad` is the block's attention call exactly as given to cr_attn_fwd / cr_stack_fwd (row_stats required).
 * Shapes (cr_stack_block_bwd_supported): one head of d = D, 8 <= D < 64, or two heads of 32 columns at D = 64 (round 4; the bias
 * gradients then come from an all-ones product); T <= 224; CR_PREC_BF16X3 or CR_PREC_BF16. */
typedef struct {
    const float* dy2;                          /* optional [M,D] */
    float* dx2;                                /* [M,D] (may be NULL with `sc` when sc->d_addend is set) */
    const float* lnf_dy2;                      /* optional, with lnf: second addend of lnf->dy (same leading dimension) */
    float* d_addend2;                          /* with sc->d_addend: [M,D] */
} cr_block_bwd1_ext;
int cr_stack_block_bwd_supported(const cr_block_bwd_desc* d, const cr_attn_desc* ad, int B, int T, int precision);   /* 1 / 0 */
int cr_stack_block_bwd(const cr_block_bwd_desc* d, const cr_attn_desc* ad, const cr_block_bwd1_ext* x, const cr_ln_bwd_desc* lnf,
                       const cr_embed_bwd_desc* sc, int B, int T, int precision, void* stream);
/* (tests) the deal of one attention pass of cr_stack_block_bwd at `tiles` 16-row tiles (1..14): pk[w] = the two tiles of wave w as
 * 5-bit numbers (low bits first, 31 = none).  Host only, no GPU. */
int cr_stack_block_bwd_deal(int tiles, int query_pass, uint32_t* pk);

/* ---- the same four row phases for hidden sizes 128 / 192 / 256 (configs C4, C5) on the bf16 matrix pipe (cr_wide.hip):
 * one launch each where the unfused path runs cr_layernorm_* + cr_gemm_rows (+ cr_eltwise) chains -- modules.py:53-80
 * (normalize), 203-205 (Q/K/V dense layers), 280-318 (feedforward) and their gradients.  Same descriptions and buffers as
 * the cr_block_* entry points; precision: CR_PREC_BF16X3 or CR_PREC_BF16.  Differences: dq_part is not taken, attn_delta is
 * per head (below).  Weight gradients: at D = 128 the backward entries form them themselves when the g_w* / g_b* pointers are given
 * (slab per workgroup, as cr_block_*), and then nothing of the chain goes through memory (g2, g1 are not written).  With
 * NULL weight-gradient pointers -- the only form at D = 192 / 256 -- they write only the dgamma / dbeta slabs,
 * cr_wide_ln_ffn_bwd returns the two operands cr_gemm_wgrad needs, g2 = dy * dropout * mask and g1 = gated(g2 W2^T), dense
 * [M, D], and cr_wide_ln_qkv_bwd OVERWRITES d_o (with the gradient of q_in, dQ Wq^T + d_o: the row of the chain that has to
 * exist in memory between the projection panels and the LayerNorm backward). */
int cr_wide_supported(const cr_block_desc* d, int precision);   /* 1 / 0 */
int cr_wide_ln_qkv_fwd(const cr_block_desc* d, int precision, void* stream);
int cr_wide_ln_ffn_fwd(const cr_block_desc* d, int precision, void* stream);
/* ... with a tail applied to the output rows while they are in registers (D = 128 only), as cr_block_ln_ffn_fwd_tail:
 * kind 1 = the NEXT block's cr_wide_ln_qkv_fwd (next->x must be this block's y), kind 2 = the stack's final LayerNorm */
int cr_wide_ln_ffn_fwd_tail(const cr_block_desc* d, const cr_block_tail_desc* t, int precision, void* stream);
/* heads: with d->attn_delta != NULL, [heads, M] floats: delta[h][m] = sum over head h's columns of d_o * (o - q_in), the row term
 * cr_attn_bwd's bf16 kernels take (cr_attn_bwd_desc.delta) to run both passes in one launch; head width a multiple of 16 */
int cr_wide_ln_ffn_bwd(const cr_block_bwd_desc* d, float* g2, float* g1, int heads, int precision, void* stream);
int cr_wide_ln_qkv_bwd(const cr_block_bwd_desc* d, int precision, void* stream);

/* cr_block_ln_qkv_bwd of a stack's FIRST block whose input x was composed by an embedding gather: instead of
 * storing dx the kernel applies that gather's backward to its rows (cr_embed_bwd, large-table mode): `sc` is the
 * descriptor that call would have taken (sc->f.out is ignored; no small-table slabs; d_addend dense [M, D]);
 * bd->dx may be NULL and bd->dx_accumulate must be 0.  One difference: sc->pos_grad is ACCUMULATED with float
 * atomics here (cr_embed_bwd writes it), so it must be zero on entry -- cr_adam_step leaves the table section
 * zeroed.  Otherwise results equal cr_block_ln_qkv_bwd followed by cr_embed_bwd up to the order of the
 * float atomics. */
int cr_block_ln_qkv_bwd_scatter(const cr_block_bwd_desc* bd, const cr_embed_bwd_desc* sc, void* stream);

/* ---- prediction head: pos/neg dot-product BCE (sasrec.py:87-115), forward + backward ---- */
typedef struct {
    const float* seq_emb; int ld;   /* [M, D] */
    const float* table;             /* item table [V, D]; row 0 reads as zeros (modules.py:154-156) */
    const int32_t* pos; const int32_t* neg;   /* [M] */
    int M, D, V;
    float* state;                   /* [0] += loss_sum, [1] += auc_sum, [2] += n_target */
    float* d_seq_emb; int ldd;      /* optional: UN-normalised gradient (times n_target) */
    float* table_grad;              /* optional [V, D]: accumulated, un-normalised */
    float* pos_logits; float* neg_logits;     /* optional [M] */
    float* coef_out;                /* optional [2, M]: d loss_sum / d pos logit, d loss_sum / d neg logit per row (0 where pos id is 0).
                                       With it and table_grad == NULL the item table's gradient is formed from the batch's occurrence
                                       index instead (cr_table_grad / cr_adam_desc.tg): no float atomics in this kernel */
} cr_head_desc;
int cr_head_fwd_bwd(const cr_head_desc* d, void* stream);

/* cr_head_fwd_bwd with the backward of the LayerNorm that produced seq_emb (sasrec.py:85) applied to each gradient
 * row while it is still in registers: `n` is the cr_layernorm_bwd call that would have followed (n->dy is ignored,
 * n->accumulate must be 0); d->d_seq_emb may be NULL (the gradient row is then never stored).  Results equal
 * cr_head_fwd_bwd followed by cr_layernorm_bwd. */
int cr_head_fwd_bwd_ln(const cr_head_desc* d, const cr_ln_bwd_desc* n, void* stream);

/* cr_stack_fwd followed by cr_head_fwd_bwd_ln(h, n) WITHOUT the second launch (round 5): the stack's last launch goes on with the
 * prediction head (sasrec.py:87-115) and the final LayerNorm's backward on the rows it has just normalised -- 14.5 us of the headline's
 * 324 us step were that launch.  `h`, `n`: the cr_head_fwd_bwd_ln call as it would have followed (h->seq_emb == d->out, n->x == the last
 * block's y, n->gamma == d->lnf_gamma); h->table_grad must be NULL (no scatter from here: h->coef_out and the occurrence index).
 * Requirements (cr_stack_fwd_head_supported) beyond cr_stack_fwd's: batches that run two workgroups per sequence (B <= 160, T > 16), at
 * most 13 tiles, n->n_slabs >= 2 B: workgroup (sequence x, half y) WRITES slab y B + x of n->dgamma / n->dbeta; slabs >= 2 B are not
 * written (cr_adam_desc.slab_counts).  Same results as the two calls. */
int cr_stack_fwd_head_supported(const cr_stack_desc* d, const cr_head_desc* h, const cr_ln_bwd_desc* n);   /* 1 / 0 */
int cr_stack_fwd_head(const cr_stack_desc* d, const cr_head_desc* h, const cr_ln_bwd_desc* n, void* stream);

/* test_logits (sasrec.py:93-97): logits[b, j] = seq_emb[b*T + T-1, :] . table'[cand[b, j], :] */
int cr_test_logits(const float* seq_emb, int ld, const float* table, const int32_t* cand,
                   int B, int T, int D, int V, int n_cand, float* logits, void* stream);

/* ---- occurrence index of a batch (round 5; csrc/cr_index.cpp, csrc/cr_tgrad.hip) ---------------------------------
 * The gradient of a looked-up table row is the sum of the gradient rows of every position that looked it up: the three
 * lookups of the item table (seq ids: modules.py:157 through sasrec.py:27; pos / neg ids: sasrec.py:89-90) and the learned
 * positional table's (sasrec.py:40-50).  TensorFlow forms those sums in its gather's gradient; rounds 1-4 formed them with float
 * atomics from the head kernel and the first block's backward (2.56 M + 1.28 M of them per step at the headline shape: ~13 us,
 * and the only sums of a step whose order -- hence whose last bits -- changed from run to run).  The index turns the scatter into
 * a gather: per table row the list of (kind, batch row) pairs, built on the host beside the batch (a stable counting sort, O(rows
 * of the batch)), travelling with the ids, summed on the device in the list's order -- bitwise reproducible, no atomics.
 *
 * Flat row numbering: item rows 0 .. V-1 (row 0, the zero pad, is never listed), then -- T_pos > 0 -- the positional rows V .. V +
 * T_pos - 1 (the positional table follows the item table in the table section, same width).
 * An occurrence is one word: kind << 30 | m.  kind 0: seq id at batch row m (gradient row = scale * rows[m]); 1 / 2: pos / neg id at
 * row m (coef[m] / coef[M + m] times seq_emb[m]); 3: positional row (1.0 * rows[m]).  Within a table row: kind 0 (or 3) by ascending
 * m, then kind 1, then kind 2.
 * The PLAN of the gather is part of the index, made for the device's geometry: a workgroup has `ng` lane groups, a lane group sums at
 * most `ent` occurrences (one batch of loads -- the sums of a Zipf corpus are bound by dependent round trips and one CU's load
 * issue, not by bytes: a hot item holds thousands of a batch's rows).  Workgroup w, lane group g has the record (four words)
 *   {flat row, first occurrence, count (0 = idle), q | k << 6 | j << 13 | n << 22}
 * a row with c occurrences takes k = ceil(c / ent) CONSECUTIVE groups of one workgroup, q = 0 .. k-1 (their partials are added in q
 * order); a row that needs more than ng groups is cut into n slices of whole workgroups w0 .. w0 + n - 1 (j = 0 .. n-1: their partial
 * rows are added in j order by the slice that finishes last -- cr_tgrad_desc.part_rows / tickets); else n = 1, j = 0.
 * Words of the index buffer (all offsets in int32 words, 16-byte aligned):
 *   [0] workgroups in use  [1] listed rows  [2] occurrences  [3] CR_INDEX_MAGIC  [4] words in use (a copy may stop there)
 *   [5] offset of the occurrences  [6] offset of the bitmap  [7] 0 | records, ng x 4 words per workgroup | occurrences | bitmap
 * (bit r & 31 of word r >> 5 set <=> flat row r is listed).  total_words is the capacity for ANY batch of the layout's shape. */
#define CR_INDEX_MAGIC 0x43524959
typedef struct {
    int M, V, T_pos;
    int ng, ent;                      /* lane groups per workgroup, occurrences per lane group (cr_tgrad_geometry) */
    int cap_blocks, cap_occ, bitmap_words;
    int64_t off_recs, total_words;
} cr_index_layout;
/* (lane groups per 1024-thread workgroup, occurrences per group) the device code uses at hidden size D; 0 where D is not taken
 * (the gather takes D a multiple of 4 up to 256, even up to 128, any up to 64) */
int cr_tgrad_geometry(int D, int* ng, int* ent);
int cr_batch_index_layout(int M, int V, int T_pos, int ng, int ent, cr_index_layout* out);
/* Host side, no HIP.  A builder keeps two int32 work arrays of V + T_pos entries between calls (so a build costs O(M), not O(V));
 * not re-entrant on one handle.  seq / pos / neg: [M] ids in [0, V) (the caller range-checks them: Engine.set_batch / feed);
 * `out`: layout.total_words words; words [0, out[4]) are written. */
typedef struct cr_index_builder cr_index_builder;
cr_index_builder* cr_index_builder_create(int M, int V, int T_pos, int ng, int ent);
int cr_index_build(cr_index_builder* b, const int32_t* seq, const int32_t* pos, const int32_t* neg, int32_t* out);
void cr_index_builder_destroy(cr_index_builder* b);

/* The table gradient of a step from the index: grad[row] = sum over the row's occurrences, in list order, for every row that has a
 * unit.  rows / rows2: the gradient rows of the stack input ([M, ld_rows]; rows2 = optional second partial, added: cr_stack_block_bwd
 * leaves them as two partials), already masked and through the embedding dropout; seq_emb: the head's input rows; coef: [2, M] the head's
 * d loss / d logit (cr_head_desc.coef_out).  `index`: the buffer of the CURRENT step -- or, ring != NULL, slot (*step mod ring_slots)
 * of a ring of slots of slot_words words holds it at word index_off (the id ring of cr_adam_desc: a slot = ids + index).
 * cr_table_grad WRITES the rows that have units into table_grad ([V + T_pos, D], plain stores: the other rows are not touched -- zero
 * where the caller keeps them zero, as cr_adam_step does) -- the data-parallel path, whose bucket is all-reduced before Adam.  The
 * one-GPU path hands the same description to cr_adam_step (cr_adam_desc.tg), which sums each listed row and applies its update in
 * place, and sweeps the rows WITHOUT units with a zero gradient (TensorFlow's dense update, modules.py:154-157 + sasrec.py:120-121):
 * no table_grad array at all. */
typedef struct {
    const int32_t* index;             /* static index buffer (used when ring == NULL) */
    const int32_t* ring; int ring_slots; int64_t slot_words, index_off;
    const uint32_t* step;             /* device word: the step number that selects the ring slot */
    cr_index_layout lay;
    const float* rows; const float* rows2; int ld_rows;
    float scale;                      /* modules.py:159-160 */
    const float* seq_emb; int ld_emb;
    const float* coef;                /* [2, M] */
    int D;
    float* part_rows; uint32_t* tickets;   /* workspace of the sliced rows: [lay.cap_blocks, D] floats and [lay.cap_blocks] words, the
                                              words ZERO before the first launch (every launch leaves them zero) */
} cr_tgrad_desc;
int cr_table_grad(const cr_tgrad_desc* d, float* table_grad, void* stream);

/* ---- Adam, TensorFlow formulation (sasrec.py:120) ------------------------------------
 * g = grad * (1 / n_target); table part: grad = table_grad[i] (zeroed after use);
 * dense part: grad = sum_s dense_slabs[s*n_dense + i].  lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
 * p -= lr_t*m/(sqrt(v)+eps), t = step counter in state[4].  Also writes state[5] = loss,
 * state[6] = auc.  {loss_sum, auc_sum, n_target} are read from `stats` when it is non-NULL (the
 * all-reduced tail of the data-parallel bucket, see cr_reduce_slabs), from state[0..2] otherwise. */
typedef struct {
    float* p; float* m; float* v;     /* flat [n_table + n_dense] */
    float* table_grad;                /* [n_table] */
    const float* dense_slabs;         /* [n_slabs, n_dense] */
    int64_t n_table;                  /* 64-bit: config C5's item table alone is 2.56e9 floats */
    int n_dense, n_slabs;
    float lr, beta1, beta2, eps;
    float* state;
    const float* stats;               /* optional [3] */
    const uint32_t* step_snapshot;    /* optional: step number t is read from here instead of state[4], and the kernel
                                         ends the step: state[0..3] = 0, state[4] = t + 1 (see the state block above) */
    float l2; int64_t n_l2;           /* embedding regulariser (modules.py:149-153, sasrec.py:109-110): the first n_l2
                                         parameters (the lookup tables: they lead the flat vector) get g += l2 * p, and
                                         the reported loss gets state[7] (written by cr_l2_penalty) added */
    /* Lazy (row-sparse) Adam for the leading item table -- a DEVIATION from the reference, off unless lazy_ids is set.
     * The reference's zero-padded lookup (modules.py:154-157) makes TensorFlow update EVERY table row densely each step
     * (rows without gradient still drift on their momentum): 7 accesses x 10.24 GB at config C5.  With lazy_ids the
     * first lazy_rows * lazy_D parameters are updated only in the rows listed (duplicates and id 0 allowed; each row
     * is claimed once through lazy_flags[row], which must hold values != the current step number): m, v, p and the
     * gradient of the other rows are not touched.  Rows updated in every step so far match dense Adam exactly. */
    const int32_t* lazy_ids; int n_lazy_ids, lazy_rows, lazy_D;
    uint32_t* lazy_flags;             /* [lazy_rows] */
    const int32_t* slab_counts;       /* optional (device) [ceil(n_dense / 256)]: slabs that hold gradient for the 256 dense
                                         parameters of block b (the others are known zero and are not read): producers that
                                         reduce more rows per workgroup (cr_gemm_wgrad at large hidden sizes) write fewer slabs */
    /* Optional: the NEXT step's id batch, moved by extra workgroups of this launch (cr_ids_ring_next's job without a launch
     * of its own): ids_dst[0 .. ids_slot_elems) <- ids_ring slot ((t + 1) mod ids_ring_slots), t = the step number this
     * launch reads.  Not together with lazy_ids (row-sparse Adam reads the current step's ids out of those buffers). */
    const int32_t* ids_ring; int ids_ring_slots; int64_t ids_slot_elems; int32_t* ids_dst;
    int64_t ids_copy_elems;           /* 0 = ids_slot_elems: how many leading words of the slot are moved (a slot may carry the batch's
                                         occurrence index behind its ids, which stays in the ring) */
    const cr_tgrad_desc* tg;          /* optional (host pointer, copied at launch): the table section's gradient comes from the batch's
                                         occurrence index (cr_table_grad's description, its ring fields = this ring); table_grad is then
                                         not read and may be NULL.  Not together with lazy_ids. */
} cr_adam_desc;
int cr_adam_step(const cr_adam_desc* d, void* stream);

/* state[7] = scale * sum_{i < n} p[i]^2  (scale = l2 / 2): the regularisation term of the loss, fixed summation order.
 * Launched before cr_adam_step (which updates p) when l2_emb != 0; every shipped run of the reference uses 0. */
int cr_l2_penalty(const float* p, int64_t n, float scale, float* state, void* stream);

/* Data-parallel path: collapse the dense slabs into one flat vector that is all-reduced over RCCL
 * together with the table gradient: out[i] = sum_s slabs[s*n_dense + i]; also copies the three loss
 * statistics state[0..2] to stats_out (the tail of the all-reduce bucket).  cr_adam_step is then
 * called with n_slabs = 1 on the reduced vector. */
int cr_reduce_slabs(const float* dense_slabs, int n_slabs, int n_dense, float* out,
                    const float* state, float* stats_out, const int32_t* slab_counts /* optional, as cr_adam_desc's */, void* stream);

/* ---- data-parallel exchange of a row-sparse table gradient (csrc/cr_dist.hip) ----------------------
 * cr_rows_pack: slot i of `packed` ([n, D + 1] floats: column 0 = the row id as int bits, then the row) gets row ids[i] of
 * `table` ([V, D]) if slot i is the first to claim that row in this call (flags: [V] words holding anything but `tag`, one
 * atomic exchange per slot; `tag` points to a DEVICE word the caller changes before every call to a value not used before --
 * read by the kernel, so that a HIP graph of the step replays with fresh tags), else id 0 and a zero row (duplicates, id 0,
 * ids outside [0, V)).  zero_rows != 0: the packed rows are zeroed in `table`.
 * cr_rows_add: table[id] += row for every slot with id != 0.  One call per rank's gathered buffer, in rank order: a call has
 * no conflicting writes (a rank's buffer holds a row once) and the sums are formed in the same order on every replica. */
int cr_rows_pack(float* table, const int32_t* ids, int n, int D, int V, uint32_t* flags, const uint32_t* tag, float* packed,
                 int zero_rows, void* stream);
int cr_rows_add(float* table, const float* packed, int n, int D, int V, void* stream);

/* ---- HIP graph capture of a whole step (launch-bound inner loop) ---------------------- */
int cr_graph_begin(void* stream);
int cr_graph_end(void* stream, void** graph_exec_out);
int cr_graph_launch(void* graph_exec, void* stream);
int cr_graph_destroy(void* graph_exec);

/* ---- native batch sampler (sampler.py:9-136), host side, bit-exact ---------------------
 * The corpus is CSR: events of user u are [offsets[u], offsets[u+1]) (row 0 empty). */
typedef struct cr_sampler cr_sampler;
cr_sampler* cr_sampler_create(const int64_t* offsets, const int32_t* items, const float* ratings,
                              const int64_t* ts, int usernum, int itemnum, int batch_size, int maxlen,
                              int bin_in_hours, int max_bins, int log_scale, double min_timedelta,
                              double max_timedelta, uint32_t seed, int queue_depth);
/* blocks until the next batch is ready; arrays are [B] / [B, maxlen] int32 */
int cr_sampler_next(cr_sampler* s, int32_t* user, int32_t* seq, int32_t* pos, int32_t* neg,
                    int32_t* timeseq, int32_t* ratings, int32_t* hours, int32_t* days);
void cr_sampler_destroy(cr_sampler* s);

#ifdef __cplusplus
}
#endif
#endif /* CASTREC_H */
