// Backward of ONE transformer block in ONE launch (round 3): the work of cr_stack_ffn_bwd -> cr_attn_bwd -> cr_stack_qkv_bwd
// per sequence, without leaving the compute unit between them.  (sasrec.py:65-83 and autodiff of it; modules.py:167-318.)
//
// Everything between the gradient of a block's output and the gradient of its input is sequence-local; only the weight
// gradients couple sequences, and they leave as slabs.  A sequence gets a PAIR of workgroups on two compute units:
//
//   Q side (blockIdx.y = 1)                                   K side (blockIdx.y = 0, dispatched first: it is the longer one)
//   1  feed-forward + LN2 backward of ALL 16-row tiles        1  the same chain, data gradients only: d_o goes straight
//      (two rounds of <= 7 tiles): d_o -> global, delta ->       into the dOut IMAGE in LDS (bf16 hi + lo), the tile's Q
//      LDS; images of hid, g2, f_in, g1 -> dW2 db2 dW1 db1       rows into the Q image, delta and the forward's row
//      dgamma2 dbeta2 (+ the stack's final LayerNorm)            statistics into LDS vectors: no staging from memory
//   2  K (natural) / V (column-permuted) images staged;       2  key-owner pass of the attention backward on the images
//      query-owner pass of the attention backward for the        (own K / V tiles from memory); dK^T, dV^T come out of the
//      wave's tiles; dQ^T comes out of the swapped product       swapped products in layout R:  dx_kv = dK Wk^T + dV Wv^T
//      in layout R and goes on through registers:                -> dx2;  dK, dV -> global (for the weight gradients)
//      dq_in = dQ Wq^T + d_o, LN1 backward -> dx
//   3  images of q_in, dQ -> dWq dbq; dgamma1 dbeta1           3  images of x, dK, dV -> dWk dbk dWv dbv
//
// The gradient of the block input is the SUM dx + dx2 of the two sides' partials (no workgroup ever waits for another:
// nothing here depends on dispatch order or co-residency); the consumer -- this kernel for the block below (dy + dy2), the
// embedding scatter applied here by both sides to their partial, cr_embed_bwd (out2), or a cr_eltwise add -- adds them.
// Both sides run the feed-forward chain (2 of the block's 16 products per tile) rather than hand d_o across compute units.
//
// Layout R (cr_rlayout.hpp) carries a tile through every row-local layer.  For the attention products a lane's registers
// are an MFMA operand when the OTHER operand's image has its columns in the k order of layout R (V on the Q side, read by
// rows only: staged permuted like the weights); images that are also read transposed stay in natural column order and
// meet operands that come from memory in natural order (Q on the Q side; K, V on the K side).  A product whose result is
// wanted in layout R is issued with its operands swapped (out^T): the same two fragments, no shuffle.
//
// One gradient slab per SEQUENCE (slab n, the two sides write disjoint parameter ranges of it): B slabs instead of 2 B.
// Fixed summation order, no atomics except the embedding scatter: bitwise reproducible like the kernels it replaces.
// Shapes: one head, 8 <= D < 64 (bias gradients ride the ones column), T <= 224, bf16 arithmetic (split or plain).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cr_rbwd.hpp"

#define B1_ROWS 224                       // rows of an attention image (14 tiles of 16)
#define B1_FSTR (B1_ROWS * 64)            // bf16 elements of one image half

struct B1Args {
    cr_block_bwd_desc bd;
    cr_attn_desc ad;
    cr_embed_bwd_desc sc;                 // scatter: the embedding backward of the block input, applied to both partials
    cr_ln_bwd_desc ln;                    // has_ln: dy = backward of the stack's final LayerNorm applied to ln.dy (+ ln_dy2)
    const float* dy2;                     // optional second addend of dy
    float* dx2;                           // K side's partial of dx
    const float* ln_dy2;
    float* d_addend2;                     // scatter: K side's partial of d_addend
    float* sbuf; float* sbuf2;            // scatter: where the masked partial rows wait for phase 3 ([M, D] each)
    int B, T, nkt, scatter, has_ln;
    float isd, isd_log2e, invT;
    unsigned qpk[8], kpk[8];              // tiles of wave w in the attention passes: two 5-bit tile numbers, 31 = none
    unsigned long long* ts;
};

#ifdef CR_TIMELINE
#define B1_TS(slot)                                                                                          \
    do {                                                                                                     \
        if (a.ts && (threadIdx.x & 63) == 0)                                                                 \
            a.ts[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SB_WAVES + (threadIdx.x >> 6)) * 32 + (slot)] = \
                ((slot) == 0 || (slot) == 31) ? wall_clock64() : clock64();                                  \
    } while (0)
#else
#define B1_TS(slot) do { } while (0)
#endif

// LDS carve-up (bytes from the start; SPLIT: 158 KB).  The image area comes first: the attention loops address it as
// (per-lane base + pair offset) + immediate.
template <bool SPLIT>
struct B1Lds {
    static constexpr int NIMG = SPLIT ? 4 : 2;                     // image halves in the area
    static constexpr int WST = SPLIT ? 2 * ST_WIMG : ST_WIMG;      // elements of a weight slot
    static constexpr int IST = SPLIT ? 2 * SB_IMG : SB_IMG;        // elements of a weight-gradient image slot (4 slots = the area)
    static constexpr int MATB = (SPLIT ? 2 : 1) * B1_FSTR * 2;     // bytes from the first matrix's images to the second's
    static constexpr int LOB = B1_FSTR * 2;                        // bytes from a hi image to its lo image
    static constexpr int W_OFF = NIMG * B1_FSTR;                   // elements
    static constexpr int F_OFF_BYTES = (W_OFF + 2 * WST) * 2;
    // float vectors behind the weights
    static constexpr int GAM = 0, GAMF = 64, PART = 128, PARTF = PART + 2 * SB_WAVES * 64, SDEL = PARTF + 2 * SB_WAVES * 64,
                         KB = SDEL + B1_ROWS, SMX = KB + B1_ROWS + 16, SINV = SMX + B1_ROWS, SUNI = SINV + B1_ROWS, SQV = SUNI + B1_ROWS,
                         TFLAG = SQV + B1_ROWS, NFLOAT = TFLAG + 16;
    static constexpr int BYTES = F_OFF_BYTES + NFLOAT * 4;
};

// accumulators D[in = 16 it + 4 lg + r][out = 16 (jt0 + j) + li] -> slab (row pitch ldw), row D = the bias gradient; `add`: the
// workgroup's second and later sequences add to what its first one stored
__device__ __forceinline__ void b1_wstore(float* dst, int ldw, float* bias_dst, const f32x4 (&acc)[2], int D, int it, int jt0, bool add) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = 16 * (jt0 + j) + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * it + 4 * lg + r;
            if (col < D && k <= D) {
                float* p = (k < D) ? dst + (size_t)k * ldw + col : bias_dst + col;
                *p = add ? *p + acc[j][r] : acc[j][r];
            }
        }
    }
}
// LayerNorm column sums: per-lane partials -> sums over the wave's 16 rows -> the wave's LDS slot (+=: a wave folds once per
// tile); b1_ln_flush adds the eight slots in wave order
__device__ __forceinline__ void b1_ln_fold(float* part, const f32x4 (&ag)[4], const f32x4 (&ab)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sg = cr_row16_sum(ag[ct][r]), sb = cr_row16_sum(ab[ct][r]);
            if (li == 0) {
                part[wave * 64 + 16 * ct + 4 * lg + r] += sg;
                part[(SB_WAVES + wave) * 64 + 16 * ct + 4 * lg + r] += sb;
            }
        }
}
__device__ __forceinline__ void b1_ln_flush(const float* part, float* dg, float* db, int D, bool add) {
    for (int c = threadIdx.x; c < D; c += SB_NT) {
        float g = 0.0f, b = 0.0f;
#pragma unroll
        for (int w = 0; w < SB_WAVES; ++w) { g += part[w * 64 + c]; b += part[(SB_WAVES + w) * 64 + c]; }
        dg[c] = add ? dg[c] + g : g;
        db[c] = add ? db[c] + b : b;
    }
}

// =====================================================================================================
// phase 1: LN2 + feed-forward backward of every tile of sequence n (both sides)
// =====================================================================================================
template <bool SPLIT, int DS, bool QSIDE>
__device__ __forceinline__ void b1_phase1(const B1Args& a, unsigned char* smem, int n, bool add) {
    typedef B1Lds<SPLIT> L;
    const cr_block_bwd_desc& bd = a.bd;
    const cr_block_desc& d = bd.f;
    __bf16* Im = reinterpret_cast<__bf16*>(smem);
    __bf16* Wi = Im + L::W_OFF;                           // slot 0: W1 (permuted), slot 1: W2
    float* fl = reinterpret_cast<float*>(smem + L::F_OFF_BYTES);
    float* gam = fl + L::GAM; float* gamF = fl + L::GAMF; float* part = fl + L::PART; float* partF = fl + L::PARTF; float* sdel = fl + L::SDEL;
    constexpr int WST = L::WST, IST = L::IST;
    const int D = DS > 0 ? DS : d.D, T = a.T;
    const DCtx dcx = d_ctx(D);
    const int wave = threadIdx.x >> 6;
    const DropCtx d2 = drop_ctx(d.drop_ffn2);
    const float scale1 = (d.drop_ffn1.rate > 0.0f) ? 1.0f / (1.0f - d.drop_ffn1.rate) : 1.0f;
    f32x4 aw1[2], aw2[2], nob[2], ag[4], ab[4], agF[4], abF[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) { aw1[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; aw2[j] = aw1[j]; nob[j] = aw1[j]; }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { ag[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; ab[ct] = ag[ct]; agF[ct] = ag[ct]; abF[ct] = ag[ct]; }
    const int it = wave >> 1, jt0 = 2 * (wave & 1);
    RRaw rdy, rdy2, rhid, rfin, ro, rq, ry;               // rq: q_in (delta); rfin: f_in (Q side) / the tile's Q rows (K side)
    const size_t MD = (size_t)d.M * D;
    auto tile_rows = [&](int rd, int& m, bool& rok) {
        const int q = 16 * (rd * SB_TPR + wave) + (lane_now() & 15);
        rok = q < T;
        m = n * T + min(q, T - 1);
    };
    auto issue = [&](int rd) {
        if (rd * SB_TPR < a.nkt && wave < min(SB_TPR, a.nkt - rd * SB_TPR)) {
            int m; bool rok;
            tile_rows(rd, m, rok);
            const u32 mo = (u32)m * (u32)(4 * D);
            if (a.has_ln) {                                          // rows beyond T: zero gradient (they must not reach the LayerNorm sums)
                r_issue(rdy, a.ln.dy, (u32)m * (u32)(4 * a.ln.lddy), dcx, rok);
                if (a.ln_dy2) r_issue(rdy2, a.ln_dy2, (u32)m * (u32)(4 * a.ln.lddy), dcx, rok);
                r_issue(ry, d.y, mo, dcx);
            } else {
                r_issue(rdy, bd.dy, mo, dcx);
                if (a.dy2) r_issue(rdy2, a.dy2, mo, dcx);
            }
            r_issue(rhid, d.hid, mo, dcx);
            r_issue(rfin, QSIDE ? d.f_in : d.qkv, mo, dcx);
            r_issue(ro, d.o, mo, dcx);
            r_issue(rq, d.q_in, mo, dcx);
        }
    };
    const int R = a.nkt > SB_TPR ? 2 : 1;
    {
        // (memory returns in order: the weights are requested first, or their staging would wait for the tile as well)
        WRegs<2, SB_NT> w;
        w_issue<2, SB_NT>(w, D, d.w1, D, 0, d.w2, D, 0, d.w2, D, 0);
        const int t = threadIdx.x;
        const float gv = (t < D) ? d.ln2_g[t] : 0.0f;
        const float gf = (a.has_ln && t < D) ? a.ln.gamma[t] : 0.0f;
        issue(0);
        w_put_perm<2, SB_NT, SPLIT>(Wi, w, D, d.w1, D, 0, d.w2, D, 0, d.w2, D, 0);
        if (t < 64) { gam[t] = gv; gamF[t] = gf; }
        if (!QSIDE && (a.nkt & 1)) {
            // an odd tile count: the key-owner pass reads the absent second tile of the last pair (with zero coefficients):
            // its rows must hold finite values (16 rows x 128 bytes per image half, 16 bytes per thread)
            const int im = t >> 7, o16 = t & 127;
            if (im < L::NIMG) *reinterpret_cast<float4*>(smem + (size_t)im * L::LOB + (size_t)(16 * a.nkt) * 128 + 16 * o16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    B1_TS(1);
#pragma unroll 1
    for (int rd = 0; rd < R; ++rd) {
        const int ntr = min(SB_TPR, a.nkt - rd * SB_TPR);          // tiles of this round (wave-uniform)
        if (wave < ntr) {
            const int lg = lane_now() >> 4;
            int m; bool rok;
            tile_rows(rd, m, rok);
            const u32 mo = (u32)m * (u32)(4 * D);
            const int trow = 16 * (rd * SB_TPR + wave);              // the tile's first row in the sequence
            const float msk = (rok && d.mask_ids[m] != 0) ? 1.0f : 0.0f;    // rows beyond T contribute nothing
            f32x4 dy[4], g2[4], hid[4];
            r_finish(dy, rdy, dcx);
            if (a.has_ln ? (a.ln_dy2 != nullptr) : (a.dy2 != nullptr)) {
                f32x4 t2[4];
                r_finish(t2, rdy2, dcx);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) dy[ct] += t2[ct];
            }
            if (a.has_ln) {
                // dy = backward of the stack's final LayerNorm (sasrec.py:85) on the gradient rows, x = this block's y
                f32x4 yv[4], dyo[4];
                r_finish(yv, ry, dcx);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) dyo[ct] = dy[ct];
                r_ln_bwd(dy, yv, dyo, gamF, agF, abF, dcx);
            }
            // g2 = dy * mask * keep2 / (1 - rate) (sasrec.py:83, modules.py:309-310)
            const uint32_t e2 = ((d.drop_ffn2.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)(4 * lg)) * CR_PHI + d2.key;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dy[ct][r] *= msk;
                    float v = dy[ct][r];
                    if (d2.on) v *= drop_factor_x(d2, e2 + (uint32_t)(16 * ct + r) * CR_PHI);
                    g2[ct][r] = v;
                }
            r_finish(hid, rhid, dcx);
            if (QSIDE) {
                img_put<SPLIT>(Im + IST, Im + IST + SB_IMG, 16 * wave, g2);
                f32x4 h1[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) h1[ct] = hid[ct];
                plant_one(h1, D);
                img_put<SPLIT>(Im, Im + SB_IMG, 16 * wave, h1);
            }
            // dhid = g2 W2^T, gated by the stored post-dropout ReLU output -> g1 (modules.py:300-304)
            bf8 gh[2], gl[2];
            f32x4 g1[4];
            r_split<SPLIT>(g2, gh, gl);
            r_gemm_t<SPLIT, false>(g1, Wi + WST, Wi + WST + ST_WIMG, gh, gl);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) g1[ct][r] = (hid[ct][r] > 0.0f) ? g1[ct][r] * scale1 : 0.0f;
            if (QSIDE) {
                img_put<SPLIT>(Im + 3 * IST, Im + 3 * IST + SB_IMG, 16 * wave, g1);
                f32x4 fin[4];
                r_finish(fin, rfin, dcx);
                plant_one(fin, D);
                img_put<SPLIT>(Im + 2 * IST, Im + 2 * IST + SB_IMG, 16 * wave, fin);
            } else {
                // the tile's Q rows -> the Q image (natural column order; rows beyond T are zero)
                f32x4 qv[4];
                r_finish(qv, rfin, dcx);
                if (!rok) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) qv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                img_put<SPLIT>(Im, Im + B1_FSTR, trow, qv);
            }
            // df = g1 W1^T + dy * mask (residual of modules.py:313)
            f32x4 df[4];
            r_split<SPLIT>(g1, gh, gl);
            r_gemm_t<SPLIT, false>(df, Wi, Wi + ST_WIMG, gh, gl);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) df[ct] += dy[ct];
            // LN2 backward: x = o, dy = df -> d_o
            f32x4 o[4], dout[4];
            r_finish(o, ro, dcx);
            r_ln_bwd(dout, o, df, gam, ag, ab, dcx);
            if (QSIDE) {
                r_store(bd.d_o, mo, dout, rok, dcx);
            } else {
                if (!rok) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) dout[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                img_put<SPLIT>(Im + (L::MATB >> 1), Im + (L::MATB >> 1) + B1_FSTR, trow, dout);
            }
            {
                // delta[row] = sum_c d_o[c] * (o[c] - q_in[c]) (the attention core's output is o - q_in, modules.py:262-269)
                f32x4 qin[4];
                r_finish(qin, rq, dcx);
                float acc = 0.0f;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = fmaf(dout[ct][r], o[ct][r] - qin[ct][r], acc);
                acc = grp_sum(acc);
                if (lg == 0) sdel[trow + (lane_now() & 15)] = rok ? acc : 0.0f;
            }
        }
        if (rd + 1 < R) issue(rd + 1);
        if (QSIDE) {
            __syncthreads();
            wgrad_accum<SPLIT, false>(aw2, nob, Im, Im + SB_IMG, Im + IST, Im + IST + SB_IMG, ntr, it, jt0);                          // dW2 (+ db2) += hid^T g2
            wgrad_accum<SPLIT, false>(aw1, nob, Im + 2 * IST, Im + 2 * IST + SB_IMG, Im + 3 * IST, Im + 3 * IST + SB_IMG, ntr, it, jt0);  // dW1 (+ db1) += f_in^T g1
            __syncthreads();
        }
    }
    B1_TS(2);
    if (QSIDE) {
        const size_t so = (size_t)blockIdx.x * bd.slab_stride;      // one slab per workgroup PAIR: the sides write disjoint ranges
        b1_wstore(bd.g_w1 + so, D, bd.g_b1 + so, aw1, D, it, jt0, add);
        b1_wstore(bd.g_w2 + so, D, bd.g_b2 + so, aw2, D, it, jt0, add);
        // LayerNorm column sums: zero the slots, fold, flush
        for (int i = threadIdx.x; i < 4 * SB_WAVES * 64; i += SB_NT) part[i] = 0.0f;       // part and partF are adjacent
        __syncthreads();
        b1_ln_fold(part, ag, ab);
        if (a.has_ln) b1_ln_fold(partF, agF, abF);
        __syncthreads();
        b1_ln_flush(part, bd.g_ln2_g + so, bd.g_ln2_b + so, D, add);
        if (a.has_ln) b1_ln_flush(partF, a.ln.dgamma + so, a.ln.dbeta + so, D, add);
    }
    (void)MD;
}
