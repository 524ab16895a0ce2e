// Backward of a block's row phases (the work of cr_block_ln_ffn_bwd and cr_block_ln_qkv_bwd) in register layout R
// (cr_rlayout.hpp), one workgroup per sequence, bf16 MFMA on split (or plain) operands.
//
// Why.  The fp32 kernels of cr_block.hip run 27-35 us per launch at the headline shape and, like every kernel of this
// step, are bound by instruction issue, not by the matrix pipe (13-20 % busy) or HBM (1.4 TB/s): a 64-row tile goes
// global -> registers -> LDS -> (ds_read_b32 + v_mfma_f32_16x16x4: 9 instructions per 4 k) -> LDS -> registers -> global
// with per-element bookkeeping in between and seven workgroup barriers per tile.  Here a wave owns a 16-row tile and
// the whole chain  dy -> g2 -> g2 W2^T -> ReLU gate -> g1 W1^T + dy -> LayerNorm backward -> d_o  (and the LN1 + Q/K/V
// one) runs through its registers: g @ W^T is out^T = W g^T, A = rows of the [in][out] weight image (one ds_read_b128
// per 8 k, the image kept in the k order of the B operand), B = the lane's own registers, D = layout R again.
//
// Weight gradients dW = a^T g need the ROW index as k, i.e. operands transposed across lanes: each wave writes its
// tile of a (with a column of ones planted at column D: row D of dW is then the bias gradient) and of g as bf16 images
// into LDS, and after a barrier every wave owns two 16 x 16 tiles of dW over ALL rows of the round
// (v_mfma_f32_16x16x16_bf16, both operands through ds_read_b64_tr_b16).  A sequence is processed in two rounds of up
// to 7 tiles (one tile per wave and round) so the images of a round fit the LDS next to the weights; the
// accumulators persist across rounds and sequences, one gradient slab per workgroup, fixed summation order
// (bitwise reproducible), no atomics.
//
// Same inputs, outputs and slab layout as the cr_block_* entry points (castrec.h); results differ from them by the
// rounding of the split products (~1e-5 relative).  Shapes: D <= 64 (D < 64: bias gradients from the ones column; D = 64:
// from an all-ones product), T <= 224 (two rounds of 7 tiles).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cr_rbwd.hpp"

struct SbArgs {
    cr_block_bwd_desc bd;
    cr_embed_bwd_desc sc;         // qkv: optional scatter of dx into the embedding tables (see cr_block_ln_qkv_bwd_scatter)
    int B, T, nkt, scatter;
    int heads;                    // ffn: attn_delta is [heads, M] (1, or 2 heads of 32 columns at D = 64)
    int has_ln;                   // ffn: bd.dy is not read; it is the backward of the stack's final LayerNorm (`ln`) applied to ln.dy
    cr_ln_bwd_desc ln;
    unsigned long long* ts;
};

#ifdef CR_TIMELINE
#define SB_TS(slot)                                                                                          \
    do {                                                                                                     \
        if (a.ts && (threadIdx.x & 63) == 0)                                                                 \
            a.ts[((size_t)blockIdx.x * SB_WAVES + (threadIdx.x >> 6)) * 64 + (slot)] =                       \
                ((slot) == 0 || (slot) == 63) ? wall_clock64() : clock64();                                  \
    } while (0)
#else
#define SB_TS(slot) do { } while (0)
#endif

// =====================================================================================================
// LN2 + feed-forward backward:  dy -> d_o, slabs of dW2 db2 dW1 db1 dgamma2 dbeta2, optional attention delta
// =====================================================================================================
template <bool SPLIT, int DS>
__global__ __launch_bounds__(SB_NT) void k_stack_ffn_bwd(SbArgs a) {
    const cr_block_bwd_desc& bd = a.bd;
    const cr_block_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WST = SPLIT ? 2 * ST_WIMG : ST_WIMG;
    constexpr int IST = SPLIT ? 2 * SB_IMG : SB_IMG;      // elements of one image slot (hi [, lo])
    __bf16* Wi = reinterpret_cast<__bf16*>(smem_raw);     // slot 0: W1 (permuted), slot 1: W2
    __bf16* Im = Wi + 2 * WST;                            // image slots: 0 hid, 1 g2, 2 f_in, 3 g1
    float* gam = reinterpret_cast<float*>(Im + 4 * IST);  // [64] gamma2, zero padded
    float* part = gam + 64;                               // [2][SB_WAVES][64]
    float* gamF = part + 2 * SB_WAVES * 64;               // [64] gamma of the stack's final LayerNorm (has_ln)
    const int D = DS > 0 ? DS : d.D, T = a.T;
    const DCtx dcx = d_ctx(D);
    const int wave = threadIdx.x >> 6;
    SB_TS(0); SB_TS(1);
    const DropCtx d2 = drop_ctx(d.drop_ffn2);
    const float scale1 = (d.drop_ffn1.rate > 0.0f) ? 1.0f / (1.0f - d.drop_ffn1.rate) : 1.0f;
    constexpr bool BIAS = DS == 64;                       // no spare column: bias gradients by an all-ones product
    f32x4 aw1[2], aw2[2], ab1[2], ab2[2], ag[4], ab[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) { aw1[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; aw2[j] = aw1[j]; ab1[j] = aw1[j]; ab2[j] = aw1[j]; }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { ag[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; ab[ct] = ag[ct]; }
    const int it = wave >> 1, jt0 = 2 * (wave & 1);       // this wave's tiles of the weight gradients
    // work items of this workgroup: (sequence n, round rd), n = blockIdx.x, + gridDim.x, ...; the inputs of the NEXT
    // item's tile are requested before the weight-gradient phase of the current one (and the first before the weights
    // are staged): a tile's five row blocks come from HBM, 2-3 us that nothing else would cover
    constexpr bool PF = DS == 50;                        // look-ahead loads only where the register budget allows
    RRaw rdy, rhid, rfin, ro, rq, ry;
    f32x4 agF[4], abF[4];                                 // final-LayerNorm gradient partials (has_ln)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { agF[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; abF[ct] = agF[ct]; }
    auto tile_rows = [&](int n, int rd, int& m, bool& rok) {
        const int q = 16 * (rd * SB_TPR + wave) + (lane_now() & 15);
        rok = q < T;
        m = n * T + min(q, T - 1);
    };
    auto issue = [&](int n, int rd) {
        if (n < a.B && wave < min(SB_TPR, a.nkt - rd * SB_TPR)) {
            int m; bool rok;
            tile_rows(n, rd, m, rok);
            const u32 mo = (u32)m * (u32)(4 * D);
            if (a.has_ln) {                                          // rows beyond T: zero gradient (they must not reach the LayerNorm sums)
                r_issue(rdy, a.ln.dy, (u32)m * (u32)(4 * a.ln.lddy), dcx, rok);
                r_issue(ry, d.y, mo, dcx);
            } else {
                r_issue(rdy, bd.dy, mo, dcx);
            }
            r_issue(rhid, d.hid, mo, dcx);
            r_issue(rfin, d.f_in, mo, dcx);
            r_issue(ro, d.o, mo, dcx);
            if (bd.attn_delta) r_issue(rq, d.q_in, mo, dcx);
        }
    };
    // (memory returns in order: the weights are requested first, or their staging would wait for the tile as well)
    const int R = a.nkt > SB_TPR ? 2 : 1, nitems = R * a.B;
    {
        WRegs<2, SB_NT> w;
        w_issue<2, SB_NT>(w, D, d.w1, D, 0, d.w2, D, 0, d.w2, D, 0);
        const int t = threadIdx.x;
        const float gv = (t < D) ? d.ln2_g[t] : 0.0f;
        const float gf = (a.has_ln && t < D) ? a.ln.gamma[t] : 0.0f;
        if (PF && (int)blockIdx.x < nitems) issue((int)blockIdx.x / R, (int)blockIdx.x % R);
        w_put_perm<2, SB_NT, SPLIT>(Wi, w, D, d.w1, D, 0, d.w2, D, 0, d.w2, D, 0);
        if (t < 64) { gam[t] = gv; gamF[t] = gf; }
    }
    __syncthreads();
    SB_TS(2);
    // work items (sequence n, round rd) are dealt round-robin to the workgroups: with n_slabs >= R * B every workgroup
    // runs ONE round (B = 128 sequences of 13 tiles on 256 CUs)
#pragma unroll 1
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        {
            const int n = item / R, rd = item % R;
            const int ntr = min(SB_TPR, a.nkt - rd * SB_TPR);       // tiles of this round (wave-uniform)
            if (!PF) issue(n, rd);
            if (wave < ntr) {
                const int lg = lane_now() >> 4;
                int m; bool rok;
                tile_rows(n, rd, m, rok);
                const u32 mo = (u32)m * (u32)(4 * D);
                const float msk = (rok && d.mask_ids[m] != 0) ? 1.0f : 0.0f;    // rows beyond T contribute nothing
                f32x4 dy[4], g2[4], hid[4];
                r_finish(dy, rdy, dcx);
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
                img_put<SPLIT>(Im + IST, Im + IST + SB_IMG, 16 * wave, g2);
                {
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
                img_put<SPLIT>(Im + 3 * IST, Im + 3 * IST + SB_IMG, 16 * wave, g1);
                {
                    f32x4 fin[4];
                    r_finish(fin, rfin, dcx);
                    plant_one(fin, D);
                    img_put<SPLIT>(Im + 2 * IST, Im + 2 * IST + SB_IMG, 16 * wave, fin);
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
                r_store(bd.d_o, mo, dout, rok, dcx);
                if (bd.attn_delta) {
                    // delta[m] = sum_c d_o[m][c] * (o[m][c] - q_in[m][c]) (the attention core's output is o - q_in, modules.py:262-269)
                    f32x4 qin[4];
                    r_finish(qin, rq, dcx);
                    float acc = 0.0f, acc2 = 0.0f;        // (two heads: column tiles 0, 1 are head 0, tiles 2, 3 head 1)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (ct < 2) acc = fmaf(dout[ct][r], o[ct][r] - qin[ct][r], acc);
                            else acc2 = fmaf(dout[ct][r], o[ct][r] - qin[ct][r], acc2);
                        }
                    if (a.heads == 2) {
                        acc = grp_sum(acc);
                        acc2 = grp_sum(acc2);
                        if (lg == 0 && rok) { bd.attn_delta[m] = acc; bd.attn_delta[(size_t)d.M + m] = acc2; }
                    } else {
                        acc = grp_sum(acc + acc2);
                        if (lg == 0 && rok) bd.attn_delta[m] = acc;
                    }
                }
            }
            if (PF && item + (int)gridDim.x < nitems) issue((item + (int)gridDim.x) / R, (item + (int)gridDim.x) % R);
            SB_TS(3);
            __syncthreads();
            SB_TS(4);
            wgrad_accum<SPLIT, BIAS>(aw2, ab2, Im, Im + SB_IMG, Im + IST, Im + IST + SB_IMG, ntr, it, jt0);                          // dW2 (+ db2) += hid^T g2
            wgrad_accum<SPLIT, BIAS>(aw1, ab1, Im + 2 * IST, Im + 2 * IST + SB_IMG, Im + 3 * IST, Im + 3 * IST + SB_IMG, ntr, it, jt0);  // dW1 (+ db1) += f_in^T g1
            SB_TS(5);
            __syncthreads();
            SB_TS(6);
        }
    }
    SB_TS(10);
    const size_t so = (size_t)blockIdx.x * bd.slab_stride;
    wgrad_store<BIAS>(bd.g_w1 + so, D, bd.g_b1 + so, aw1, ab1, D, it, jt0);
    wgrad_store<BIAS>(bd.g_w2 + so, D, bd.g_b2 + so, aw2, ab2, D, it, jt0);
    SB_TS(11);
    ln_grads_store(part, ag, ab, bd.g_ln2_g + so, bd.g_ln2_b + so, D);
    if (a.has_ln) ln_grads_store(part, agF, abF, a.ln.dgamma + so, a.ln.dbeta + so, D);
    SB_TS(63);
}

// =====================================================================================================
// LN1 + Q/K/V projections backward:  (dQ|dK|dV, d_o) -> dx (= or +=), slabs of dWqkv dbqkv dgamma1 dbeta1
// =====================================================================================================
template <bool SPLIT, int DS>
__global__ __launch_bounds__(SB_NT) void k_stack_qkv_bwd(SbArgs a) {
    const cr_block_bwd_desc& bd = a.bd;
    const cr_block_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WST = SPLIT ? 2 * ST_WIMG : ST_WIMG;
    constexpr int IST = SPLIT ? 2 * SB_IMG : SB_IMG;
    __bf16* Wi = reinterpret_cast<__bf16*>(smem_raw);     // Wq, Wk, Wv (permuted)
    __bf16* Im = Wi + 3 * WST;                            // image slots: first {q_in, dQ}, then {x, dK, dV}
    float* gam = reinterpret_cast<float*>(Im + 3 * IST);  // [64] gamma1
    float* part = gam + 64;
    // scatter: fp32 scratch [7 waves][16][64]: split build: image slot 2 (free until the second image phase); else appended
    float* scat = SPLIT ? reinterpret_cast<float*>(Im + 2 * IST) : part + 2 * SB_WAVES * 64;
    const int D = DS > 0 ? DS : d.D, T = a.T;
    const DCtx dcx = d_ctx(D);
    const int wave = threadIdx.x >> 6;
    const size_t MD = (size_t)d.M * D;
    // the next work item's first inputs (dQ, q_in, d_o, dK) are requested ahead, as in the feed-forward kernel; dV, x
    // (and the old dx) follow at the head of the item, under its first product.  (Not in the generic-D build: its
    // column-tile predicates are live values too, and the look-ahead registers on top of them spilled 270.)
    constexpr bool PF = DS == 50;
    RRaw rdq, rqin, rdo, rdk;
    auto tile_rows = [&](int n, int rd, int& m, bool& rok) {
        const int q = 16 * (rd * SB_TPR + wave) + (lane_now() & 15);
        rok = q < T;
        m = n * T + min(q, T - 1);
    };
    auto issue = [&](int n, int rd) {
        if (n < a.B && wave < min(SB_TPR, a.nkt - rd * SB_TPR)) {
            int m; bool rok;
            tile_rows(n, rd, m, rok);
            const u32 mo = (u32)m * (u32)(4 * D);
            r_issue(rdq, bd.dqkv, mo, dcx, rok);                     // rows beyond T: zero gradients
            r_issue(rqin, d.q_in, mo, dcx);
            r_issue(rdo, bd.d_o, mo, dcx, rok);
            r_issue(rdk, bd.dqkv + MD, mo, dcx, rok);
        }
    };
    const int R = a.nkt > SB_TPR ? 2 : 1, nitems = R * a.B;
    {
        WRegs<3, SB_NT> w;
        w_issue<3, SB_NT>(w, D, d.wqkv, 3 * D, 0, d.wqkv, 3 * D, D, d.wqkv, 3 * D, 2 * D);
        const int t = threadIdx.x;
        const float gv = (t < D) ? d.ln1_g[t] : 0.0f;
        if (PF && (int)blockIdx.x < nitems) issue((int)blockIdx.x / R, (int)blockIdx.x % R);
        w_put_perm<3, SB_NT, SPLIT>(Wi, w, D, d.wqkv, 3 * D, 0, d.wqkv, 3 * D, D, d.wqkv, 3 * D, 2 * D);
        if (t < 64) gam[t] = gv;
    }
    constexpr bool BIAS = DS == 64;
    f32x4 awq[2], awk[2], awv[2], abq[2], abk[2], abv[2], ag[4], ab[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) { awq[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; awk[j] = awq[j]; awv[j] = awq[j]; abq[j] = awq[j]; abk[j] = awq[j]; abv[j] = awq[j]; }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { ag[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; ab[ct] = ag[ct]; }
    const int it = wave >> 1, jt0 = 2 * (wave & 1);
    __syncthreads();
#pragma unroll 1
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        {
            const int n = item / R, rd = item % R;
            const int ntr = min(SB_TPR, a.nkt - rd * SB_TPR);
            const bool active = wave < ntr;
            f32x4 x[4], dK[4], dV[4];                     // kept for the second image phase
            if (!PF) issue(n, rd);
            if (active) {
                int m; bool rok;
                tile_rows(n, rd, m, rok);
                const u32 mo = (u32)m * (u32)(4 * D);
                RRaw rdv, rx, rdx;
                r_issue(rdv, bd.dqkv + 2 * MD, mo, dcx, rok);
                r_issue(rx, d.x, mo, dcx);
                if (bd.dx_accumulate) r_issue(rdx, bd.dx, mo, dcx, rok);
                f32x4 dQ[4], dqin[4], dxp[4];
                bf8 gh[2], gl[2];
                r_finish(dQ, rdq, dcx);
                img_put<SPLIT>(Im + IST, Im + IST + SB_IMG, 16 * wave, dQ);
                {
                    f32x4 qin[4];
                    r_finish(qin, rqin, dcx);
                    plant_one(qin, D);
                    img_put<SPLIT>(Im, Im + SB_IMG, 16 * wave, qin);
                }
                // dq_in = dQ Wq^T + d_o (residual branch, modules.py:269)
                r_split<SPLIT>(dQ, gh, gl);
                r_gemm_t<SPLIT, false>(dqin, Wi, Wi + ST_WIMG, gh, gl);
                {
                    f32x4 dob[4];
                    r_finish(dob, rdo, dcx);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) dqin[ct] += dob[ct];
                }
                // dx_part = dK Wk^T + dV Wv^T
                r_finish(dK, rdk, dcx);
                r_split<SPLIT>(dK, gh, gl);
                r_gemm_t<SPLIT, false>(dxp, Wi + WST, Wi + WST + ST_WIMG, gh, gl);
                r_finish(dV, rdv, dcx);
                r_split<SPLIT>(dV, gh, gl);
                r_gemm_t<SPLIT, true>(dxp, Wi + 2 * WST, Wi + 2 * WST + ST_WIMG, gh, gl);
                // dx = dx_part + LN1 backward(dq_in; x)
                r_finish(x, rx, dcx);
                f32x4 dxl[4];
                r_ln_bwd(dxl, x, dqin, gam, ag, ab, dcx);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) dxl[ct] += dxp[ct];
                if (bd.dx_accumulate) {
                    f32x4 old[4];
                    r_finish(old, rdx, dcx);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) dxl[ct] += old[ct];
                }
                if (!a.scatter) {
                    r_store(bd.dx, mo, dxl, rok, dcx);
                } else {
                    // x was composed by an embedding gather (cr_embed_fwd): its backward applied to the tile right here instead
                    // of storing dx (see cr_block_ln_qkv_bwd_scatter).  g = dx * mask with the gather's dropout regenerated;
                    // d_addend straight from the registers; for the table rows the tile goes through a wave-private fp32
                    // scratch (the image slot the second phase fills later) so that one row = one contiguous float-atomic burst.
                    const cr_embed_desc& e = a.sc.f;
                    const DropCtx dce = drop_ctx(e.drop);
                    const int ln = lane_now(), li = ln & 15, lg = ln >> 4;
                    const int mk = e.mask_ids ? e.mask_ids[m] : 1;
                    const float kf = (rok && mk != 0) ? 1.0f : 0.0f;
                    const uint32_t eb = ((e.drop.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)(4 * lg)) * CR_PHI + dce.key;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = dxl[ct][r] * kf;
                            if (dce.on) v *= drop_factor_x(dce, eb + (uint32_t)(16 * ct + r) * CR_PHI);
                            dxl[ct][r] = v;
                        }
                    if (a.sc.d_addend) r_store(a.sc.d_addend, mo, dxl, rok, dcx);
                    if (a.sc.table_grad || a.sc.pos_grad) {
                        float* scr = scat + wave * (16 * 64);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)     // 16-byte piece (4 ct + lg) of row li, XOR-ed with li: conflict-free both ways
                            *reinterpret_cast<float4*>(scr + li * 64 + 4 * ((4 * ct + lg) ^ li)) = make_float4(dxl[ct][0], dxl[ct][1], dxl[ct][2], dxl[ct][3]);
                        const int my_id = e.ids[m];                                  // lanes 0..15: the id of row li
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const int nr = min(16, T - 16 * (rd * SB_TPR + wave));
                        const int col = ln;
#pragma unroll 4
                        for (int r = 0; r < nr; ++r) {
                            const int id = __shfl(my_id, r, 64);
                            const float g = scr[r * 64 + 4 * ((col >> 2) ^ r) + (col & 3)];
                            if (col < D) {
                                if (a.sc.table_grad && !(e.zero_pad && id == 0)) atomicAdd(a.sc.table_grad + (size_t)id * D + col, g * e.scale);
                                if (a.sc.pos_grad) atomicAdd(a.sc.pos_grad + (size_t)((n * T + 16 * (rd * SB_TPR + wave) + r) % e.T) * D + col, g);
                            }
                        }
                    }
                }
            }
            if (PF && item + (int)gridDim.x < nitems) issue((item + (int)gridDim.x) / R, (item + (int)gridDim.x) % R);
            __syncthreads();
            wgrad_accum<SPLIT, BIAS>(awq, abq, Im, Im + SB_IMG, Im + IST, Im + IST + SB_IMG, ntr, it, jt0);       // dWq (+ dbq) += q_in^T dQ
            __syncthreads();
            if (active) {
                plant_one(x, D);
                img_put<SPLIT>(Im, Im + SB_IMG, 16 * wave, x);
                img_put<SPLIT>(Im + IST, Im + IST + SB_IMG, 16 * wave, dK);
                img_put<SPLIT>(Im + 2 * IST, Im + 2 * IST + SB_IMG, 16 * wave, dV);
            }
            __syncthreads();
            wgrad_accum<SPLIT, BIAS>(awk, abk, Im, Im + SB_IMG, Im + IST, Im + IST + SB_IMG, ntr, it, jt0);                // dWk (+ dbk) += x^T dK
            wgrad_accum<SPLIT, BIAS>(awv, abv, Im, Im + SB_IMG, Im + 2 * IST, Im + 2 * IST + SB_IMG, ntr, it, jt0);        // dWv (+ dbv) += x^T dV
            __syncthreads();
        }
    }
    const size_t so = (size_t)blockIdx.x * bd.slab_stride;
    wgrad_store<BIAS>(bd.g_wqkv + so, 3 * D, bd.g_bqkv + so, awq, abq, D, it, jt0);
    wgrad_store<BIAS>(bd.g_wqkv + so + D, 3 * D, bd.g_bqkv + so + D, awk, abk, D, it, jt0);
    wgrad_store<BIAS>(bd.g_wqkv + so + 2 * D, 3 * D, bd.g_bqkv + so + 2 * D, awv, abv, D, it, jt0);
    ln_grads_store(part, ag, ab, bd.g_ln1_g + so, bd.g_ln1_b + so, D);
}

// =====================================================================================================
// host side
// =====================================================================================================
static const char* sb_unsupported(const cr_block_bwd_desc* bd, int B, int T, int precision) {
    if (!bd) return "NULL description";
    const cr_block_desc& d = bd->f;
    if (d.D < 8 || d.D > 64) return "hidden size 8..64";
    if (precision != CR_PREC_BF16X3 && precision != CR_PREC_BF16) return "bf16 arithmetic (precision) only";
    if (B < 1 || T < 1 || d.M != B * T) return "M = B T";
    if ((T + 15) / 16 > 2 * SB_TPR) return "T <= 224 (two rounds of 7 row tiles)";
    if ((size_t)d.M * d.D * 4 >= ((size_t)1 << 32)) return "activations of 4 GiB or more (32-bit row offsets)";
    if (bd->n_slabs < 1) return "n_slabs";
    return nullptr;
}
extern "C" int cr_stack_bwd_supported(const cr_block_bwd_desc* bd, int B, int T, int precision) {
    return sb_unsupported(bd, B, T, precision) == nullptr;
}

static size_t sb_lds(int nw, int nimg, bool split) {
    return (size_t)nw * ST_WIMG * 2 * (split ? 2 : 1) + (size_t)nimg * SB_IMG * 2 * (split ? 2 : 1) + 64 * 4 + 2 * SB_WAVES * 64 * 4;
}

template <bool SPLIT, int DS>
static int launch_ffn_bwd(const SbArgs& a, hipStream_t s) {
    static cr_devmask attr = 0;
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_stack_ffn_bwd<SPLIT, DS>), &attr);
    if (rc) return rc;
    hipLaunchKernelGGL((k_stack_ffn_bwd<SPLIT, DS>), dim3(a.bd.n_slabs), dim3(SB_NT), sb_lds(2, 4, SPLIT) + 64 * 4, s, a);
    return cr_check_launch("cr_stack_ffn_bwd");
}
template <bool SPLIT, int DS>
static int launch_qkv_bwd(const SbArgs& a, hipStream_t s) {
    static cr_devmask attr = 0;
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_stack_qkv_bwd<SPLIT, DS>), &attr);
    if (rc) return rc;
    hipLaunchKernelGGL((k_stack_qkv_bwd<SPLIT, DS>), dim3(a.bd.n_slabs), dim3(SB_NT), sb_lds(3, 3, SPLIT) + ((a.scatter && !SPLIT) ? (size_t)SB_TPR * 16 * 64 * 4 : 0), s, a);
    return cr_check_launch("cr_stack_qkv_bwd");
}

static int sb_args(SbArgs* a, const cr_block_bwd_desc* bd, int B, int T, int precision, const char* who) {
    const char* why = sb_unsupported(bd, B, T, precision);
    CR_REQUIRE(why == nullptr, "%s: unsupported (%s)", who, why ? why : "");
    memset(static_cast<void*>(a), 0, sizeof(*a));
    a->bd = *bd;
    a->B = B; a->T = T; a->nkt = (T + 15) / 16;
    a->ts = nullptr;
    return CR_OK;
}

static int stack_ffn_bwd_any(const cr_block_bwd_desc* bd, const cr_ln_bwd_desc* n, int B, int T, int heads, int precision, void* stream, const char* who) {
    SbArgs a;
    int rc = sb_args(&a, bd, B, T, precision, who);
    if (rc) return rc;
    const cr_block_desc* d = &bd->f;
    CR_REQUIRE(heads == 1 || (heads == 2 && d->D == 64), "%s: attn_delta per head is formed for one head, or two heads at D = 64", who);
    a.heads = heads;
    CR_REQUIRE(bd->d_o && d->hid && d->f_in && d->o && d->mask_ids && d->w1 && d->w2 && d->ln2_g, "%s: NULL pointer", who);
    CR_REQUIRE(bd->g_w1 && bd->g_b1 && bd->g_w2 && bd->g_b2 && bd->g_ln2_g && bd->g_ln2_b, "%s: NULL gradient pointer", who);
    CR_REQUIRE(bd->attn_delta == nullptr || d->q_in != nullptr, "%s: attn_delta needs q_in", who);
    if (n) {
        CR_REQUIRE(n->x == d->y && n->ldx == d->D && n->M == d->M && n->D == d->D, "%s: the LayerNorm's input must be this block's y", who);
        CR_REQUIRE(n->gamma && n->dy && n->dgamma && n->dbeta && n->accumulate == 0, "%s: LayerNorm backward arguments", who);
        CR_REQUIRE(n->slab_stride == bd->slab_stride && n->n_slabs == bd->n_slabs, "%s: the LayerNorm's slabs must be the block's", who);
        CR_REQUIRE((size_t)n->M * n->lddy * 4 < ((size_t)1 << 32), "%s: dy of 4 GiB or more", who);
        a.ln = *n;
        a.has_ln = 1;
    } else {
        CR_REQUIRE(bd->dy, "%s: dy is NULL", who);
    }
    a.ts = g_attn_ts_which == 8 ? g_attn_ts : nullptr;
    const bool split = precision == CR_PREC_BF16X3;
    hipStream_t s = cr_stream(stream);
    if (d->D == 50) return split ? launch_ffn_bwd<true, 50>(a, s) : launch_ffn_bwd<false, 50>(a, s);
    if (d->D == 64) return split ? launch_ffn_bwd<true, 64>(a, s) : launch_ffn_bwd<false, 64>(a, s);
    return split ? launch_ffn_bwd<true, 0>(a, s) : launch_ffn_bwd<false, 0>(a, s);
}
extern "C" int cr_stack_ffn_bwd(const cr_block_bwd_desc* bd, int B, int T, int precision, void* stream) {
    return stack_ffn_bwd_any(bd, nullptr, B, T, 1, precision, stream, "cr_stack_ffn_bwd");
}
extern "C" int cr_stack_ffn_bwd_ln(const cr_block_bwd_desc* bd, const cr_ln_bwd_desc* n, int B, int T, int precision, void* stream) {
    CR_REQUIRE(n != nullptr, "cr_stack_ffn_bwd_ln: NULL LayerNorm description");
    return stack_ffn_bwd_any(bd, n, B, T, 1, precision, stream, "cr_stack_ffn_bwd_ln");
}
extern "C" int cr_stack_ffn_bwd_heads(const cr_block_bwd_desc* bd, const cr_ln_bwd_desc* n, int B, int T, int heads, int precision, void* stream) {
    return stack_ffn_bwd_any(bd, n, B, T, heads, precision, stream, "cr_stack_ffn_bwd_heads");
}

static int stack_qkv_bwd_any(const cr_block_bwd_desc* bd, const cr_embed_bwd_desc* sc, int B, int T, int precision, void* stream, const char* who) {
    SbArgs a;
    int rc = sb_args(&a, bd, B, T, precision, who);
    if (rc) return rc;
    const cr_block_desc* d = &bd->f;
    CR_REQUIRE(bd->dqkv && bd->d_o && d->q_in && d->x && d->wqkv && d->ln1_g, "%s: NULL pointer", who);
    CR_REQUIRE(bd->g_wqkv && bd->g_bqkv && bd->g_ln1_g && bd->g_ln1_b, "%s: NULL gradient pointer", who);
    CR_REQUIRE(bd->dq_part == nullptr, "%s: dq_part (single-pass fp32 attention backward) is not taken", who);
    if (sc) {
        const cr_embed_desc* e = &sc->f;
        CR_REQUIRE(e->ids && e->M == d->M && e->D == d->D && e->ld_out == d->D && e->col_off == 0 && e->T > 0 && e->V > 0,
                   "%s: the embedding recipe must describe the block's dense input x", who);
        CR_REQUIRE(sc->n_slabs == 0, "%s: small-table mode is not fused", who);
        CR_REQUIRE(!bd->dx_accumulate, "%s: dx_accumulate is not supported (this kernel must be the only producer of dx)", who);
        CR_REQUIRE(sc->table_grad || sc->d_addend || sc->pos_grad, "%s: nothing to scatter into", who);
        CR_REQUIRE(sc->d_addend == nullptr || e->ld_add == d->D, "%s: d_addend must be dense [M, D]", who);
        a.sc = *sc;
        a.scatter = 1;
    } else {
        CR_REQUIRE(bd->dx, "%s: dx is NULL", who);
    }
    const bool split = precision == CR_PREC_BF16X3;
    hipStream_t s = cr_stream(stream);
    if (d->D == 50) return split ? launch_qkv_bwd<true, 50>(a, s) : launch_qkv_bwd<false, 50>(a, s);
    if (d->D == 64) return split ? launch_qkv_bwd<true, 64>(a, s) : launch_qkv_bwd<false, 64>(a, s);
    return split ? launch_qkv_bwd<true, 0>(a, s) : launch_qkv_bwd<false, 0>(a, s);
}
extern "C" int cr_stack_qkv_bwd(const cr_block_bwd_desc* bd, int B, int T, int precision, void* stream) {
    return stack_qkv_bwd_any(bd, nullptr, B, T, precision, stream, "cr_stack_qkv_bwd");
}
extern "C" int cr_stack_qkv_bwd_scatter(const cr_block_bwd_desc* bd, const cr_embed_bwd_desc* sc, int B, int T, int precision, void* stream) {
    CR_REQUIRE(sc != nullptr, "cr_stack_qkv_bwd_scatter: NULL embedding recipe");
    return stack_qkv_bwd_any(bd, sc, B, T, precision, stream, "cr_stack_qkv_bwd_scatter");
}
