// Row phases of a transformer block for hidden sizes 128 / 192 / 256 (configs C4 / C5) on the bf16 matrix pipe:
// the same four steps as cr_block.hip (D <= 64), modules.py:53-80,203-205,280-318 and their backward,
//   cr_wide_ln_qkv_fwd : q_in = LN1(x) (+ key / query masks); Q = q_in Wq + bq; K = x Wk + bk; V = x Wv + bv
//   cr_wide_ln_ffn_fwd : f_in = LN2(o); hid = drop(relu(f_in W1 + b1)); y = (drop(hid W2 + b2) + f_in) * mask
//   cr_wide_ln_ffn_bwd : dy -> g2, g1 (the operands of the weight-gradient products), d_o, slabs of dgamma2 dbeta2
//   cr_wide_ln_qkv_bwd : (dQ|dK|dV, d_o) -> dx (= or +=), slabs of dgamma1 dbeta1
// each ONE launch where the unfused path runs 3 to 6 (cr_layernorm_*, cr_gemm_rows, cr_eltwise).  The weight gradients
// stay with cr_gemm_wgrad (contraction over rows).
//
// Structure.  A workgroup of 8 waves owns 128 rows, a wave 16 of them, in register layout R (cr_rlayout.hpp): lane
// (li, lg) holds row li, columns 16 ct + 4 lg + r -- the D-operand layout of v_mfma_f32_16x16x32_bf16 for the
// transposed product out^T = W^T x^T and, read as two column tiles per k-step, its B operand: LayerNorm -> projection
// -> ... chains run through registers with no transposition.  A [D x D] weight does not fit LDS beside its
// neighbours at these sizes (256 x 256 split into bf16 hi + lo is 256 KB), so the weights STREAM through LDS as
// PANELS of 64 output columns, double buffered: while the waves multiply panel i, every thread holds its share of
// panel i + 1 in registers (issued before the multiply, converted and written after it); one barrier per panel.
//   forward  (out = x W):    panel = W[:, 64 p .. 64 p + 63], image [D][64], A operand by transposed reads
//                            (ds_read_b64_tr_b16), k order = layout R's column order;
//   backward (out = g W^T):  panel = rows 64 p .. 64 p + 63 of W, image [64][D] stored in that same k order
//                            (position 32 ks + 8 lg + 4 h + r <-> column 32 ks + 16 h + 4 lg + r), A operand by 16-byte
//                            row reads.
// Both images use the conflict-free swizzle of cr_bf16.hpp on [rows][64] pieces.
// Column sums (dgamma, dbeta) are DPP row reductions per wave, folded over the waves through LDS slots in a fixed order
// and written as one slab per workgroup (no atomics, bitwise reproducible), as cr_layernorm_bwd does.
#include "cr_attn_common.hpp"
#include "cr_bf16.hpp"

#define WD_NT 512                 // threads per workgroup
#define WD_ROWS 128               // rows per workgroup pass

// ---- layout R rows of a dense [*, D] matrix, D = 16 NCT exactly (no boundary cases at these sizes) -------------
template <int NCT>
__device__ __forceinline__ void wr_load(f32x4 (&x)[NCT], const float* base, int m, bool rok) {
    const int lg = (threadIdx.x & 63) >> 4;
    const float* p = base + (size_t)(rok ? m : 0) * (16 * NCT) + 4 * lg;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const f4u t = *reinterpret_cast<const f4u*>(p + 16 * ct);
        x[ct] = rok ? (f32x4){t.x, t.y, t.z, t.w} : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}
template <int NCT>
__device__ __forceinline__ void wr_store(float* base, int m, bool rok, const f32x4 (&x)[NCT]) {
    const int lg = (threadIdx.x & 63) >> 4;
    float* p = base + (size_t)(rok ? m : 0) * (16 * NCT) + 4 * lg;
    if (rok) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f4u*>(p + 16 * ct) = (f4u){x[ct][0], x[ct][1], x[ct][2], x[ct][3]};
    }
}
// the four column tiles of panel p (columns 64 p .. 64 p + 63)
__device__ __forceinline__ void wr_load4(f32x4 (&x)[4], const float* base, int D, int m, bool rok, int p) {
    const int lg = (threadIdx.x & 63) >> 4;
    const float* q = base + (size_t)(rok ? m : 0) * D + 64 * p + 4 * lg;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const f4u t = *reinterpret_cast<const f4u*>(q + 16 * ct);
        x[ct] = rok ? (f32x4){t.x, t.y, t.z, t.w} : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ void wr_store4(float* base, int D, int m, bool rok, int p, const f32x4 (&x)[4]) {
    const int lg = (threadIdx.x & 63) >> 4;
    float* q = base + (size_t)(rok ? m : 0) * D + 64 * p + 4 * lg;
    if (rok) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<f4u*>(q + 16 * ct) = (f4u){x[ct][0], x[ct][1], x[ct][2], x[ct][3]};
    }
}
// a [D] vector's entries at the lane's columns of panel p
__device__ __forceinline__ void wr_vec4(f32x4 (&v)[4], const float* vec, int p) {
    const int lg = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const f4u t = *reinterpret_cast<const f4u*>(vec + 64 * p + 16 * ct + 4 * lg);
        v[ct] = (f32x4){t.x, t.y, t.z, t.w};
    }
}
template <int NCT>
__device__ __forceinline__ float wr_rowsum(const f32x4 (&x)[NCT]) {
    float s = 0.0f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s += (x[ct][0] + x[ct][1]) + (x[ct][2] + x[ct][3]);
    return grp_sum(s);
}
// mean and 1 / sd of the lane's row (modules.py:74-76: variance + epsilon inside the root)
template <int NCT>
__device__ __forceinline__ void wr_stats(const f32x4 (&x)[NCT], float& mean, float& rs, float& sum) {
    constexpr float invD = 1.0f / (16 * NCT);
    sum = wr_rowsum<NCT>(x);
    mean = sum * invD;
    float v = 0.0f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float c = x[ct][r] - mean;
            v = fmaf(c, c, v);
        }
    rs = 1.0f / sqrtf(grp_sum(v) * invD + 1e-8f);
}
// operand form of a layout-R row: k-step ks <- column tiles 2 ks, 2 ks + 1
template <int NCT, bool SPLIT>
__device__ __forceinline__ void wr_split(const f32x4 (&x)[NCT], bf8 (&h)[NCT / 2], bf8 (&l)[NCT / 2]) {
#pragma unroll
    for (int ks = 0; ks < NCT / 2; ++ks) {
        const float v[8] = {x[2 * ks][0], x[2 * ks][1], x[2 * ks][2], x[2 * ks][3],
                            x[2 * ks + 1][0], x[2 * ks + 1][1], x[2 * ks + 1][2], x[2 * ks + 1][3]};
        split8<SPLIT>(v, h[ks], l[ks]);
    }
}
template <bool SPLIT>
__device__ __forceinline__ void wr_split2(const f32x4& a, const f32x4& b, bf8& h, bf8& l) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    split8<SPLIT>(v, h, l);
}

// ---- weight panels ----------------------------------------------------------------------------------------
// An item is 8 consecutive floats of a weight row (two dword-aligned 16-byte loads).  D * 8 items per panel,
// D / 64 per thread.
template <int NCT> struct PanelRegs { float v[NCT / 4][8]; };

// forward panel: W[k][c0 .. c0 + 63], k < D; item = (k, chunk of 8 columns)
template <int NCT>
__device__ __forceinline__ void fpanel_issue(PanelRegs<NCT>& r, const float* W, int ld, int c0) {
#pragma unroll
    for (int u = 0; u < NCT / 4; ++u) {
        const int item = threadIdx.x + WD_NT * u;
        const int k = item >> 3, ch = item & 7;
        const float* p = W + (size_t)k * ld + c0 + 8 * ch;
        const f4u a = *reinterpret_cast<const f4u*>(p), b = *reinterpret_cast<const f4u*>(p + 4);
        r.v[u][0] = a.x; r.v[u][1] = a.y; r.v[u][2] = a.z; r.v[u][3] = a.w;
        r.v[u][4] = b.x; r.v[u][5] = b.y; r.v[u][6] = b.z; r.v[u][7] = b.w;
    }
}
template <int NCT, bool SPLIT>
__device__ __forceinline__ void fpanel_put(const PanelRegs<NCT>& r, __bf16* img) {
#pragma unroll
    for (int u = 0; u < NCT / 4; ++u) {
        const int item = threadIdx.x + WD_NT * u;
        const int k = item >> 3, ch = item & 7;
        bf8 h, l;
        split8<SPLIT>(r.v[u], h, l);
        const int o = img_off<2>(k, ch);
        *reinterpret_cast<bf8*>(img + o) = h;
        if (SPLIT) *reinterpret_cast<bf8*>(img + 16 * NCT * 64 + o) = l;
    }
}
// backward panel: W[j0 + j][c0 + n], j < 64, n < D; item = (j, chunk of 8 columns n); stored in the operand's k order
template <int NCT>
__device__ __forceinline__ void bpanel_issue(PanelRegs<NCT>& r, const float* W, int ld, int j0, int c0) {
#pragma unroll
    for (int u = 0; u < NCT / 4; ++u) {
        const int item = threadIdx.x + WD_NT * u;
        const int j = item / (2 * NCT), c8 = item % (2 * NCT);
        const float* p = W + (size_t)(j0 + j) * ld + c0 + 8 * c8;
        const f4u a = *reinterpret_cast<const f4u*>(p), b = *reinterpret_cast<const f4u*>(p + 4);
        r.v[u][0] = a.x; r.v[u][1] = a.y; r.v[u][2] = a.z; r.v[u][3] = a.w;
        r.v[u][4] = b.x; r.v[u][5] = b.y; r.v[u][6] = b.z; r.v[u][7] = b.w;
    }
}
template <int NCT, bool SPLIT>
__device__ __forceinline__ void bpanel_put(const PanelRegs<NCT>& r, __bf16* img) {
#pragma unroll
    for (int u = 0; u < NCT / 4; ++u) {
        const int item = threadIdx.x + WD_NT * u;
        const int j = item / (2 * NCT), c8 = item % (2 * NCT);
        const int ks = c8 >> 2, q = c8 & 3, h4 = 4 * (q >> 1);
        const int chA = 4 * (ks & 1) + 2 * (q & 1);
        bf8 h, l;
        split8<SPLIT>(r.v[u], h, l);
        __bf16* s = img + (ks >> 1) * 4096;
        const int oa = img_off<2>(j, chA) + h4, ob = img_off<2>(j, chA + 1) + h4;
        *reinterpret_cast<bf4*>(s + oa) = __builtin_shufflevector(h, h, 0, 1, 2, 3);
        *reinterpret_cast<bf4*>(s + ob) = __builtin_shufflevector(h, h, 4, 5, 6, 7);
        if (SPLIT) {
            *reinterpret_cast<bf4*>(s + 16 * NCT * 64 + oa) = __builtin_shufflevector(l, l, 0, 1, 2, 3);
            *reinterpret_cast<bf4*>(s + 16 * NCT * 64 + ob) = __builtin_shufflevector(l, l, 4, 5, 6, 7);
        }
    }
}

__device__ __forceinline__ bf8 wd_tr(const __bf16* img, int ra, int rb, int jt, int lane) {
    const int lg = lane >> 4, idx = lane & 15, q = idx >> 2, p = idx & 3;
    const int ch = 2 * jt + (p >> 1), sub = 4 * (p & 1);
    const bf4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + img_off<2>(ra + 4 * lg + q, ch) + sub));
    const bf4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + img_off<2>(rb + 4 * lg + q, ch) + sub));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// acc[ct] (+)= the lane's row times the panel: four output column tiles; per k-step the four fragments are one batch of reads
template <int NCT, bool SPLIT, bool FWD>
__device__ __forceinline__ void panel_mma(f32x4 (&acc)[4], const __bf16* img, const bf8 (&xh)[NCT / 2], const bf8 (&xl)[NCT / 2]) {
    const int lane = threadIdx.x & 63;
    const __bf16* lo = img + 16 * NCT * 64;
#pragma unroll
    for (int ks = 0; ks < NCT / 2; ++ks) {
        bf8 wh[4], wl[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (FWD) {
                wh[ct] = wd_tr(img, 32 * ks, 32 * ks + 16, ct, lane);
                wl[ct] = SPLIT ? wd_tr(lo, 32 * ks, 32 * ks + 16, ct, lane) : wh[ct];
            } else {
                const int o = (ks >> 1) * 4096 + img_off<2>(16 * ct + (lane & 15), (lane >> 4) + 4 * (ks & 1));
                wh[ct] = *reinterpret_cast<const bf8*>(img + o);
                wl[ct] = SPLIT ? *reinterpret_cast<const bf8*>(lo + o) : wh[ct];
            }
        }
        if (SPLIT) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ct], xh[ks], acc[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ct], xl[ks], acc[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ct], xh[ks], acc[ct], 0, 0, 0);
    }
}
__device__ __forceinline__ void acc_zero(f32x4 (&a)[4]) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) a[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

template <int NCT, bool SPLIT>
struct WideLds {
    static constexpr int HALF = 16 * NCT * 64;                        // bf16 elements of one image half
    static constexpr int BUF = (SPLIT ? 2 : 1) * HALF;                // one panel buffer
    static constexpr size_t PANEL_BYTES = 2 * (size_t)BUF * 2;        // double buffered
    static constexpr size_t SLOT_BYTES = 8 * 2 * 16 * NCT * 4;        // [8 waves][gamma | beta][D] floats (backward kernels)
};

// =====================================================================================================
// forward: LN1 + Q / K / V projections
// =====================================================================================================
template <int NCT, bool SPLIT>
__global__ __launch_bounds__(WD_NT) void k_wide_qkv_fwd(cr_block_desc d) {
    constexpr int D = 16 * NCT, NKS = NCT / 2, NP = NCT / 4, NPAN = 3 * NP;
    typedef WideLds<NCT, SPLIT> LD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* pb = reinterpret_cast<__bf16*>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15;
    const int m = blockIdx.x * WD_ROWS + 16 * wave + li;
    const bool rok = m < d.M;
    PanelRegs<NCT> pr;
    fpanel_issue<NCT>(pr, d.wqkv, 3 * D, 0);
    f32x4 x[NCT];
    wr_load<NCT>(x, d.x, m, rok);
    bf8 oh[NKS], ol[NKS];                                 // the current operand: q_in for Q, x for K and V
    {
        float mean, rs, sum;
        wr_stats<NCT>(x, mean, rs, sum);
        f32x4 q[NCT];
        float ys = 0.0f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            f32x4 g[4], b[4];
            wr_vec4(g, d.ln1_g, p);
            wr_vec4(b, d.ln1_b, p);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = fmaf(g[ct][r], (x[4 * p + ct][r] - mean) * rs, b[ct][r]);
                    q[4 * p + ct][r] = y;
                    ys += y;
                }
        }
        ys = grp_sum(ys);
        wr_store<NCT>(d.q_in, m, rok, q);
        if (rok && (lane >> 4) == 0) {                   // modules.py:222 (keys = x), 248-249 (queries = LN1(x))
            d.k_valid[m] = (sum != 0.0f) ? 1.0f : 0.0f;
            d.q_valid[m] = (ys != 0.0f) ? 1.0f : 0.0f;
        }
        wr_split<NCT, SPLIT>(q, oh, ol);
    }
    fpanel_put<NCT, SPLIT>(pr, pb);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPAN; ++i) {
        const int part = i / NP, p = i % NP;
        if (i + 1 < NPAN) fpanel_issue<NCT>(pr, d.wqkv, 3 * D, ((i + 1) / NP) * D + 64 * ((i + 1) % NP));
        if (i == NP) wr_split<NCT, SPLIT>(x, oh, ol);     // K and V take the un-normalised rows (modules.py:204-205)
        f32x4 acc[4], bias[4];
        wr_vec4(bias, d.bqkv + part * D, p);
        acc_zero(acc);
        panel_mma<NCT, SPLIT, true>(acc, pb + (i & 1) * LD::BUF, oh, ol);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] += bias[ct];
        wr_store4(d.qkv + (size_t)part * d.M * D, D, m, rok, p, acc);
        if (i + 1 < NPAN) fpanel_put<NCT, SPLIT>(pr, pb + ((i + 1) & 1) * LD::BUF);
        __syncthreads();
    }
}

// =====================================================================================================
// forward: LN2 + feed-forward
// =====================================================================================================
template <int NCT, bool SPLIT>
__global__ __launch_bounds__(WD_NT) void k_wide_ffn_fwd(cr_block_desc d) {
    constexpr int D = 16 * NCT, NKS = NCT / 2, NP = NCT / 4, NPAN = 2 * NP;
    typedef WideLds<NCT, SPLIT> LD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* pb = reinterpret_cast<__bf16*>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int m = blockIdx.x * WD_ROWS + 16 * wave + li;
    const bool rok = m < d.M;
    const DropCtx dc1 = drop_ctx(d.drop_ffn1), dc2 = drop_ctx(d.drop_ffn2);
    PanelRegs<NCT> pr;
    fpanel_issue<NCT>(pr, d.w1, D, 0);
    bf8 fh[NKS], fl[NKS], hh[NKS], hl[NKS];
    {
        f32x4 x[NCT];
        wr_load<NCT>(x, d.o, m, rok);
        float mean, rs, sum;
        wr_stats<NCT>(x, mean, rs, sum);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            f32x4 g[4], b[4];
            wr_vec4(g, d.ln2_g, p);
            wr_vec4(b, d.ln2_b, p);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[4 * p + ct][r] = fmaf(g[ct][r], (x[4 * p + ct][r] - mean) * rs, b[ct][r]);
        }
        wr_store<NCT>(d.f_in, m, rok, x);
        wr_split<NCT, SPLIT>(x, fh, fl);
    }
    const int id = rok ? d.mask_ids[m] : 0;
    const uint32_t xrow = ((d.drop_ffn1.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)(4 * lg)) * CR_PHI;
    fpanel_put<NCT, SPLIT>(pr, pb);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPAN; ++i) {
        const int part = i / NP, p = i % NP;
        if (i + 1 < NPAN) fpanel_issue<NCT>(pr, (i + 1) / NP ? d.w2 : d.w1, D, 64 * ((i + 1) % NP));
        f32x4 acc[4], bias[4];
        wr_vec4(bias, part ? d.b2 : d.b1, p);
        acc_zero(acc);
        if (part == 0) {
            panel_mma<NCT, SPLIT, true>(acc, pb + (i & 1) * LD::BUF, fh, fl);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = fmaxf(acc[ct][r] + bias[ct][r], 0.0f);                          // modules.py:300
                    if (dc1.on) v *= drop_factor_x(dc1, xrow + (uint32_t)(64 * p + 16 * ct + r) * CR_PHI + dc1.key);
                    acc[ct][r] = v;
                }
            wr_store4(d.hid, D, m, rok, p, acc);
            wr_split2<SPLIT>(acc[0], acc[1], hh[2 * p], hl[2 * p]);
            wr_split2<SPLIT>(acc[2], acc[3], hh[2 * p + 1], hl[2 * p + 1]);
        } else {
            f32x4 res[4];
            wr_load4(res, d.f_in, D, m, rok, p);          // this lane's own stores (residual = LN2 output, modules.py:313)
            panel_mma<NCT, SPLIT, true>(acc, pb + (i & 1) * LD::BUF, hh, hl);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[ct][r] + bias[ct][r];
                    if (dc2.on) v *= drop_factor_x(dc2, xrow + (uint32_t)(64 * p + 16 * ct + r) * CR_PHI + dc2.key);
                    v += res[ct][r];
                    acc[ct][r] = id ? v : 0.0f;                                              // sasrec.py:83
                }
            wr_store4(d.y, D, m, rok, p, acc);
        }
        if (i + 1 < NPAN) fpanel_put<NCT, SPLIT>(pr, pb + ((i + 1) & 1) * LD::BUF);
        __syncthreads();
    }
}

// ---- column sums of the workgroup's rows into per-thread accumulators ---------------------------------------
// a[ct][r], b[ct][r]: the lane's contributions (row li) to dgamma / dbeta at columns 16 ct + 4 lg + r.  After the call
// thread c < D has added this pass's dgamma[c], thread D + c its dbeta[c] into `tot`.  Two barriers.
template <int NCT>
__device__ __forceinline__ void colsum_fold(float* slots, const f32x4 (&a)[NCT], const f32x4 (&b)[NCT], float& tot) {
    constexpr int D = 16 * NCT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    float* sg = slots + (size_t)wave * 2 * D;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        f32x4 ga, gb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ga[r] = cr_row16_sum(a[ct][r]);
            gb[r] = cr_row16_sum(b[ct][r]);
        }
        if (li == 0) {
            *reinterpret_cast<float4*>(sg + 16 * ct + 4 * lg) = make_float4(ga[0], ga[1], ga[2], ga[3]);
            *reinterpret_cast<float4*>(sg + D + 16 * ct + 4 * lg) = make_float4(gb[0], gb[1], gb[2], gb[3]);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * D) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += slots[(size_t)w * 2 * D + threadIdx.x];
        tot += s;
    }
    __syncthreads();
}

// =====================================================================================================
// backward: feed-forward + LN2
// =====================================================================================================
template <int NCT, bool SPLIT>
__global__ __launch_bounds__(WD_NT) void k_wide_ffn_bwd(cr_block_bwd_desc bd, float* g2out, float* g1out) {
    constexpr int D = 16 * NCT, NKS = NCT / 2, NP = NCT / 4, NPAN = 2 * NP;
    typedef WideLds<NCT, SPLIT> LD;
    const cr_block_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* pb = reinterpret_cast<__bf16*>(smem_raw);
    float* slots = reinterpret_cast<float*>(smem_raw + LD::PANEL_BYTES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const DropCtx dc2 = drop_ctx(d.drop_ffn2);
    const float gate_scale = d.drop_ffn1.rate > 0.0f ? 1.0f / (1.0f - d.drop_ffn1.rate) : 1.0f;
    const int nblk = (d.M + WD_ROWS - 1) / WD_ROWS;
    float tot = 0.0f;
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int m = blk * WD_ROWS + 16 * wave + li;
        const bool rok = m < d.M;
        PanelRegs<NCT> pr;
        bpanel_issue<NCT>(pr, d.w2, D, 0, 0);
        const int id = rok ? d.mask_ids[m] : 0;
        const uint32_t xrow = ((d.drop_ffn2.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)(4 * lg)) * CR_PHI;
        bf8 ah[NKS], al[NKS], gh[NKS], gl[NKS];
        {
            // g2 = dy * dropout(ffn2) * mask: gradient of the second dense layer's output
            f32x4 g[NCT];
            wr_load<NCT>(g, bd.dy, m, rok);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = g[ct][r];
                    if (dc2.on) v *= drop_factor_x(dc2, xrow + (uint32_t)(16 * ct + r) * CR_PHI + dc2.key);
                    g[ct][r] = id ? v : 0.0f;
                }
            wr_store<NCT>(g2out, m, rok, g);
            wr_split<NCT, SPLIT>(g, ah, al);
        }
        f32x4 df[NCT];
        bpanel_put<NCT, SPLIT>(pr, pb);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NPAN; ++i) {
            const int part = i / NP, p = i % NP;
            if (i + 1 < NPAN) bpanel_issue<NCT>(pr, (i + 1) / NP ? d.w1 : d.w2, D, 64 * ((i + 1) % NP), 0);
            f32x4 acc[4], aux[4];
            wr_load4(aux, part ? bd.dy : d.hid, D, m, rok, p);
            acc_zero(acc);
            if (part == 0) {
                // g1 = (g2 W2^T) gated by the stored post-dropout ReLU output (modules.py:300-303)
                panel_mma<NCT, SPLIT, false>(acc, pb + (i & 1) * LD::BUF, ah, al);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ct][r] = aux[ct][r] > 0.0f ? acc[ct][r] * gate_scale : 0.0f;
                wr_store4(g1out, D, m, rok, p, acc);
                wr_split2<SPLIT>(acc[0], acc[1], gh[2 * p], gl[2 * p]);
                wr_split2<SPLIT>(acc[2], acc[3], gh[2 * p + 1], gl[2 * p + 1]);
            } else {
                // df_in = (g1 W1^T + dy) * mask   (residual branch of modules.py:313)
                panel_mma<NCT, SPLIT, false>(acc, pb + (i & 1) * LD::BUF, gh, gl);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) df[4 * p + ct][r] = id ? acc[ct][r] + aux[ct][r] : 0.0f;
            }
            if (i + 1 < NPAN) bpanel_put<NCT, SPLIT>(pr, pb + ((i + 1) & 1) * LD::BUF);
            __syncthreads();
        }
        // LN2 backward (modules.py:74-78): d_o = rstd * (df * gamma - mean(df * gamma) - xhat * mean(df * gamma * xhat))
        {
            f32x4 xh[NCT];
            wr_load<NCT>(xh, d.o, m, rok);
            float mean, rs, sum;
            wr_stats<NCT>(xh, mean, rs, sum);
            float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                f32x4 g[4];
                wr_vec4(g, d.ln2_g, p);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float h = (xh[4 * p + ct][r] - mean) * rs;
                        const float dg = df[4 * p + ct][r] * g[ct][r];
                        xh[4 * p + ct][r] = h;
                        c1 += dg;
                        c2 = fmaf(dg, h, c2);
                    }
            }
            constexpr float invD = 1.0f / D;
            c1 = grp_sum(c1) * invD;
            c2 = grp_sum(c2) * invD;
            float delta = 0.0f;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                f32x4 g[4], dxo[4];
                wr_vec4(g, d.ln2_g, p);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dxo[ct][r] = rs * (df[4 * p + ct][r] * g[ct][r] - c1 - xh[4 * p + ct][r] * c2);
                wr_store4(bd.d_o, D, m, rok, p, dxo);
            }
            (void)delta;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) xh[ct][r] *= df[ct][r];                 // dgamma contributions; dbeta's are df
            colsum_fold<NCT>(slots, xh, df, tot);
        }
    }
    if ((int)threadIdx.x < D) bd.g_ln2_g[(size_t)blockIdx.x * bd.slab_stride + threadIdx.x] = tot;
    else if ((int)threadIdx.x < 2 * D) bd.g_ln2_b[(size_t)blockIdx.x * bd.slab_stride + threadIdx.x - D] = tot;
}

// =====================================================================================================
// backward: Q / K / V projections + LN1
// =====================================================================================================
template <int NCT, bool SPLIT>
__global__ __launch_bounds__(WD_NT) void k_wide_qkv_bwd(cr_block_bwd_desc bd) {
    constexpr int D = 16 * NCT, NKS = NCT / 2, NP = NCT / 4, NPAN = 3 * NP;
    typedef WideLds<NCT, SPLIT> LD;
    const cr_block_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* pb = reinterpret_cast<__bf16*>(smem_raw);
    float* slots = reinterpret_cast<float*>(smem_raw + LD::PANEL_BYTES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15;
    const size_t MD = (size_t)d.M * D;
    const int nblk = (d.M + WD_ROWS - 1) / WD_ROWS;
    float tot = 0.0f;
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int m = blk * WD_ROWS + 16 * wave + li;
        const bool rok = m < d.M;
        PanelRegs<NCT> pr;
        bpanel_issue<NCT>(pr, d.wqkv, 3 * D, 0, 0);
        bf8 ah[NKS], al[NKS];
        {
            f32x4 g[NCT];
            wr_load<NCT>(g, bd.dqkv, m, rok);                              // dQ rows
            wr_split<NCT, SPLIT>(g, ah, al);
        }
        f32x4 dq[NCT];                                                    // dq_in = dQ Wq^T + d_o (residual, modules.py:269)
        bpanel_put<NCT, SPLIT>(pr, pb);
        __syncthreads();
        // panel order: Wq rows p = 0..NP-1; then for every p: Wk rows p, Wv rows p (both into the same accumulators)
#pragma unroll
        for (int i = 0; i < NPAN; ++i) {
            if (i + 1 < NPAN) {
                const int n = i + 1;
                const int part = n < NP ? 0 : 1 + ((n - NP) & 1), p = n < NP ? n : (n - NP) >> 1;
                bpanel_issue<NCT>(pr, d.wqkv, 3 * D, 64 * p, part * D);
            }
            const __bf16* img = pb + (i & 1) * LD::BUF;
            if (i < NP) {
                f32x4 acc[4], res[4];
                wr_load4(res, bd.d_o, D, m, rok, i);
                acc_zero(acc);
                panel_mma<NCT, SPLIT, false>(acc, img, ah, al);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) dq[4 * i + ct] = acc[ct] + res[ct];
                if (i == NP - 1) {
                    // LN1 backward of dq_in with respect to x; the result starts dx (this lane re-reads its own stores below)
                    f32x4 xh[NCT];
                    wr_load<NCT>(xh, d.x, m, rok);
                    float mean, rs, sum;
                    wr_stats<NCT>(xh, mean, rs, sum);
                    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        f32x4 g[4];
                        wr_vec4(g, d.ln1_g, p);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float h = (xh[4 * p + ct][r] - mean) * rs;
                                const float dg = dq[4 * p + ct][r] * g[ct][r];
                                xh[4 * p + ct][r] = h;
                                c1 += dg;
                                c2 = fmaf(dg, h, c2);
                            }
                    }
                    constexpr float invD = 1.0f / D;
                    c1 = grp_sum(c1) * invD;
                    c2 = grp_sum(c2) * invD;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        f32x4 g[4], dxo[4], old[4];
                        wr_vec4(g, d.ln1_g, p);
                        if (bd.dx_accumulate) wr_load4(old, bd.dx, D, m, rok, p);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                dxo[ct][r] = rs * (dq[4 * p + ct][r] * g[ct][r] - c1 - xh[4 * p + ct][r] * c2);
                                if (bd.dx_accumulate) dxo[ct][r] += old[ct][r];
                            }
                        wr_store4(bd.dx, D, m, rok, p, dxo);
                    }
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) xh[ct][r] *= dq[ct][r];
                    // (the fold's barriers are workgroup-uniform: every wave passes here exactly once per row block)
                    colsum_fold<NCT>(slots, xh, dq, tot);
                    f32x4 g[NCT];
                    wr_load<NCT>(g, bd.dqkv + MD, m, rok);                 // dK rows: the operand of the next panel
                    wr_split<NCT, SPLIT>(g, ah, al);
                }
            } else {
                const int j = i - NP, p = j >> 1;
                static_assert(NPAN == 3 * NP, "panel order");
                if ((j & 1) == 0) {
                    f32x4 acc[4];
                    acc_zero(acc);
                    panel_mma<NCT, SPLIT, false>(acc, img, ah, al);        // dK Wk^T
                    f32x4 g[NCT];
                    wr_load<NCT>(g, bd.dqkv + 2 * MD, m, rok);             // dV rows
                    bf8 vh[NKS], vl[NKS];
                    wr_split<NCT, SPLIT>(g, vh, vl);
                    // keep the partial sums in dq's registers of this panel (dq is dead after the LayerNorm backward)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) dq[4 * p + ct] = acc[ct];
                    // the operand swap: next panel is Wv rows p with dV, the one after Wk rows p + 1 with dK again
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) { ah[ks] = vh[ks]; al[ks] = vl[ks]; }
                } else {
                    f32x4 acc[4], old[4];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[ct] = dq[4 * p + ct];
                    panel_mma<NCT, SPLIT, false>(acc, img, ah, al);        // + dV Wv^T
                    wr_load4(old, bd.dx, D, m, rok, p);                    // the LayerNorm part (own stores)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[ct] += old[ct];
                    wr_store4(bd.dx, D, m, rok, p, acc);
                    if (i + 1 < NPAN) {
                        f32x4 g[NCT];
                        wr_load<NCT>(g, bd.dqkv + MD, m, rok);             // dK rows again
                        wr_split<NCT, SPLIT>(g, ah, al);
                    }
                }
            }
            if (i + 1 < NPAN) bpanel_put<NCT, SPLIT>(pr, pb + ((i + 1) & 1) * LD::BUF);
            __syncthreads();
        }
    }
    if ((int)threadIdx.x < D) bd.g_ln1_g[(size_t)blockIdx.x * bd.slab_stride + threadIdx.x] = tot;
    else if ((int)threadIdx.x < 2 * D) bd.g_ln1_b[(size_t)blockIdx.x * bd.slab_stride + threadIdx.x - D] = tot;
}

// =====================================================================================================
// host side
// =====================================================================================================
static const char* wide_why(const cr_block_desc* d, int precision) {
    if (precision != CR_PREC_BF16X3 && precision != CR_PREC_BF16) return "precision must be CR_PREC_BF16X3 or CR_PREC_BF16";
    if (d->D != 128 && d->D != 192 && d->D != 256) return "D must be 128, 192 or 256";
    if (d->M <= 0) return "M <= 0";
    if ((long long)d->M * d->D * 4 >= (1ll << 40)) return "matrix too large";
    return nullptr;
}
extern "C" int cr_wide_supported(const cr_block_desc* d, int precision) { return d && wide_why(d, precision) == nullptr; }

template <typename K, typename... A>
static int wide_launch(K kern, cr_devmask* done, int grid, size_t lds, hipStream_t s, const char* who, A... args) {
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(kern), done);
    if (rc != CR_OK) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WD_NT), lds, s, args...);
    return cr_check_launch(who);
}

#define WIDE_DISPATCH(KERN, GRID, SLOTS, WHO, ...)                                                                        \
    do {                                                                                                                  \
        static cr_devmask done[6];                                                                                        \
        const bool sp = precision == CR_PREC_BF16X3;                                                                      \
        switch (d->D / 16) {                                                                                              \
            case 8:                                                                                                       \
                return sp ? wide_launch(KERN<8, true>, &done[0], GRID, WideLds<8, true>::PANEL_BYTES + (SLOTS ? WideLds<8, true>::SLOT_BYTES : 0), s, WHO, __VA_ARGS__)      \
                          : wide_launch(KERN<8, false>, &done[1], GRID, WideLds<8, false>::PANEL_BYTES + (SLOTS ? WideLds<8, false>::SLOT_BYTES : 0), s, WHO, __VA_ARGS__);  \
            case 12:                                                                                                      \
                return sp ? wide_launch(KERN<12, true>, &done[2], GRID, WideLds<12, true>::PANEL_BYTES + (SLOTS ? WideLds<12, true>::SLOT_BYTES : 0), s, WHO, __VA_ARGS__)   \
                          : wide_launch(KERN<12, false>, &done[3], GRID, WideLds<12, false>::PANEL_BYTES + (SLOTS ? WideLds<12, false>::SLOT_BYTES : 0), s, WHO, __VA_ARGS__); \
            default:                                                                                                      \
                return sp ? wide_launch(KERN<16, true>, &done[4], GRID, WideLds<16, true>::PANEL_BYTES + (SLOTS ? WideLds<16, true>::SLOT_BYTES : 0), s, WHO, __VA_ARGS__)   \
                          : wide_launch(KERN<16, false>, &done[5], GRID, WideLds<16, false>::PANEL_BYTES + (SLOTS ? WideLds<16, false>::SLOT_BYTES : 0), s, WHO, __VA_ARGS__); \
        }                                                                                                                 \
    } while (0)

extern "C" int cr_wide_ln_qkv_fwd(const cr_block_desc* d, int precision, void* stream) {
    CR_REQUIRE(d, "cr_wide_ln_qkv_fwd: NULL description");
    const char* why = wide_why(d, precision);
    if (why) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_wide_ln_qkv_fwd: %s", why);
    CR_REQUIRE(d->x && d->q_in && d->qkv && d->k_valid && d->q_valid && d->wqkv && d->bqkv && d->ln1_g && d->ln1_b,
               "cr_wide_ln_qkv_fwd: NULL pointer");
    hipStream_t s = cr_stream(stream);
    WIDE_DISPATCH(k_wide_qkv_fwd, cr_ceil_div(d->M, WD_ROWS), false, "cr_wide_ln_qkv_fwd", *d);
}

extern "C" int cr_wide_ln_ffn_fwd(const cr_block_desc* d, int precision, void* stream) {
    CR_REQUIRE(d, "cr_wide_ln_ffn_fwd: NULL description");
    const char* why = wide_why(d, precision);
    if (why) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_wide_ln_ffn_fwd: %s", why);
    CR_REQUIRE(d->o && d->f_in && d->hid && d->y && d->mask_ids && d->w1 && d->b1 && d->w2 && d->b2 && d->ln2_g && d->ln2_b,
               "cr_wide_ln_ffn_fwd: NULL pointer");
    hipStream_t s = cr_stream(stream);
    WIDE_DISPATCH(k_wide_ffn_fwd, cr_ceil_div(d->M, WD_ROWS), false, "cr_wide_ln_ffn_fwd", *d);
}

extern "C" int cr_wide_ln_ffn_bwd(const cr_block_bwd_desc* bd, float* g2, float* g1, int precision, void* stream) {
    CR_REQUIRE(bd, "cr_wide_ln_ffn_bwd: NULL description");
    const cr_block_desc* d = &bd->f;
    const char* why = wide_why(d, precision);
    if (why) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_wide_ln_ffn_bwd: %s", why);
    CR_REQUIRE(bd->dy && bd->d_o && g2 && g1 && d->hid && d->o && d->mask_ids && d->w1 && d->w2 && d->ln2_g, "cr_wide_ln_ffn_bwd: NULL pointer");
    CR_REQUIRE(bd->g_ln2_g && bd->g_ln2_b && bd->n_slabs > 0 && bd->slab_stride > 0, "cr_wide_ln_ffn_bwd: NULL gradient pointer / no slabs");
    hipStream_t s = cr_stream(stream);
    WIDE_DISPATCH(k_wide_ffn_bwd, bd->n_slabs, true, "cr_wide_ln_ffn_bwd", *bd, g2, g1);
}

extern "C" int cr_wide_ln_qkv_bwd(const cr_block_bwd_desc* bd, int precision, void* stream) {
    CR_REQUIRE(bd, "cr_wide_ln_qkv_bwd: NULL description");
    const cr_block_desc* d = &bd->f;
    const char* why = wide_why(d, precision);
    if (why) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_wide_ln_qkv_bwd: %s", why);
    CR_REQUIRE(bd->dqkv && bd->d_o && bd->dx && d->x && d->wqkv && d->ln1_g, "cr_wide_ln_qkv_bwd: NULL pointer");
    CR_REQUIRE(bd->g_ln1_g && bd->g_ln1_b && bd->n_slabs > 0 && bd->slab_stride > 0, "cr_wide_ln_qkv_bwd: NULL gradient pointer / no slabs");
    CR_REQUIRE(bd->dq_part == nullptr, "cr_wide_ln_qkv_bwd: dq_part (single-pass fp32 attention backward) is not taken");
    hipStream_t s = cr_stream(stream);
    WIDE_DISPATCH(k_wide_qkv_bwd, bd->n_slabs, true, "cr_wide_ln_qkv_bwd", *bd);
}
