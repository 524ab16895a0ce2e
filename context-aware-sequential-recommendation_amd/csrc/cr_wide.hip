// Row phases of a transformer block for hidden sizes 128 / 192 / 256 (configs C4 / C5) on the bf16 matrix pipe:
// the same four steps as cr_block.hip (D <= 64), modules.py:53-80,203-205,280-318 and their backward,
//   cr_wide_ln_qkv_fwd : q_in = LN1(x) (+ key / query masks); Q = q_in Wq + bq; K = x Wk + bk; V = x Wv + bv
//   cr_wide_ln_ffn_fwd : f_in = LN2(o); hid = drop(relu(f_in W1 + b1)); y = (drop(hid W2 + b2) + f_in) * mask
//   cr_wide_ln_ffn_bwd : dy -> d_o, slabs of dgamma2 dbeta2 (+ dW2 db2 dW1 db1 at D = 128; else g2, g1 for cr_gemm_wgrad)
//   cr_wide_ln_qkv_bwd : (dQ|dK|dV, d_o) -> dx (= or +=), slabs of dgamma1 dbeta1 (+ dWqkv dbqkv at D = 128)
// each ONE launch where the unfused path runs 3 to 6 (cr_layernorm_*, cr_gemm_rows, cr_eltwise, cr_gemm_wgrad).
// DESIGN.md section 4 has the measurements behind the choices (timing variants, per-wave timeline, HBM counters).
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
// Weight gradients (D = 128, where the panel loop is unrolled and a [D, D] slab per 128 rows costs what a row block costs):
// the workgroup's rows of `a` and `g` become [128][128] bf16 images in LDS -- `g` from the registers that hold it, behind the
// panel buffers; `a` over them once the panels are done -- both MFMA operands are transposed reads with k = row, and wave w
// owns row tile w of dW.  At D = 128 nothing of a chain goes through memory between its parts (static register indices); above,
// a part's result is stored and the next part's operand re-read by the lane that stored it.
#include "cr_attn_common.hpp"
#include "cr_bf16.hpp"

// Threads per workgroup.  Forward kernels: 8 waves (128 rows) at every size.  Backward kernels: 8 waves at D = 128, 4 waves
// (64 rows) above, where a row's operands and partial results need more than the 256 registers a wave gets at two waves per
// SIMD (one wave per SIMD may use all 512).  More rows per workgroup = fewer passes of the weights through LDS: at D = 256
// a workgroup reads 768 KB of weights per 64 KB of activation rows.
// MODE 0: forward, 8 waves; 1: backward; 2: forward, 4 waves (row counts that leave CUs idle at 128 rows per workgroup)
template <int NCT, int MODE> struct WdCfg {
    static constexpr int NT = MODE == 0 ? 512 : (MODE == 2 ? 256 : (NCT <= 8 ? 512 : 256)), ROWS = NT / 4, NW = NT / 64, ITEMS = 128 * NCT / NT;
};
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
template <int NCT, int MODE> struct PanelRegs { float v[WdCfg<NCT, MODE>::ITEMS][8]; };

// forward panel: W[k][c0 .. c0 + 63], k < D; item = (k, chunk of 8 columns)
template <int NCT, int MODE>
__device__ __forceinline__ void fpanel_issue(PanelRegs<NCT, MODE>& r, const float* W, int ld, int c0) {
#pragma unroll
    for (int u = 0; u < WdCfg<NCT, MODE>::ITEMS; ++u) {
        const int item = threadIdx.x + WdCfg<NCT, MODE>::NT * u;
        const int k = item >> 3, ch = item & 7;
        const float* p = W + (size_t)k * ld + c0 + 8 * ch;
        const f4u a = *reinterpret_cast<const f4u*>(p), b = *reinterpret_cast<const f4u*>(p + 4);
        r.v[u][0] = a.x; r.v[u][1] = a.y; r.v[u][2] = a.z; r.v[u][3] = a.w;
        r.v[u][4] = b.x; r.v[u][5] = b.y; r.v[u][6] = b.z; r.v[u][7] = b.w;
    }
}
template <int NCT, bool SPLIT, int MODE>
__device__ __forceinline__ void fpanel_put(const PanelRegs<NCT, MODE>& r, __bf16* img) {
#pragma unroll
    for (int u = 0; u < WdCfg<NCT, MODE>::ITEMS; ++u) {
        const int item = threadIdx.x + WdCfg<NCT, MODE>::NT * u;
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
__device__ __forceinline__ void bpanel_issue(PanelRegs<NCT, 1>& r, const float* W, int ld, int j0, int c0) {
#pragma unroll
    for (int u = 0; u < WdCfg<NCT, 1>::ITEMS; ++u) {
        const int item = threadIdx.x + WdCfg<NCT, 1>::NT * u;
        const int j = item / (2 * NCT), c8 = item % (2 * NCT);
        const float* p = W + (size_t)(j0 + j) * ld + c0 + 8 * c8;
        const f4u a = *reinterpret_cast<const f4u*>(p), b = *reinterpret_cast<const f4u*>(p + 4);
        r.v[u][0] = a.x; r.v[u][1] = a.y; r.v[u][2] = a.z; r.v[u][3] = a.w;
        r.v[u][4] = b.x; r.v[u][5] = b.y; r.v[u][6] = b.z; r.v[u][7] = b.w;
    }
}
template <int NCT, bool SPLIT>
__device__ __forceinline__ void bpanel_put(const PanelRegs<NCT, 1>& r, __bf16* img) {
#pragma unroll
    for (int u = 0; u < WdCfg<NCT, 1>::ITEMS; ++u) {
        const int item = threadIdx.x + WdCfg<NCT, 1>::NT * u;
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
    static constexpr int NV = 5;                                       // column-sum vectors of a backward kernel (dgamma, dbeta, 3 bias gradients)
    static constexpr size_t SLOT_BYTES = (size_t)WdCfg<NCT, 1>::NW * NV * 16 * NCT * 4;  // [waves][NV][D] floats (backward kernels)
    static constexpr size_t TOTAL = PANEL_BYTES + SLOT_BYTES + 5 * 16 * NCT * 4;      // + [5 D] floats of biases (forward kernels)
    // weight-gradient phase of the backward kernels: two [ROWS][D] images (a, g), each hi (+ lo), over the panel buffers
    static constexpr int IMG_HALF = WdCfg<NCT, 1>::ROWS * 16 * NCT;       // bf16 elements of one half of one image
    static constexpr int IMG = (SPLIT ? 2 : 1) * IMG_HALF;
    static constexpr size_t IMG_BYTES = 2 * (size_t)IMG * 2;
    static constexpr size_t TOTAL_BWD = (IMG_BYTES > PANEL_BYTES ? IMG_BYTES : PANEL_BYTES) + SLOT_BYTES;
    static constexpr size_t TOTAL_BWD_NOWG = PANEL_BYTES + SLOT_BYTES;
};

// Per-wave phase stamps (compiled only with -DWD_TS=1; tools/wide_ts.py): slot k of wave w of workgroup b, core clocks
#ifdef WD_TS
__device__ unsigned long long g_wd_ts[512 * 8 * 64];
#define WTS(slot)                                                                                                   \
    do {                                                                                                            \
        if (ts_on && (threadIdx.x & 63) == 0 && blockIdx.x < 512 && (slot) < 64)                                    \
            g_wd_ts[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + (slot)] = clock64();                       \
    } while (0)
extern "C" int cr_wide_ts_read(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wd_ts), sizeof(g_wd_ts)) == hipSuccess ? 0 : -3;
}
#define WTS_KERNEL(id) const bool ts_on = (WD_TS == (id) + 1)
#else
#define WTS(slot) do { } while (0)
#define WTS_KERNEL(id) const bool ts_on = false
#endif

// ---- the panel pipeline -------------------------------------------------------------------------------------
// One iteration per panel, every global LOAD of the steady state at the top of an iteration and for a LATER iteration: the
// weights of panel i + 2 (D = 128: two register sets; above that i + 1, one set) and the auxiliary rows (residual, gate,
// running sum) of panel i + 1.  What an iteration consumes was requested one iteration earlier, BEFORE that iteration's result
// stores: vector-memory operations retire in issue order, so a load issued after a store cannot be waited for without
// waiting for the store too (the first version loaded bias and residual rows right before the epilogue that used them: every
// iteration then waited for the previous iteration's stores; 2.9 us per panel for 0.65 us of MFMA time).  Biases sit in LDS.
// Nothing depends on the loop being unrolled except the choice between the two register sets (D = 128 only, where the loop IS
// unrolled): the operand of a part is ONE register array, reloaded -- from rows this lane stored itself -- and split again at
// a part boundary behind a workgroup-uniform branch, and every panel's result goes to memory.
//   issue(regs, n)      request the weights of panel n
//   auxload(aux, n)     request panel n's auxiliary rows (may do nothing)
//   boundary(i) -> bool part-boundary work before iteration i (operand reload, LayerNorm backward ...); true = the
//                       auxiliary rows prefetched for i are stale (written by this boundary): they are loaded again
//   epilogue(i, acc, aux)
template <int NCT, bool SPLIT, int MODE, int NPAN, class Issue, class AuxLoad, class Boundary, class Epilogue>
__device__ __forceinline__ void panel_pipeline(__bf16* pb, PanelRegs<NCT, MODE> (&pr)[NCT <= 8 ? 2 : 1], const bf8 (&oh)[NCT / 2],
                                               const bf8 (&ol)[NCT / 2], Issue issue, AuxLoad auxload, Boundary boundary,
                                               Epilogue epilogue, const bool ts_on) {
    (void)ts_on;
    typedef WideLds<NCT, SPLIT> LD;
    constexpr bool D2 = NCT <= 8, FWD = MODE != 1;
    f32x4 aux_n[4];
    auxload(aux_n, 0);
    if constexpr (FWD) fpanel_put<NCT, SPLIT, MODE>(pr[0], pb); else bpanel_put<NCT, SPLIT>(pr[0], pb);
    __syncthreads();
    _Pragma("clang loop unroll_count(NCT <= 8 ? NPAN : 1)")
    for (int i = 0; i < NPAN; ++i) {
        WTS(8 + 4 * i);
        const bool stale = boundary(i);
        f32x4 aux[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) aux[ct] = aux_n[ct];
        if (stale) auxload(aux, i);
        if (D2) { if (i + 2 < NPAN) issue(pr[D2 ? (i & 1) : 0], i + 2); }
        else if (i + 1 < NPAN) issue(pr[0], i + 1);
        if (i + 1 < NPAN) auxload(aux_n, i + 1);
        f32x4 acc[4];
        acc_zero(acc);
        panel_mma<NCT, SPLIT, FWD>(acc, pb + (i & 1) * LD::BUF, oh, ol);
#ifdef WD_TS
        asm volatile("s_nop 0" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));     // the products are done
#endif
        WTS(9 + 4 * i);
        epilogue(i, acc, aux);
        WTS(10 + 4 * i);
        if (i + 1 < NPAN) {
            __bf16* dst = pb + ((i + 1) & 1) * LD::BUF;
            if constexpr (FWD) fpanel_put<NCT, SPLIT, MODE>(pr[D2 ? ((i + 1) & 1) : 0], dst); else bpanel_put<NCT, SPLIT>(pr[D2 ? ((i + 1) & 1) : 0], dst);
        }
        WTS(11 + 4 * i);
        __syncthreads();
    }
    WTS(8 + 4 * NPAN);
}
// n floats global -> LDS (biases), visible after the next barrier
__device__ __forceinline__ void vec_to_lds(float* dst, const float* src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
__device__ __forceinline__ void lds_vec4(f32x4 (&v)[4], const float* vec, int p) {
    const int lg = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const float4 t = *reinterpret_cast<const float4*>(vec + 64 * p + 16 * ct + 4 * lg);
        v[ct] = (f32x4){t.x, t.y, t.z, t.w};
    }
}

// =====================================================================================================
// forward: LN1 + Q / K / V projections
// =====================================================================================================
template <int NCT, bool SPLIT, int MODE>
__global__ __launch_bounds__((WdCfg<NCT, MODE>::NT)) void k_wide_qkv_fwd(cr_block_desc d) {
    constexpr int D = 16 * NCT, NKS = NCT / 2, NP = NCT / 4, NPAN = 3 * NP;
    constexpr bool D2 = NCT <= 8;
    typedef WideLds<NCT, SPLIT> LD;
    WTS_KERNEL(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* pb = reinterpret_cast<__bf16*>(smem_raw);
    float* lbias = reinterpret_cast<float*>(smem_raw + LD::PANEL_BYTES + LD::SLOT_BYTES);       // [3 D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15;
    const int m = blockIdx.x * WdCfg<NCT, MODE>::ROWS + 16 * wave + li;
    const bool rok = m < d.M;
    WTS(0);
    auto issue = [&](PanelRegs<NCT, MODE>& r, int n) { fpanel_issue<NCT, MODE>(r, d.wqkv, 3 * D, (n / NP) * D + 64 * (n % NP)); };
    PanelRegs<NCT, MODE> pr[D2 ? 2 : 1];
    issue(pr[0], 0);
    if constexpr (D2) issue(pr[1], 1);
    vec_to_lds(lbias, d.bqkv, 3 * D);
    bf8 oh[NKS], ol[NKS];                                 // the current operand: q_in for Q, x for K and V
    bf8 xh[NKS], xl[NKS];                                 // D = 128: x's operand form, kept for the K and V parts (else x is re-read)
    {
        f32x4 x[NCT];
        wr_load<NCT>(x, d.x, m, rok);
        float mean, rs, sum;
        wr_stats<NCT>(x, mean, rs, sum);
        if constexpr (D2) wr_split<NCT, SPLIT>(x, xh, xl);
        WTS(1);
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
                    x[4 * p + ct][r] = y;
                    ys += y;
                }
        }
        ys = grp_sum(ys);
        wr_store<NCT>(d.q_in, m, rok, x);
        if (rok && (lane >> 4) == 0) {                   // modules.py:222 (keys = x), 248-249 (queries = LN1(x))
            d.k_valid[m] = (sum != 0.0f) ? 1.0f : 0.0f;
            d.q_valid[m] = (ys != 0.0f) ? 1.0f : 0.0f;
        }
        wr_split<NCT, SPLIT>(x, oh, ol);
    }
    WTS(2);
    panel_pipeline<NCT, SPLIT, MODE, NPAN>(pb, pr, oh, ol, issue,
        [&](f32x4 (&)[4], int) {},
        [&](int i) {
            if (i == NP) {                                // K and V take the un-normalised rows (modules.py:204-205)
                if constexpr (D2) {
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) { oh[ks] = xh[ks]; ol[ks] = xl[ks]; }
                } else {
                    f32x4 x[NCT];
                    wr_load<NCT>(x, d.x, m, rok);
                    wr_split<NCT, SPLIT>(x, oh, ol);
                }
            }
            return false;
        },
        [&](int i, f32x4 (&acc)[4], const f32x4 (&)[4]) {
            const int part = i / NP, p = i % NP;
            f32x4 bias[4];
            lds_vec4(bias, lbias + part * D, p);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] += bias[ct];
            wr_store4(d.qkv + (size_t)part * d.M * D, D, m, rok, p, acc);
        }, ts_on);
}

// =====================================================================================================
// forward: LN2 + feed-forward (+ a tail on the output rows while they are in registers, D = 128:
//          TAIL 1 = the NEXT block's LN1 + Q / K / V projections, TAIL 2 = the stack's final LayerNorm)
// =====================================================================================================
struct WideTail {
    cr_block_desc next;                                   // TAIL 1
    const float* lnf_g; const float* lnf_b; float* out; int ld_out, col_out;   // TAIL 2
};
template <int NCT, bool SPLIT, int MODE, int TAIL>
__global__ __launch_bounds__((WdCfg<NCT, MODE>::NT)) void k_wide_ffn_fwd(cr_block_desc d, WideTail tl) {
    constexpr int D = 16 * NCT, NKS = NCT / 2, NP = NCT / 4, NPAN = (TAIL == 1 ? 5 : 2) * NP;
    constexpr bool D2 = NCT <= 8;
    static_assert(TAIL == 0 || D2, "the tails keep the output rows in registers (unrolled panel loop)");
    typedef WideLds<NCT, SPLIT> LD;
    WTS_KERNEL(1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* pb = reinterpret_cast<__bf16*>(smem_raw);
    float* lbias = reinterpret_cast<float*>(smem_raw + LD::PANEL_BYTES + LD::SLOT_BYTES);       // [5 D]: b1, b2, the next block's bqkv
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int m = blockIdx.x * WdCfg<NCT, MODE>::ROWS + 16 * wave + li;
    const bool rok = m < d.M;
    const DropCtx dc1 = drop_ctx(d.drop_ffn1), dc2 = drop_ctx(d.drop_ffn2);
    auto issue = [&](PanelRegs<NCT, MODE>& r, int n) {
        if (TAIL == 1 && n >= 2 * NP) fpanel_issue<NCT, MODE>(r, tl.next.wqkv, 3 * D, ((n - 2 * NP) / NP) * D + 64 * (n % NP));
        else fpanel_issue<NCT, MODE>(r, n / NP ? d.w2 : d.w1, D, 64 * (n % NP));
    };
    PanelRegs<NCT, MODE> pr[D2 ? 2 : 1];
    issue(pr[0], 0);
    if constexpr (D2) issue(pr[1], 1);
    vec_to_lds(lbias, d.b1, D);
    vec_to_lds(lbias + D, d.b2, D);
    if (TAIL == 1) vec_to_lds(lbias + 2 * D, tl.next.bqkv, 3 * D);
    bf8 oh[NKS], ol[NKS];                                 // f_in, then hid (then the next block's q_in, x)
    bf8 nh[NKS], nl[NKS];                                 // D = 128: hid as the next operand, built panel by panel (else re-read)
    f32x4 fin[NCT];                                       // D = 128: the residual rows (else re-read per panel); then the output rows y
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
        wr_split<NCT, SPLIT>(x, oh, ol);
        if constexpr (D2) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) fin[ct] = x[ct];
        }
    }
    const int id = rok ? d.mask_ids[m] : 0;
    const uint32_t xrow = ((d.drop_ffn1.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)(4 * lg)) * CR_PHI;
    panel_pipeline<NCT, SPLIT, MODE, NPAN>(pb, pr, oh, ol, issue,
        [&](f32x4 (&aux)[4], int n) {
            if (!D2 && n >= NP) wr_load4(aux, d.f_in, D, m, rok, n - NP);     // own stores (residual = LN2 output, modules.py:313)
        },
        [&](int i) {
            if (i == NP) {                                // the second layer's operand: the hidden rows
                if constexpr (D2) {
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) { oh[ks] = nh[ks]; ol[ks] = nl[ks]; }
                } else {
                    f32x4 h[NCT];
                    wr_load<NCT>(h, d.hid, m, rok);       // own stores
                    wr_split<NCT, SPLIT>(h, oh, ol);
                }
            }
            if constexpr (TAIL == 1) {
                if (i == 2 * NP) {
                    // the next block's first phase on y (in `fin` now): x' = y; q_in' = LN1'(y); key / query masks; operand q_in'
                    wr_split<NCT, SPLIT>(fin, nh, nl);    // y's operand form, for the K and V parts
                    float mean, rs, sum;
                    wr_stats<NCT>(fin, mean, rs, sum);
                    float ys = 0.0f;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        f32x4 g[4], b[4];
                        wr_vec4(g, tl.next.ln1_g, p);
                        wr_vec4(b, tl.next.ln1_b, p);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float y = fmaf(g[ct][r], (fin[4 * p + ct][r] - mean) * rs, b[ct][r]);
                                fin[4 * p + ct][r] = y;
                                ys += y;
                            }
                    }
                    ys = grp_sum(ys);
                    wr_store<NCT>(tl.next.q_in, m, rok, fin);
                    if (rok && lg == 0) {                 // modules.py:222 (keys = x), 248-249 (queries = LN1(x))
                        tl.next.k_valid[m] = (sum != 0.0f) ? 1.0f : 0.0f;
                        tl.next.q_valid[m] = (ys != 0.0f) ? 1.0f : 0.0f;
                    }
                    wr_split<NCT, SPLIT>(fin, oh, ol);
                }
                if (i == 3 * NP) {
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) { oh[ks] = nh[ks]; ol[ks] = nl[ks]; }
                }
            }
            return false;
        },
        [&](int i, f32x4 (&acc)[4], const f32x4 (&res)[4]) {
            const int part = i / NP, p = i % NP;
            f32x4 bias[4];
            lds_vec4(bias, lbias + part * D, p);
            if (TAIL == 1 && part >= 2) {                 // the next block's Q / K / V
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct] += bias[ct];
                wr_store4(tl.next.qkv + (size_t)(part - 2) * d.M * D, D, m, rok, p, acc);
                return;
            }
            DropCtx dc;                                   // (field by field: a selected struct reference went through scratch)
            dc.on = dc1.on;
            dc.key = part ? dc2.key : dc1.key;
            dc.thresh = part ? dc2.thresh : dc1.thresh;
            dc.scale = part ? dc2.scale : dc1.scale;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[ct][r] + bias[ct][r];
                    if (!part) v = fmaxf(v, 0.0f);                                           // modules.py:300
                    if (dc.on) v *= drop_factor_x(dc, xrow + (uint32_t)(64 * p + 16 * ct + r) * CR_PHI + dc.key);
                    if (part) v = id ? v + (D2 ? fin[4 * p + ct][r] : res[ct][r]) : 0.0f;    // modules.py:313, sasrec.py:83
                    acc[ct][r] = v;
                }
            if constexpr (D2) {
                if (part == 0) {
                    wr_split2<SPLIT>(acc[0], acc[1], nh[2 * p], nl[2 * p]);
                    wr_split2<SPLIT>(acc[2], acc[3], nh[2 * p + 1], nl[2 * p + 1]);
                } else if (TAIL != 0) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) fin[4 * p + ct] = acc[ct];                // y takes the residual's registers, panel by panel
                }
            }
            wr_store4(part ? d.y : d.hid, D, m, rok, p, acc);
        }, ts_on);
    if constexpr (TAIL == 2) {
        // the stack's final LayerNorm (sasrec.py:85) on y, into a column block of `out`
        float mean, rs, sum;
        wr_stats<NCT>(fin, mean, rs, sum);
        float* q = tl.out + (size_t)(rok ? m : 0) * tl.ld_out + tl.col_out + 4 * lg;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            f32x4 g[4], b[4];
            wr_vec4(g, tl.lnf_g, p);
            wr_vec4(b, tl.lnf_b, p);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                f4u v;
                v.x = fmaf(g[ct][0], (fin[4 * p + ct][0] - mean) * rs, b[ct][0]);
                v.y = fmaf(g[ct][1], (fin[4 * p + ct][1] - mean) * rs, b[ct][1]);
                v.z = fmaf(g[ct][2], (fin[4 * p + ct][2] - mean) * rs, b[ct][2]);
                v.w = fmaf(g[ct][3], (fin[4 * p + ct][3] - mean) * rs, b[ct][3]);
                if (rok) *reinterpret_cast<f4u*>(q + 64 * p + 16 * ct) = v;
            }
        }
    }
}

// ---- column sums of the workgroup's rows into per-thread accumulators ---------------------------------------
// colsum_put: a[ct][r] = the lane's contributions (row li) at columns 16 ct + 4 lg + r; the wave's sums go to vector `which`
// of its LDS slot.  colsum_fold: after the puts, thread c < D adds this pass's sums of column c to t[0 .. NV-1], waves in a
// fixed order.  Two barriers.
template <int NCT>
__device__ __forceinline__ void colsum_put(float* slots, const f32x4 (&a)[NCT], int which) {
    constexpr int D = 16 * NCT, NV = WideLds<NCT, true>::NV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    float* sg = slots + ((size_t)wave * NV + which) * D;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        f32x4 ga;
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[r] = cr_row16_sum(a[ct][r]);
        if (li == 0) *reinterpret_cast<float4*>(sg + 16 * ct + 4 * lg) = make_float4(ga[0], ga[1], ga[2], ga[3]);
    }
}
template <int NCT, int NUSE>
__device__ __forceinline__ void colsum_fold(const float* slots, float (&t)[NUSE]) {
    constexpr int D = 16 * NCT, NW = WdCfg<NCT, 1>::NW, NV = WideLds<NCT, true>::NV;
    __syncthreads();
    if ((int)threadIdx.x < D) {
#pragma unroll
        for (int v = 0; v < NUSE; ++v) {
            float s0 = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s0 += slots[((size_t)w * NV + v) * D + threadIdx.x];
            t[v] += s0;
        }
    }
    __syncthreads();
}

// ---- weight gradients dW = a^T g of the workgroup's rows (contraction over rows) ----------------------------------
// The rows of a and g go to LDS as [ROWS][D] bf16 images (pieces of 64 columns, the swizzle of cr_bf16.hpp); both MFMA
// operands are transposed reads (k = row).  Wave w owns the 16-row tiles w, w + NW, ... of dW (k = columns of a) over all D
// columns: accumulators acc[NCT].  The result is written -- or, from the workgroup's second row block on, added -- to the
// workgroup's slab: the same lanes touch the same words, no synchronisation.
template <int NCT, bool SPLIT>
__device__ __forceinline__ void wg_split_to_image(__bf16* img, const bf8 (&h)[NCT / 2], const bf8 (&l)[NCT / 2]) {
    typedef WideLds<NCT, SPLIT> LD;
    constexpr int SUB = WdCfg<NCT, 1>::ROWS * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int row = 16 * wave + li;
#pragma unroll
    for (int c2 = 0; c2 < NCT / 2; ++c2) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ct = 2 * c2 + e;
            const int o = (ct >> 2) * SUB + img_off<2>(row, 2 * (ct & 3) + (lg >> 1)) + 4 * (lg & 1);
            *reinterpret_cast<bf4*>(img + o) = e ? __builtin_shufflevector(h[c2], h[c2], 4, 5, 6, 7) : __builtin_shufflevector(h[c2], h[c2], 0, 1, 2, 3);
            if (SPLIT) *reinterpret_cast<bf4*>(img + LD::IMG_HALF + o) = e ? __builtin_shufflevector(l[c2], l[c2], 4, 5, 6, 7) : __builtin_shufflevector(l[c2], l[c2], 0, 1, 2, 3);
        }
    }
}
template <int NCT, bool SPLIT>
__device__ __forceinline__ void wg_row_to_image(__bf16* img, const f32x4 (&x)[NCT]) {
    bf8 h[NCT / 2], l[NCT / 2];
    wr_split<NCT, SPLIT>(x, h, l);
    wg_split_to_image<NCT, SPLIT>(img, h, l);
}
// the column sums of one panel's four tiles (columns 64 p ..) into vector `which` of the wave's slot
template <int NCT>
__device__ __forceinline__ void colsum_put4(float* slots, const f32x4 (&a)[4], int which, int p) {
    constexpr int D = 16 * NCT, NV = WideLds<NCT, true>::NV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    float* sg = slots + ((size_t)wave * NV + which) * D + 64 * p;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        f32x4 ga;
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[r] = cr_row16_sum(a[ct][r]);
        if (li == 0) *reinterpret_cast<float4*>(sg + 16 * ct + 4 * lg) = make_float4(ga[0], ga[1], ga[2], ga[3]);
    }
}
template <int NCT, bool SPLIT>
__device__ __forceinline__ void wg_product(const __bf16* ia, const __bf16* ig, float* dW, int ldw, bool first) {
    typedef WideLds<NCT, SPLIT> LD;
    constexpr int ROWS = WdCfg<NCT, 1>::ROWS, NW = WdCfg<NCT, 1>::NW, SUB = ROWS * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
#pragma unroll 1
    for (int kt = wave; kt < NCT; kt += NW) {
        f32x4 acc[NCT];
#pragma unroll
        for (int nt = 0; nt < NCT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const __bf16* pa = ia + (kt >> 2) * SUB;
#pragma unroll
        for (int s = 0; s < ROWS / 32; ++s) {
            const bf8 ah = wd_tr(pa, 32 * s, 32 * s + 16, kt & 3, lane);
            const bf8 al = SPLIT ? wd_tr(pa + LD::IMG_HALF, 32 * s, 32 * s + 16, kt & 3, lane) : ah;
#pragma unroll
            for (int n0 = 0; n0 < NCT; n0 += 2) {
                bf8 gh[2], gl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    gh[j] = wd_tr(ig + (n0 >> 2) * SUB, 32 * s, 32 * s + 16, (n0 & 3) + j, lane);
                    gl[j] = SPLIT ? wd_tr(ig + LD::IMG_HALF + (n0 >> 2) * SUB, 32 * s, 32 * s + 16, (n0 & 3) + j, lane) : gh[j];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[n0 + j] = mma<SPLIT>(ah, al, gh[j], gl[j], acc[n0 + j]);
            }
        }
        float* q0 = dW + (size_t)(16 * kt + 4 * lg) * ldw + li;
        if (first) {                                      // workgroup-uniform
#pragma unroll
            for (int nt = 0; nt < NCT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(acc[nt][r], q0 + (size_t)r * ldw + 16 * nt);   // (streaming: Adam reads the slab many
                                                                                                                  //  launches later; -1.25 % per C4 step)
        } else {
#pragma unroll 2
            for (int nt = 0; nt < NCT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) q0[(size_t)r * ldw + 16 * nt] += acc[nt][r];
        }
    }
}

// LayerNorm backward of the lane's row (modules.py:74-78): g = gradient of the output row (the dbeta contributions), x = the
// LayerNorm's input row; returns dx in g and the dgamma contributions (g * xhat) in x.  Two passes over two arrays (a third
// array for xhat cost the registers the D = 256 kernels do not have).
template <int NCT>
__device__ __forceinline__ void wr_ln_bwd(f32x4 (&g)[NCT], f32x4 (&x)[NCT], const float* gamma) {
    constexpr int NP = NCT / 4;
    constexpr float invD = 1.0f / (16 * NCT);
    float mean, rs, sum;
    wr_stats<NCT>(x, mean, rs, sum);
    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        f32x4 gm[4];
        wr_vec4(gm, gamma, p);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = (x[4 * p + ct][r] - mean) * rs;
                const float dg = g[4 * p + ct][r] * gm[ct][r];
                c1 += dg;
                c2 = fmaf(dg, h, c2);
            }
    }
    c1 = grp_sum(c1) * invD;
    c2 = grp_sum(c2) * invD;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        f32x4 gm[4];
        wr_vec4(gm, gamma, p);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = (x[4 * p + ct][r] - mean) * rs;
                const float go = g[4 * p + ct][r];
                g[4 * p + ct][r] = cr_ln_bwd_tail(go * gm[ct][r], c1, h, c2, rs);             // three scalar instructions (cr_common.hpp)
                x[4 * p + ct][r] = go * h;
            }
    }
}

// =====================================================================================================
// backward: feed-forward + LN2
// =====================================================================================================
template <int NCT, bool SPLIT, bool WG>
__global__ __launch_bounds__((WdCfg<NCT, 1>::NT)) void k_wide_ffn_bwd(cr_block_bwd_desc bd, float* g2out, float* g1out, int heads) {
    constexpr int D = 16 * NCT, NKS = NCT / 2, NP = NCT / 4, NPAN = 2 * NP;
    constexpr bool D2 = NCT <= 8;
    typedef WideLds<NCT, SPLIT> LD;
    const cr_block_desc& d = bd.f;
    WTS_KERNEL(2);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* pb = reinterpret_cast<__bf16*>(smem_raw);
    float* slots = reinterpret_cast<float*>(smem_raw + (WG ? LD::TOTAL_BWD : LD::TOTAL_BWD_NOWG) - LD::SLOT_BYTES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const DropCtx dc2 = drop_ctx(d.drop_ffn2);
    const float gate_scale = d.drop_ffn1.rate > 0.0f ? 1.0f / (1.0f - d.drop_ffn1.rate) : 1.0f;
    const int nblk = (d.M + WdCfg<NCT, 1>::ROWS - 1) / WdCfg<NCT, 1>::ROWS;
    float tot[4] = {0.f, 0.f, 0.f, 0.f};                  // dgamma2, dbeta2, db2, db1 of column threadIdx.x
    constexpr bool wgrad = WG;                            // (a template flag: the phase's registers cost the other shapes a spill)
    auto issue = [&](PanelRegs<NCT, 1>& r, int n) { bpanel_issue<NCT>(r, n / NP ? d.w1 : d.w2, D, 64 * (n % NP), 0); };
#pragma unroll 1
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int m = blk * WdCfg<NCT, 1>::ROWS + 16 * wave + li;
        const bool rok = m < d.M;
        PanelRegs<NCT, 1> pr[D2 ? 2 : 1];
        issue(pr[0], 0);
        if constexpr (D2) issue(pr[1], 1);
        const int id = rok ? d.mask_ids[m] : 0;
        const uint32_t xrow = ((d.drop_ffn2.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)(4 * lg)) * CR_PHI;
        bf8 oh[NKS], ol[NKS];                             // g2, then g1
        // WG (D = 128: the panel loop is unrolled, every index below is static): g1 and df_in stay in registers and the g2 /
        // g1 rows reach the weight-gradient products as LDS images written from registers -- none of them goes through
        // memory (the first version stored g2, g1 and df_in and read them back: 218 MB per launch for 117 algorithmic)
        bf8 nh[NKS], nl[NKS];                             // g1 as the next operand (WG)
        f32x4 df[NCT];                                    // df_in (WG)
        __bf16* ia = pb;                                  // the `a` image lies over the panel buffers, the `g` image behind them
        __bf16* ig = pb + LD::IMG;
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
            wr_split<NCT, SPLIT>(g, oh, ol);
            if constexpr (WG) {
                wg_split_to_image<NCT, SPLIT>(ig, oh, ol);
                colsum_put<NCT>(slots, g, 2);
            } else {
                wr_store<NCT>(g2out, m, rok, g);
            }
        }
        panel_pipeline<NCT, SPLIT, 1, NPAN>(pb, pr, oh, ol, issue,
            [&](f32x4 (&aux)[4], int n) { wr_load4(aux, n / NP ? bd.dy : d.hid, D, m, rok, n % NP); },
            [&](int i) {
                if (i == NP) {
                    if constexpr (WG) {
#pragma unroll
                        for (int ks = 0; ks < NKS; ++ks) { oh[ks] = nh[ks]; ol[ks] = nl[ks]; }
                    } else {
                        f32x4 g[NCT];
                        wr_load<NCT>(g, g1out, m, rok);   // own stores
                        wr_split<NCT, SPLIT>(g, oh, ol);
                    }
                }
                return false;
            },
            [&](int i, f32x4 (&acc)[4], const f32x4 (&aux)[4]) {
                const int part = i / NP, p = i % NP;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // part 0: g1 = (g2 W2^T) gated by the stored post-dropout ReLU output (modules.py:300-303)
                        // part 1: df_in = (g1 W1^T + dy) * mask (residual branch of modules.py:313)
                        const float v0 = aux[ct][r] > 0.0f ? acc[ct][r] * gate_scale : 0.0f;
                        const float v1 = id ? acc[ct][r] + aux[ct][r] : 0.0f;
                        acc[ct][r] = part ? v1 : v0;
                    }
                if constexpr (WG) {
                    if (part == 0) {
                        wr_split2<SPLIT>(acc[0], acc[1], nh[2 * p], nl[2 * p]);
                        wr_split2<SPLIT>(acc[2], acc[3], nh[2 * p + 1], nl[2 * p + 1]);
                        colsum_put4<NCT>(slots, acc, 3, p);
                    } else {
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) df[4 * p + ct] = acc[ct];
                    }
                } else {
                    wr_store4(part ? bd.d_o : g1out, D, m, rok, p, acc);      // (df_in parked in d_o)
                }
            }, ts_on);
        {
            f32x4 xh[NCT];
            if constexpr (!WG) wr_load<NCT>(df, bd.d_o, m, rok);           // own stores
            wr_load<NCT>(xh, d.o, m, rok);
            colsum_put<NCT>(slots, df, 1);
            wr_ln_bwd<NCT>(df, xh, d.ln2_g);
            wr_store<NCT>(bd.d_o, m, rok, df);
            colsum_put<NCT>(slots, xh, 0);
            if (heads > 0) {
                // the softmax-backward row term of every head, delta[h][m] = sum over the head's columns of d_o * (o - q_in)
                // (o - q_in is the head's A V, modules.py:262-269): with it cr_attn_bwd runs its two passes in ONE launch
                const int tph = NCT / heads;              // column tiles per head (the host checks the head dim is 16 k)
                float a = 0.0f;
                int hcur = 0;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int lg4 = 4 * ((threadIdx.x & 63) >> 4);
                    const size_t o = (size_t)(rok ? m : 0) * D + 16 * ct + lg4;
                    const f4u ov = *reinterpret_cast<const f4u*>(d.o + o), qv = *reinterpret_cast<const f4u*>(d.q_in + o);
                    a += df[ct][0] * (ov.x - qv.x) + df[ct][1] * (ov.y - qv.y) + df[ct][2] * (ov.z - qv.z) + df[ct][3] * (ov.w - qv.w);
                    if ((ct + 1) % tph == 0) {            // wave-uniform
                        const float sm = grp_sum(a);
                        if (rok && lg4 == 0) bd.attn_delta[(size_t)hcur * d.M + m] = sm;
                        a = 0.0f;
                        ++hcur;
                    }
                }
            }
        }
        if constexpr (WG) {
            // dW2 = hid^T g2, dW1 = f_in^T g1 (the bias gradients, column sums of g2 and g1, are in the slots already);
            // every wave is past the pipeline's last barrier: the panel buffers are free for the `a` image
            const bool first = blk == (int)blockIdx.x;
            const size_t so = (size_t)blockIdx.x * bd.slab_stride;
            {
                f32x4 t[NCT];
                wr_load<NCT>(t, d.hid, m, rok);
                wg_row_to_image<NCT, SPLIT>(ia, t);
            }
            __syncthreads();
            wg_product<NCT, SPLIT>(ia, ig, bd.g_w2 + so, D, first);
            __syncthreads();
            wg_split_to_image<NCT, SPLIT>(ig, oh, ol);                     // g1: the second part's operand, still in registers
            {
                f32x4 t[NCT];
                wr_load<NCT>(t, d.f_in, m, rok);
                wg_row_to_image<NCT, SPLIT>(ia, t);
            }
            __syncthreads();
            wg_product<NCT, SPLIT>(ia, ig, bd.g_w1 + so, D, first);
            __syncthreads();
        }
        if (wgrad) colsum_fold<NCT, 4>(slots, tot);
        else {
            float t2[2] = {0.f, 0.f};
            colsum_fold<NCT, 2>(slots, t2);
            tot[0] += t2[0]; tot[1] += t2[1];
        }
    }
    if ((int)threadIdx.x < D) {
        const size_t o = (size_t)blockIdx.x * bd.slab_stride + threadIdx.x;
        bd.g_ln2_g[o] = tot[0];
        bd.g_ln2_b[o] = tot[1];
        if (wgrad) { bd.g_b2[o] = tot[2]; bd.g_b1[o] = tot[3]; }
    }
}

// =====================================================================================================
// backward: Q / K / V projections + LN1
// =====================================================================================================
template <int NCT, bool SPLIT, bool WG>
__global__ __launch_bounds__((WdCfg<NCT, 1>::NT)) void k_wide_qkv_bwd(cr_block_bwd_desc bd) {
    constexpr int D = 16 * NCT, NKS = NCT / 2, NP = NCT / 4, NPAN = 3 * NP;
    constexpr bool D2 = NCT <= 8;
    typedef WideLds<NCT, SPLIT> LD;
    const cr_block_desc& d = bd.f;
    WTS_KERNEL(3);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* pb = reinterpret_cast<__bf16*>(smem_raw);
    float* slots = reinterpret_cast<float*>(smem_raw + (WG ? LD::TOTAL_BWD : LD::TOTAL_BWD_NOWG) - LD::SLOT_BYTES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15;
    const size_t MD = (size_t)d.M * D;
    const int nblk = (d.M + WdCfg<NCT, 1>::ROWS - 1) / WdCfg<NCT, 1>::ROWS;
    float tot[5] = {0.f, 0.f, 0.f, 0.f, 0.f};             // dgamma1, dbeta1, dbq, dbk, dbv of column threadIdx.x
    constexpr bool wgrad = WG;
    auto issue = [&](PanelRegs<NCT, 1>& r, int n) { bpanel_issue<NCT>(r, d.wqkv, 3 * D, 64 * (n % NP), (n / NP) * D); };
#pragma unroll 1
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int m = blk * WdCfg<NCT, 1>::ROWS + 16 * wave + li;
        const bool rok = m < d.M;
        PanelRegs<NCT, 1> pr[D2 ? 2 : 1];
        issue(pr[0], 0);
        if constexpr (D2) issue(pr[1], 1);
        bf8 oh[NKS], ol[NKS];                             // dQ, dK, dV rows in turn
        // WG (D = 128, unrolled panel loop): dq_in, then dx, accumulate in registers -- no parking in d_o, no read-modify-write
        // of dx between the parts (that was 78 of the 247 MB a launch moved)
        f32x4 dq[NCT];
        __bf16* ia = pb;
        __bf16* ig = pb + LD::IMG;
        {
            f32x4 g[NCT];
            wr_load<NCT>(g, bd.dqkv, m, rok);
            wr_split<NCT, SPLIT>(g, oh, ol);
            if constexpr (WG) {                           // dQ's image for dWq, from the registers that hold it now
                wg_split_to_image<NCT, SPLIT>(ig, oh, ol);
                colsum_put<NCT>(slots, g, 2);
            }
        }
        // parts: Wq rows (dq_in = dQ Wq^T + d_o), then LN1 backward -> dx; Wk rows (dx += dK Wk^T); Wv rows (dx += dV Wv^T)
        panel_pipeline<NCT, SPLIT, 1, NPAN>(pb, pr, oh, ol, issue,
            [&](f32x4 (&aux)[4], int n) {
                if (!WG || n < NP) wr_load4(aux, n / NP ? bd.dx : bd.d_o, D, m, rok, n % NP);   // residual (modules.py:269) / the sum so far
            },
            [&](int i) {
                if (i == NP) {
                    // LN1 backward of dq_in with respect to x starts dx
                    f32x4 xh[NCT];
                    if constexpr (!WG) wr_load<NCT>(dq, bd.d_o, m, rok);     // own stores
                    wr_load<NCT>(xh, d.x, m, rok);
                    colsum_put<NCT>(slots, dq, 1);
                    wr_ln_bwd<NCT>(dq, xh, d.ln1_g);
                    colsum_put<NCT>(slots, xh, 0);
                    if (bd.dx_accumulate) {
                        wr_load<NCT>(xh, bd.dx, m, rok);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) dq[ct] += xh[ct];
                    }
                    if constexpr (!WG) wr_store<NCT>(bd.dx, m, rok, dq);     // (this lane re-reads its own stores in the panels below)
                }
                if (i == NP || i == 2 * NP) {
                    f32x4 g[NCT];
                    wr_load<NCT>(g, bd.dqkv + (size_t)(i / NP) * MD, m, rok);      // dK / dV rows
                    wr_split<NCT, SPLIT>(g, oh, ol);
                }
                return !WG && i == NP;                    // dx was just written: the prefetched rows are stale
            },
            [&](int i, f32x4 (&acc)[4], const f32x4 (&old)[4]) {
                const int part = i / NP, p = i % NP;
                if constexpr (WG) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) dq[4 * p + ct] = part ? dq[4 * p + ct] + acc[ct] : acc[ct] + old[ct];
                } else {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[ct] += old[ct];
                    wr_store4(part ? bd.dx : bd.d_o, D, m, rok, p, acc);   // (dq_in parked in d_o)
                }
            }, ts_on);
        if constexpr (WG) {
            wr_store<NCT>(bd.dx, m, rok, dq);
            // dWq = q_in^T dQ, dWk = x^T dK, dWv = x^T dV: column blocks of the [D, 3 D] gradient (+ bias gradients)
            const bool first = blk == (int)blockIdx.x;
            const size_t so = (size_t)blockIdx.x * bd.slab_stride;
#pragma unroll 1
            for (int w = 0; w < 3; ++w) {
                f32x4 t[NCT];
                if (w < 2) {                              // the x image serves dWk and dWv
                    wr_load<NCT>(t, w ? d.x : d.q_in, m, rok);
                    wg_row_to_image<NCT, SPLIT>(ia, t);
                }
                if (w > 0) {                              // (dQ's image was written in the prologue)
                    wr_load<NCT>(t, bd.dqkv + (size_t)w * MD, m, rok);
                    wg_row_to_image<NCT, SPLIT>(ig, t);
                    colsum_put<NCT>(slots, t, 2 + w);
                }
                __syncthreads();
                wg_product<NCT, SPLIT>(ia, ig, bd.g_wqkv + so + w * D, 3 * D, first);
                __syncthreads();
            }
            colsum_fold<NCT, 5>(slots, tot);
        } else {
            float t2[2] = {0.f, 0.f};
            colsum_fold<NCT, 2>(slots, t2);
            tot[0] += t2[0]; tot[1] += t2[1];
        }
    }
    if ((int)threadIdx.x < D) {
        const size_t o = (size_t)blockIdx.x * bd.slab_stride + threadIdx.x;
        bd.g_ln1_g[o] = tot[0];
        bd.g_ln1_b[o] = tot[1];
        if (wgrad) { bd.g_bqkv[o] = tot[2]; bd.g_bqkv[o + D] = tot[3]; bd.g_bqkv[o + 2 * D] = tot[4]; }
    }
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

// grid < 0: one workgroup per row block (forward); else the slab count
template <int NCT, int MODE, typename K, typename... A>
static int wide_launch(K kern, cr_devmask* done, int M, int grid, size_t lds, hipStream_t s, const char* who, A... args) {
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(kern), done);
    if (rc != CR_OK) return rc;
    if (grid < 0) grid = cr_ceil_div(M, WdCfg<NCT, MODE>::ROWS);          // forward: one workgroup per row block
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WdCfg<NCT, MODE>::NT), lds, s, args...);
    return cr_check_launch(who);
}
// forward: 128 rows per workgroup when that still gives every CU one (or at D = 128, where the registers allow two waves per
// SIMD anyway), else 64
template <int NCT, bool SPLIT>
static int wide_launch_fwd(bool qkv, const cr_block_desc* d, hipStream_t s) {
    static cr_devmask done[4];
    const bool w8 = NCT <= 8 || cr_ceil_div(d->M, 128) >= 256;
    const size_t lds = WideLds<NCT, SPLIT>::TOTAL;
    if (qkv) {
        if (w8) return wide_launch<NCT, 0>(k_wide_qkv_fwd<NCT, SPLIT, 0>, &done[0], d->M, -1, lds, s, "cr_wide_ln_qkv_fwd", *d);
        return wide_launch<NCT, 2>(k_wide_qkv_fwd<NCT, SPLIT, 2>, &done[1], d->M, -1, lds, s, "cr_wide_ln_qkv_fwd", *d);
    }
    WideTail none = {};
    if (w8) return wide_launch<NCT, 0>(k_wide_ffn_fwd<NCT, SPLIT, 0, 0>, &done[2], d->M, -1, lds, s, "cr_wide_ln_ffn_fwd", *d, none);
    return wide_launch<NCT, 2>(k_wide_ffn_fwd<NCT, SPLIT, 2, 0>, &done[3], d->M, -1, lds, s, "cr_wide_ln_ffn_fwd", *d, none);
}
static int wide_dispatch_fwd(bool qkv, const cr_block_desc* d, int precision, hipStream_t s) {
    const bool sp = precision == CR_PREC_BF16X3;
    switch (d->D / 16) {
        case 8: return sp ? wide_launch_fwd<8, true>(qkv, d, s) : wide_launch_fwd<8, false>(qkv, d, s);
        case 12: return sp ? wide_launch_fwd<12, true>(qkv, d, s) : wide_launch_fwd<12, false>(qkv, d, s);
        default: return sp ? wide_launch_fwd<16, true>(qkv, d, s) : wide_launch_fwd<16, false>(qkv, d, s);
    }
}

extern "C" int cr_wide_ln_qkv_fwd(const cr_block_desc* d, int precision, void* stream) {
    CR_REQUIRE(d, "cr_wide_ln_qkv_fwd: NULL description");
    const char* why = wide_why(d, precision);
    if (why) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_wide_ln_qkv_fwd: %s", why);
    CR_REQUIRE(d->x && d->q_in && d->qkv && d->k_valid && d->q_valid && d->wqkv && d->bqkv && d->ln1_g && d->ln1_b,
               "cr_wide_ln_qkv_fwd: NULL pointer");
    return wide_dispatch_fwd(true, d, precision, cr_stream(stream));
}

extern "C" int cr_wide_ln_ffn_fwd(const cr_block_desc* d, int precision, void* stream) {
    CR_REQUIRE(d, "cr_wide_ln_ffn_fwd: NULL description");
    const char* why = wide_why(d, precision);
    if (why) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_wide_ln_ffn_fwd: %s", why);
    CR_REQUIRE(d->o && d->f_in && d->hid && d->y && d->mask_ids && d->w1 && d->b1 && d->w2 && d->b2 && d->ln2_g && d->ln2_b,
               "cr_wide_ln_ffn_fwd: NULL pointer");
    return wide_dispatch_fwd(false, d, precision, cr_stream(stream));
}

template <int NCT, bool SPLIT, bool WG>
static int wide_launch_ffn_bwd(const cr_block_bwd_desc* bd, float* g2, float* g1, int heads, hipStream_t s) {
    static cr_devmask done = 0;
    return wide_launch<NCT, 1>(k_wide_ffn_bwd<NCT, SPLIT, WG>, &done, bd->f.M, bd->n_slabs,
                            WG ? WideLds<NCT, SPLIT>::TOTAL_BWD : WideLds<NCT, SPLIT>::TOTAL_BWD_NOWG, s, "cr_wide_ln_ffn_bwd", *bd, g2, g1, heads);
}
template <int NCT, bool SPLIT, bool WG>
static int wide_launch_qkv_bwd(const cr_block_bwd_desc* bd, hipStream_t s) {
    static cr_devmask done = 0;
    return wide_launch<NCT, 1>(k_wide_qkv_bwd<NCT, SPLIT, WG>, &done, bd->f.M, bd->n_slabs,
                            WG ? WideLds<NCT, SPLIT>::TOTAL_BWD : WideLds<NCT, SPLIT>::TOTAL_BWD_NOWG, s, "cr_wide_ln_qkv_bwd", *bd);
}

extern "C" int cr_wide_ln_ffn_bwd(const cr_block_bwd_desc* bd, float* g2, float* g1, int heads, int precision, void* stream) {
    CR_REQUIRE(bd, "cr_wide_ln_ffn_bwd: NULL description");
    const cr_block_desc* d = &bd->f;
    const char* why = wide_why(d, precision);
    if (why) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_wide_ln_ffn_bwd: %s", why);
    CR_REQUIRE(bd->dy && bd->d_o && d->hid && d->o && d->mask_ids && d->w1 && d->w2 && d->ln2_g, "cr_wide_ln_ffn_bwd: NULL pointer");
    if (bd->attn_delta) {
        CR_REQUIRE(heads > 0 && d->D % heads == 0 && (d->D / heads) % 16 == 0 && d->q_in,
                   "cr_wide_ln_ffn_bwd: attn_delta needs heads whose width is a multiple of 16 columns (and q_in)");
    } else {
        heads = 0;
    }
    CR_REQUIRE(bd->g_ln2_g && bd->g_ln2_b && bd->n_slabs > 0 && bd->slab_stride > 0, "cr_wide_ln_ffn_bwd: NULL gradient pointer / no slabs");
    const bool wg = bd->g_w1 || bd->g_w2 || bd->g_b1 || bd->g_b2;
    CR_REQUIRE(wg || (g2 && g1), "cr_wide_ln_ffn_bwd: g2 / g1 are needed when the weight gradients are left to cr_gemm_wgrad");
    if (wg) {
        CR_REQUIRE(d->D == 128, "cr_wide_ln_ffn_bwd: weight gradients are formed at D = 128 only (pass NULL g_w1 g_b1 g_w2 g_b2 and use cr_gemm_wgrad)");
        CR_REQUIRE(bd->g_w1 && bd->g_w2 && bd->g_b1 && bd->g_b2 && d->f_in, "cr_wide_ln_ffn_bwd: all of g_w1 g_b1 g_w2 g_b2 (and f_in) or none");
    }
    hipStream_t s = cr_stream(stream);
    const bool sp = precision == CR_PREC_BF16X3;
    switch (d->D / 16) {
        case 8:
            if (wg) return sp ? wide_launch_ffn_bwd<8, true, true>(bd, g2, g1, heads, s) : wide_launch_ffn_bwd<8, false, true>(bd, g2, g1, heads, s);
            return sp ? wide_launch_ffn_bwd<8, true, false>(bd, g2, g1, heads, s) : wide_launch_ffn_bwd<8, false, false>(bd, g2, g1, heads, s);
        case 12: return sp ? wide_launch_ffn_bwd<12, true, false>(bd, g2, g1, heads, s) : wide_launch_ffn_bwd<12, false, false>(bd, g2, g1, heads, s);
        default: return sp ? wide_launch_ffn_bwd<16, true, false>(bd, g2, g1, heads, s) : wide_launch_ffn_bwd<16, false, false>(bd, g2, g1, heads, s);
    }
}

extern "C" int cr_wide_ln_qkv_bwd(const cr_block_bwd_desc* bd, int precision, void* stream) {
    CR_REQUIRE(bd, "cr_wide_ln_qkv_bwd: NULL description");
    const cr_block_desc* d = &bd->f;
    const char* why = wide_why(d, precision);
    if (why) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_wide_ln_qkv_bwd: %s", why);
    CR_REQUIRE(bd->dqkv && bd->d_o && bd->dx && d->x && d->wqkv && d->ln1_g, "cr_wide_ln_qkv_bwd: NULL pointer");
    CR_REQUIRE(bd->g_ln1_g && bd->g_ln1_b && bd->n_slabs > 0 && bd->slab_stride > 0, "cr_wide_ln_qkv_bwd: NULL gradient pointer / no slabs");
    CR_REQUIRE(bd->dq_part == nullptr, "cr_wide_ln_qkv_bwd: dq_part (single-pass fp32 attention backward) is not taken");
    const bool wg = bd->g_wqkv || bd->g_bqkv;
    if (wg) {
        CR_REQUIRE(d->D == 128, "cr_wide_ln_qkv_bwd: weight gradients are formed at D = 128 only (pass NULL g_wqkv g_bqkv and use cr_gemm_wgrad)");
        CR_REQUIRE(bd->g_wqkv && bd->g_bqkv && d->q_in, "cr_wide_ln_qkv_bwd: both of g_wqkv g_bqkv (and q_in) or none");
    }
    hipStream_t s = cr_stream(stream);
    const bool sp = precision == CR_PREC_BF16X3;
    switch (d->D / 16) {
        case 8:
            if (wg) return sp ? wide_launch_qkv_bwd<8, true, true>(bd, s) : wide_launch_qkv_bwd<8, false, true>(bd, s);
            return sp ? wide_launch_qkv_bwd<8, true, false>(bd, s) : wide_launch_qkv_bwd<8, false, false>(bd, s);
        case 12: return sp ? wide_launch_qkv_bwd<12, true, false>(bd, s) : wide_launch_qkv_bwd<12, false, false>(bd, s);
        default: return sp ? wide_launch_qkv_bwd<16, true, false>(bd, s) : wide_launch_qkv_bwd<16, false, false>(bd, s);
    }
}

extern "C" int cr_wide_ln_ffn_fwd_tail(const cr_block_desc* d, const cr_block_tail_desc* t, int precision, void* stream) {
    CR_REQUIRE(d && t, "cr_wide_ln_ffn_fwd_tail: NULL description");
    if (t->kind == 0) return cr_wide_ln_ffn_fwd(d, precision, stream);
    const char* why = wide_why(d, precision);
    if (why) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_wide_ln_ffn_fwd_tail: %s", why);
    if (d->D != 128) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_wide_ln_ffn_fwd_tail: the tails are taken at D = 128 only");
    CR_REQUIRE(d->o && d->f_in && d->hid && d->y && d->mask_ids && d->w1 && d->b1 && d->w2 && d->b2 && d->ln2_g && d->ln2_b,
               "cr_wide_ln_ffn_fwd_tail: NULL pointer");
    WideTail tl = {};
    if (t->kind == 1) {
        const cr_block_desc* n = t->next;
        CR_REQUIRE(n && n->x == d->y && n->M == d->M && n->D == d->D, "cr_wide_ln_ffn_fwd_tail: next->x must be this block's y (same shape)");
        CR_REQUIRE(n->q_in && n->qkv && n->k_valid && n->q_valid && n->wqkv && n->bqkv && n->ln1_g && n->ln1_b, "cr_wide_ln_ffn_fwd_tail: NULL pointer in next");
        tl.next = *n;
    } else {
        CR_REQUIRE(t->kind == 2 && t->lnf_gamma && t->lnf_beta && t->out && t->ld_out >= t->col_out + d->D && t->col_out % 4 == 0 && t->ld_out % 4 == 0,
                   "cr_wide_ln_ffn_fwd_tail: bad final-LayerNorm tail");
        tl.lnf_g = t->lnf_gamma; tl.lnf_b = t->lnf_beta; tl.out = t->out; tl.ld_out = t->ld_out; tl.col_out = t->col_out;
    }
    hipStream_t s = cr_stream(stream);
    static cr_devmask done[4];
    const bool sp = precision == CR_PREC_BF16X3;
    const size_t lds = sp ? WideLds<8, true>::TOTAL : WideLds<8, false>::TOTAL;
    if (t->kind == 1)
        return sp ? wide_launch<8, 0>(k_wide_ffn_fwd<8, true, 0, 1>, &done[0], d->M, -1, lds, s, "cr_wide_ln_ffn_fwd_tail", *d, tl)
                  : wide_launch<8, 0>(k_wide_ffn_fwd<8, false, 0, 1>, &done[1], d->M, -1, lds, s, "cr_wide_ln_ffn_fwd_tail", *d, tl);
    return sp ? wide_launch<8, 0>(k_wide_ffn_fwd<8, true, 0, 2>, &done[2], d->M, -1, lds, s, "cr_wide_ln_ffn_fwd_tail", *d, tl)
              : wide_launch<8, 0>(k_wide_ffn_fwd<8, false, 0, 2>, &done[3], d->M, -1, lds, s, "cr_wide_ln_ffn_fwd_tail", *d, tl);
}
