// Register layout R and its helpers, shared by the whole-stack forward kernel (cr_stack.hip) and the row-phase
// backward kernels (cr_stack_bwd.hip).
//
// Layout R: a wave owns a 16-row tile; lane (li = lane & 15, lg = lane >> 4) holds row li of the tile, columns
// 16 ct + 4 lg + r (ct, r = 0..3) -- the D-operand layout of v_mfma_f32_16x16x32_bf16 for the TRANSPOSED product
// out^T = W^T x^T (see cr_stack.hip), so chains of row-local layers run through registers.
#pragma once
#include "cr_attn_common.hpp"
#include "cr_bf16.hpp"

#define ST_WIMG 4096              // bf16 elements of one [64][64] weight image

// ---- layout R <-> rows of a dense [*, ld] matrix ----------------------------------------------------------
// Columns come in 16-byte pieces (column tile ct, lane group lg: columns 16 ct + 4 lg .. + 3).  With nfull = D / 16 and
// rem = D % 16 (wave-uniform), tiles ct < nfull are whole for every lane: plain loads / stores, no masks, no address
// clamps -- scalar branches decide.  Only tile ct == nfull needs lane predicates: pieces with 4 lg + 4 <= rem are
// whole, the piece with 4 lg < rem < 4 lg + 4 (D % 4 != 0) is moved element by element, the rest do not exist (read as
// 0).  A lane only ever reads elements that r_store of the SAME lane wrote, so a tile written earlier by this wave can
// be re-read without a barrier in between.  Nothing outside a row is touched.
// (The first version clamped and masked every piece of every tile per lane: 5 VALU instructions per element, and the
//  kernel is VALU-issue bound -- about 700 of the 1100 instructions of a phase-A tile were this bookkeeping.)
struct DCtx { int D, nfull, rem, np; };                   // np = D % 4: elements of the crossing piece
// NF >= 0: the number of whole column tiles as a COMPILE-TIME constant (the caller guarantees NF == D / 16): the tile classification
// of every helper below then folds at any hidden size of that family -- what is left at run time are two scalars (rem, np) and the
// lane predicates of the ONE boundary tile.  With nfull a run-time value every tile of every helper keeps all three forms alive, and
// the one-launch block backward spilled 108-228 bytes per lane (round 4's generic instantiation).
template <int NF = -1>
__device__ __forceinline__ DCtx d_ctx(int D) { DCtx c; c.D = D; c.nfull = NF >= 0 ? NF : (D >> 4); c.rem = D & 15; c.np = D & 3; return c; }
// the template argument for a kernel's hidden-size parameter DS (> 0: exact size; 0: anything; < 0: the family nfull = -DS - 1)
#define D_NF(DS) ((DS) > 0 ? (DS) / 16 : ((DS) < 0 ? -(DS) - 1 : -1))
// Thread / lane number the optimiser cannot hoist or share between uses: everything derived from it (column offsets,
// pad masks) is then recomputed where it is used -- a few VALU ops -- instead of being kept live across the whole kernel
// (the loop-invariant per-lane values of all helpers together spilled 160 registers).
__device__ __forceinline__ int tid_now() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
__device__ __forceinline__ int lane_now() { return tid_now() & 63; }
typedef uint32_t u32;
struct RRaw { f4u v[4]; float p[3]; };

// base: a wave-uniform pointer; rowb: the row's BYTE offset (32 bits: the host checks M * D * 4 < 2^32)
// rok = false: the lane's row does not exist -- nothing is loaded, the row reads as zeros
__device__ __forceinline__ void r_issue(RRaw& w, const float* base, u32 rowb, const DCtx& dc, bool rok = true) {
    const int lgb = (lane_now() >> 4) * 4;
    const char* b = reinterpret_cast<const char*>(base);
    const u32 ob = rowb + 4u * (u32)lgb;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        w.v[ct] = (f4u){0.f, 0.f, 0.f, 0.f};
        if (ct < dc.nfull) {
            if (rok) w.v[ct] = *reinterpret_cast<const f4u*>(b + (ob + 64u * ct));
        } else if (ct == dc.nfull) {
            if (rok && lgb + 4 <= dc.rem) w.v[ct] = *reinterpret_cast<const f4u*>(b + (ob + 64u * ct));
        }
    }
    w.p[0] = w.p[1] = w.p[2] = 0.0f;
    if (dc.np) {                                          // wave-uniform
        // the crossing piece (columns 16 nfull + (rem & ~3) ..: np elements) is read by EVERY lane of the row, from an address that
        // does not depend on the lane group -- inside the row, same cache line -- and r_finish keeps it in the one lane group that owns
        // it: no lane predicate, i.e. no exec-mask branch around the load (seven per tile in the block backward's chains)
        if (rok) {
            const float* q = reinterpret_cast<const float*>(b + (rowb + 64u * (u32)dc.nfull + 4u * (u32)(dc.rem & ~3)));
            w.p[0] = q[0];
            if (dc.np > 1) w.p[1] = q[1];
            if (dc.np > 2) w.p[2] = q[2];
        }
    }
}
__device__ __forceinline__ void r_finish(f32x4 (&x)[4], const RRaw& w, const DCtx& dc) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) x[ct] = (f32x4){w.v[ct].x, w.v[ct].y, w.v[ct].z, w.v[ct].w};
    if (dc.np) {
        const int lgb = (lane_now() >> 4) * 4;
        const bool part = lgb < dc.rem && lgb + 4 > dc.rem;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            if (ct == dc.nfull) {                         // wave-uniform
                x[ct][0] = part ? w.p[0] : x[ct][0];
                x[ct][1] = part ? w.p[1] : x[ct][1];
                x[ct][2] = part ? w.p[2] : x[ct][2];
            }
    }
}
// NT: a streaming store (written once here, read by a LATER launch -- saved activations, gradient slabs): it does not stay in the
// L2 as a dirty line that the next launch's first loads wait behind
template <bool NT = false>
__device__ __forceinline__ void r_store(float* base, u32 rowb, const f32x4 (&x)[4], bool rok, const DCtx& dc) {
    const int lgb = (lane_now() >> 4) * 4;
    char* b = reinterpret_cast<char*>(base);
    const u32 ob = rowb + 4u * (u32)lgb;
    auto st4 = [](char* p, const f32x4& v) {
        const f4u t = (f4u){v[0], v[1], v[2], v[3]};
        if (NT) __builtin_nontemporal_store(t, reinterpret_cast<f4u*>(p));
        else *reinterpret_cast<f4u*>(p) = t;
    };
    auto st1 = [](float* p, float v) {
        if (NT) __builtin_nontemporal_store(v, p);
        else *p = v;
    };
    if (rok) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct < dc.nfull) {
                st4(b + (ob + 64u * ct), x[ct]);
            } else if (ct == dc.nfull) {
                if (lgb + 4 <= dc.rem) st4(b + (ob + 64u * ct), x[ct]);
                if (dc.np && lgb < dc.rem && lgb + 4 > dc.rem) {
                    float* q = reinterpret_cast<float*>(b + (ob + 64u * ct));
                    st1(q, x[ct][0]);
                    if (dc.np > 1) st1(q + 1, x[ct][1]);
                    if (dc.np > 2) st1(q + 2, x[ct][2]);
                }
            }
        }
    }
}
// a zero-padded 64-float LDS vector in layout R (the lane's 16 columns)
__device__ __forceinline__ void r_vec(f32x4 (&v)[4], const float* vec) {
    const int lg = lane_now() >> 4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const float4 t = *reinterpret_cast<const float4*>(vec + 16 * ct + 4 * lg);
        v[ct] = (f32x4){t.x, t.y, t.z, t.w};
    }
}
__device__ __forceinline__ float r_rowsum(const f32x4 (&x)[4]) {
    float s = 0.0f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) s += (x[ct][0] + x[ct][1]) + (x[ct][2] + x[ct][3]);
    return grp_sum(s);
}
// LayerNorm of the lane's row (modules.py:74-78).  x is 0 in the pad columns; gam / bet are zero padded, so pad columns
// come out 0 without a mask; the centred values are masked in the one column tile that holds the boundary.
// RECIP: y = g * (xc * (1 / sd)) + b (the block kernels' form), else g * (xc / sd) + b (cr_layernorm_fwd's)
template <bool RECIP>
__device__ __forceinline__ void r_layernorm(f32x4 (&y)[4], const f32x4 (&x)[4], const float* gam, const float* bet, const DCtx& dc) {
    const float invD = 1.0f / (float)dc.D;
    const float mean = r_rowsum(x) * invD;
    f32x4 xc[4];
    float v = 0.0f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        if (ct < dc.nfull) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xc[ct][r] = x[ct][r] - mean;
        } else if (ct == dc.nfull) {
            const int lgb = (lane_now() >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) xc[ct][r] = (lgb + r < dc.rem) ? x[ct][r] - mean : 0.0f;
        } else {
            xc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v = fmaf(xc[ct][r], xc[ct][r], v);
    }
    const float sd = sqrtf(grp_sum(v) * invD + 1e-8f);
    const float rs = 1.0f / sd;
    f32x4 g[4], b[4];
    r_vec(g, gam);
    r_vec(b, bet);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[ct][r] = RECIP ? fmaf(g[ct][r], xc[ct][r] * rs, b[ct][r]) : fmaf(g[ct][r], xc[ct][r] / sd, b[ct][r]);
}

// cr_bf16.hpp's operand reads with the lane number passed in (an opaque copy per tile iteration: the address sums of
// one iteration's reads are then formed where they are used instead of being hoisted out of the tile loop -- 56 of
// them -- and spilled)
__device__ __forceinline__ bf8 row_frag_l(const __bf16* img, int row0, int ks, int lane) {
    return *reinterpret_cast<const bf8*>(img + img_off<2>(row0 + (lane & 15), (lane >> 4) + 4 * ks));
}
__device__ __forceinline__ bf8 tr_frag_l(const __bf16* img, int ra, int rb, int jt, int lane) {
    const int lg = lane >> 4, idx = lane & 15, q = idx >> 2, p = idx & 3;
    const int ch = 2 * jt + (p >> 1), sub = 4 * (p & 1);
    const bf4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + img_off<2>(ra + 4 * lg + q, ch) + sub));
    const bf4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + img_off<2>(rb + 4 * lg + q, ch) + sub));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Two image layouts.  ATTN: cr_bf16.hpp's (16-byte chunks XOR-ed with row & 6: one image serves row reads AND transposed reads) --
// a tile written from layout R hits it with 4-way bank conflicts (16 rows, one chunk: four chunk positions), which the attention
// images accept.  Weight-gradient images are only ever read transposed, so they use layout W: 8-byte slots (four elements), slot
// index XOR-ed with a bijection of the row's low four bits -- rows 2, 4, 6 of an aligned 8-row group move to other 32-byte quads
// (the transposed read's 32-lane half: 8 rows x 32 bytes, conflict-free), rows that differ in bits 0 / 3 permute inside the quad
// (the write's 16 lanes: 16 rows, one slot each, all 32 banks once).  Counters before: a third of the LDS-active cycles of the
// block backward were bank conflicts, all of them these writes (1 144 per sequence pair x 12 extra cycles).
__device__ __forceinline__ int wimg_swz(int row) { return (((row >> 1) & 3) << 2) | ((row & 1) << 1) | ((row >> 3) & 1); }
__device__ __forceinline__ int wimg_off(int row, int slot) { return row * 64 + ((slot ^ wimg_swz(row)) << 2); }      // bf16 elements

// four rows (k = 4 lg + 0..3 of the tile at row0) of image column 16 jt + li of a layout-W image: the K = 16 MFMA's A or B operand
__device__ __forceinline__ bf4 tr4(const __bf16* img, int row0, int jt, int lane) {
    const int lg = lane >> 4, idx = lane & 15, q = idx >> 2, p = idx & 3;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + wimg_off(row0 + 4 * lg + q, 4 * jt + p)));
}
// B operand with k = row out of a layout-W image (cr_bf16.hpp's tr_frag for that layout): k index 8 lg + j <-> row (j < 4 ? ra : rb) +
// 4 lg + (j & 3), output column 16 jt + li
__device__ __forceinline__ bf8 tr_frag_w(const __bf16* img, int ra, int rb, int jt, int lane) {
    const bf4 t0 = tr4(img, ra, jt, lane), t1 = tr4(img, rb, jt, lane);
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- out^T = W^T x^T on the lane's row -------------------------------------------------------------------
template <bool SPLIT>
__device__ __forceinline__ void r_split(const f32x4 (&x)[4], bf8 (&h)[2], bf8 (&l)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const float v[8] = {x[2 * ks][0], x[2 * ks][1], x[2 * ks][2], x[2 * ks][3],
                            x[2 * ks + 1][0], x[2 * ks + 1][1], x[2 * ks + 1][2], x[2 * ks + 1][3]};
        split8<SPLIT>(v, h[ks], l[ks]);
    }
}
// acc (layout R) = x W for the [in][out] image W (hi, lo).  Per k-step the four output-column tiles' fragments are
// read as one batch; consecutive MFMAs then belong to four independent accumulators (no back-to-back dependency).
template <bool SPLIT>
__device__ __forceinline__ void r_gemm(f32x4 (&acc)[4], const __bf16* Wh, const __bf16* Wl, const bf8 (&xh)[2], const bf8 (&xl)[2]) {
    const int lane = lane_now();
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf8 wh[4], wl[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            wh[ct] = tr_frag_l(Wh, 32 * ks, 32 * ks + 16, ct, lane);
            wl[ct] = SPLIT ? tr_frag_l(Wl, 32 * ks, 32 * ks + 16, ct, lane) : wh[ct];
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

// ---- NWT weights [D][D] (row pitch ld, column offset c0) -> [64][64] images in consecutive slots -----------
// 512 items (row k, 16-byte chunk) per weight, spread over the NT threads of the workgroup; issue and put are apart
// so the loads fly across a barrier.
struct WSrc { const float* p; int ld, c0; };
template <int NWT, int NT> struct WRegs { float v[(NWT * 512 + NT - 1) / NT][8]; };
// (scalars, not an array of sources: an indexed array of them ended up in scratch memory)
#define W3_PARAMS const float* p0, int ld0, int c00, const float* p1, int ld1, int c01, const float* p2, int ld2, int c02
#define W3_ARGS p0, ld0, c00, p1, ld1, c01, p2, ld2, c02
__device__ __forceinline__ WSrc w_pick(int wi, W3_PARAMS) {
    WSrc s;
    s.p = wi == 0 ? p0 : (wi == 1 ? p1 : p2);
    s.ld = wi == 0 ? ld0 : (wi == 1 ? ld1 : ld2);
    s.c0 = wi == 0 ? c00 : (wi == 1 ? c01 : c02);
    return s;
}
template <int NWT, int NT>
__device__ __forceinline__ void w_issue(WRegs<NWT, NT>& r, int D, W3_PARAMS) {
#pragma unroll
    for (int u = 0; u < (NWT * 512 + NT - 1) / NT; ++u) {
        const int item = min(tid_now() + NT * u, NWT * 512 - 1);
        const WSrc s = w_pick(item >> 9, W3_ARGS);          // wave-uniform
        const int k = (item & 511) >> 3, ch = item & 7;
        const bool rok = k < D;
        item_issue(r.v[u], s.p + (size_t)(rok ? k : 0) * s.ld + s.c0, 8 * ch, D, item_fix(rok, k == D - 1, 8 * ch, D));
    }
}
template <int NWT, int NT, bool SPLIT>
__device__ __forceinline__ void w_put(__bf16* Wi, WRegs<NWT, NT>& r, int D, W3_PARAMS) {
    constexpr int WST = SPLIT ? 2 * ST_WIMG : ST_WIMG;
#pragma unroll
    for (int u = 0; u < (NWT * 512 + NT - 1) / NT; ++u) {
        const int item = tid_now() + NT * u;
        if (item < NWT * 512) {
            const int wi = item >> 9, k = (item & 511) >> 3, ch = item & 7;
            const bool rok = k < D;
            const bool fix = item_fix(rok, k == D - 1, 8 * ch, D);
            item_mask(r.v[u], 8 * ch, D, rok, fix);
            if (__builtin_expect(fix, 0)) {                // one thread per weight
                const WSrc s = w_pick(wi, W3_ARGS);
                item_refill(r.v[u], s.p + (size_t)k * s.ld + s.c0, 8 * ch, D);
            }
            bf8 h, l;
            split8<SPLIT>(r.v[u], h, l);
            const int o = wi * WST + img_off<2>(k, ch);
            *reinterpret_cast<bf8*>(Wi + o) = h;
            if (SPLIT) *reinterpret_cast<bf8*>(Wi + o + ST_WIMG) = l;
        }
    }
}
