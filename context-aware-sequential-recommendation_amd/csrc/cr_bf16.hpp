// bf16 matrix-pipe primitives shared by the attention kernels (cr_attn_bf.hip) and the dense-layer GEMMs (cr_gemm_bf.hip):
// the [rows][64] bf16 LDS image and its two kinds of operand reads, the hi + lo split of fp32 operands, the three-product
// MFMA, and the branch-free 8-float items global -> register traffic is made of.  See cr_attn_bf.hip for the design notes.
#pragma once
#include "cr_common.hpp"

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf4 lds_bf4;

#ifndef BF_NO_SGB
#define BF_SGB(mask, n, id) __builtin_amdgcn_sched_group_barrier(mask, n, id)
#else
#define BF_SGB(mask, n, id) do { } while (0)
#endif

// Element offset of 16-byte chunk `ch` of row `row` in an image of 32 * NKS columns (NKS = 2: 128-byte rows, 8 chunks,
// chunk ^ (row & 6); NKS = 1: 64-byte rows, 4 chunks, chunk ^ ((row & 4) >> 1)).  Both maps are conflict-free for the row
// reads and for the transposed reads (tools/lds_banks.py: brute force over the XOR-linear maps).
template <int NKS = 2>
__device__ __forceinline__ int img_off(int row, int ch) {
    return NKS == 2 ? row * 64 + ((ch ^ (row & 6)) << 3) : row * 32 + ((ch ^ ((row & 4) >> 1)) << 3);
}

// A / B operand with k = head dim: row `row0 + li`, columns 32 ks + 8 lg .. + 7
template <int NKS = 2>
__device__ __forceinline__ bf8 row_frag(const __bf16* img, int row0, int ks) {
    const int lane = threadIdx.x & 63;
    return *reinterpret_cast<const bf8*>(img + img_off<NKS>(row0 + (lane & 15), (lane >> 4) + 4 * ks));
}

// B operand with k = row: k index 8 lg + j <-> row (j < 4 ? ra : rb) + 4 lg + (j & 3), output column 16 jt + li
template <int NKS = 2>
__device__ __forceinline__ bf8 tr_frag(const __bf16* img, int ra, int rb, int jt) {
    const int lane = threadIdx.x & 63, lg = lane >> 4, idx = lane & 15, q = idx >> 2, p = idx & 3;
    const int ch = 2 * jt + (p >> 1), sub = 4 * (p & 1);
    const bf4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + img_off<NKS>(ra + 4 * lg + q, ch) + sub));
    const bf4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + img_off<NKS>(rb + 4 * lg + q, ch) + sub));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool SPLIT>
__device__ __forceinline__ f32x4 mma(const bf8& ah, const bf8& al, const bf8& bh, const bf8& bl, f32x4 c) {
    if (SPLIT) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);      // small terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}

// hi = bf16(x), lo = bf16(x - hi), two elements at a time: v_cvt_pk_bf16_f32, the bf16 pair widened with a shift and a
// mask, v_pk_add_f32, v_cvt_pk_bf16_f32 -- 2.5 instructions per element (the element-wise form compiled to 5)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
template <bool SPLIT>
__device__ __forceinline__ void split8(const float (&x)[8], bf8& hi, bf8& lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 v = {x[j], x[j + 1]};
        const bf2 h = __builtin_convertvector(v, bf2);
        hi[j] = h[0];
        hi[j + 1] = h[1];
        if (SPLIT) {
            const bf2 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), bf2);
            lo[j] = l[0];
            lo[j + 1] = l[1];
        }
    }
}

// ---- global -> register / LDS traffic -------------------------------------------------------------------
// An item is 8 consecutive floats of a row: columns c .. c+7 of a d-column head block, two dword-aligned 16-byte
// loads.  Nothing here branches, so every load of a batch is in flight before the first use: a chunk that crosses
// column d is read whole (it runs into the next row of the matrix) and masked; a chunk beyond d, and the one chunk
// in the whole matrix whose overrun would leave it (the partial chunk of the LAST row), read the 8 floats that end
// at column d instead -- always inside the matrix, the host checks it holds 8 floats -- and are masked to zero;
// the thread that owns that last-row chunk then re-reads its valid columns one by one (a divergent branch that a
// single wave of the grid ever takes).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ bool item_fix(bool rok, bool last_row, int c, int d) { return rok && last_row && c < d && c + 8 > d; }
__device__ __forceinline__ void item_issue(float (&v)[8], const float* p, int c, int d, bool fix) {
    const int cl = (c >= d || fix) ? d - 8 : c;
    const f4u a = *reinterpret_cast<const f4u*>(p + cl), b = *reinterpret_cast<const f4u*>(p + cl + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
// (only the eight floats stay live between issue and mask: row / column facts are recomputed)
__device__ __forceinline__ void item_mask(float (&v)[8], int c, int d, bool rok, bool fix) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (rok && !fix && c + j < d) ? v[j] : 0.0f;
}
// the partial chunk of the matrix's last row, re-read column by column (behind branches one wave of the grid takes)
__device__ __forceinline__ void item_refill(float (&v)[8], const float* p, int c, int d) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[c + j < d ? c + j : d - 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c + j < d) ? v[j] : 0.0f;
}

// Register fragment of the wave's own 16-row tile straight from global memory (operand with k = head dim):
// lane (li, lg) holds row grow0 + li, columns 32 ks + 8 lg + j.  Issue early, finish (mask, split) at first use.
template <int NKS>
struct GFrag { float v[NKS][8]; };
template <int NKS>
__device__ __forceinline__ void gfrag_issue(GFrag<NKS>& f, const float* src, int ld, int grow0, int hoff, int nvalid, int d, int M) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const bool rok = li < nvalid;
    const int grow = grow0 + (rok ? li : 0);
    const float* p = src + (size_t)grow * ld + hoff;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) item_issue(f.v[ks], p, 32 * ks + 8 * lg, d, item_fix(rok, grow == M - 1, 32 * ks + 8 * lg, d));
}
template <int NKS>
__device__ __forceinline__ void gfrag_mask(GFrag<NKS>& f, const float* src, int ld, int grow0, int hoff, int nvalid, int d, int M) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const bool rok = li < nvalid;
    const int grow = grow0 + (rok ? li : 0);
    bool any_fix = false;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const bool fix = item_fix(rok, grow == M - 1, 32 * ks + 8 * lg, d);
        item_mask(f.v[ks], 32 * ks + 8 * lg, d, rok, fix);
        any_fix |= fix;
    }
    if (__builtin_expect(grow0 + 16 >= M && __any(any_fix ? 1 : 0), 0)) {        // wave-uniform, true for one tile of the grid
        const float* p = src + (size_t)grow * ld + hoff;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            if (item_fix(rok, grow == M - 1, 32 * ks + 8 * lg, d)) item_refill(f.v[ks], p, 32 * ks + 8 * lg, d);
    }
}
template <bool SPLIT, int NKS>
__device__ __forceinline__ void gfrag_finish(GFrag<NKS>& f, const float* src, int ld, int grow0, int hoff, int nvalid, int d, int M,
                                             bf8 (&hi)[NKS], bf8 (&lo)[NKS]) {
    gfrag_mask<NKS>(f, src, ld, grow0, hoff, nvalid, d, M);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) split8<SPLIT>(f.v[ks], hi[ks], lo[ks]);
}

