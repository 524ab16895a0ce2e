// Helpers of the register-layout backward kernels (cr_stack_bwd.hip: one launch per row phase; cr_stack_bwd1.hip: one launch
// per block): permuted weight images for out^T = W g^T, the wave's tile -> bf16 image, weight gradients a^T g from images
// through transposed reads, LayerNorm backward of the lane's row and the fixed-order fold of its column sums.
#pragma once
#include "cr_rlayout.hpp"

#define SB_WAVES 8
#define SB_NT (64 * SB_WAVES)
#define SB_TPR 7                  // tiles per round (16 rows each)
#define SB_IMG (SB_TPR * 16 * 64) // bf16 elements of one activation / gradient image


// ---- weight [D][D] -> [64][64] image whose columns are in the k order of layout R's B operand -------------
// position 32 ks + 8 lg + 4 h + r  <->  column 32 ks + 16 h + 4 lg + r: row_frag_l then delivers A[i = row][k] for
// out^T = W g^T with one 16-byte read per k-step.  An item (row, columns 8 ch .. 8 ch + 7) lands as two 8-byte pieces.
template <int NWT, int NT, bool SPLIT>
__device__ __forceinline__ void w_put_perm(__bf16* Wi, WRegs<NWT, NT>& r, int D, W3_PARAMS) {
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
            const int ks = ch >> 2, c4 = ch & 3, hh = c4 >> 1, lga = 2 * (c4 & 1);
            const int oa = wi * WST + img_off<2>(k, 4 * ks + lga) + 4 * hh;
            const int ob = wi * WST + img_off<2>(k, 4 * ks + lga + 1) + 4 * hh;
            *reinterpret_cast<bf4*>(Wi + oa) = __builtin_shufflevector(h, h, 0, 1, 2, 3);
            *reinterpret_cast<bf4*>(Wi + ob) = __builtin_shufflevector(h, h, 4, 5, 6, 7);
            if (SPLIT) {
                *reinterpret_cast<bf4*>(Wi + oa + ST_WIMG) = __builtin_shufflevector(l, l, 0, 1, 2, 3);
                *reinterpret_cast<bf4*>(Wi + ob + ST_WIMG) = __builtin_shufflevector(l, l, 4, 5, 6, 7);
            }
        }
    }
}

// acc (layout R) (+)= g W^T for the permuted [in][out] image W (hi, lo): out^T[in][row] = sum_out W[in][out] g^T[out][row]
template <bool SPLIT, bool ACC>
__device__ __forceinline__ void r_gemm_t(f32x4 (&acc)[4], const __bf16* Wh, const __bf16* Wl, const bf8 (&gh)[2], const bf8 (&gl)[2]) {
    const int lane = lane_now();
    if (!ACC) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf8 wh[4], wl[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            wh[ct] = row_frag_l(Wh, 16 * ct, ks, lane);
            wl[ct] = SPLIT ? row_frag_l(Wl, 16 * ct, ks, lane) : wh[ct];
        }
        if (SPLIT) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ct], gh[ks], acc[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ct], gl[ks], acc[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ct], gh[ks], acc[ct], 0, 0, 0);
    }
}

// (the two image layouts, ATTN and W: cr_rlayout.hpp)
// the wave's tile (layout R) -> rows [row0, row0 + 16) of an image in natural column order (read transposed)
template <bool SPLIT, bool ATTN = false>
__device__ __forceinline__ void img_put(__bf16* Ih, __bf16* Il, int row0, const f32x4 (&x)[4]) {
    const int lane = lane_now(), li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        bf4 h, l;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const f32x2 v = {x[ct][r], x[ct][r + 1]};
            const bf2 hh = __builtin_convertvector(v, bf2);
            h[r] = hh[0]; h[r + 1] = hh[1];
            if (SPLIT) {
                const bf2 ll = __builtin_convertvector(v - __builtin_convertvector(hh, f32x2), bf2);
                l[r] = ll[0]; l[r + 1] = ll[1];
            }
        }
        const int o = ATTN ? img_off<2>(row0 + li, 2 * ct + (lg >> 1)) + 4 * (lg & 1) : wimg_off(row0 + li, 4 * ct + lg);
        *reinterpret_cast<bf4*>(Ih + o) = h;
        if (SPLIT) *reinterpret_cast<bf4*>(Il + o) = l;
    }
}

// acc[j] += a^T g over the rows of `ntr` tiles: output tile (in-column tile it, out-column tiles jt0, jt0 + 1).
// Two tiles per k-step: v_mfma_f32_16x16x16_bf16 holds the matrix pipe for the same 16 cycles as the 16x16x32 shape
// (tools/probes/probe_issue_cost.hip: 6.9 ns per instruction either way, and the pipe is shared by the SIMD's waves), so a product
// over 16 rows wastes half of it.  Both operands take tile t's four rows in k slots 0..3 and tile t + 1's in 4..7 (two transposed
// reads each); an odd last tile goes through the K = 16 shape.
// BIAS (D == 64: no spare column for the ones trick): the waves with it == 0 also form accb[j] += 1^T g, the column sums
// of g (an all-ones A operand: every row of the result is the bias gradient).
template <bool SPLIT, bool BIAS, int TMAX = SB_TPR>
__device__ __forceinline__ void wgrad_accum(f32x4 (&acc)[2], f32x4 (&accb)[2], const __bf16* Ah, const __bf16* Al, const __bf16* Gh, const __bf16* Gl,
                                            int ntr, int it, int jt0) {
    const int lane = lane_now();
    const __bf16 one = (__bf16)1.0f;
    const bf8 ones8 = (bf8){one, one, one, one, one, one, one, one};
    auto cat = [](const bf4& x, const bf4& y) { return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7); };
#pragma unroll
    for (int t = 0; t < TMAX; t += 2) {                  // unrolled, wave-uniform guards: several tiles' reads in flight
        if (t + 1 < ntr) {
            const bf8 ah = cat(tr4(Ah, 16 * t, it, lane), tr4(Ah, 16 * t + 16, it, lane));
            const bf8 al = SPLIT ? cat(tr4(Al, 16 * t, it, lane), tr4(Al, 16 * t + 16, it, lane)) : ah;
            bf8 gh[2], gl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                gh[j] = cat(tr4(Gh, 16 * t, jt0 + j, lane), tr4(Gh, 16 * t + 16, jt0 + j, lane));
                gl[j] = SPLIT ? cat(tr4(Gl, 16 * t, jt0 + j, lane), tr4(Gl, 16 * t + 16, jt0 + j, lane)) : gh[j];
            }
            if (SPLIT) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, gh[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gl[j], acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gh[j], acc[j], 0, 0, 0);
            if (BIAS && it == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (SPLIT) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, gl[j], accb[j], 0, 0, 0);
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, gh[j], accb[j], 0, 0, 0);
                }
            }
        } else if (t < ntr) {
            // an odd last tile: the SAME K = 32 instruction with zeros in k slots 4..7.  (It used to go through the K = 16 shape; in the plain-bf16
            // build -- two products per pair instead of six -- that 4-pass product then read, as its SrcC, the accumulator of the 8-pass product
            // issued just before it, and registers 0 / 1 of that accumulator came out wrong, run-dependent: tools/probes/bf16_w2_blocks.py,
            // DESIGN.md section 4 "Round 5: a mixed-shape accumulate".  One shape along the whole chain is the form the hardware forwards.)
            const bf4 z4 = (bf4){(__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
            const bf8 ah = cat(tr4(Ah, 16 * t, it, lane), z4);
            const bf8 al = SPLIT ? cat(tr4(Al, 16 * t, it, lane), z4) : ah;
            bf8 gh[2], gl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                gh[j] = cat(tr4(Gh, 16 * t, jt0 + j, lane), z4);
                gl[j] = SPLIT ? cat(tr4(Gl, 16 * t, jt0 + j, lane), z4) : gh[j];
            }
            if (SPLIT) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, gh[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gl[j], acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gh[j], acc[j], 0, 0, 0);
            if (BIAS && it == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (SPLIT) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, gl[j], accb[j], 0, 0, 0);
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, gh[j], accb[j], 0, 0, 0);
                }
            }
        }
    }
}
// accumulators D[in = 16 it + 4 lg + r][out = 16 (jt0 + j) + li] -> slab (row pitch ldw); row D is the bias gradient
// (D < 64), or it comes from accb (BIAS)
template <bool BIAS>
__device__ __forceinline__ void wgrad_store(float* dst, int ldw, float* bias_dst, const f32x4 (&acc)[2], const f32x4 (&accb)[2], int D, int it, int jt0) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = 16 * (jt0 + j) + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * it + 4 * lg + r;
            if (col < D) {
                if (k < D) dst[(size_t)k * ldw + col] = acc[j][r];
                else if (k == D) bias_dst[col] = acc[j][r];
            }
        }
        if (BIAS && it == 0 && lg == 0 && col < D) bias_dst[col] = accb[j][0];
    }
}
// plant 1.0 at column D (D < 64) of the lane's row: the bias-gradient row of a^T g
__device__ __forceinline__ void plant_one(f32x4 (&x)[4], int D) {
    const int lgb = (lane_now() >> 4) * 4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (16 * ct + lgb + r == D) x[ct][r] = 1.0f;
}

// LayerNorm backward of the lane's row (x pad columns 0, dy pad columns 0, gam zero padded):
// dx = rstd * (dy g - mean(dy g) - xhat mean(dy g xhat)); ag += dy xhat, ab += dy
__device__ __forceinline__ void r_ln_bwd(f32x4 (&dx)[4], const f32x4 (&x)[4], const f32x4 (&dy)[4], const float* gam,
                                         f32x4 (&ag)[4], f32x4 (&ab)[4], const DCtx& dc) {
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
    const float rstd = 1.0f / sqrtf(grp_sum(v) * invD + 1e-8f);
    f32x4 g[4];
    r_vec(g, gam);
    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xc[ct][r] *= rstd;                                       // xhat (0 in the pad columns)
            const float dg = dy[ct][r] * g[ct][r];
            c1 += dg;
            c2 = fmaf(dg, xc[ct][r], c2);
            ag[ct][r] = fmaf(dy[ct][r], xc[ct][r], ag[ct][r]);
            ab[ct][r] += dy[ct][r];
            dx[ct][r] = dg;
        }
    c1 = grp_sum(c1) * invD;
    c2 = grp_sum(c2) * invD;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        if (ct < dc.nfull) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dx[ct][r] = cr_ln_bwd_tail(dx[ct][r], c1, xc[ct][r], c2, rstd);     // three scalar instructions, see there
        } else if (ct == dc.nfull) {
            const int lgb = (lane_now() >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) dx[ct][r] = (lgb + r < dc.rem) ? cr_ln_bwd_tail(dx[ct][r], c1, xc[ct][r], c2, rstd) : 0.0f;
        } else {
            dx[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
}
// per-lane LayerNorm-gradient partials (the lane's row, 16 columns) -> sums over the wave's 16 rows -> per-wave LDS
// slots -> fixed-order sum over the waves -> slab
__device__ __forceinline__ void ln_grads_store(float* part /* [2][SB_WAVES][64] */, f32x4 (&ag)[4], f32x4 (&ab)[4], float* dg, float* db, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sg = cr_row16_sum(ag[ct][r]), sb = cr_row16_sum(ab[ct][r]);
            if (li == 0) {
                part[wave * 64 + 16 * ct + 4 * lg + r] = sg;
                part[(SB_WAVES + wave) * 64 + 16 * ct + 4 * lg + r] = sb;
            }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += SB_NT) {
        float g = 0.0f, b = 0.0f;
#pragma unroll
        for (int w = 0; w < SB_WAVES; ++w) { g += part[w * 64 + c]; b += part[(SB_WAVES + w) * 64 + c]; }
        dg[c] = g;
        db[c] = b;
    }
}

