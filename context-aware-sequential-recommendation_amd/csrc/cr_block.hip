// Fused row-phase kernels of one transformer block for hidden sizes D <= 64 (sasrec.py:65-83 minus the
// attention core): a 64-row tile of activations stays in LDS from LayerNorm through the projections,
// so each activation is read from HBM/L2 once per phase and ~10 small launches per block disappear.
//
// Tile shapes: workgroup = 4 waves = 64 rows (wave w owns rows [16w, 16w+16)), all D (<= 64) columns.
// MFMA v_mfma_f32_16x16x4_f32.  LDS pitches: row tiles / row-read weights P = 4*ceil(D/4) + 2 (A-pattern
// conflict-free, % 4 == 2); column-read weights PW = 80 (B-pattern conflict-free).  All LDS is dynamic and
// sized by D, so the forward kernels fit two workgroups per CU.
//
// Latency structure (profiles/r01_d_*): forward kernels stage the row tile and ALL their weights with one
// burst of loads and one barrier -- after it every wave works only on rows it owns.  The persistent backward
// kernels prefetch the next tile into registers under the MFMAs of the current one, and obtain the bias
// gradients for free by planting a column of ones in the A-tile of the weight-gradient MFMA (D < 64).
#include <stdlib.h>

#include "cr_common.hpp"

#define BK_PW 80

struct BlockGeom {
    int P;        // row-tile pitch
    int ks;       // k-steps of 4 covering D
    int ones;     // column holding 1.0 for the bias-gradient trick, or -1 (D == 64)
    int dbg;      // timing-only ablation switches (env CR_BLOCK_DBG); 0 in production
    uint32_t invD;  // floor(2^32 / D) + 1: e / D == umulhi(e, invD) for e < 2^16
};

// ---- small helpers ---------------------------------------------------------------------------------
// 16 elements per thread of a [64 x 64] window of a row-major matrix: rows m0 + tr + 4i, column c0 + tc
__device__ __forceinline__ void fetch_tile(float (&v)[16], const float* src, int ld, int c0, int m0, int m_end, int D) {
    const int tr = threadIdx.x >> 6, tc = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + tr + 4 * i;
        v[i] = (m < m_end && tc < D) ? src[(size_t)m * ld + c0 + tc] : 0.0f;
    }
}
// ... and their place in a row tile (pitch P); column `ones` of valid rows is set to 1 (bias-gradient trick)
__device__ __forceinline__ void put_tile(float* dst, const float (&v)[16], int P, int ones, int m0, int m_end) {
    const int tr = threadIdx.x >> 6, tc = threadIdx.x & 63;
    if (tc < P) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = tr + 4 * i;
            dst[r * P + tc] = (tc == ones && m0 + r < m_end) ? 1.0f : v[i];
        }
    }
}
__device__ __forceinline__ void load_tile(float* dst, const float* src, int ld, int c0, int m0, int m_end, int D, int P, int ones) {
    float v[16];
    fetch_tile(v, src, ld, c0, m0, m_end, D);
    put_tile(dst, v, P, ones, m0, m_end);
}

// weight [K=D rows][N=D cols] (row pitch ldw, column offset c0) -> Ws[k][pitch]; zero padded to 64 x 64.
// pitch = BK_PW: B operand of x @ W (column-read).  pitch = P: read by rows it is the B operand of g @ W^T.
__device__ __forceinline__ void load_w(float* Ws, int pitch, const float* W, int ldw, int c0, int D, int rows = 64) {
    const int tr = threadIdx.x >> 6, tc = threadIdx.x & 63;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = tr + 4 * i;
        v[i] = (k < D && tc < D) ? W[(size_t)k * ldw + c0 + tc] : 0.0f;
    }
    if (tc < pitch) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (tr + 4 * i < rows) Ws[(tr + 4 * i) * pitch + tc] = v[i];
    }
}

// acc[j] (+)= As[rows 16w..][k] * Ws[k][16j..]   (x @ W), ks k-steps
__device__ __forceinline__ void tile_mma(f32x4 (&acc)[4], const float* As, int P, const float* Ws, int ks, int wave) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const float* ap = As + (16 * wave + li) * P + lg;
    const float* bp = Ws + lg * BK_PW + li;
#pragma unroll 2
    for (int kk = 0; kk < ks; ++kk) {
        const float a = ap[4 * kk];
        float b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = bp[4 * kk * BK_PW + 16 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = mfma16(a, b[j], acc[j]);
    }
}

// acc[i][k] (+)= sum_n As[i][n] * Wr[k][n]   (g @ W^T; lanes walk the ROWS k = 16j + li of the staged W)
__device__ __forceinline__ void tile_mma_t(f32x4 (&acc)[4], const float* As, const float* Wr, int P, int ks, int wave) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const float* ap = As + (16 * wave + li) * P + lg;
    const float* bp = Wr + li * P + lg;
#pragma unroll 2
    for (int kk = 0; kk < ks; ++kk) {
        const float a = ap[4 * kk];
        float b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = bp[16 * j * P + 4 * kk];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = mfma16(a, b[j], acc[j]);
    }
}

// accw[j] += sum_m As[m][16w + li] * Gs[m][16j + li] over the 64 rows of the tile (A^T G: weight-gradient
// strip of k-rows [16w, 16w+16) owned by wave w; with a ones column in As, row `ones` is the bias gradient)
__device__ __forceinline__ void tile_wgrad(f32x4 (&accw)[4], const float* As, const float* Gs, int P, int wave) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const float* ap = As + lg * P + 16 * wave + li;
    const float* gp = Gs + lg * P + li;
#pragma unroll 4
    for (int mm = 0; mm < 16; ++mm) {
        const float a = ap[4 * mm * P];
        float b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = gp[4 * mm * P + 16 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j) accw[j] = mfma16(a, b[j], accw[j]);
    }
}

// The wave's 16 rows of a DENSE [M, D] matrix (ld == D) are one contiguous 16*D-float block: move it as
// a 16-byte-per-lane stream (full cache lines) instead of 4-byte lanes on 4*D-byte row segments.
__device__ __forceinline__ void wave_store_rows(float* gdst, const float* Ts, int P, int D, int nrows, uint32_t invD) {
    const int lane = threadIdx.x & 63;
    const int total = nrows * D, nf4 = total >> 2;
    for (int f = lane; f < nf4; f += 64) {
        const int e = 4 * f;
        int r = (int)__umulhi((uint32_t)e, invD), c = e - r * D;
        float4 v;
        float* pv = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pv[u] = Ts[r * P + c];
            if (++c == D) { c = 0; ++r; }
        }
        reinterpret_cast<float4*>(gdst)[f] = v;
    }
    for (int e = 4 * nf4 + lane; e < total; e += 64) {
        const int r = (int)__umulhi((uint32_t)e, invD), c = e - r * D;
        gdst[e] = Ts[r * P + c];
    }
}
__device__ __forceinline__ void wave_load_rows(float* Ts, const float* gsrc, int P, int D, int nrows, uint32_t invD) {
    const int lane = threadIdx.x & 63;
    const int total = nrows * D, nf4 = total >> 2;
    for (int f = lane; f < nf4; f += 64) {
        const int e = 4 * f;
        int r = (int)__umulhi((uint32_t)e, invD), c = e - r * D;
        const float4 v = reinterpret_cast<const float4*>(gsrc)[f];
        const float* pv = reinterpret_cast<const float*>(&v);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Ts[r * P + c] = pv[u];
            if (++c == D) { c = 0; ++r; }
        }
    }
    for (int e = 4 * nf4 + lane; e < total; e += 64) {
        const int r = (int)__umulhi((uint32_t)e, invD), c = e - r * D;
        Ts[r * P + c] = gsrc[e];
    }
}

__device__ __forceinline__ void zero_acc(f32x4 (&acc)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ float sum16(float v) { return cr_row16_sum(v); }   // over the 16 lanes of a row group

// LayerNorm of the wave's 16 rows of Xs (modules.py:74-78): 16 lanes per row, 4 rows per pass.
// Writes y to Ys (LDS, zero beyond D) and to global `yg`; optional row-nonzero flags of x and y.
__device__ __forceinline__ void ln_rows(const float* Xs, float* Ys, int P, const float* gamma, const float* beta, float* yg,
                                        float* x_nz, float* y_nz, int m0, int M, int D, int wave) {
    const int lane = threadIdx.x & 63, l = lane & 15, sub = lane >> 4;
    const float invD = 1.0f / (float)D;
    float g[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = l + 16 * i;
        g[i] = (c < D) ? gamma[c] : 0.0f;
        b[i] = (c < D) ? beta[c] : 0.0f;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = 16 * wave + 4 * p + sub, m = m0 + r;
        float x[4], s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = (l + 16 * i < D) ? Xs[r * P + l + 16 * i] : 0.0f; s += x[i]; }
        s = sum16(s);
        const float mean = s * invD;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float dx = (l + 16 * i < D) ? x[i] - mean : 0.0f; v += dx * dx; }
        const float sd = sqrtf(sum16(v) * invD + 1e-8f);
        float ys = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = l + 16 * i;
            const float y = (c < D) ? g[i] * ((x[i] - mean) / sd) + b[i] : 0.0f;
            if (c < P) Ys[r * P + c] = y;
            if (yg && c < D && m < M) yg[(size_t)m * D + c] = y;
            ys += y;
        }
        ys = sum16(ys);
        if (l == 0 && m < M) {
            if (x_nz) x_nz[m] = (s != 0.0f) ? 1.0f : 0.0f;
            if (y_nz) y_nz[m] = (ys != 0.0f) ? 1.0f : 0.0f;
        }
    }
}

// ---- F1: LN1 + Q/K/V projections --------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_block_ln_qkv_fwd(cr_block_desc d, BlockGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = g.P, D = d.D;
    float* Xs = smem;                       // [64][P]
    float* Qs = Xs + 64 * P;                // [64][P]
    float* Ws = Qs + 64 * P;                // 3 x [4*ks][PW]: Wq, Wk, Wv (k rows beyond D are zero)
    const int wsz = 4 * g.ks * BK_PW;
    const int m0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    load_tile(Xs, d.x, D, 0, m0, d.M, D, P, -1);
#pragma unroll 1
    for (int part = 0; part < 3; ++part) load_w(Ws + part * wsz, BK_PW, d.wqkv, 3 * D, part * D, D, 4 * g.ks);
    __syncthreads();                        // the only barrier: from here on a wave touches only rows it owns
    ln_rows(Xs, Qs, P, d.ln1_g, d.ln1_b, d.q_in, d.k_valid, d.q_valid, m0, d.M, D, wave);   // sasrec.py:69; masks modules.py:222,248
#pragma unroll 1
    for (int part = 0; part < 3; ++part) {                                                  // modules.py:203-205
        f32x4 acc[4];
        zero_acc(acc);
        tile_mma(acc, part == 0 ? Qs : Xs, P, Ws + part * wsz, g.ks, wave);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 16 * j + li;
            if (col < D) {
                const float bias = d.bqkv[part * D + col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * wave + 4 * lg + r;
                    if (m < d.M) d.qkv[(size_t)m * 3 * D + part * D + col] = acc[j][r] + bias;
                }
            }
        }
    }
}

// ---- F3: LN2 + point-wise feed-forward + residual + mask -----------------------------------------------
// Pad-tolerant, branch-free element code: tiles have pitch 66 and ALL 64 columns are computed and written to
// LDS unconditionally -- weights, biases and inputs are zero beyond D, so pad columns come out as exact zeros
// (relu(0) = 0, dropout(0) = 0) without a single per-lane condition.  Global traffic goes through the
// wave-contiguous row streams, which know the valid row count.
#define F3_P 66
__device__ __forceinline__ void ln_rows_fast(const float* Xs, float* Ys, const float* gam, const float* bet, int D, int wave) {
    // LayerNorm (modules.py:74-78) of the wave's 16 rows; gam/bet are zero-padded LDS arrays of 64 floats
    const int lane = threadIdx.x & 63, l = lane & 15, sub = lane >> 4;
    const float invD = 1.0f / (float)D;
    float g[4], b[4], in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[i] = gam[l + 16 * i]; b[i] = bet[l + 16 * i]; in[i] = (l + 16 * i < D) ? 1.0f : 0.0f; }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = 16 * wave + 4 * p + sub;
        float x[4], s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = Xs[r * F3_P + l + 16 * i]; s += x[i]; }        // pad columns hold 0
        const float mean = sum16(s) * invD;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = (x[i] - mean) * in[i]; v += x[i] * x[i]; }
        const float rs = 1.0f / sqrtf(sum16(v) * invD + 1e-8f);
#pragma unroll
        for (int i = 0; i < 4; ++i) Ys[r * F3_P + l + 16 * i] = g[i] * (x[i] * rs) + b[i];  // pad: 0*.. + 0 = 0
    }
}

__global__ __launch_bounds__(256) void k_block_ln_ffn_fwd(cr_block_desc d, BlockGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = d.D;
    float* Os = smem;                       // [64][66] input tile, reused for the hidden tile and the output tile
    float* Fs = Os + 64 * F3_P;             // [64][66]
    float* W1s = Fs + 64 * F3_P;            // [4*ks][PW]
    float* W2s = W1s + 4 * g.ks * BK_PW;    // [4*ks][PW]
    float* vec = W2s + 4 * g.ks * BK_PW;    // 4 x [64]: gamma2, beta2, b1, b2 (zero padded)
    float* msk = vec + 256;                 // [64] row mask (sasrec.py:83)
    float* Hs = Os;
    const int m0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int mw = m0 + 16 * wave;                                  // first row of this wave
    const int nr = max(0, min(16, d.M - mw));                       // its valid rows
    for (int e = lane; e < 16 * F3_P; e += 64) Os[16 * wave * F3_P + e] = 0.0f;
    if (nr > 0) wave_load_rows(Os + 16 * wave * F3_P, d.o + (size_t)mw * D, F3_P, D, nr, g.invD);
    load_w(W1s, BK_PW, d.w1, D, 0, D, 4 * g.ks);
    load_w(W2s, BK_PW, d.w2, D, 0, D, 4 * g.ks);
    {
        const int t = threadIdx.x, c = t & 63, which = t >> 6;
        const float* src = which == 0 ? d.ln2_g : (which == 1 ? d.ln2_b : (which == 2 ? d.b1 : d.b2));
        vec[t] = (c < D) ? src[c] : 0.0f;
        if (t < 64) msk[t] = (m0 + t < d.M && d.mask_ids[m0 + t] != 0) ? 1.0f : 0.0f;
    }
    __syncthreads();                        // the only barrier
    ln_rows_fast(Os, Fs, vec, vec + 64, D, wave);                                           // sasrec.py:81
    if (nr > 0) wave_store_rows(d.f_in + (size_t)mw * D, Fs + 16 * wave * F3_P, F3_P, D, nr, g.invD);
    const DropCtx d1 = drop_ctx(d.drop_ffn1), d2 = drop_ctx(d.drop_ffn2);
    // per-lane hashing bases: idx = (row_offset + m) * D + col
    uint32_t rb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rb[r] = (d.drop_ffn1.row_offset + (uint32_t)(mw + 4 * lg + r)) * (uint32_t)D + (uint32_t)li;
    f32x4 acc[4];
    zero_acc(acc);
    tile_mma(acc, Fs, F3_P, W1s, g.ks, wave);                                               // modules.py:300-302
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float bias = vec[128 + 16 * j + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = fmaxf(acc[j][r] + bias, 0.0f);
            if (d1.on) v *= (cr_fmix32((rb[r] + 16u * j) * 0x9E3779B1u + d1.key) >= d1.thresh) ? d1.scale : 0.0f;   // modules.py:303-304
            Hs[(16 * wave + 4 * lg + r) * F3_P + 16 * j + li] = v;
        }
    }
    if (nr > 0) wave_store_rows(d.hid + (size_t)mw * D, Hs + 16 * wave * F3_P, F3_P, D, nr, g.invD);
    zero_acc(acc);
    tile_mma(acc, Hs, F3_P, W2s, g.ks, wave);                                               // modules.py:306-308
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float bias = vec[192 + 16 * j + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + 4 * lg + r;
            float v = acc[j][r] + bias;
            if (d2.on) v *= (cr_fmix32((rb[r] + 16u * j) * 0x9E3779B1u + d2.key) >= d2.thresh) ? d2.scale : 0.0f;   // modules.py:309-310
            v = (v + Fs[row * F3_P + 16 * j + li]) * msk[row];                               // modules.py:313, sasrec.py:83
            Hs[row * F3_P + 16 * j + li] = v;              // the MFMAs above have consumed the wave's Hs rows
        }
    }
    if (nr > 0) wave_store_rows(d.y + (size_t)mw * D, Hs + 16 * wave * F3_P, F3_P, D, nr, g.invD);
}

// =====================================================================================================
// backward
// =====================================================================================================
// the wave's 16 rows of a [M, ld] matrix -> its rows of a tile (no block barrier needed)
__device__ __forceinline__ void load_rows_wave(float* dst, int P, const float* src, int ld, int c0, int m0, int m_end, int D, int wave) {
    const int lane = threadIdx.x & 63;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + 16 * wave + i;
        v[i] = (m < m_end && lane < D) ? src[(size_t)m * ld + c0 + lane] : 0.0f;
    }
    if (lane < P) {
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[(16 * wave + i) * P + lane] = v[i];
    }
}

// strip of a [D,D] weight gradient held as (wave, lg, r) x (j, li) accumulators -> slab (row pitch ldw);
// with the ones-column trick accumulator row `ones` (== D) is the bias gradient
__device__ __forceinline__ void store_wgrad(float* dst, int ldw, float* bias_dst, const f32x4 (&accw)[4], int D, int ones, int wave) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = 16 * j + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * wave + 4 * lg + r;
            if (col < D) {
                if (k < D) dst[(size_t)k * ldw + col] = accw[j][r];
                else if (k == ones) bias_dst[col] = accw[j][r];
            }
        }
    }
}

__device__ __forceinline__ float colsum64(const float* Ts, int P) {     // D == 64 fallback: thread tid < 64 sums column tid
    float s = 0.0f;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) s += Ts[r * P + threadIdx.x];
    return s;
}

// LayerNorm backward on the wave's 16 rows (row layout, 16 lanes per row): x from Xs, incoming gradient
// from DYs, optional extra addend ADs (already-computed part of dx).  Accumulates dgamma / dbeta partials.
__device__ __forceinline__ void ln_bwd_rows(const float* Xs, const float* DYs, const float* ADs, int P, const float (&g)[4],
                                            float (&ag)[4], float (&ab)[4], float* dxg, int accumulate, int m0, int m_end,
                                            int D, int wave) {
    const int lane = threadIdx.x & 63, l = lane & 15, sub = lane >> 4;
    const float invD = 1.0f / (float)D;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = 16 * wave + 4 * p + sub, m = m0 + r;
        float x[4], dy[4], s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = l + 16 * i < D;
            x[i] = in ? Xs[r * P + l + 16 * i] : 0.0f;
            dy[i] = in ? DYs[r * P + l + 16 * i] : 0.0f;
            s += x[i];
        }
        const float mean = sum16(s) * invD;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float dx = (l + 16 * i < D) ? x[i] - mean : 0.0f; v += dx * dx; }
        const float rstd = 1.0f / sqrtf(sum16(v) * invD + 1e-8f);
        float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xh = (l + 16 * i < D) ? (x[i] - mean) * rstd : 0.0f;
            x[i] = xh;
            const float dg = dy[i] * g[i];
            c1 += dg; c2 += dg * xh;
            if (m < m_end) { ag[i] += dy[i] * xh; ab[i] += dy[i]; }
        }
        c1 = sum16(c1) * invD;
        c2 = sum16(c2) * invD;
        if (m < m_end) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = l + 16 * i;
                if (c < D) {
                    float dx = rstd * (dy[i] * g[i] - c1 - x[i] * c2);
                    if (ADs) dx += ADs[r * P + c];
                    float* pp = dxg + (size_t)m * D + c;
                    *pp = accumulate ? (*pp + dx) : dx;
                }
            }
        }
    }
}

// fold the per-lane LayerNorm partials (4 row groups per wave via shuffles, then the 4 waves through
// per-wave LDS slots summed in a fixed order: bitwise reproducible) and write the slab entries
__device__ __forceinline__ void store_ln_grads(float* sg, float* sb, float (&ag)[4], float (&ab)[4], float* dg, float* db, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, sub = lane >> 4;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ag[i] += __shfl_xor(ag[i], 16, 64); ag[i] += __shfl_xor(ag[i], 32, 64);
        ab[i] += __shfl_xor(ab[i], 16, 64); ab[i] += __shfl_xor(ab[i], 32, 64);
        const int c = l + 16 * i;
        if (sub == 0) { sg[wave * 64 + c] = ag[i]; sb[wave * 64 + c] = ab[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        dg[c] = (sg[c] + sg[64 + c]) + (sg[128 + c] + sg[192 + c]);
        db[c] = (sb[c] + sb[64 + c]) + (sb[128 + c] + sb[192 + c]);
    }
}

__device__ __forceinline__ int rows_per_wg(int M, int nwg) {       // multiple of the 64-row tile
    const int rps = (M + nwg - 1) / nwg;
    return (rps + 63) / 64 * 64;
}

// B3 staging: gradient wrt the FFN2 pre-dropout output, g2 = dy * mask * keep/(1-rate) (sasrec.py:83, modules.py:309-310)
__device__ __forceinline__ void fetch_g2(float (&v)[16], const cr_block_bwd_desc& bd, const DropCtx& d2, int m0, int m_end) {
    const cr_block_desc& d = bd.f;
    const int tr = threadIdx.x >> 6, tc = threadIdx.x & 63, D = d.D;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + tr + 4 * i;
        float x = 0.0f;
        if (m < m_end && tc < D) {
            x = bd.dy[(size_t)m * D + tc];
            if (d.mask_ids[m] == 0) x = 0.0f;
            x = drop_apply(d2, (d.drop_ffn2.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)tc, x);
        }
        v[i] = x;
    }
}

// ---- B3: backward of LN2 + FFN --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_block_ln_ffn_bwd(cr_block_bwd_desc bd, BlockGeom gm) {
    const cr_block_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = gm.P, D = d.D, ks = gm.ks, ones = gm.ones;
    float* T1 = smem;                      // g2                             -> later df
    float* T2 = T1 + 64 * P;               // hid (+ ones column)            -> later g1
    float* T3 = T2 + 64 * P;               // f_in (+ ones column)           -> later o
    float* W1r = T3 + 64 * P;              // W1 [k][P], read by rows
    float* W2r = W1r + 64 * P;             // W2 [k][P]
    float* sg = W2r + 64 * P;              // [4][64] + [4][64]
    float* sb = sg + 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4, l = lane & 15;
    const int rps = rows_per_wg(d.M, gridDim.x);
    const int mb = blockIdx.x * rps, me = min(d.M, mb + rps);
    load_w(W1r, P, d.w1, D, 0, D);
    load_w(W2r, P, d.w2, D, 0, D);
    const DropCtx d2 = drop_ctx(d.drop_ffn2);
    const float scale1 = (d.drop_ffn1.rate > 0.0f) ? 1.0f / (1.0f - d.drop_ffn1.rate) : 1.0f;
    f32x4 aw1[4], aw2[4];
    zero_acc(aw1); zero_acc(aw2);
    float b1s = 0.0f, b2s = 0.0f;
    float g[4], ag[4], ab[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[i] = (l + 16 * i < D) ? d.ln2_g[l + 16 * i] : 0.0f; ag[i] = 0.0f; ab[i] = 0.0f; }
    float p1[16], p2[16], p3[16];                                  // register prefetch of the next tile
    if (mb < me) {
        fetch_g2(p1, bd, d2, mb, me);
        fetch_tile(p2, d.hid, D, 0, mb, me, D);
        fetch_tile(p3, d.f_in, D, 0, mb, me, D);
    }
    for (int m0 = mb; m0 < me; m0 += 64) {
        put_tile(T1, p1, P, -1, m0, me);
        put_tile(T2, p2, P, ones, m0, me);
        put_tile(T3, p3, P, ones, m0, me);
        __syncthreads();
        if (m0 + 64 < me) {                                        // next tile's loads fly under this tile's MFMAs
            fetch_g2(p1, bd, d2, m0 + 64, me);
            fetch_tile(p2, d.hid, D, 0, m0 + 64, me, D);
            fetch_tile(p3, d.f_in, D, 0, m0 + 64, me, D);
        }
        // dW2 (+ db2 in row `ones`) += hid^T g2
        tile_wgrad(aw2, T2, T1, P, wave);
        if (ones < 0 && threadIdx.x < 64) b2s += colsum64(T1, P);
        __syncthreads();
        // dhid = g2 W2^T, gated by the stored post-dropout ReLU output -> g1 (in place over hid, own rows)
        {
            f32x4 acc[4];
            zero_acc(acc);
            tile_mma_t(acc, T1, W2r, P, ks, wave);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * wave + 4 * lg + r, col = 16 * j + li;
                    if (col < P) {
                        const float h = T2[row * P + col];
                        T2[row * P + col] = (h > 0.0f && col < D) ? acc[j][r] * scale1 : 0.0f;           // modules.py:300-304
                    }
                }
        }
        __syncthreads();
        // dW1 (+ db1) += f_in^T g1
        tile_wgrad(aw1, T3, T2, P, wave);
        if (ones < 0 && threadIdx.x < 64) b1s += colsum64(T2, P);
        // df = (g1 W1^T + dy) * mask  (residual of modules.py:313) -> T1 (own rows)
        {
            f32x4 acc[4];
            zero_acc(acc);
            tile_mma_t(acc, T2, W1r, P, ks, wave);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * wave + 4 * lg + r, col = 16 * j + li, m = m0 + row;
                    float v = 0.0f;
                    if (m < me && col < D && d.mask_ids[m] != 0) v = acc[j][r] + bd.dy[(size_t)m * D + col];
                    if (col < P) T1[row * P + col] = v;
                }
        }
        __syncthreads();                                           // all waves are done with every row of T3
        // LN2 backward: x = o (own rows into T3), dy = df
        load_rows_wave(T3, P, d.o, D, 0, m0, me, D, wave);
        ln_bwd_rows(T3, T1, nullptr, P, g, ag, ab, bd.d_o, 0, m0, me, D, wave);
        __syncthreads();
    }
    const size_t so = (size_t)blockIdx.x * bd.slab_stride;
    store_wgrad(bd.g_w1 + so, D, bd.g_b1 + so, aw1, D, ones, wave);
    store_wgrad(bd.g_w2 + so, D, bd.g_b2 + so, aw2, D, ones, wave);
    if (ones < 0 && threadIdx.x < D) { bd.g_b1[so + threadIdx.x] = b1s; bd.g_b2[so + threadIdx.x] = b2s; }
    store_ln_grads(sg, sb, ag, ab, bd.g_ln2_g + so, bd.g_ln2_b + so, D);
}

// ---- B1: backward of LN1 + Q/K/V projections --------------------------------------------------------------
__global__ __launch_bounds__(256) void k_block_ln_qkv_bwd(cr_block_bwd_desc bd, BlockGeom gm) {
    const cr_block_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = gm.P, D = d.D, ks = gm.ks, ones = gm.ones;
    float* TA = smem;                      // dQ                 -> later dq_in
    float* TB = TA + 64 * P;               // dK                 -> later dx part (dK Wk^T + dV Wv^T)
    float* TC = TB + 64 * P;               // dV
    float* TQ = TC + 64 * P;               // q_in (+ ones column)
    float* TX = TQ + 64 * P;               // x    (+ ones column)
    float* Wqr = TX + 64 * P;              // Wq, Wk, Wv as [k][P], read by rows
    float* Wkr = Wqr + 64 * P;
    float* Wvr = Wkr + 64 * P;
    float* sg = Wvr + 64 * P;
    float* sb = sg + 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4, l = lane & 15;
    const int rps = rows_per_wg(d.M, gridDim.x);
    const int mb = blockIdx.x * rps, me = min(d.M, mb + rps);
    load_w(Wqr, P, d.wqkv, 3 * D, 0, D);
    load_w(Wkr, P, d.wqkv, 3 * D, D, D);
    load_w(Wvr, P, d.wqkv, 3 * D, 2 * D, D);
    f32x4 awq[4], awk[4], awv[4];
    zero_acc(awq); zero_acc(awk); zero_acc(awv);
    float bqs = 0.0f, bks = 0.0f, bvs = 0.0f;
    float g[4], ag[4], ab[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[i] = (l + 16 * i < D) ? d.ln1_g[l + 16 * i] : 0.0f; ag[i] = 0.0f; ab[i] = 0.0f; }
    for (int m0 = mb; m0 < me; m0 += 64) {
        {   // one burst of loads for the five tiles, then one barrier
            float va[16], vb[16], vc[16], vq[16], vx[16];
            fetch_tile(va, bd.dqkv, 3 * D, 0, m0, me, D);
            fetch_tile(vb, bd.dqkv, 3 * D, D, m0, me, D);
            fetch_tile(vc, bd.dqkv, 3 * D, 2 * D, m0, me, D);
            fetch_tile(vq, d.q_in, D, 0, m0, me, D);
            fetch_tile(vx, d.x, D, 0, m0, me, D);
            put_tile(TA, va, P, -1, m0, me);
            put_tile(TB, vb, P, -1, m0, me);
            put_tile(TC, vc, P, -1, m0, me);
            put_tile(TQ, vq, P, ones, m0, me);
            put_tile(TX, vx, P, ones, m0, me);
        }
        __syncthreads();
        // weight (+ bias) gradients: reductions over all 64 rows of the tiles
        tile_wgrad(awq, TQ, TA, P, wave);
        tile_wgrad(awk, TX, TB, P, wave);
        tile_wgrad(awv, TX, TC, P, wave);
        if (ones < 0 && threadIdx.x < 64) { bqs += colsum64(TA, P); bks += colsum64(TB, P); bvs += colsum64(TC, P); }
        // data gradients on own rows: dx_part = dK Wk^T + dV Wv^T ; dq_in = dQ Wq^T + d_o (modules.py:269)
        f32x4 dxa[4], acc[4];
        zero_acc(dxa); zero_acc(acc);
        tile_mma_t(dxa, TB, Wkr, P, ks, wave);
        tile_mma_t(dxa, TC, Wvr, P, ks, wave);
        tile_mma_t(acc, TA, Wqr, P, ks, wave);
        __syncthreads();                                           // every wave is done reading all rows of TA / TB
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * wave + 4 * lg + r, col = 16 * j + li, m = m0 + row;
                if (col < P) {
                    float v = 0.0f;
                    if (m < me && col < D) v = acc[j][r] + bd.d_o[(size_t)m * D + col];
                    TA[row * P + col] = v;
                    TB[row * P + col] = dxa[j][r];
                }
            }
        // LN1 backward on own rows: dx = dx_part + LNbwd(dq_in; x)
        ln_bwd_rows(TX, TA, TB, P, g, ag, ab, bd.dx, bd.dx_accumulate, m0, me, D, wave);
        __syncthreads();
    }
    const size_t so = (size_t)blockIdx.x * bd.slab_stride;
    store_wgrad(bd.g_wqkv + so, 3 * D, bd.g_bqkv + so, awq, D, ones, wave);
    store_wgrad(bd.g_wqkv + so + D, 3 * D, bd.g_bqkv + so + D, awk, D, ones, wave);
    store_wgrad(bd.g_wqkv + so + 2 * D, 3 * D, bd.g_bqkv + so + 2 * D, awv, D, ones, wave);
    if (ones < 0 && threadIdx.x < D) {
        bd.g_bqkv[so + threadIdx.x] = bqs;
        bd.g_bqkv[so + D + threadIdx.x] = bks;
        bd.g_bqkv[so + 2 * D + threadIdx.x] = bvs;
    }
    store_ln_grads(sg, sb, ag, ab, bd.g_ln1_g + so, bd.g_ln1_b + so, D);
}

// ---- host side ------------------------------------------------------------------------------------------
static int block_check(const cr_block_desc* d, BlockGeom* g, const char* who) {
    CR_REQUIRE(d != nullptr, "%s: NULL desc", who);
    CR_REQUIRE(d->M > 0 && d->D > 0, "%s: bad shape", who);
    if (d->D > 64) return cr_set_error(CR_ERR_UNSUPPORTED, "%s: D=%d > 64 (use the unfused kernels)", who, d->D);
    g->ks = (d->D + 3) / 4;
    g->P = 4 * g->ks + 2;
    g->ones = d->D < 64 ? d->D : -1;
    g->invD = (uint32_t)(4294967296.0 / d->D) + 1u;
    const char* e = getenv("CR_BLOCK_DBG");
    g->dbg = e ? atoi(e) : 0;
    return CR_OK;
}

static int block_lds_attr(const void* fn, bool* done) {
    if (!*done) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        *done = true;
    }
    return CR_OK;
}

extern "C" int cr_block_ln_qkv_fwd(const cr_block_desc* d, void* stream) {
    BlockGeom g;
    int rc = block_check(d, &g, "cr_block_ln_qkv_fwd");
    if (rc) return rc;
    CR_REQUIRE(d->x && d->q_in && d->qkv && d->k_valid && d->q_valid && d->ln1_g && d->ln1_b && d->wqkv && d->bqkv,
               "cr_block_ln_qkv_fwd: NULL pointer");
    static bool attr = false;
    rc = block_lds_attr(reinterpret_cast<const void*>(&k_block_ln_qkv_fwd), &attr);
    if (rc) return rc;
    const size_t lds = sizeof(float) * (2 * 64 * g.P + 3 * 4 * g.ks * BK_PW);
    hipLaunchKernelGGL(k_block_ln_qkv_fwd, dim3(cr_ceil_div(d->M, 64)), dim3(256), lds, cr_stream(stream), *d, g);
    return cr_check_launch("cr_block_ln_qkv_fwd");
}

extern "C" int cr_block_ln_ffn_fwd(const cr_block_desc* d, void* stream) {
    BlockGeom g;
    int rc = block_check(d, &g, "cr_block_ln_ffn_fwd");
    if (rc) return rc;
    CR_REQUIRE(d->o && d->f_in && d->hid && d->y && d->mask_ids && d->ln2_g && d->ln2_b && d->w1 && d->b1 && d->w2 && d->b2,
               "cr_block_ln_ffn_fwd: NULL pointer");
    static bool attr = false;
    rc = block_lds_attr(reinterpret_cast<const void*>(&k_block_ln_ffn_fwd), &attr);
    if (rc) return rc;
    const size_t lds = sizeof(float) * (2 * 64 * F3_P + 2 * 4 * g.ks * BK_PW + 256 + 64);
    hipLaunchKernelGGL(k_block_ln_ffn_fwd, dim3(cr_ceil_div(d->M, 64)), dim3(256), lds, cr_stream(stream), *d, g);
    return cr_check_launch("cr_block_ln_ffn_fwd");
}

extern "C" int cr_block_ln_ffn_bwd(const cr_block_bwd_desc* bd, void* stream) {
    CR_REQUIRE(bd != nullptr, "cr_block_ln_ffn_bwd: NULL desc");
    BlockGeom g;
    int rc = block_check(&bd->f, &g, "cr_block_ln_ffn_bwd");
    if (rc) return rc;
    const cr_block_desc* d = &bd->f;
    CR_REQUIRE(bd->dy && bd->d_o && d->hid && d->f_in && d->o && d->mask_ids && d->w1 && d->w2 && d->ln2_g, "cr_block_ln_ffn_bwd: NULL pointer");
    CR_REQUIRE(bd->g_w1 && bd->g_b1 && bd->g_w2 && bd->g_b2 && bd->g_ln2_g && bd->g_ln2_b && bd->n_slabs > 0, "cr_block_ln_ffn_bwd: NULL gradient pointer");
    static bool attr = false;
    rc = block_lds_attr(reinterpret_cast<const void*>(&k_block_ln_ffn_bwd), &attr);
    if (rc) return rc;
    const size_t lds = sizeof(float) * (5 * 64 * g.P + 512);
    hipLaunchKernelGGL(k_block_ln_ffn_bwd, dim3(bd->n_slabs), dim3(256), lds, cr_stream(stream), *bd, g);
    return cr_check_launch("cr_block_ln_ffn_bwd");
}

extern "C" int cr_block_ln_qkv_bwd(const cr_block_bwd_desc* bd, void* stream) {
    CR_REQUIRE(bd != nullptr, "cr_block_ln_qkv_bwd: NULL desc");
    BlockGeom g;
    int rc = block_check(&bd->f, &g, "cr_block_ln_qkv_bwd");
    if (rc) return rc;
    const cr_block_desc* d = &bd->f;
    CR_REQUIRE(bd->dqkv && bd->d_o && bd->dx && d->q_in && d->x && d->wqkv && d->ln1_g, "cr_block_ln_qkv_bwd: NULL pointer");
    CR_REQUIRE(bd->g_wqkv && bd->g_bqkv && bd->g_ln1_g && bd->g_ln1_b && bd->n_slabs > 0, "cr_block_ln_qkv_bwd: NULL gradient pointer");
    static bool attr = false;
    rc = block_lds_attr(reinterpret_cast<const void*>(&k_block_ln_qkv_bwd), &attr);
    if (rc) return rc;
    const size_t lds = sizeof(float) * (8 * 64 * g.P + 512);
    hipLaunchKernelGGL(k_block_ln_qkv_bwd, dim3(bd->n_slabs), dim3(256), lds, cr_stream(stream), *bd, g);
    return cr_check_launch("cr_block_ln_qkv_bwd");
}
