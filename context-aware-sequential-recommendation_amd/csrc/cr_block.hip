// Fused row-phase kernels of one transformer block for hidden sizes D <= 64 (sasrec.py:65-83 minus the
// attention core): a 64-row tile of activations stays in LDS from LayerNorm through the projections,
// so each activation is read from HBM/L2 once per phase and ~10 small launches per block disappear.
//
// Tile shapes: a group of 4 waves owns a 64-row tile (wave w its rows [16w, 16w+16)), all D (<= 64) columns;
// the forward kernels run one group per workgroup, the backward kernels two (512 threads, shared weights).
// MFMA v_mfma_f32_16x16x4_f32.  Row tiles have pitch 66 (all 64 columns are computed and written, pad columns
// are exact zeros: no per-lane conditions in the element code); weights live in LDS as packed conflict-free
// images (see BK_WROW), the backward kernels use the packed TRANSPOSED images so g @ W^T is the same MFMA loop.
//
// Latency structure (profiles/, tools/block_ts.py): forward kernels stage the row tile and ALL their weights with
// one burst of 16-byte loads and one barrier -- after it every wave works only on rows it owns.  The backward
// kernels accumulate the weight gradients in MFMA accumulators across tiles and obtain the bias gradients for
// free by planting a column of ones in the A-tile of the weight-gradient MFMA (D < 64).
#include <stdlib.h>

#include "cr_common.hpp"

// Column-read weights (B operand of x @ W) are stored PACKED: element (k, n) at
//   (k >> 1) * 128 + (n >> 4) * 32 + (k & 1) * 16 + (n & 15)
// so the 32 lanes of a ds_read_b32 half (li = n & 15, lg in {0,1} or {2,3}: k = 4kk + lg) hit 32 distinct banks
// with no padding at all: 64 floats per k-row instead of the 80-float pitch a plain row layout needs.  At D = 50
// that takes the QKV kernel from 85 KB to 75 KB of LDS -- two workgroups per CU instead of one, which the
// per-wave timeline (tools/block_ts.py) showed to be the difference between one and two serial rounds.
#define BK_WROW 64
#define F3_P 66          // row-tile pitch of the pad-tolerant kernels (64 columns + 2, % 4 == 2)
__device__ __forceinline__ int bk_waddr(int k, int n) { return (k >> 1) * 128 + (n >> 4) * 32 + (k & 1) * 16 + (n & 15); }

struct BlockGeom {
    int P;        // row-tile pitch
    int ks;       // k-steps of 4 covering D
    int ones;     // column holding 1.0 for the bias-gradient trick, or -1 (D == 64)
    int dbg;      // timing-only ablation switches (env CR_BLOCK_DBG); 0 in production
    uint32_t invD;  // floor(2^32 / D) + 1: e / D == umulhi(e, invD) for e < 2^16
    unsigned long long* ts;   // debug: per-wave phase timestamps [n_wg][4 waves][16] (tools/block_ts.py); NULL in production
};

#ifdef CR_TIMELINE
// debug-only phase stamps: slots 0 / 15 = wall clock (100 MHz, comparable across the chip), others = s_memtime
#define BK_TSG(geom, slot)                                                                                   \
    do {                                                                                                     \
        if ((geom).ts && (threadIdx.x & 63) == 0)                                                            \
            (geom).ts[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] =        \
                ((slot) == 0 || (slot) == 15) ? wall_clock64() : clock64();                                  \
    } while (0)
#define BK_TS(slot) BK_TSG(g, slot)
static unsigned long long* g_block_ts = nullptr;
extern "C" void cr_debug_block_ts(void* p) { g_block_ts = static_cast<unsigned long long*>(p); }
#else
#define BK_TSG(geom, slot) do { } while (0)
#define BK_TS(slot) do { } while (0)
static unsigned long long* const g_block_ts = nullptr;
#endif

// Weight [K=D rows][N=D cols] (row pitch ldw, column offset c0) -> packed LDS image of 4*ks k-rows x 64, zero padded.
// The vector-memory pipe of a CU retires one wave-instruction per ~16 clocks whatever its width (per-wave
// timelines, tools/block_ts.py), so the weights are fetched as 16-byte chunks -- 4 loads per thread instead of
// 16 dword loads; gfx950 global loads need only dword alignment.  Item (k, q) = row k, columns [4q, 4q+4);
// a chunk that would cross column D is read shifted back to [D-4, D) and rotated, so nothing outside the
// weight is touched.  Two halves: fetch_w (loads) ... put_w_packed / put_wt_packed (LDS writes); the latter
// stores the TRANSPOSED weight, B(k = n, col = k') = W[k'][n], which makes g @ W^T a plain tile_mma.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
// NT = threads of the workgroup (256 or 512): 1024 items, 1024 / NT per thread
template <int NT> struct WFragT { f4u v[1024 / NT]; };
typedef WFragT<256> WFrag;
// Item <-> (k, q) mapping.  KFAST = false: consecutive lanes take consecutive chunks q of one row (coalesced rows;
// right for the plain image, whose 16-byte LDS writes then differ in q).  KFAST = true: consecutive lanes take
// consecutive ROWS k -- what the TRANSPOSED image needs: its element (n, k) lands on LDS bank (k & 15) + 16 (n & 1),
// so lanes that differ only in q would all hit one bank (16-way conflicts: 2.8 us of LDS time per workgroup at the
// start of the backward kernels); the loads are 16 bytes from 64 different rows, cheap for an L2-resident 10 KB weight.
template <bool KFAST>
__device__ __forceinline__ void w_item(int item, int& k, int& q) {
    if (KFAST) { k = item & 63; q = item >> 6; } else { k = item >> 4; q = item & 15; }
}
template <int NT, bool KFAST = false>
__device__ __forceinline__ void fetch_w(WFragT<NT>& w, const float* W, int ldw, int c0, int D) {
#pragma unroll
    for (int it = 0; it < 1024 / NT; ++it) {
        int k, q;
        w_item<KFAST>(threadIdx.x + NT * it, k, q);
        const bool valid = (k < D) && (4 * q < D);
        const int col = valid ? min(4 * q, D - 4) : 0;
        w.v[it] = *reinterpret_cast<const f4u*>(W + (size_t)(valid ? k : 0) * ldw + c0 + col);
    }
}
// element t of item `it` after the zero padding / back-shift fix-up
template <int NT, bool KFAST = false>
__device__ __forceinline__ void wfrag_item(const WFragT<NT>& w, int it, int D, float (&e)[4]) {
    int k, q;
    w_item<KFAST>(threadIdx.x + NT * it, k, q);
    const bool valid = (k < D) && (4 * q < D);
    const int shift = valid ? 4 * q - min(4 * q, D - 4) : 0;
    const float x0 = w.v[it].x, x1 = w.v[it].y, x2 = w.v[it].z, x3 = w.v[it].w;
    const float r0 = shift == 0 ? x0 : (shift == 1 ? x1 : (shift == 2 ? x2 : x3));
    const float r1 = shift == 0 ? x1 : (shift == 1 ? x2 : (shift == 2 ? x3 : 0.0f));
    const float r2 = shift == 0 ? x2 : (shift == 1 ? x3 : 0.0f);
    const float r3 = shift == 0 ? x3 : 0.0f;
    e[0] = (valid && 4 * q + 0 < D) ? r0 : 0.0f;
    e[1] = (valid && 4 * q + 1 < D) ? r1 : 0.0f;
    e[2] = (valid && 4 * q + 2 < D) ? r2 : 0.0f;
    e[3] = (valid && 4 * q + 3 < D) ? r3 : 0.0f;
}
template <int NT>
__device__ __forceinline__ void put_w_packed(float* Ws, const WFragT<NT>& w, int D, int nrows) {
#pragma unroll
    for (int it = 0; it < 1024 / NT; ++it) {
        const int item = threadIdx.x + NT * it, k = item >> 4, q = item & 15;
        float e[4];
        wfrag_item(w, it, D, e);
        if (k < nrows) *reinterpret_cast<float4*>(Ws + bk_waddr(k, 4 * q)) = make_float4(e[0], e[1], e[2], e[3]);
    }
}
template <int NT>
__device__ __forceinline__ void put_wt_packed(float* Ws, const WFragT<NT>& w, int D, int nrows) {     // pair with fetch_w<NT, true>
#pragma unroll
    for (int it = 0; it < 1024 / NT; ++it) {
        int k, q;
        w_item<true>(threadIdx.x + NT * it, k, q);
        float e[4];
        wfrag_item<NT, true>(w, it, D, e);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * q + t < nrows) Ws[bk_waddr(4 * q + t, k)] = e[t];
    }
}
__device__ __forceinline__ void load_w_packed(float* Ws, const float* W, int ldw, int c0, int D, int rows) {
    WFrag w;
    fetch_w(w, W, ldw, c0, D);
    put_w_packed(Ws, w, D, rows);
}

// acc[j] (+)= As[rows 16w..][k] * W[k][16j..]   (x @ W), ks k-steps, W in the packed layout.
// The operands of CH k-steps (CH A values + 4*CH B values) are read as ONE batch with clamped addresses, then
// the MFMAs of those steps issue back to back behind wave-uniform guards.  (The plain loop compiled to
// read -> wait -> 2 MFMAs -> read -> wait -> 2 MFMAs: two exposed LDS latencies per k-step, 2.3x the MFMA time.)
template <int CH = 8>
__device__ __forceinline__ void tile_mma(f32x4 (&acc)[4], const float* As, int P, const float* Ws, int ks, int wave) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const float* ap = As + (16 * wave + li) * P + lg;
    const float* bp = Ws + (lg >> 1) * 128 + (lg & 1) * 16 + li;
#pragma unroll 1
    for (int k0 = 0; k0 < ks; k0 += CH) {
        float a[CH], b[CH][4];
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int k = min(k0 + s, ks - 1);
            a[s] = ap[4 * k];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[s][j] = bp[k * 256 + 32 * j];
        }
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            if (k0 + s < ks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = mfma16(a[s], b[s][j], acc[j]);
            }
        }
    }
}

// accw[j] += sum_m As[m][16w + li] * Gs[m][16j + li] over the 64 rows of a tile (A^T G: weight-gradient strip of
// k-rows [16w, 16w+16) owned by wave w; with a ones column in As, row `ones` is the bias gradient).  Operand reads
// batched CH steps at a time like tile_mma.
// Weight-gradient strip over the tiles of NGRP tile groups at once (group g's tiles at + g * gstride): wave `wave`
// owns k-rows [16 wave, 16 wave + 16) and the NJ n-tiles starting at n-tile NJ * half, summed over ALL rows of all
// groups.  With two groups per workgroup (NJ = 2: each group's waves take one half of the columns) every accumulator is
// final for its (k-strip, column-half) -- no cross-group fold through LDS, half the accumulator registers.
template <int NJ, int NGRP, int CH = 4>
__device__ __forceinline__ void tile_wgrad_g(f32x4 (&accw)[NJ], const float* As0, const float* Gs0, int gstride, int P,
                                             int wave, int half) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int gi = 0; gi < NGRP; ++gi) {
        const float* ap = As0 + gi * gstride + lg * P + 16 * wave + li;
        const float* gp = Gs0 + gi * gstride + lg * P + 16 * NJ * half + li;
#pragma unroll 1
        for (int m0 = 0; m0 < 16; m0 += CH) {
            float a[CH], b[CH][NJ];
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                a[s] = ap[4 * (m0 + s) * P];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[s][j] = gp[4 * (m0 + s) * P + 16 * j];
            }
#pragma unroll
            for (int s = 0; s < CH; ++s)
#pragma unroll
                for (int j = 0; j < NJ; ++j) accw[j] = mfma16(a[s], b[s][j], accw[j]);
        }
    }
}
template <int NJ>
__device__ __forceinline__ void zero_acc_n(f32x4 (&acc)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// Row I/O of a wave's 16-row strip of a DENSE [M, D] matrix (ld == D) <-> its rows of a pitch-66 LDS tile.
// An item is (row r, 4-column chunk q): lane -> q = lane & 15, r = (lane >> 4) + 4 i, i = 0..3, so one wave
// instruction moves four 4*D-byte rows (contiguous in memory) as 16-byte pieces, dword aligned (all gfx950
// needs), and the address arithmetic is a multiply-add per 4 elements.  The chunk that crosses column D is
// accessed shifted back to [D-4, D) and rotated (D >= 4), so nothing outside the rows is touched; on the LDS
// side ALL 64 columns of all 16 rows are written (zeros outside the valid rows / columns).
// (The first version walked a flat element stream and recovered (row, col) per element: 5x the VALU work and
//  a branch per element, which the instruction census showed to dominate these kernels.)
typedef float f4r __attribute__((ext_vector_type(4), aligned(4)));
struct Stream4 { f4r v[4]; };
__device__ __forceinline__ void stream_fetch(Stream4& s, const float* gsrc, int D, int total) {
    const int lane = threadIdx.x & 63, q = lane & 15, r0 = lane >> 4;
    const int col0 = min(4 * q, D - 4);
    if (total <= 0) {                                   // wave-uniform: a strip beyond the matrix must not be touched at all
#pragma unroll
        for (int i = 0; i < 4; ++i) s.v[i] = (f4r){0.f, 0.f, 0.f, 0.f};
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 4 * i;
        const bool valid = (r * D < total) && (4 * q < D);
        s.v[i] = *reinterpret_cast<const f4r*>(gsrc + (valid ? r * D + col0 : 0));
    }
}
// place the strip into the wave's rows of a pitch-66 tile; fn(e, r, v) maps (flat element e = r*D + c, row r) -> value
template <class F>
__device__ __forceinline__ void stream_put(float* Ts, const Stream4& s, int D, int total, uint32_t, F fn) {
    const int lane = threadIdx.x & 63, q = lane & 15, r0 = lane >> 4;
    const int shift = 4 * q - min(4 * q, D - 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 4 * i;
        const bool valid = (r * D < total) && (4 * q < D);
        const float x0 = s.v[i].x, x1 = s.v[i].y, x2 = s.v[i].z, x3 = s.v[i].w;
        float e[4];
        e[0] = shift == 0 ? x0 : (shift == 1 ? x1 : (shift == 2 ? x2 : x3));
        e[1] = shift == 0 ? x1 : (shift == 1 ? x2 : (shift == 2 ? x3 : 0.0f));
        e[2] = shift == 0 ? x2 : (shift == 1 ? x3 : 0.0f);
        e[3] = shift == 0 ? x3 : 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) e[t] = (valid && 4 * q + t < D) ? fn(r * D + 4 * q + t, r, e[t]) : 0.0f;
        float2* pt = reinterpret_cast<float2*>(Ts + r * F3_P + 4 * q);
        pt[0] = make_float2(e[0], e[1]);
        pt[1] = make_float2(e[2], e[3]);
    }
}
struct PutPlain { __device__ __forceinline__ float operator()(int, int, float v) const { return v; } };
// Ts[valid elements] += strip
__device__ __forceinline__ void stream_add(float* Ts, const Stream4& s, int D, int total) {
    const int lane = threadIdx.x & 63, q = lane & 15, r0 = lane >> 4;
    const int shift = 4 * q - min(4 * q, D - 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 4 * i;
        const bool valid = (r * D < total) && (4 * q < D);
        const float x0 = s.v[i].x, x1 = s.v[i].y, x2 = s.v[i].z, x3 = s.v[i].w;
        float e[4];
        e[0] = shift == 0 ? x0 : (shift == 1 ? x1 : (shift == 2 ? x2 : x3));
        e[1] = shift == 0 ? x1 : (shift == 1 ? x2 : (shift == 2 ? x3 : 0.0f));
        e[2] = shift == 0 ? x2 : (shift == 1 ? x3 : 0.0f);
        e[3] = shift == 0 ? x3 : 0.0f;
        float2* pt = reinterpret_cast<float2*>(Ts + r * F3_P + 4 * q);
        float2 a = pt[0], b = pt[1];
        a.x += (valid && 4 * q + 0 < D) ? e[0] : 0.0f;
        a.y += (valid && 4 * q + 1 < D) ? e[1] : 0.0f;
        b.x += (valid && 4 * q + 2 < D) ? e[2] : 0.0f;
        b.y += (valid && 4 * q + 3 < D) ? e[3] : 0.0f;
        pt[0] = a;
        pt[1] = b;
    }
}
__device__ __forceinline__ void wave_load_rows(float* Ts, const float* gsrc, int, int D, int nrows, uint32_t invD) {
    Stream4 s;
    stream_fetch(s, gsrc, D, nrows * D);
    stream_put(Ts, s, D, nrows * D, invD, PutPlain());
}
__device__ __forceinline__ void wave_store_rows(float* gdst, const float* Ts, int, int D, int nrows, uint32_t) {
    const int lane = threadIdx.x & 63, q = lane & 15, r0 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 4 * i;
        if (r < nrows && 4 * q < D) {
            const float2* pt = reinterpret_cast<const float2*>(Ts + r * F3_P + 4 * q);
            const float2 a = pt[0], b = pt[1];
            float* gp = gdst + r * D + 4 * q;
            if (4 * q + 3 < D) {
                *reinterpret_cast<f4r*>(gp) = (f4r){a.x, a.y, b.x, b.y};
            } else {
                gp[0] = a.x;
                if (4 * q + 1 < D) gp[1] = a.y;
                if (4 * q + 2 < D) gp[2] = b.x;
            }
        }
    }
}

// The wave's 16-row strip of x built in place from the embedding recipe of cr_embed_fwd (same arithmetic, same
// dropout counters): x[m] = mask * dropout(table'[id[m]] * scale + pos[m % T] + addend[m]).  Row-chunk items as in
// stream_fetch; the strip is returned UNROTATED (lane's vector = columns [col0, col0 + 4)), ready for stream_put,
// and written to the dense x matrix on the way (the backward kernels and the residual read it).
__device__ __forceinline__ void gather_issue_ids(const cr_embed_desc& e, int mw, int nrows, int (&id)[4], int (&mk)[4]) {
    const int r0 = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mw + min(r0 + 4 * i, max(nrows - 1, 0));
        id[i] = (nrows > 0) ? e.ids[m] : 0;
        mk[i] = (nrows > 0 && e.mask_ids) ? e.mask_ids[m] : 1;
    }
}
__device__ __forceinline__ void gather_rows(Stream4& s, const cr_embed_desc& e, const int (&id)[4], const int (&mk)[4],
                                            int mw, int nrows, int D) {
    const int lane = threadIdx.x & 63, q = lane & 15, r0 = lane >> 4;
    const int col0 = min(4 * q, D - 4);
    if (nrows <= 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s.v[i] = (f4r){0.f, 0.f, 0.f, 0.f};
        return;
    }
    const DropCtx dc = drop_ctx(e.drop);
    f4r tv[4], pv[4], av[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mw + min(r0 + 4 * i, nrows - 1);
        tv[i] = *reinterpret_cast<const f4r*>(e.table + (size_t)id[i] * D + col0);       // row 0 exists; zeroed below when zero_pad
        pv[i] = (f4r){0.f, 0.f, 0.f, 0.f};
        av[i] = (f4r){0.f, 0.f, 0.f, 0.f};
        if (e.pos_table) pv[i] = *reinterpret_cast<const f4r*>(e.pos_table + (size_t)(m % e.T) * D + col0);
        if (e.addend) av[i] = *reinterpret_cast<const f4r*>(e.addend + (size_t)m * e.ld_add + col0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 4 * i, m = mw + min(r, nrows - 1);
        const bool padrow = e.zero_pad && id[i] == 0;
        const bool dead = mk[i] == 0;
        const uint32_t base = (e.drop.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)col0;
        float y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = (padrow ? 0.0f : tv[i][t]) * e.scale + pv[i][t] + av[i][t];
            v = drop_apply(dc, base + (uint32_t)t, v);
            y[t] = dead ? 0.0f : v;
        }
        s.v[i] = (f4r){y[0], y[1], y[2], y[3]};
        // the chunk that crosses column D was shifted back: it rewrites up to three columns of its left neighbour
        // with the same values
        if (r < nrows && 4 * q < D) *reinterpret_cast<f4r*>(e.out + (size_t)m * D + col0) = s.v[i];
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
// Q, K, V are stored as three dense [M, D] matrices ([3, M, D]) so each wave's 16 output rows of each are one
// contiguous block.  f1_body is everything after the staging barrier: Xs holds the block input rows, Ws the three
// packed weights, vec = gamma1 | beta1 | bq | bk | bv (zero padded); every wave works only on the rows it owns.
__device__ __forceinline__ void f1_body(const cr_block_desc& d, const BlockGeom& g, float* Xs, float* Qs, const float* Ws,
                                        const float* vec, int m0) {
    const int D = d.D;
    const int wsz = 4 * g.ks * BK_WROW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int mw = m0 + 16 * wave;
    const int nr = max(0, min(16, d.M - mw));
    // LN1 with the data-dependent key / query masks (modules.py:222,248-249)
    {
        const int l = lane & 15, sub = lane >> 4;
        const float invD = 1.0f / (float)D;
        float gm[4], bt[4], in[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { gm[i] = vec[l + 16 * i]; bt[i] = vec[64 + l + 16 * i]; in[i] = (l + 16 * i < D) ? 1.0f : 0.0f; }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = 16 * wave + 4 * p + sub, m = m0 + r;
            float x[4], s = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = Xs[r * F3_P + l + 16 * i]; s += x[i]; }
            s = sum16(s);
            const float mean = s * invD;
            float v = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = (x[i] - mean) * in[i]; v += x[i] * x[i]; }
            const float rs = 1.0f / sqrtf(sum16(v) * invD + 1e-8f);
            float ys = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float y = gm[i] * (x[i] * rs) + bt[i]; Qs[r * F3_P + l + 16 * i] = y; ys += y; }
            ys = sum16(ys);
            if (l == 0 && m < d.M) {
                d.k_valid[m] = (s != 0.0f) ? 1.0f : 0.0f;
                d.q_valid[m] = (ys != 0.0f) ? 1.0f : 0.0f;
            }
        }
    }
    if (nr > 0) wave_store_rows(d.q_in + (size_t)mw * D, Qs + 16 * wave * F3_P, F3_P, D, nr, g.invD);
    BK_TS(4);
#pragma unroll 1
    for (int part = 0; part < 3; ++part) {                                                  // modules.py:203-205
        f32x4 acc[4];
        zero_acc(acc);
        tile_mma(acc, part == 0 ? Qs : Xs, F3_P, Ws + part * wsz, g.ks, wave);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float bias = vec[128 + 64 * part + 16 * j + li];
#pragma unroll
            for (int r = 0; r < 4; ++r) Qs[(16 * wave + 4 * lg + r) * F3_P + 16 * j + li] = acc[j][r] + bias;
        }
        if (nr > 0) wave_store_rows(d.qkv + ((size_t)part * d.M + mw) * D, Qs + 16 * wave * F3_P, F3_P, D, nr, g.invD);
        if (part == 0) BK_TS(5);
    }
}

// gamma1 | beta1 | bq | bk | bv of a block, two values per thread (320 = 5 x 64 slots)
__device__ __forceinline__ void f1_fetch_vec(const cr_block_desc& d, float (&v)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = threadIdx.x + 256 * u, c = t & 63, which = (t >> 6) % 5;
        const float* src = which == 0 ? d.ln1_g : (which == 1 ? d.ln1_b : d.bqkv + (which - 2) * d.D);
        v[u] = src[c < d.D ? c : 0];
    }
}
__device__ __forceinline__ void f1_put_vec(float* vec, const float (&v)[2], int D) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = threadIdx.x + 256 * u;
        if (t < 320) vec[t] = ((t & 63) < D) ? v[u] : 0.0f;
    }
}

template <bool GATHER>
__global__ __launch_bounds__(256) void k_block_ln_qkv_fwd(cr_block_desc d, BlockGeom g, cr_embed_desc e) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = d.D;
    float* Xs = smem;                       // [64][66]
    float* Qs = Xs + 64 * F3_P;             // [64][66] LN1 output, then staging of each projection's result
    float* Ws = Qs + 64 * F3_P;             // 3 x packed [4*ks][64]: Wq, Wk, Wv
    const int wsz = 4 * g.ks * BK_WROW;
    float* vec = Ws + 3 * wsz;              // 5 x [64]: gamma1, beta1, bq, bk, bv (zero padded)
    const int m0 = blockIdx.x * 64;
    const int wave = threadIdx.x >> 6;
    const int mw = m0 + 16 * wave;
    const int nr = max(0, min(16, d.M - mw));
    BK_TS(0); BK_TS(1);
    int id[4], mk[4];
    if (GATHER) gather_issue_ids(e, mw, nr, id, mk);                // x does not exist yet: this kernel composes it
    else wave_load_rows(Xs + 16 * wave * F3_P, d.x + (size_t)mw * D, F3_P, D, nr, g.invD);
    {
        WFrag w3[3];
        float vv[2];
#pragma unroll
        for (int part = 0; part < 3; ++part) fetch_w(w3[part], d.wqkv, 3 * D, part * D, D);
        f1_fetch_vec(d, vv);
        if (GATHER) {
            Stream4 sx;
            gather_rows(sx, e, id, mk, mw, nr, D);
            stream_put(Xs + 16 * wave * F3_P, sx, D, nr * D, g.invD, PutPlain());
        }
#pragma unroll
        for (int part = 0; part < 3; ++part) put_w_packed(Ws + part * wsz, w3[part], D, 4 * g.ks);
        f1_put_vec(vec, vv, D);
    }
    BK_TS(2);
    __syncthreads();                        // the only barrier
    BK_TS(3);
    f1_body(d, g, Xs, Qs, Ws, vec, m0);
    BK_TS(6); BK_TS(15);
}

// ---- F3: LN2 + point-wise feed-forward + residual + mask -----------------------------------------------
// Pad-tolerant, branch-free element code: tiles have pitch 66 and ALL 64 columns are computed and written to
// LDS unconditionally -- weights, biases and inputs are zero beyond D, so pad columns come out as exact zeros
// (relu(0) = 0, dropout(0) = 0) without a single per-lane condition.  Global traffic goes through the
// wave-contiguous row streams, which know the valid row count.
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

// TAIL = 0: plain.  TAIL = 1: the NEXT block's LN1 + Q/K/V projections run on the output rows while they are still
// in LDS (one launch, one staging and one round trip of y less per block boundary).  TAIL = 2: the stack's final
// LayerNorm (sasrec.py:85) is applied to the output rows and written to `out` (a column block of a [M, ld] buffer).
struct BlockTail {
    cr_block_desc next;                     // TAIL 1
    const float* lnf_g; const float* lnf_b; float* out; int ld_out, col_out;   // TAIL 2
};
__device__ __forceinline__ void f3_dummy() {}
template <int TAIL>
__global__ __launch_bounds__(256) void k_block_ln_ffn_fwd(cr_block_desc d, BlockGeom g, BlockTail tl) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = d.D;
    const int wsz = 4 * g.ks * BK_WROW;
    float* Os = smem;                       // [64][66] input tile, reused for the hidden tile and the output tile
    float* Fs = Os + 64 * F3_P;             // [64][66]
    float* W1s = Fs + 64 * F3_P;            // packed [4*ks][64]
    float* W2s = W1s + wsz;                 // packed [4*ks][64]   (TAIL 1: a third slot follows for Wq|Wk|Wv)
    float* vec = W2s + (TAIL == 1 ? 2 : 1) * wsz;   // 5 x [64]: gamma2, beta2, b1, b2 (zero padded) -- later the next block's vectors
    float* msk = vec + 320;                 // [64] row mask (sasrec.py:83)
    float* Hs = Os;
    const DropCtx d1 = drop_ctx(d.drop_ffn1), d2 = drop_ctx(d.drop_ffn2);   // step counter load: requested first, needed late
    const int m0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int mw = m0 + 16 * wave;                                  // first row of this wave
    const int nr = max(0, min(16, d.M - mw));                       // its valid rows
    {   // every global request of the prologue is issued before the first one is consumed
        Stream4 so;
        stream_fetch(so, d.o + (size_t)mw * D, D, nr * D);
        WFrag wa, wb;
        fetch_w(wa, d.w1, D, 0, D);
        fetch_w(wb, d.w2, D, 0, D);
        const int t = threadIdx.x, c = t & 63, which = t >> 6;
        const float* src = which == 0 ? d.ln2_g : (which == 1 ? d.ln2_b : (which == 2 ? d.b1 : d.b2));
        const float vv = src[c < D ? c : 0];
        const int mk = (t < 64 && m0 + t < d.M) ? d.mask_ids[m0 + t] : 0;
        stream_put(Os + 16 * wave * F3_P, so, D, nr * D, g.invD, PutPlain());
        put_w_packed(W1s, wa, D, 4 * g.ks);
        put_w_packed(W2s, wb, D, 4 * g.ks);
        vec[t] = (c < D) ? vv : 0.0f;
        if (t < 64) msk[t] = mk != 0 ? 1.0f : 0.0f;
    }
    float lng[4], lnb[4];                                           // TAIL 2: final LayerNorm gamma / beta of this lane's columns
    if (TAIL == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = li + 16 * i;
            lng[i] = tl.lnf_g[c < D ? c : 0];
            lnb[i] = tl.lnf_b[c < D ? c : 0];
        }
    }
    __syncthreads();
    ln_rows_fast(Os, Fs, vec, vec + 64, D, wave);                                           // sasrec.py:81
    if (nr > 0) wave_store_rows(d.f_in + (size_t)mw * D, Fs + 16 * wave * F3_P, F3_P, D, nr, g.invD);
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
            if (d1.on) v *= drop_factor_x(d1, rb[r] * CR_PHI + d1.key + (16u * j) * CR_PHI);   // modules.py:303-304
            Hs[(16 * wave + 4 * lg + r) * F3_P + 16 * j + li] = v;
        }
    }
    if (nr > 0) wave_store_rows(d.hid + (size_t)mw * D, Hs + 16 * wave * F3_P, F3_P, D, nr, g.invD);
    // TAIL 1: the next block's weights and vectors are requested now and fly under the second GEMM and its epilogue
    WFrag nw3[3];
    float nvv[2];
    if (TAIL == 1) {
#pragma unroll
        for (int part = 0; part < 3; ++part) fetch_w(nw3[part], tl.next.wqkv, 3 * D, part * D, D);
        f1_fetch_vec(tl.next, nvv);
    }
    zero_acc(acc);
    tile_mma(acc, Hs, F3_P, W2s, g.ks, wave);                                               // modules.py:306-308
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float bias = vec[192 + 16 * j + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + 4 * lg + r;
            float v = acc[j][r] + bias;
            if (d2.on) v *= drop_factor_x(d2, rb[r] * CR_PHI + d2.key + (16u * j) * CR_PHI);   // modules.py:309-310
            v = (v + Fs[row * F3_P + 16 * j + li]) * msk[row];                               // modules.py:313, sasrec.py:83
            Hs[row * F3_P + 16 * j + li] = v;              // the MFMAs above have consumed the wave's Hs rows
        }
    }
    if (nr > 0) wave_store_rows(d.y + (size_t)mw * D, Hs + 16 * wave * F3_P, F3_P, D, nr, g.invD);
    if (TAIL == 1) {
        __syncthreads();                                            // every wave is done with W1 / W2 / vec
#pragma unroll
        for (int part = 0; part < 3; ++part) put_w_packed(W1s + part * wsz, nw3[part], D, 4 * g.ks);
        f1_put_vec(vec, nvv, D);
        __syncthreads();
        f1_body(tl.next, g, Hs, Fs, W1s, vec, m0);                  // y rows (masked) are the next block's x
    }
    if (TAIL == 2) {
        // final LayerNorm of the stack on the wave's own output rows -> Fs -> out[:, col_out : col_out + D]
        const int l = lane & 15, sub = lane >> 4;
        const float invD = 1.0f / (float)D;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = 16 * wave + 4 * p + sub;
            float x[4], sm = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = Hs[r * F3_P + l + 16 * i]; sm += x[i]; }
            const float mean = sum16(sm) * invD;
            float v = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = (l + 16 * i < D) ? x[i] - mean : 0.0f; v += x[i] * x[i]; }
            const float sd = sqrtf(sum16(v) * invD + 1e-8f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = l + 16 * i;
                if (c < D && 4 * p + sub < nr)
                    tl.out[(size_t)(mw + 4 * p + sub) * tl.ld_out + tl.col_out + c] = lng[i] * (x[i] / sd) + lnb[i];
            }
        }
    }
}

// =====================================================================================================
// backward
// =====================================================================================================

// zero the wave's 16 rows of a tile (pad columns / rows beyond the valid count must read as 0)
__device__ __forceinline__ void zero_rows(float* Ts) {
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < 16 * F3_P; e += 64) Ts[e] = 0.0f;
}

// strip of a [D,D] weight gradient held as (wave, lg, r) x (j, li) accumulators -> slab (row pitch ldw);
// with the ones-column trick accumulator row `ones` (== D) is the bias gradient
template <int NJ>
__device__ __forceinline__ void store_wgrad_g(float* dst, int ldw, float* bias_dst, const f32x4 (&accw)[NJ], int D, int ones,
                                              int wave, int half) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = 16 * (NJ * half + j) + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * wave + 4 * lg + r;
            if (col < D) {
                if (k < D) dst[k * ldw + col] = accw[j][r];
                else if (k == ones) bias_dst[col] = accw[j][r];
            }
        }
    }
}

__device__ __forceinline__ float colsum64(const float* Ts, int col) {   // D == 64 fallback: one thread sums one column
    float s = 0.0f;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) s += Ts[r * F3_P + col];
    return s;
}

// LayerNorm backward on the wave's 16 rows, in place: DYs rows hold dy on entry and dx on exit
// (dx = rstd*(dy*g - c1 - xhat*c2) [+ ADs]).  Accumulates dgamma / dbeta partials.  x from Xs (pad columns 0).
__device__ __forceinline__ void ln_bwd_rows(const float* Xs, float* DYs, const float* ADs, const float* gam,
                                            float (&ag)[4], float (&ab)[4], int D, int wave) {
    const int lane = threadIdx.x & 63, l = lane & 15, sub = lane >> 4;
    const float invD = 1.0f / (float)D;
    float g[4], in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[i] = gam[l + 16 * i]; in[i] = (l + 16 * i < D) ? 1.0f : 0.0f; }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = 16 * wave + 4 * p + sub;
        float x[4], dy[4], s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = Xs[r * F3_P + l + 16 * i] * in[i]; dy[i] = DYs[r * F3_P + l + 16 * i] * in[i]; s += x[i]; }
        const float mean = sum16(s) * invD;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = (x[i] - mean) * in[i]; v += x[i] * x[i]; }
        const float rstd = 1.0f / sqrtf(sum16(v) * invD + 1e-8f);
        float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[i] *= rstd;
            const float dg = dy[i] * g[i];
            c1 += dg; c2 += dg * x[i];
            ag[i] += dy[i] * x[i]; ab[i] += dy[i];                 // rows beyond the valid count hold dy = 0
        }
        c1 = sum16(c1) * invD;
        c2 = sum16(c2) * invD;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float dx = rstd * (dy[i] * g[i] - c1 - x[i] * c2) * in[i];
            if (ADs) dx += ADs[r * F3_P + l + 16 * i];
            DYs[r * F3_P + l + 16 * i] = dx;
        }
    }
}

// fold the per-lane LayerNorm partials (4 row groups per wave via shuffles, then the 4 waves through
// per-wave LDS slots summed in a fixed order: bitwise reproducible) and write the slab entries
template <int NW = 4>
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
    for (int c = threadIdx.x; c < D; c += 64 * NW) {
        float g = (sg[c] + sg[64 + c]) + (sg[128 + c] + sg[192 + c]);
        float b = (sb[c] + sb[64 + c]) + (sb[128 + c] + sb[192 + c]);
        if (NW == 8) {
            g += (sg[256 + c] + sg[320 + c]) + (sg[384 + c] + sg[448 + c]);
            b += (sb[256 + c] + sb[320 + c]) + (sb[384 + c] + sb[448 + c]);
        }
        dg[c] = g;
        db[c] = b;
    }
}

__device__ __forceinline__ int rows_per_wg(int M, int nwg) {       // multiple of the 64-row tile
    const int rps = (M + nwg - 1) / nwg;
    return (rps + 63) / 64 * 64;
}

// set column `ones` of the wave's valid rows to 1 (bias-gradient trick)
__device__ __forceinline__ void plant_ones(float* Ts, int ones, int nr) {
    const int lane = threadIdx.x & 63;
    if (ones >= 0 && lane < nr) Ts[lane * F3_P + ones] = 1.0f;
}

// ---- B3: backward of LN2 + FFN --------------------------------------------------------------------------
// NG groups of 4 waves per workgroup; each group works on its own 64-row tile (own T1..T3, own row mask), all
// share one copy of the weights.  NG = 2 puts two waves on every SIMD (the kernel is a chain of dependent
// load -> LDS -> MFMA -> LDS steps; a single wave per SIMD leaves each of them exposed) without doubling the
// number of gradient slabs.  Barriers are workgroup-wide; a group whose tile lies beyond the workgroup's rows
// runs the phases on zero rows.
template <int NG>
__global__ __launch_bounds__(256 * NG) void k_block_ln_ffn_bwd(cr_block_bwd_desc bd, BlockGeom gm) {
    constexpr int NT = 256 * NG;
    const cr_block_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = d.D, ks = gm.ks, ones = gm.ones;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int grp = wave >> 2, w4 = wave & 3, gtid = threadIdx.x & 255;
    float* W1t = smem;                     // W1^T, packed [4*ks][64]
    float* W2t = W1t + 4 * ks * BK_WROW;   // W2^T, packed
    float* gam = W2t + 4 * ks * BK_WROW;   // [64] gamma2, zero padded
    float* sg = gam + 64;                  // [4*NG][64] + [4*NG][64]
    float* sb = sg + 256 * NG;
    float* Tg = sb + 256 * NG + grp * (3 * 64 * F3_P + 64);
    float* T1 = Tg;                        // g2                             -> later df -> d_o
    float* T2 = T1 + 64 * F3_P;            // hid (+ ones column)            -> later g1
    float* T3 = T2 + 64 * F3_P;            // f_in (+ ones column)           -> later o
    float* msk = T3 + 64 * F3_P;           // [64] row mask of the group's tile
    const int rps = (d.M + gridDim.x - 1) / gridDim.x;             // rows of this workgroup (any count)
    const int mb = blockIdx.x * rps, me = min(d.M, mb + rps);
    // loads first (weights, gamma, the first tile's streams), LDS writes after
    WFragT<NT> vw1, vw2;
    fetch_w<NT, true>(vw1, d.w1, D, 0, D);
    fetch_w<NT, true>(vw2, d.w2, D, 0, D);
    const float gam_v = d.ln2_g[threadIdx.x < D ? threadIdx.x : 0];
    const DropCtx d2 = drop_ctx(d.drop_ffn2);
    const float scale1 = (d.drop_ffn1.rate > 0.0f) ? 1.0f / (1.0f - d.drop_ffn1.rate) : 1.0f;
    // weight-gradient accumulators: with two tile groups each wave owns a k-strip x one HALF of the columns over the
    // rows of both groups (tile_wgrad_g): final sums, no cross-group fold
    constexpr int NJ = (NG == 2) ? 2 : 4;
    constexpr int GST = 3 * 64 * F3_P + 64;          // LDS distance between the groups' tile sets
    float* Tg0 = sb + 256 * NG;                       // group 0's tiles
    const int half = (NG == 2) ? grp : 0;
    f32x4 aw1[NJ], aw2[NJ];
    zero_acc_n<NJ>(aw1); zero_acc_n<NJ>(aw2);
    float b1s = 0.0f, b2s = 0.0f;
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    Stream4 sdy, shid, sfin;
    BK_TSG(gm, 0); BK_TSG(gm, 1);
    auto fetch = [&](int m0) {
        const int mw = m0 + 16 * w4;
        const int tot = max(0, min(16, me - mw)) * D;
        stream_fetch(sdy, bd.dy + (size_t)mw * D, D, tot);
        stream_fetch(shid, d.hid + (size_t)mw * D, D, tot);
        stream_fetch(sfin, d.f_in + (size_t)mw * D, D, tot);
    };
    if (mb < me) fetch(mb + 64 * grp);
    BK_TSG(gm, 2);
    put_wt_packed<NT>(W1t, vw1, D, 4 * ks);
    put_wt_packed<NT>(W2t, vw2, D, 4 * ks);
    if (threadIdx.x < 64) gam[threadIdx.x] = (threadIdx.x < D) ? gam_v : 0.0f;
    BK_TSG(gm, 3);
    for (int base = mb; base < me; base += 64 * NG) {
        const int m0 = base + 64 * grp;
        const int mw = m0 + 16 * w4;
        const int nr = max(0, min(16, me - mw)), tot = nr * D;
        float* t1 = T1 + 16 * w4 * F3_P; float* t2 = T2 + 16 * w4 * F3_P;
        float* t3 = T3 + 16 * w4 * F3_P;
        if (lane < 16) msk[16 * w4 + lane] = (lane < nr && d.mask_ids[mw + lane] != 0) ? 1.0f : 0.0f;
        if (base == mb) BK_TSG(gm, 4);
        // g2 = dy * mask * keep2/(1-rate) (sasrec.py:83, modules.py:309-310)
        {
            const float* mrow = msk + 16 * w4;
            const uint32_t hb = (d.drop_ffn2.row_offset + (uint32_t)mw) * (uint32_t)D;
            stream_put(t1, sdy, D, tot, gm.invD, [&](int e, int r, float v) {
                float x = v * mrow[r];
                if (d2.on) x *= drop_factor_x(d2, (hb + (uint32_t)e) * CR_PHI + d2.key);
                return x;
            });
        }
        if (base == mb) BK_TSG(gm, 5);
        stream_put(t2, shid, D, tot, gm.invD, PutPlain());
        stream_put(t3, sfin, D, tot, gm.invD, PutPlain());
        plant_ones(t2, ones, nr);
        plant_ones(t3, ones, nr);
        if (base == mb) BK_TSG(gm, 6);
        __syncthreads();
        if (base + 64 * NG < me) fetch(m0 + 64 * NG);              // next tile's loads fly under this tile's MFMAs
        // dW2 (+ db2 in row `ones`) += hid^T g2
        tile_wgrad_g<NJ, NG>(aw2, Tg0 + 64 * F3_P, Tg0, GST, F3_P, w4, half);
        if (ones < 0 && threadIdx.x < 64) {
            b2s += colsum64(Tg0, threadIdx.x);
            if (NG == 2) b2s += colsum64(Tg0 + GST, threadIdx.x);
        }
        __syncthreads();
        if (base == mb) BK_TSG(gm, 7);
        // dhid = g2 W2^T, gated by the stored post-dropout ReLU output -> g1 (in place over hid, own rows)
        {
            f32x4 acc[4];
            zero_acc(acc);
            tile_mma<4>(acc, T1, F3_P, W2t, ks, w4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* ph = T2 + (16 * w4 + 4 * lg + r) * F3_P + 16 * j + li;
                    const bool gate = (*ph > 0.0f) && (16 * j + li != ones);
                    *ph = gate ? acc[j][r] * scale1 : 0.0f;                                 // modules.py:300-304
                }
        }
        __syncthreads();
        if (base == mb) BK_TSG(gm, 8);
        // dW1 (+ db1) += f_in^T g1
        tile_wgrad_g<NJ, NG>(aw1, Tg0 + 2 * 64 * F3_P, Tg0 + 64 * F3_P, GST, F3_P, w4, half);
        if (ones < 0 && threadIdx.x < 64) {
            b1s += colsum64(Tg0 + 64 * F3_P, threadIdx.x);
            if (NG == 2) b1s += colsum64(Tg0 + GST + 64 * F3_P, threadIdx.x);
        }
        // df = g1 W1^T + dy*mask  (residual of modules.py:313; g1 rows of masked positions are 0) -> T1 (own rows)
        {
            // residual dy * mask in the accumulator layout, re-read from L2 (this tile's dy was streamed a moment ago)
            float res[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = (4 * lg + r < nr) && (16 * j + li < D);
                    res[j][r] = bd.dy[ok ? (size_t)(mw + 4 * lg + r) * D + 16 * j + li : (size_t)mb * D];
                }
            f32x4 acc[4];
            zero_acc(acc);
            tile_mma<4>(acc, T2, F3_P, W1t, ks, w4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * w4 + 4 * lg + r;
                    const bool ok = (4 * lg + r < nr) && (16 * j + li < D);
                    T1[row * F3_P + 16 * j + li] = acc[j][r] + (ok ? res[j][r] * msk[row] : 0.0f);
                }
        }
        __syncthreads();                                           // all waves are done with every row of T3
        if (base == mb) BK_TSG(gm, 9);
        // LN2 backward in place on own rows: x = o (streamed into T3), dy = df (T1) -> d_o (T1)
        Stream4 sqin;                                              // q_in rows for the attention delta (T2 is free by now)
        if (bd.attn_delta) stream_fetch(sqin, d.q_in + (size_t)mw * D, D, tot);
        wave_load_rows(t3, d.o + (size_t)mw * D, F3_P, D, nr, gm.invD);
        ln_bwd_rows(T3, T1, nullptr, gam, ag, ab, D, w4);
        if (nr > 0) wave_store_rows(bd.d_o + (size_t)mw * D, t1, F3_P, D, nr, gm.invD);
        if (bd.attn_delta) {
            // delta[m] = sum_c d_o[m][c] * (o[m][c] - q_in[m][c]): the softmax-backward row term of the attention
            // core (its output is o - q_in, modules.py:262-269), for the single-pass cr_attn_bwd
            stream_put(t2, sqin, D, tot, gm.invD, PutPlain());
            const int l = lane & 15, sub = lane >> 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = 16 * w4 + 4 * p + sub;
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = l + 16 * i;
                    acc += T1[r * F3_P + c] * (T3[r * F3_P + c] - T2[r * F3_P + c]);      // pad columns hold 0
                }
                acc = sum16(acc);
                if (l == 0 && 4 * p + sub < nr) bd.attn_delta[mw + 4 * p + sub] = acc;
            }
        }
        __syncthreads();
        if (base == mb) BK_TSG(gm, 10);
    }
    BK_TSG(gm, 14);
    const size_t so = (size_t)blockIdx.x * bd.slab_stride;
    store_wgrad_g<NJ>(bd.g_w1 + so, D, bd.g_b1 + so, aw1, D, ones, w4, half);
    store_wgrad_g<NJ>(bd.g_w2 + so, D, bd.g_b2 + so, aw2, D, ones, w4, half);
    if (ones < 0 && threadIdx.x < D) { bd.g_b1[so + threadIdx.x] = b1s; bd.g_b2[so + threadIdx.x] = b2s; }
    store_ln_grads<4 * NG>(sg, sb, ag, ab, bd.g_ln2_g + so, bd.g_ln2_b + so, D);
    BK_TSG(gm, 15);
}

// ---- B1: backward of LN1 + Q/K/V projections --------------------------------------------------------------
// dqkv is [3, M, D] (dQ rows, dK rows, dV rows), like qkv.
// NG groups of 4 waves per workgroup (see B3), three 64-row tiles per group, reused phase by phase:
//   TG: dQ -> dK -> dV -> dq_in -> dx      TA: q_in -> x      TB: dK Wk^T + dV Wv^T
// so that two groups and the three packed transposed weights fit the 160 KB of a CU (two waves per SIMD).
// SCATTER: the block is the first of its stack and x was composed by an embedding gather (cr_embed_fwd /
// cr_block_ln_qkv_fwd_gather): instead of storing dx, the wave applies that gather's backward to its 16 rows right
// here -- mask and dropout regenerated, d_addend written, table rows scatter-added with one contiguous float-atomic
// burst per row (cr_embed_bwd's large-table mode; `sc` is the descriptor that call would have taken).
// SCATTER: 0 none, 1 table rows (+ d_addend), 2 also a learned positional table (kept out of variant 1: the extra
// live values pushed the kernel, which sits at the 256-VGPR cap, from 48 to 88 bytes of scratch and cost 3.5 us).
template <int NG, int SCATTER>
__global__ __launch_bounds__(256 * NG) void k_block_ln_qkv_bwd(cr_block_bwd_desc bd, BlockGeom gm, cr_embed_bwd_desc sc) {
    constexpr int NT = 256 * NG;
    const cr_block_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = d.D, ks = gm.ks, ones = gm.ones;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int grp = wave >> 2, w4 = wave & 3, gtid = threadIdx.x & 255;
    float* Wqt = smem;                     // Wq^T, Wk^T, Wv^T packed [4*ks][64]
    float* Wkt = Wqt + 4 * ks * BK_WROW;
    float* Wvt = Wkt + 4 * ks * BK_WROW;
    float* gam = Wvt + 4 * ks * BK_WROW;   // [64] gamma1
    float* sg = gam + 64;                  // [4*NG][64] x 2
    float* sb = sg + 256 * NG;
    float* TG = sb + 256 * NG + grp * (3 * 64 * F3_P);
    float* TA = TG + 64 * F3_P;
    float* TB = TA + 64 * F3_P;
    const int rps = (d.M + gridDim.x - 1) / gridDim.x;
    const int mb = blockIdx.x * rps, me = min(d.M, mb + rps);
    const size_t MD = (size_t)d.M * D;
    WFragT<NT> vq, vk, vv;
    fetch_w<NT, true>(vq, d.wqkv, 3 * D, 0, D);
    fetch_w<NT, true>(vk, d.wqkv, 3 * D, D, D);
    fetch_w<NT, true>(vv, d.wqkv, 3 * D, 2 * D, D);
    const float gam_v = d.ln1_g[threadIdx.x < D ? threadIdx.x : 0];
    // weight-gradient accumulators as in the FFN backward: k-strip x column half over the rows of both groups
    constexpr int NJ = (NG == 2) ? 2 : 4;
    constexpr int GST = 3 * 64 * F3_P;               // LDS distance between the groups' tile sets
    float* TG0 = sb + 256 * NG;                       // group 0's TG; its TA follows at + 64 * F3_P
    const int half = (NG == 2) ? grp : 0;
    f32x4 awq[NJ], awk[NJ], awv[NJ];
    zero_acc_n<NJ>(awq); zero_acc_n<NJ>(awk); zero_acc_n<NJ>(awv);
    float bqs = 0.0f, bks = 0.0f, bvs = 0.0f;
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    Stream4 s0, s1, s2, s3, sp;             // dQ, q_in, dK, x  (then s0, s1 again: dV, d_o); sp: second partial of dQ
    auto fetch4 = [&](int m0) {
        const int mw = m0 + 16 * w4;
        const int tot = max(0, min(16, me - mw)) * D;
        if (bd.dq_part) stream_fetch(sp, bd.dq_part + (size_t)mw * D, D, tot);
        stream_fetch(s0, bd.dqkv + (size_t)mw * D, D, tot);
        stream_fetch(s1, d.q_in + (size_t)mw * D, D, tot);
        stream_fetch(s2, bd.dqkv + MD + (size_t)mw * D, D, tot);
        stream_fetch(s3, d.x + (size_t)mw * D, D, tot);
    };
    BK_TSG(gm, 0); BK_TSG(gm, 1);
    if (mb < me) fetch4(mb + 64 * grp);
    put_wt_packed<NT>(Wqt, vq, D, 4 * ks);
    put_wt_packed<NT>(Wkt, vk, D, 4 * ks);
    put_wt_packed<NT>(Wvt, vv, D, 4 * ks);
    if (threadIdx.x < 64) gam[threadIdx.x] = (threadIdx.x < D) ? gam_v : 0.0f;
    for (int base = mb; base < me; base += 64 * NG) {
        const int m0 = base + 64 * grp;
        const int mw = m0 + 16 * w4;
        const int nr = max(0, min(16, me - mw)), tot = nr * D;
        float* tg = TG + 16 * w4 * F3_P; float* ta = TA + 16 * w4 * F3_P; float* tb = TB + 16 * w4 * F3_P;
        if (base != mb) fetch4(m0);
        if (base == mb) BK_TSG(gm, 2);
        // ---- phase 1: dQ, q_in -> dWq (+ dbq), dq_in = dQ Wq^T
        stream_put(tg, s0, D, tot, gm.invD, PutPlain());
        if (bd.dq_part) stream_add(tg, sp, D, tot);                // dQ = the two partial sums of the single-pass attention backward
        stream_put(ta, s1, D, tot, gm.invD, PutPlain());
        plant_ones(ta, ones, nr);
        stream_fetch(s0, bd.dqkv + 2 * MD + (size_t)mw * D, D, tot);      // dV and the residual gradient d_o, for later phases
        stream_fetch(s1, bd.d_o + (size_t)mw * D, D, tot);
        if (base == mb) BK_TSG(gm, 3);
        __syncthreads();
        if (base == mb) BK_TSG(gm, 4);
        tile_wgrad_g<NJ, NG>(awq, TG0 + 64 * F3_P, TG0, GST, F3_P, w4, half);
        if (ones < 0 && threadIdx.x < 64) {
            bqs += colsum64(TG0, threadIdx.x);
            if (NG == 2) bqs += colsum64(TG0 + GST, threadIdx.x);
        }
        f32x4 acc[4], dxa[4];
        zero_acc(acc); zero_acc(dxa);
        tile_mma<4>(acc, TG, F3_P, Wqt, ks, w4);
        if (base == mb) BK_TSG(gm, 5);
        __syncthreads();                                           // all rows of TG (dQ) and TA (q_in) have been read
        // ---- phase 2: dK, x -> dWk (+ dbk), dx_part = dK Wk^T
        stream_put(tg, s2, D, tot, gm.invD, PutPlain());
        stream_put(ta, s3, D, tot, gm.invD, PutPlain());
        plant_ones(ta, ones, nr);
        __syncthreads();
        tile_wgrad_g<NJ, NG>(awk, TG0 + 64 * F3_P, TG0, GST, F3_P, w4, half);
        if (ones < 0 && threadIdx.x < 64) {
            bks += colsum64(TG0, threadIdx.x);
            if (NG == 2) bks += colsum64(TG0 + GST, threadIdx.x);
        }
        tile_mma<4>(dxa, TG, F3_P, Wkt, ks, w4);
        if (base == mb) BK_TSG(gm, 6);
        __syncthreads();                                           // all rows of TG (dK) have been read
        // ---- phase 3: dV -> dWv (+ dbv), dx_part += dV Wv^T
        stream_put(tg, s0, D, tot, gm.invD, PutPlain());
        __syncthreads();
        tile_wgrad_g<NJ, NG>(awv, TG0 + 64 * F3_P, TG0, GST, F3_P, w4, half);
        if (ones < 0 && threadIdx.x < 64) {
            bvs += colsum64(TG0, threadIdx.x);
            if (NG == 2) bvs += colsum64(TG0 + GST, threadIdx.x);
        }
        tile_mma<4>(dxa, TG, F3_P, Wvt, ks, w4);
        if (base == mb) BK_TSG(gm, 7);
        __syncthreads();                                           // all rows of TG (dV) and TA (x) have been read
        if (base == mb) BK_TSG(gm, 8);
        // ---- phase 4 (own rows): dq_in = dQ Wq^T + d_o (modules.py:269); dx = dx_part + LN1bwd(dq_in; x)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (16 * w4 + 4 * lg + r) * F3_P + 16 * j + li;
                TG[o] = acc[j][r];
                TB[o] = dxa[j][r];
            }
        stream_add(tg, s1, D, tot);                                 // tg += d_o (residual branch)
        if (ones >= 0 && lane < 16) ta[lane * F3_P + ones] = 0.0f;     // remove the ones column before LN reads x
        ln_bwd_rows(TA, TG, TB, gam, ag, ab, D, w4);
        if (nr > 0 && !SCATTER) {
            float* gdx = bd.dx + (size_t)mw * D;
            if (bd.dx_accumulate) {
                wave_load_rows(tb, gdx, F3_P, D, nr, gm.invD);
                for (int e = lane; e < 16 * F3_P; e += 64) tg[e] += tb[e];
            }
            wave_store_rows(gdx, tg, F3_P, D, nr, gm.invD);
        }
        if (nr > 0 && SCATTER) {
            const cr_embed_desc& e = sc.f;
            const DropCtx dc = drop_ctx(e.drop);
            const int mrow = mw + min(lane & 15, nr - 1);
            const int my_id = e.ids[mrow];                              // lane r (< 16) holds the id / mask of row r
            const int my_mk = e.mask_ids ? e.mask_ids[mrow] : 1;
            {   // g = dx * mask, dropout regenerated (same counters as the forward), in place in the wave's strip
                const int q = lane & 15, r0 = lane >> 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = r0 + 4 * i;
                    const int mk = __shfl(my_mk, r, 64);
                    if (r < nr && 4 * q < D) {
                        float2* pt = reinterpret_cast<float2*>(tg + r * F3_P + 4 * q);
                        float2 a = pt[0], b = pt[1];
                        const uint32_t base_idx = (e.drop.row_offset + (uint32_t)(mw + r)) * (uint32_t)D + (uint32_t)(4 * q);
                        const float k = mk != 0 ? 1.0f : 0.0f;
                        a.x = drop_apply(dc, base_idx + 0u, a.x * k);
                        a.y = drop_apply(dc, base_idx + 1u, a.y * k);
                        b.x = drop_apply(dc, base_idx + 2u, b.x * k);
                        b.y = drop_apply(dc, base_idx + 3u, b.y * k);
                        pt[0] = a;
                        pt[1] = b;
                    }
                }
            }
            if (sc.d_addend) wave_store_rows(sc.d_addend + (size_t)mw * D, tg, F3_P, D, nr, gm.invD);
            if (sc.table_grad) {
#pragma unroll 4
                for (int r = 0; r < nr; ++r) {
                    const int id = __shfl(my_id, r, 64);
                    if (!(e.zero_pad && id == 0) && lane < D)
                        atomicAdd(sc.table_grad + (size_t)id * D + lane, tg[r * F3_P + lane] * e.scale);
                }
            }
            if (SCATTER == 2 && sc.pos_grad) {                          // learned positional table: row m % T, no scale
#pragma unroll 4
                for (int r = 0; r < nr; ++r)
                    if (lane < D) atomicAdd(sc.pos_grad + (size_t)((mw + r) % e.T) * D + lane, tg[r * F3_P + lane]);
            }
        }
        if (base == mb) BK_TSG(gm, 9);
    }
    BK_TSG(gm, 14);
    const size_t so = (size_t)blockIdx.x * bd.slab_stride;
    store_wgrad_g<NJ>(bd.g_wqkv + so, 3 * D, bd.g_bqkv + so, awq, D, ones, w4, half);
    store_wgrad_g<NJ>(bd.g_wqkv + so + D, 3 * D, bd.g_bqkv + so + D, awk, D, ones, w4, half);
    store_wgrad_g<NJ>(bd.g_wqkv + so + 2 * D, 3 * D, bd.g_bqkv + so + 2 * D, awv, D, ones, w4, half);
    if (ones < 0 && threadIdx.x < D) {
        bd.g_bqkv[so + threadIdx.x] = bqs;
        bd.g_bqkv[so + D + threadIdx.x] = bks;
        bd.g_bqkv[so + 2 * D + threadIdx.x] = bvs;
    }
    store_ln_grads<4 * NG>(sg, sb, ag, ab, bd.g_ln1_g + so, bd.g_ln1_b + so, D);
    BK_TSG(gm, 15);
}

// ---- host side ------------------------------------------------------------------------------------------
static int block_check(const cr_block_desc* d, BlockGeom* g, const char* who) {
    CR_REQUIRE(d != nullptr, "%s: NULL desc", who);
    CR_REQUIRE(d->M > 0 && d->D > 0, "%s: bad shape", who);
    if (d->D > 64 || d->D < 4)
        return cr_set_error(CR_ERR_UNSUPPORTED, "%s: D=%d outside [4, 64] (use the unfused kernels)", who, d->D);
    g->ks = (d->D + 3) / 4;
    g->P = 4 * g->ks + 2;
    g->ones = d->D < 64 ? d->D : -1;
    g->invD = (uint32_t)(4294967296.0 / d->D) + 1u;
    const char* e = getenv("CR_BLOCK_DBG");
    g->dbg = e ? atoi(e) : 0;
    g->ts = g_block_ts;
    return CR_OK;
}

static int block_qkv_fwd_launch(const cr_block_desc* d, const cr_embed_desc* e, void* stream, const char* who) {
    BlockGeom g;
    int rc = block_check(d, &g, who);
    if (rc) return rc;
    CR_REQUIRE(d->x && d->q_in && d->qkv && d->k_valid && d->q_valid && d->ln1_g && d->ln1_b && d->wqkv && d->bqkv,
               "%s: NULL pointer", who);
    const size_t lds = sizeof(float) * (2 * 64 * F3_P + 3 * 4 * g.ks * BK_WROW + 320);
    static cr_devmask attr[2] = {0, 0};
    if (e) {
        CR_REQUIRE(e->ids && e->table && e->out == d->x && e->ld_out == d->D && e->col_off == 0 && e->M == d->M && e->D == d->D &&
                   e->T > 0 && e->M % e->T == 0 && e->V > 0 && (e->addend == nullptr || e->ld_add >= e->D),
                   "%s: the embedding recipe must describe the block's dense input x (out == x, ld_out == D, col_off == 0)", who);
        rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_qkv_fwd<true>), &attr[1]);
        if (rc) return rc;
        hipLaunchKernelGGL(k_block_ln_qkv_fwd<true>, dim3(cr_ceil_div(d->M, 64)), dim3(256), lds, cr_stream(stream), *d, g, *e);
    } else {
        cr_embed_desc none = {};
        rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_qkv_fwd<false>), &attr[0]);
        if (rc) return rc;
        hipLaunchKernelGGL(k_block_ln_qkv_fwd<false>, dim3(cr_ceil_div(d->M, 64)), dim3(256), lds, cr_stream(stream), *d, g, none);
    }
    return cr_check_launch(who);
}

extern "C" int cr_block_ln_qkv_fwd(const cr_block_desc* d, void* stream) {
    return block_qkv_fwd_launch(d, nullptr, stream, "cr_block_ln_qkv_fwd");
}

extern "C" int cr_block_ln_qkv_fwd_gather(const cr_block_desc* d, const cr_embed_desc* e, void* stream) {
    CR_REQUIRE(e, "cr_block_ln_qkv_fwd_gather: NULL embedding recipe");
    return block_qkv_fwd_launch(d, e, stream, "cr_block_ln_qkv_fwd_gather");
}

static int block_ffn_fwd_launch(const cr_block_desc* d, const cr_block_tail_desc* t, void* stream, const char* who) {
    BlockGeom g;
    int rc = block_check(d, &g, who);
    if (rc) return rc;
    CR_REQUIRE(d->o && d->f_in && d->hid && d->y && d->mask_ids && d->ln2_g && d->ln2_b && d->w1 && d->b1 && d->w2 && d->b2,
               "%s: NULL pointer", who);
    BlockTail tl = {};
    const int kind = t ? t->kind : 0;
    CR_REQUIRE(kind >= 0 && kind <= 2, "%s: tail kind %d", who, kind);
    if (kind == 1) {
        CR_REQUIRE(t->next != nullptr, "%s: tail 1 needs the next block's description", who);
        const cr_block_desc* n = t->next;
        CR_REQUIRE(n->M == d->M && n->D == d->D, "%s: next block has a different shape", who);
        CR_REQUIRE(n->x == d->y, "%s: the next block's input must be this block's output", who);
        CR_REQUIRE(n->q_in && n->qkv && n->k_valid && n->q_valid && n->ln1_g && n->ln1_b && n->wqkv && n->bqkv, "%s: next block: NULL pointer", who);
        tl.next = *n;
    } else if (kind == 2) {
        CR_REQUIRE(t->lnf_gamma && t->lnf_beta && t->out && t->ld_out >= t->col_out + d->D, "%s: tail 2: bad final-LayerNorm output", who);
        tl.lnf_g = t->lnf_gamma; tl.lnf_b = t->lnf_beta; tl.out = t->out; tl.ld_out = t->ld_out; tl.col_out = t->col_out;
    }
    const size_t lds = sizeof(float) * (2 * 64 * F3_P + (kind == 1 ? 3 : 2) * 4 * g.ks * BK_WROW + 320 + 64);
    static cr_devmask attr[3] = {0, 0, 0};
    if (kind == 0) {
        rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_ffn_fwd<0>), &attr[0]);
        if (rc) return rc;
        hipLaunchKernelGGL(k_block_ln_ffn_fwd<0>, dim3(cr_ceil_div(d->M, 64)), dim3(256), lds, cr_stream(stream), *d, g, tl);
    } else if (kind == 1) {
        rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_ffn_fwd<1>), &attr[1]);
        if (rc) return rc;
        hipLaunchKernelGGL(k_block_ln_ffn_fwd<1>, dim3(cr_ceil_div(d->M, 64)), dim3(256), lds, cr_stream(stream), *d, g, tl);
    } else {
        rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_ffn_fwd<2>), &attr[2]);
        if (rc) return rc;
        hipLaunchKernelGGL(k_block_ln_ffn_fwd<2>, dim3(cr_ceil_div(d->M, 64)), dim3(256), lds, cr_stream(stream), *d, g, tl);
    }
    return cr_check_launch(who);
}

extern "C" int cr_block_ln_ffn_fwd(const cr_block_desc* d, void* stream) {
    return block_ffn_fwd_launch(d, nullptr, stream, "cr_block_ln_ffn_fwd");
}

extern "C" int cr_block_ln_ffn_fwd_tail(const cr_block_desc* d, const cr_block_tail_desc* t, void* stream) {
    return block_ffn_fwd_launch(d, t, stream, "cr_block_ln_ffn_fwd_tail");
}

extern "C" int cr_block_ln_ffn_bwd(const cr_block_bwd_desc* bd, void* stream) {
    CR_REQUIRE(bd != nullptr, "cr_block_ln_ffn_bwd: NULL desc");
    BlockGeom g;
    int rc = block_check(&bd->f, &g, "cr_block_ln_ffn_bwd");
    if (rc) return rc;
    const cr_block_desc* d = &bd->f;
    CR_REQUIRE(bd->dy && bd->d_o && d->hid && d->f_in && d->o && d->mask_ids && d->w1 && d->w2 && d->ln2_g, "cr_block_ln_ffn_bwd: NULL pointer");
    CR_REQUIRE(bd->g_w1 && bd->g_b1 && bd->g_w2 && bd->g_b2 && bd->g_ln2_g && bd->g_ln2_b && bd->n_slabs > 0, "cr_block_ln_ffn_bwd: NULL gradient pointer");
    // two 64-row tile groups per workgroup unless the workgroups have a single tile's worth of rows anyway
    const int rps = (d->M + bd->n_slabs - 1) / bd->n_slabs;
    const int ng = (rps > 64 && !(g.dbg & 1)) ? 2 : 1;
    const size_t lds = sizeof(float) * (2 * 4 * g.ks * BK_WROW + 64 + 512 * ng + ng * (3 * 64 * F3_P + 64));
    static cr_devmask attr1 = 0, attr2 = 0;
    if (ng == 2) {
        rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_ffn_bwd<2>), &attr2);
        if (rc) return rc;
        hipLaunchKernelGGL(k_block_ln_ffn_bwd<2>, dim3(bd->n_slabs), dim3(512), lds, cr_stream(stream), *bd, g);
    } else {
        rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_ffn_bwd<1>), &attr1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_block_ln_ffn_bwd<1>, dim3(bd->n_slabs), dim3(256), lds, cr_stream(stream), *bd, g);
    }
    return cr_check_launch("cr_block_ln_ffn_bwd");
}

static int block_qkv_bwd_launch(const cr_block_bwd_desc* bd, const cr_embed_bwd_desc* sc, void* stream, const char* who) {
    CR_REQUIRE(bd != nullptr, "%s: NULL desc", who);
    BlockGeom g;
    int rc = block_check(&bd->f, &g, who);
    if (rc) return rc;
    const cr_block_desc* d = &bd->f;
    CR_REQUIRE(bd->dqkv && bd->d_o && d->q_in && d->x && d->wqkv && d->ln1_g, "%s: NULL pointer", who);
    CR_REQUIRE(bd->g_wqkv && bd->g_bqkv && bd->g_ln1_g && bd->g_ln1_b && bd->n_slabs > 0, "%s: NULL gradient pointer", who);
    const int rps = (d->M + bd->n_slabs - 1) / bd->n_slabs;
    const int ng = (rps > 64 && !(g.dbg & 2)) ? 2 : 1;
    const size_t lds = sizeof(float) * (3 * 4 * g.ks * BK_WROW + 64 + 512 * ng + ng * (3 * 64 * F3_P));
    static cr_devmask attr[6] = {0, 0, 0, 0, 0, 0};
    hipStream_t s = cr_stream(stream);
    if (sc) {
        const cr_embed_desc* e = &sc->f;
        CR_REQUIRE(e->ids && e->M == d->M && e->D == d->D && e->ld_out == d->D && e->col_off == 0 && e->T > 0 && e->V > 0,
                   "%s: the embedding recipe must describe the block's dense input x", who);
        CR_REQUIRE(sc->n_slabs == 0, "%s: small-table mode is not fused", who);
        CR_REQUIRE(!bd->dx_accumulate, "%s: dx_accumulate is not supported (this kernel must be the only producer of dx)", who);
        CR_REQUIRE(sc->table_grad || sc->d_addend || sc->pos_grad, "%s: nothing to scatter into", who);
        CR_REQUIRE(sc->d_addend == nullptr || e->ld_add == d->D, "%s: d_addend must be dense [M, D]", who);
        const bool pos = sc->pos_grad != nullptr;
        if (ng == 2 && !pos) {
            rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_qkv_bwd<2, 1>), &attr[3]);
            if (rc) return rc;
            hipLaunchKernelGGL((k_block_ln_qkv_bwd<2, 1>), dim3(bd->n_slabs), dim3(512), lds, s, *bd, g, *sc);
        } else if (ng == 2) {
            rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_qkv_bwd<2, 2>), &attr[5]);
            if (rc) return rc;
            hipLaunchKernelGGL((k_block_ln_qkv_bwd<2, 2>), dim3(bd->n_slabs), dim3(512), lds, s, *bd, g, *sc);
        } else if (!pos) {
            rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_qkv_bwd<1, 1>), &attr[2]);
            if (rc) return rc;
            hipLaunchKernelGGL((k_block_ln_qkv_bwd<1, 1>), dim3(bd->n_slabs), dim3(256), lds, s, *bd, g, *sc);
        } else {
            rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_qkv_bwd<1, 2>), &attr[4]);
            if (rc) return rc;
            hipLaunchKernelGGL((k_block_ln_qkv_bwd<1, 2>), dim3(bd->n_slabs), dim3(256), lds, s, *bd, g, *sc);
        }
        return cr_check_launch(who);
    }
    CR_REQUIRE(bd->dx, "%s: dx is NULL", who);
    cr_embed_bwd_desc none = {};
    if (ng == 2) {
        rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_qkv_bwd<2, 0>), &attr[1]);
        if (rc) return rc;
        hipLaunchKernelGGL((k_block_ln_qkv_bwd<2, 0>), dim3(bd->n_slabs), dim3(512), lds, s, *bd, g, none);
    } else {
        rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_block_ln_qkv_bwd<1, 0>), &attr[0]);
        if (rc) return rc;
        hipLaunchKernelGGL((k_block_ln_qkv_bwd<1, 0>), dim3(bd->n_slabs), dim3(256), lds, s, *bd, g, none);
    }
    return cr_check_launch(who);
}

extern "C" int cr_block_ln_qkv_bwd(const cr_block_bwd_desc* bd, void* stream) {
    return block_qkv_bwd_launch(bd, nullptr, stream, "cr_block_ln_qkv_bwd");
}

extern "C" int cr_block_ln_qkv_bwd_scatter(const cr_block_bwd_desc* bd, const cr_embed_bwd_desc* sc, void* stream) {
    CR_REQUIRE(sc != nullptr, "cr_block_ln_qkv_bwd_scatter: NULL embedding recipe");
    return block_qkv_bwd_launch(bd, sc, stream, "cr_block_ln_qkv_bwd_scatter");
}
