// fp32 MFMA GEMMs for the dense layers (tf.layers.dense modules.py:203-205,333-334; conv1d k=1
// modules.py:300-310) -- forward / data-gradient (k_gemm_rows) and weight-gradient (k_gemm_wgrad).
//
// MFMA shape: v_mfma_f32_16x16x4_f32 (exact fp32).  Workgroup = 4 waves, 64x64 output tile, wave w
// owns rows [16w,16w+16) x 64 columns (4 accumulators, the A fragment is reused by 4 MFMAs).
// LDS pitches are chosen so the fragment reads (ds_read_b32, 32-lane groups) are conflict-free:
//   A-pattern  (lanes vary ROW by li, k by lg):  pitch % 4 == 2           (66)
//   B-pattern  (lanes vary COL by li, k by lg):  pitch % 32 == 16         (80)
// A K-chunk is 64 deep (one pass for hidden sizes <= 64); the next chunk is prefetched into
// registers while the MFMAs of the current one run.
#include "cr_common.hpp"

struct GemmBatch {
    cr_gemm_desc p[CR_MAX_BATCH];
};

#define G_BM 64
#define G_BN 64
#define G_KC 64
#define G_PA 66      // A-pattern pitch (% 4 == 2)
#define G_PB 80      // B-pattern pitch (% 32 == 16)
#define G_NLD ((G_BM * G_KC) / 256)   // elements per thread per staged tile (16)

__global__ __launch_bounds__(256) void k_gemm_rows(GemmBatch batch) {
    const cr_gemm_desc& d = batch.p[blockIdx.y];
    const int ntiles = (d.N + G_BN - 1) / G_BN;
    const int mtiles = (d.M + G_BM - 1) / G_BM;
    if ((int)blockIdx.x >= ntiles * mtiles) return;
    const int m0 = ((int)blockIdx.x / ntiles) * G_BM, n0 = ((int)blockIdx.x % ntiles) * G_BN;
    __shared__ float As[G_BM * G_PA];
    __shared__ float Bs[G_KC * G_PB];   // trans_b: used as Bt[64][G_PA] (64*66 = 4224 <= 5120)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int tr = tid >> 6, tc = tid & 63;     // staging coordinates: 4 rows x 64 columns per pass
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ra[G_NLD], rb[G_NLD];

    // global -> registers for the K-chunk starting at k0 (issued back to back: 32 loads in flight)
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < G_NLD; ++i) {
            const int r = tr + 4 * i;           // tile row (A: m, B: k or n)
            const int gm = m0 + r, gk = k0 + tc;
            ra[i] = (gm < d.M && gk < d.K) ? d.A[(size_t)gm * d.lda + gk] : 0.0f;
            if (!d.trans_b) {                   // B[k0 + r][n0 + tc]
                const int bk = k0 + r, bn = n0 + tc;
                rb[i] = (bk < d.K && bn < d.N) ? d.B[(size_t)bk * d.ldb + bn] : 0.0f;
            } else {                            // B[n0 + r][k0 + tc]
                const int bn = n0 + r, bk = k0 + tc;
                rb[i] = (bk < d.K && bn < d.N) ? d.B[(size_t)bn * d.ldb + bk] : 0.0f;
            }
        }
    };
    load_chunk(0);
    for (int k0 = 0; k0 < d.K; k0 += G_KC) {
#pragma unroll
        for (int i = 0; i < G_NLD; ++i) {
            const int r = tr + 4 * i;
            As[r * G_PA + tc] = ra[i];
            if (!d.trans_b) Bs[r * G_PB + tc] = rb[i];
            else Bs[r * G_PA + tc] = rb[i];
        }
        __syncthreads();
        if (k0 + G_KC < d.K) load_chunk(k0 + G_KC);      // prefetch under the MFMAs
        const int ksteps = min(G_KC, d.K - k0);
        if (!d.trans_b) {
            for (int kk = 0; kk < ksteps; kk += 4) {
                const float a = As[(16 * wave + li) * G_PA + kk + lg];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = mfma16(a, Bs[(kk + lg) * G_PB + 16 * j + li], acc[j]);
            }
        } else {
            for (int kk = 0; kk < ksteps; kk += 4) {
                const float a = As[(16 * wave + li) * G_PA + kk + lg];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = mfma16(a, Bs[(16 * j + li) * G_PA + kk + lg], acc[j]);
            }
        }
        __syncthreads();
    }

    const DropCtx dc = drop_ctx(d.drop);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + 16 * j + li;
        if (col >= d.N) continue;
        const float bias = d.bias ? d.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * wave + 4 * lg + r;
            if (row >= d.M) continue;
            float v = acc[j][r] + bias;
            if (d.relu) v = fmaxf(v, 0.0f);
            v = drop_apply(dc, (d.drop.row_offset + (uint32_t)row) * (uint32_t)d.N + (uint32_t)col, v);
            if (d.residual) v += d.residual[(size_t)row * d.ldr + col];
            if (d.mask_ids && d.mask_ids[row] == 0) v = 0.0f;
            float* p = d.C + (size_t)row * d.ldc + col;
            *p = d.accumulate ? (*p + v) : v;
        }
    }
}

// bf16-MFMA forms (cr_gemm_bf.hip)
bool cr_gemm_rows_bf_supported(const cr_gemm_desc* d, int n);
int cr_gemm_rows_bf_launch(const cr_gemm_desc* d, int n, hipStream_t s);
bool cr_gemm_wgrad_bf_supported(const cr_wgrad_desc* d, int n);
int cr_gemm_wgrad_bf_launch(const cr_wgrad_desc* d, int n, int slab_stride, int n_slabs, hipStream_t s);

extern "C" int cr_gemm_rows(const cr_gemm_desc* d, int n, void* stream) {
    CR_REQUIRE(d && n >= 1 && n <= CR_MAX_BATCH, "cr_gemm_rows: n_problems=%d out of [1,%d]", n, CR_MAX_BATCH);
    GemmBatch b;
    int maxtiles = 0;
    for (int i = 0; i < n; ++i) {
        CR_REQUIRE(d[i].A && d[i].B && d[i].C, "cr_gemm_rows[%d]: NULL pointer", i);
        CR_REQUIRE(d[i].M > 0 && d[i].N > 0 && d[i].K > 0, "cr_gemm_rows[%d]: bad shape %dx%dx%d", i, d[i].M, d[i].N, d[i].K);
        CR_REQUIRE(d[i].lda >= d[i].K && d[i].ldc >= d[i].N, "cr_gemm_rows[%d]: leading dimension too small", i);
        CR_REQUIRE(d[i].ldb >= (d[i].trans_b ? d[i].K : d[i].N), "cr_gemm_rows[%d]: ldb too small", i);
        CR_REQUIRE(d[i].precision >= CR_PREC_F32 && d[i].precision <= CR_PREC_BF16, "cr_gemm_rows[%d]: unknown precision %d", i, d[i].precision);
        b.p[i] = d[i];
        const int tiles = cr_ceil_div(d[i].M, G_BM) * cr_ceil_div(d[i].N, G_BN);
        if (tiles > maxtiles) maxtiles = tiles;
    }
    if (cr_gemm_rows_bf_supported(d, n)) return cr_gemm_rows_bf_launch(d, n, cr_stream(stream));
    for (int i = n; i < CR_MAX_BATCH; ++i) b.p[i] = d[0];
    hipLaunchKernelGGL(k_gemm_rows, dim3(maxtiles, n), dim3(256), 0, cr_stream(stream), b);
    return cr_check_launch("cr_gemm_rows");
}

// -------------------------------------------------------------------------------------------
// Weight gradient: dW[K,N] = A^T G, db = colsum(G).  The M reduction is split over gridDim.y
// workgroups; workgroup s reduces rows [s*rps, (s+1)*rps) and writes slab s.
// -------------------------------------------------------------------------------------------
struct WgradBatch {
    cr_wgrad_desc p[CR_MAX_BATCH];
    int n;
    int slab_stride;
};

#define W_MC 32
#define W_NLD ((W_MC * 64) / 256)     // 8 elements per thread per staged tile

__global__ __launch_bounds__(256) void k_gemm_wgrad(WgradBatch batch) {
    int bx = blockIdx.x, pi = 0;
    for (; pi < batch.n; ++pi) {
        const int cnt = ((batch.p[pi].K + 63) / 64) * ((batch.p[pi].N + 63) / 64);
        if (bx < cnt) break;
        bx -= cnt;
    }
    if (pi >= batch.n) return;
    const cr_wgrad_desc& d = batch.p[pi];
    const int ntiles = (d.N + 63) / 64;
    const int k0 = (bx / ntiles) * 64, n0 = (bx % ntiles) * 64;
    const int s = blockIdx.y;
    const int rps = (d.M + gridDim.y - 1) / gridDim.y;
    const int mb = s * rps, me = min(d.M, mb + rps);
    __shared__ float As[W_MC * G_PB];
    __shared__ float Gs[W_MC * G_PB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int tr = tid >> 6, tc = tid & 63;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    float ra[W_NLD], rg[W_NLD];
    auto load_chunk = [&](int mc) {
#pragma unroll
        for (int i = 0; i < W_NLD; ++i) {
            const int gm = mc + tr + 4 * i;
            ra[i] = (gm < me && k0 + tc < d.K) ? d.A[(size_t)gm * d.lda + k0 + tc] : 0.0f;
            rg[i] = (gm < me && n0 + tc < d.N) ? d.G[(size_t)gm * d.ldg + n0 + tc] : 0.0f;
        }
    };
    if (mb < me) load_chunk(mb);
    for (int mc = mb; mc < me; mc += W_MC) {
#pragma unroll
        for (int i = 0; i < W_NLD; ++i) {
            As[(tr + 4 * i) * G_PB + tc] = ra[i];
            Gs[(tr + 4 * i) * G_PB + tc] = rg[i];
        }
        __syncthreads();
        if (mc + W_MC < me) load_chunk(mc + W_MC);       // prefetch under the MFMAs
#pragma unroll
        for (int mm = 0; mm < W_MC; mm += 4) {
            const float a = As[(mm + lg) * G_PB + 16 * wave + li];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = mfma16(a, Gs[(mm + lg) * G_PB + 16 * j + li], acc[j]);
        }
        if (k0 == 0 && tid < 64) {
#pragma unroll
            for (int mm = 0; mm < W_MC; ++mm) bsum += Gs[mm * G_PB + tid];
        }
        __syncthreads();
    }
    float* dW = d.dW + (size_t)s * batch.slab_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + 16 * j + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int krow = k0 + 16 * wave + 4 * lg + r;
            if (krow < d.K && col < d.N) dW[(size_t)krow * d.ldw + col] = acc[j][r];
        }
    }
    if (d.db && k0 == 0 && tid < 64 && n0 + tid < d.N) d.db[(size_t)s * batch.slab_stride + n0 + tid] = bsum;
}

extern "C" int cr_gemm_wgrad(const cr_wgrad_desc* d, int n, int slab_stride, int n_slabs, void* stream) {
    CR_REQUIRE(d && n >= 1 && n <= CR_MAX_BATCH, "cr_gemm_wgrad: n_problems=%d out of [1,%d]", n, CR_MAX_BATCH);
    CR_REQUIRE(n_slabs >= 1 && slab_stride >= 0, "cr_gemm_wgrad: bad slab geometry");
    WgradBatch b;
    b.n = n;
    b.slab_stride = slab_stride;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        CR_REQUIRE(d[i].A && d[i].G && d[i].dW, "cr_gemm_wgrad[%d]: NULL pointer", i);
        CR_REQUIRE(d[i].M > 0 && d[i].N > 0 && d[i].K > 0, "cr_gemm_wgrad[%d]: bad shape", i);
        CR_REQUIRE(d[i].lda >= d[i].K && d[i].ldg >= d[i].N && d[i].ldw >= d[i].N, "cr_gemm_wgrad[%d]: leading dimension too small", i);
        b.p[i] = d[i];
        tiles += cr_ceil_div(d[i].K, 64) * cr_ceil_div(d[i].N, 64);
    }
    if (cr_gemm_wgrad_bf_supported(d, n)) return cr_gemm_wgrad_bf_launch(d, n, slab_stride, n_slabs, cr_stream(stream));
    for (int i = n; i < CR_MAX_BATCH; ++i) b.p[i] = d[0];
    hipLaunchKernelGGL(k_gemm_wgrad, dim3(tiles, n_slabs), dim3(256), 0, cr_stream(stream), b);
    return cr_check_launch("cr_gemm_wgrad");
}
