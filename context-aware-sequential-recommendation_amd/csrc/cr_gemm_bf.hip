// Dense-layer GEMMs on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, fp32 accumulation) for the UNFUSED row phases
// (hidden sizes above 64: configs C4 / C5; the CAST mlp): forward / data-gradient (cr_gemm_rows) and weight-gradient
// (cr_gemm_wgrad) when the descriptor's `precision` is CR_PREC_BF16X3 (fp32 operands split into bf16 hi + lo, three
// products: ~1e-5) or CR_PREC_BF16.  The exact fp32 kernels of cr_gemm.hip stay the CR_PREC_F32 path and take the
// shapes these do not (K or N below 8).
//
// Why: the fp32 kernels issue one ds_read_b32 per operand per 16x16x4 MFMA and stage with dword loads; at
// M = 25 600, N = K = 128 they run 48.8 us per product (17 TF/s), and the C4 step is 28 of them.  Here a 64 x 64 output
// tile takes a 64-deep K chunk as two [64][64] bf16 images (cr_bf16.hpp: 16-byte global loads converted on the fly,
// conflict-free swizzle) and 24 (split) MFMAs per wave with 16-byte / hardware-transposed operand reads:
//   C = A B      B is [K, N]: B's operand has k on the image ROW -> transposed reads, whose k order within a 32-step
//                (rows 4 lg + j and 16 + 4 lg + j) the A operand follows with two 8-byte reads of its row;
//   C = A B^T    B is [N, K]: both operands are plain row reads;
//   dW = A^T G   contraction over rows: both operands are transposed reads of the two row images.
#include "cr_bf16.hpp"

struct GemmBatchBf {
    cr_gemm_desc p[CR_MAX_BATCH];
};

// Stage a [64 rows][64 columns] tile of a row-major fp32 matrix into an image pair: rows row0.., columns col0.. .
// rows_total / cols_total bound the matrix (zero fill beyond); all loads of the tile are in flight before the first
// conversion (512 items for 256 threads: two per thread).
template <bool SPLIT>
__device__ __forceinline__ void stage_tile(__bf16* hi, __bf16* lo, const float* src, int ld, int row0, int rows_total,
                                           int col0, int cols_total) {
    const int d = min(64, cols_total - col0);            // valid columns of this tile (>= 1)
    float v[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int item = threadIdx.x + 256 * u;
        const int r = item >> 3, ch = item & 7;
        const bool rok = row0 + r < rows_total;
        const int grow = rok ? row0 + r : row0;
        item_issue(v[u], src + (size_t)grow * ld + col0, 8 * ch, d, item_fix(rok, grow == rows_total - 1, 8 * ch, d));
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int item = threadIdx.x + 256 * u;
        const int r = item >> 3, ch = item & 7;
        const bool rok = row0 + r < rows_total;
        const int grow = rok ? row0 + r : row0;
        const bool fix = item_fix(rok, grow == rows_total - 1, 8 * ch, d);
        item_mask(v[u], 8 * ch, d, rok, fix);
        if (__builtin_expect(fix, 0)) item_refill(v[u], src + (size_t)grow * ld + col0, 8 * ch, d);
        bf8 h, l;
        split8<SPLIT>(v[u], h, l);
        *reinterpret_cast<bf8*>(hi + img_off(r, ch)) = h;
        if (SPLIT) *reinterpret_cast<bf8*>(lo + img_off(r, ch)) = l;
    }
}

// stage_tile in two halves, so that a tile's loads fly while the previous tile is multiplied
struct TileRegs { float v[2][8]; };
__device__ __forceinline__ void stage_issue(TileRegs& t, const float* src, int ld, int row0, int rows_total, int col0, int cols_total) {
    const int d = min(64, cols_total - col0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int item = threadIdx.x + 256 * u;
        const int r = item >> 3, ch = item & 7;
        const bool rok = row0 + r < rows_total;
        const int grow = rok ? row0 + r : row0;
        item_issue(t.v[u], src + (size_t)grow * ld + col0, 8 * ch, d, item_fix(rok, grow == rows_total - 1, 8 * ch, d));
    }
}
template <bool SPLIT>
__device__ __forceinline__ void stage_put(TileRegs& t, __bf16* hi, __bf16* lo, const float* src, int ld, int row0, int rows_total,
                                          int col0, int cols_total) {
    const int d = min(64, cols_total - col0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int item = threadIdx.x + 256 * u;
        const int r = item >> 3, ch = item & 7;
        const bool rok = row0 + r < rows_total;
        const int grow = rok ? row0 + r : row0;
        const bool fix = item_fix(rok, grow == rows_total - 1, 8 * ch, d);
        item_mask(t.v[u], 8 * ch, d, rok, fix);
        if (__builtin_expect(fix, 0)) item_refill(t.v[u], src + (size_t)grow * ld + col0, 8 * ch, d);
        bf8 h, l;
        split8<SPLIT>(t.v[u], h, l);
        *reinterpret_cast<bf8*>(hi + img_off(r, ch)) = h;
        if (SPLIT) *reinterpret_cast<bf8*>(lo + img_off(r, ch)) = l;
    }
}

// A operand whose k order follows tr_frag's: element j <-> column 32 ks + 16 (j >> 2) + 4 lg + (j & 3) of row row0 + li
__device__ __forceinline__ bf8 row_frag_perm(const __bf16* img, int row0, int ks) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const int ch = 4 * ks + (lg >> 1), sub = 4 * (lg & 1);
    const bf4 t0 = *reinterpret_cast<const bf4*>(img + img_off(row0 + li, ch) + sub);
    const bf4 t1 = *reinterpret_cast<const bf4*>(img + img_off(row0 + li, ch + 2) + sub);
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void k_gemm_rows_bf(GemmBatchBf batch) {
    const cr_gemm_desc& d = batch.p[blockIdx.y];
    const int ntiles = (d.N + 63) / 64, mtiles = (d.M + 63) / 64;
    if ((int)blockIdx.x >= ntiles * mtiles) return;
    const int m0 = ((int)blockIdx.x / ntiles) * 64, n0 = ((int)blockIdx.x % ntiles) * 64;
    __shared__ __attribute__((aligned(16))) __bf16 smem[4 * 64 * 64];
    __bf16* Ah = smem;
    __bf16* Al = Ah + (SPLIT ? 64 * 64 : 0);
    __bf16* Bh = Al + 64 * 64;
    __bf16* Bl = Bh + (SPLIT ? 64 * 64 : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the next K chunk's loads are requested before this chunk's products (stage -> barrier -> multiply in series exposed one
    // memory round trip per chunk)
    TileRegs ta, tb;
    stage_issue(ta, d.A, d.lda, m0, d.M, 0, d.K);
    if (!d.trans_b) stage_issue(tb, d.B, d.ldb, 0, d.K, n0, d.N);
    else stage_issue(tb, d.B, d.ldb, n0, d.N, 0, d.K);
    for (int k0 = 0; k0 < d.K; k0 += 64) {
        if (k0) __syncthreads();
        stage_put<SPLIT>(ta, Ah, Al, d.A, d.lda, m0, d.M, k0, d.K);
        if (!d.trans_b) stage_put<SPLIT>(tb, Bh, Bl, d.B, d.ldb, k0, d.K, n0, d.N);   // image [k][n]
        else stage_put<SPLIT>(tb, Bh, Bl, d.B, d.ldb, n0, d.N, k0, d.K);               // image [n][k]
        __syncthreads();
        if (k0 + 64 < d.K) {
            stage_issue(ta, d.A, d.lda, m0, d.M, k0 + 64, d.K);
            if (!d.trans_b) stage_issue(tb, d.B, d.ldb, k0 + 64, d.K, n0, d.N);
            else stage_issue(tb, d.B, d.ldb, n0, d.N, k0 + 64, d.K);
        }
        const int ksteps = (min(64, d.K - k0) + 31) / 32;
        for (int ks = 0; ks < ksteps; ++ks) {
            bf8 ah, al, bh[4], bl[4];
            if (!d.trans_b) {
                ah = row_frag_perm(Ah, 16 * wave, ks);
                al = SPLIT ? row_frag_perm(Al, 16 * wave, ks) : ah;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bh[j] = tr_frag(Bh, 32 * ks, 32 * ks + 16, j);
                    bl[j] = SPLIT ? tr_frag(Bl, 32 * ks, 32 * ks + 16, j) : bh[j];
                }
            } else {
                ah = row_frag(Ah, 16 * wave, ks);
                al = SPLIT ? row_frag(Al, 16 * wave, ks) : ah;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bh[j] = row_frag(Bh, 16 * j, ks);
                    bl[j] = SPLIT ? row_frag(Bl, 16 * j, ks) : bh[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = mma<SPLIT>(ah, al, bh[j], bl[j], acc[j]);
        }
    }
    // epilogue (order as cr_gemm.hip): +bias -> relu -> dropout -> +residual -> *row mask -> (accumulate)
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

bool cr_gemm_rows_bf_supported(const cr_gemm_desc* d, int n) {
    for (int i = 0; i < n; ++i) {
        if (d[i].precision == CR_PREC_F32 || d[i].precision != d[0].precision) return false;
        if (d[i].K < 8 || d[i].N < 8) return false;       // the 8-float items need 8 columns in a row
    }
    return true;
}

int cr_gemm_rows_bf_launch(const cr_gemm_desc* d, int n, hipStream_t s) {
    GemmBatchBf b;
    int maxtiles = 0;
    for (int i = 0; i < n; ++i) {
        b.p[i] = d[i];
        const int tiles = cr_ceil_div(d[i].M, 64) * cr_ceil_div(d[i].N, 64);
        if (tiles > maxtiles) maxtiles = tiles;
    }
    for (int i = n; i < CR_MAX_BATCH; ++i) b.p[i] = d[0];
    if (d[0].precision == CR_PREC_BF16X3) hipLaunchKernelGGL(k_gemm_rows_bf<true>, dim3(maxtiles, n), dim3(256), 0, s, b);
    else hipLaunchKernelGGL(k_gemm_rows_bf<false>, dim3(maxtiles, n), dim3(256), 0, s, b);
    return cr_check_launch("cr_gemm_rows(bf16)");
}

// -------------------------------------------------------------------------------------------
// Weight gradient: dW[K,N] = A^T G, db = colsum(G); the M reduction is split over gridDim.y workgroups, workgroup s
// reduces rows [s*rps, (s+1)*rps) and writes slab s (no atomics, bitwise reproducible).
// -------------------------------------------------------------------------------------------
struct WgradBatchBf {
    cr_wgrad_desc p[CR_MAX_BATCH];
    int n;
    int slab_stride;
};

template <bool SPLIT>
__global__ __launch_bounds__(256) void k_gemm_wgrad_bf(WgradBatchBf batch) {
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
    __shared__ __attribute__((aligned(16))) __bf16 smem[4 * 64 * 64];
    __bf16* Ah = smem;
    __bf16* Al = Ah + (SPLIT ? 64 * 64 : 0);
    __bf16* Gh = Al + 64 * 64;
    __bf16* Gl = Gh + (SPLIT ? 64 * 64 : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    // the next 64-row chunk's loads are requested before this chunk's products (with 32 slabs a workgroup walks 13 chunks at
    // the C4 shape; stage -> barrier -> multiply in series left the memory pipe idle during the products and vice versa)
    TileRegs ta, tg;
    if (mb < me) {
        stage_issue(ta, d.A, d.lda, mb, me, k0, d.K);                  // rows beyond `me` belong to the next slab: zero filled
        stage_issue(tg, d.G, d.ldg, mb, me, n0, d.N);
    }
    for (int mc = mb; mc < me; mc += 64) {
        if (mc != mb) __syncthreads();
        stage_put<SPLIT>(ta, Ah, Al, d.A, d.lda, mc, me, k0, d.K);
        stage_put<SPLIT>(tg, Gh, Gl, d.G, d.ldg, mc, me, n0, d.N);
        __syncthreads();
        if (mc + 64 < me) {
            stage_issue(ta, d.A, d.lda, mc + 64, me, k0, d.K);
            stage_issue(tg, d.G, d.ldg, mc + 64, me, n0, d.N);
        }
        const int msteps = (min(64, me - mc) + 31) / 32;
        for (int ms = 0; ms < msteps; ++ms) {
            const bf8 ah = tr_frag(Ah, 32 * ms, 32 * ms + 16, wave);  // A^T[k = 16 wave + li][rows of the step]
            const bf8 al = SPLIT ? tr_frag(Al, 32 * ms, 32 * ms + 16, wave) : ah;
            bf8 gh[4], gl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gh[j] = tr_frag(Gh, 32 * ms, 32 * ms + 16, j);
                gl[j] = SPLIT ? tr_frag(Gl, 32 * ms, 32 * ms + 16, j) : gh[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = mma<SPLIT>(ah, al, gh[j], gl[j], acc[j]);
        }
        if (d.db && k0 == 0) {                                // column sums of G from the images (hi + lo = the fp32 value to 2^-17):
            const int c = threadIdx.x & 63, r0 = 16 * wave;   // wave w takes rows 16 w .. 16 w + 15 (one wave doing all 64 was the
#pragma unroll 4                                              // longest chain of its workgroup: 128 dependent LDS reads per chunk)
            for (int r = r0; r < r0 + 16; ++r) {
                const int o = img_off(r, c >> 3) + (c & 7);
                bsum += (float)Gh[o] + (SPLIT ? (float)Gl[o] : 0.0f);
            }
        }
    }
    if (d.db && k0 == 0) {                                    // fold the four waves' partial column sums, wave order
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        red[threadIdx.x] = bsum;
        __syncthreads();
        if (threadIdx.x < 64) bsum = (red[threadIdx.x] + red[64 + threadIdx.x]) + (red[128 + threadIdx.x] + red[192 + threadIdx.x]);
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
    if (d.db && k0 == 0 && threadIdx.x < 64 && n0 + (int)threadIdx.x < d.N) d.db[(size_t)s * batch.slab_stride + n0 + threadIdx.x] = bsum;
}

bool cr_gemm_wgrad_bf_supported(const cr_wgrad_desc* d, int n) {
    for (int i = 0; i < n; ++i) {
        if (d[i].precision == CR_PREC_F32 || d[i].precision != d[0].precision) return false;
        if (d[i].K < 8 || d[i].N < 8) return false;
    }
    return true;
}

int cr_gemm_wgrad_bf_launch(const cr_wgrad_desc* d, int n, int slab_stride, int n_slabs, hipStream_t s) {
    WgradBatchBf b;
    b.n = n;
    b.slab_stride = slab_stride;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        b.p[i] = d[i];
        tiles += cr_ceil_div(d[i].K, 64) * cr_ceil_div(d[i].N, 64);
    }
    for (int i = n; i < CR_MAX_BATCH; ++i) b.p[i] = d[0];
    if (d[0].precision == CR_PREC_BF16X3) hipLaunchKernelGGL(k_gemm_wgrad_bf<true>, dim3(tiles, n_slabs), dim3(256), 0, s, b);
    else hipLaunchKernelGGL(k_gemm_wgrad_bf<false>, dim3(tiles, n_slabs), dim3(256), 0, s, b);
    return cr_check_launch("cr_gemm_wgrad(bf16)");
}
