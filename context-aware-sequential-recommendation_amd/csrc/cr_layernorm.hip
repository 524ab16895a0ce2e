// Layer normalisation (modules.py:53-80 `normalize`) forward / backward.
// Rows are reduced by xor-shuffles.  For the small hidden sizes of this model family (D <= 64) a
// 64-lane wave would idle on most lanes and serialise on the row's dependent reduction chain, so a
// row gets LPR = 16 lanes and a wave works on 4 rows at once; D > 64 uses one wave per row.
#include "cr_common.hpp"

template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
    if (LPR == 16) return cr_row16_sum(v);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int LPR, int MAXC>
__global__ __launch_bounds__(256) void k_ln_fwd(cr_ln_desc d) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    const int m = (blockIdx.x * 4 + wave) * RPW + sub;
    const bool act = m < d.M;
    const int mm = act ? m : d.M - 1;
    float x[MAXC];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = l + LPR * i;
        x[i] = (c < d.D) ? d.x[(size_t)mm * d.ldx + c] : 0.0f;
        s += x[i];
    }
    s = row_sum<LPR>(s);
    const float mean = s / (float)d.D;
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = l + LPR * i;
        const float dx = (c < d.D) ? (x[i] - mean) : 0.0f;
        v += dx * dx;
    }
    v = row_sum<LPR>(v) / (float)d.D;
    const float sd = sqrtf(v + d.eps);
    float ys = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = l + LPR * i;
        if (c < d.D) {
            const float y = d.gamma[c] * ((x[i] - mean) / sd) + d.beta[c];
            if (act) d.y[(size_t)m * d.ldy + c] = y;
            ys += y;
        }
    }
    ys = row_sum<LPR>(ys);
    if (act && l == 0) {
        if (d.y_nonzero) d.y_nonzero[m] = (ys != 0.0f) ? 1.0f : 0.0f;
        if (d.x_nonzero) d.x_nonzero[m] = (s != 0.0f) ? 1.0f : 0.0f;
    }
}

extern "C" int cr_layernorm_fwd(const cr_ln_desc* d, void* stream) {
    CR_REQUIRE(d && d->x && d->y && d->gamma && d->beta, "cr_layernorm_fwd: NULL pointer");
    CR_REQUIRE(d->M > 0 && d->D > 0 && d->ldx >= d->D && d->ldy >= d->D, "cr_layernorm_fwd: bad shape");
    if (d->D > 512) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_layernorm_fwd: D=%d > 512", d->D);
    if (d->D <= 64)
        hipLaunchKernelGGL((k_ln_fwd<16, 4>), dim3(cr_ceil_div(d->M, 16)), dim3(256), 0, cr_stream(stream), *d);
    else
        hipLaunchKernelGGL((k_ln_fwd<64, 8>), dim3(cr_ceil_div(d->M, 4)), dim3(256), 0, cr_stream(stream), *d);
    return cr_check_launch("cr_layernorm_fwd");
}

// Backward.  n_slabs workgroups of 16 waves; workgroup s owns rows [s*rps, (s+1)*rps) and writes its
// partial dgamma / dbeta into slab s (summed later by cr_adam_step) -- no global atomics.
template <int LPR, int MAXC>
__global__ __launch_bounds__(1024) void k_ln_bwd(cr_ln_bwd_desc d) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    const int rps = (d.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * rps, m1 = min(d.M, m0 + rps);
    float g[MAXC], ag[MAXC], ab[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = l + LPR * i;
        g[i] = (c < d.D) ? d.gamma[c] : 0.0f;
        ag[i] = 0.0f; ab[i] = 0.0f;
    }
    const float invD = 1.0f / (float)d.D;
    for (int mb = m0 + wave * RPW; mb < m1; mb += 16 * RPW) {
        const int m = mb + sub;
        const bool act = m < m1;
        const int mm = act ? m : m0;
        float x[MAXC], dy[MAXC];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = l + LPR * i;
            x[i] = (c < d.D) ? d.x[(size_t)mm * d.ldx + c] : 0.0f;
            dy[i] = (c < d.D && act) ? d.dy[(size_t)mm * d.lddy + c] : 0.0f;
            s += x[i];
        }
        const float mean = row_sum<LPR>(s) * invD;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = l + LPR * i;
            const float dx = (c < d.D) ? (x[i] - mean) : 0.0f;
            v += dx * dx;
        }
        const float rstd = 1.0f / sqrtf(row_sum<LPR>(v) * invD + d.eps);
        float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = l + LPR * i;
            const float xh = (c < d.D) ? (x[i] - mean) * rstd : 0.0f;
            x[i] = xh;
            const float dg = dy[i] * g[i];
            c1 += dg;
            c2 += dg * xh;
            ag[i] += dy[i] * xh;
            ab[i] += dy[i];
        }
        c1 = row_sum<LPR>(c1) * invD;
        c2 = row_sum<LPR>(c2) * invD;
        if (act) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = l + LPR * i;
                if (c < d.D) {
                    const float dx = rstd * (dy[i] * g[i] - c1 - x[i] * c2);
                    float* p = d.dx + (size_t)m * d.lddx + c;
                    *p = d.accumulate ? (*p + dx) : dx;
                }
            }
        }
    }
    // fold the RPW row groups of the wave (shuffles), then the 16 waves through per-wave LDS slots summed
    // in a fixed order (bitwise reproducible)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
            ag[i] += __shfl_xor(ag[i], o, 64);
            ab[i] += __shfl_xor(ab[i], o, 64);
        }
    }
    if (LPR == 16) {
        // D <= 64: every wave has its own [64]-float slot, one barrier, then column c is summed over the 16 slots in a
        // fixed order by one thread (the wave-by-wave fold below costs 16 barriers: ~4 us of a 10.8 us kernel)
        __shared__ float wg[16][64], wb[16][64];
        if (sub == 0) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = l + LPR * i;
                if (c < 64) { wg[wave][c] = ag[i]; wb[wave][c] = ab[i]; }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < d.D) {
            float g = 0.0f, b = 0.0f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { g += wg[w][threadIdx.x]; b += wb[w][threadIdx.x]; }
            d.dgamma[(size_t)blockIdx.x * d.slab_stride + threadIdx.x] = g;
            d.dbeta[(size_t)blockIdx.x * d.slab_stride + threadIdx.x] = b;
        }
        return;
    }
    // D <= 512: the same with [16][512] slots, gamma and beta one after the other through one 32 KiB array
    __shared__ float wide[16][512];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (sub == 0) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = l + LPR * i;
                if (c < 512) wide[wave][c] = pass == 0 ? ag[i] : ab[i];
            }
        }
        __syncthreads();
        float* dst = pass == 0 ? d.dgamma : d.dbeta;
        for (int c = threadIdx.x; c < d.D; c += 1024) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 16; ++w) v += wide[w][c];
            dst[(size_t)blockIdx.x * d.slab_stride + c] = v;
        }
        __syncthreads();
    }
}

extern "C" int cr_layernorm_bwd(const cr_ln_bwd_desc* d, void* stream) {
    CR_REQUIRE(d && d->x && d->gamma && d->dy && d->dx && d->dgamma && d->dbeta, "cr_layernorm_bwd: NULL pointer");
    CR_REQUIRE(d->M > 0 && d->D > 0 && d->n_slabs > 0, "cr_layernorm_bwd: bad shape");
    if (d->D > 512) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_layernorm_bwd: D=%d > 512", d->D);
    if (d->D <= 64)
        hipLaunchKernelGGL((k_ln_bwd<16, 4>), dim3(d->n_slabs), dim3(1024), 0, cr_stream(stream), *d);
    else
        hipLaunchKernelGGL((k_ln_bwd<64, 8>), dim3(d->n_slabs), dim3(1024), 0, cr_stream(stream), *d);
    return cr_check_launch("cr_layernorm_bwd");
}
