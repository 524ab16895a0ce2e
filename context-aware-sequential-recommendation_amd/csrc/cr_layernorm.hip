// Layer normalisation (modules.py:53-80 `normalize`) forward / backward.
// One wavefront per row; row statistics by 64-lane xor-shuffle reductions.
#include "cr_common.hpp"

#define LN_MAXC 8   // columns per lane: D <= 512

__global__ __launch_bounds__(256) void k_ln_fwd(cr_ln_desc d) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= d.M) return;
    float x[LN_MAXC];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        x[i] = (c < d.D) ? d.x[(size_t)m * d.ldx + c] : 0.0f;
        s += x[i];
    }
    s = wave_sum(s);
    const float mean = s / (float)d.D;
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        const float dx = (c < d.D) ? (x[i] - mean) : 0.0f;
        v += dx * dx;
    }
    v = wave_sum(v) / (float)d.D;
    const float sd = sqrtf(v + d.eps);
    float ys = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < d.D) {
            const float y = d.gamma[c] * ((x[i] - mean) / sd) + d.beta[c];
            d.y[(size_t)m * d.ldy + c] = y;
            ys += y;
        }
    }
    if (d.y_nonzero) {
        ys = wave_sum(ys);
        if (lane == 0) d.y_nonzero[m] = (ys != 0.0f) ? 1.0f : 0.0f;
    }
    if (d.x_nonzero && lane == 0) d.x_nonzero[m] = (s != 0.0f) ? 1.0f : 0.0f;
}

extern "C" int cr_layernorm_fwd(const cr_ln_desc* d, void* stream) {
    CR_REQUIRE(d && d->x && d->y && d->gamma && d->beta, "cr_layernorm_fwd: NULL pointer");
    CR_REQUIRE(d->M > 0 && d->D > 0 && d->ldx >= d->D && d->ldy >= d->D, "cr_layernorm_fwd: bad shape");
    if (d->D > 64 * LN_MAXC) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_layernorm_fwd: D=%d > %d", d->D, 64 * LN_MAXC);
    hipLaunchKernelGGL(k_ln_fwd, dim3(cr_ceil_div(d->M, 4)), dim3(256), 0, cr_stream(stream), *d);
    return cr_check_launch("cr_layernorm_fwd");
}

// Backward.  n_slabs persistent workgroups; workgroup s owns rows [s*rps, (s+1)*rps) and writes
// its partial dgamma / dbeta into slab s (reduced later by cr_adam_step) -- no atomics.
__global__ __launch_bounds__(256) void k_ln_bwd(cr_ln_bwd_desc d) {
    __shared__ float red[2][4][64 * LN_MAXC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rps = (d.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * rps, m1 = min(d.M, m0 + rps);
    float g[LN_MAXC], ag[LN_MAXC], ab[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        g[i] = (c < d.D) ? d.gamma[c] : 0.0f;
        ag[i] = 0.0f; ab[i] = 0.0f;
    }
    const float invD = 1.0f / (float)d.D;
    for (int m = m0 + wave; m < m1; m += 4) {
        float x[LN_MAXC], dy[LN_MAXC];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            x[i] = (c < d.D) ? d.x[(size_t)m * d.ldx + c] : 0.0f;
            dy[i] = (c < d.D) ? d.dy[(size_t)m * d.lddy + c] : 0.0f;
            s += x[i];
        }
        const float mean = wave_sum(s) * invD;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            const float dx = (c < d.D) ? (x[i] - mean) : 0.0f;
            v += dx * dx;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(v) * invD + d.eps);
        float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            const float xh = (c < d.D) ? (x[i] - mean) * rstd : 0.0f;
            x[i] = xh;
            const float dg = dy[i] * g[i];
            c1 += dg;
            c2 += dg * xh;
            ag[i] += dy[i] * xh;
            ab[i] += dy[i];
        }
        c1 = wave_sum(c1) * invD;
        c2 = wave_sum(c2) * invD;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < d.D) {
                const float dx = rstd * (dy[i] * g[i] - c1 - x[i] * c2);
                float* p = d.dx + (size_t)m * d.lddx + c;
                *p = d.accumulate ? (*p + dx) : dx;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        red[0][wave][lane + 64 * i] = ag[i];
        red[1][wave][lane + 64 * i] = ab[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d.D; c += 256) {
        d.dgamma[(size_t)blockIdx.x * d.slab_stride + c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        d.dbeta[(size_t)blockIdx.x * d.slab_stride + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    }
}

extern "C" int cr_layernorm_bwd(const cr_ln_bwd_desc* d, void* stream) {
    CR_REQUIRE(d && d->x && d->gamma && d->dy && d->dx && d->dgamma && d->dbeta, "cr_layernorm_bwd: NULL pointer");
    CR_REQUIRE(d->M > 0 && d->D > 0 && d->n_slabs > 0, "cr_layernorm_bwd: bad shape");
    if (d->D > 64 * LN_MAXC) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_layernorm_bwd: D=%d > %d", d->D, 64 * LN_MAXC);
    hipLaunchKernelGGL(k_ln_bwd, dim3(d->n_slabs), dim3(256), 0, cr_stream(stream), *d);
    return cr_check_launch("cr_layernorm_bwd");
}
