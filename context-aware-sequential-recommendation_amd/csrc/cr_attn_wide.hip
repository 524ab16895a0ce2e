// General-shape fallback of the attention core (modules.py:208-269) for shapes outside the envelope of the
// LDS-resident MFMA kernels (T > 256 or head dim > 64; e.g. BASELINE config 5: maxlen 512).
// One wavefront per (sample, head, query row): lanes own keys (k = lane + 64 j) for scores / softmax and
// columns (c = lane + 64 i) for the value products; K / V rows are read through L2.  dK and dV are
// accumulated with float atomics (zeroed by the host first), so summation order -- only here -- is not fixed.
// Semantics are identical to the MFMA kernels (finite -2^32+1 fill => uniform 1/T rows, query mask,
// counter-based dropout, dead rows).  Correctness first: this path is not tuned.
#include <math.h>

#include "cr_attn_common.hpp"

#define W_MAXK 16     // keys per lane: T <= 1024
#define W_MAXC 4      // columns per lane: head dim <= 256

struct WideRow {
    float p[W_MAXK];  // softmax probability (before query mask / dropout) of key lane + 64 j
    float mx, inv;
    bool uniform;
};

// scores + masks + softmax of one query row (all lanes participate)
__device__ __forceinline__ void wide_softmax(const cr_attn_desc& d, const float* Qrow, const float* Kbase, const float* kvalid,
                                             int q, bool is_dead, float isd, WideRow& r) {
    const int lane = threadIdx.x & 63;
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < W_MAXK; ++j) {
        const int k = lane + 64 * j;
        float s = -INFINITY;
        if (k <= q && k < d.T && !is_dead && kvalid[k] != 0.0f) {        // key mask + causal (modules.py:222-241)
            const float* Kr = Kbase + (size_t)k * d.ld;
            float acc = 0.0f;
            for (int c = 0; c < d.d; ++c) acc = fmaf(Qrow[c], Kr[c], acc);
            s = acc * isd;                                                // modules.py:219
        }
        r.p[j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    r.uniform = (mx == -INFINITY) && !is_dead;
    const float off = (mx == -INFINITY) ? 0.0f : mx;
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < W_MAXK; ++j) {
        const float e = (r.p[j] == -INFINITY) ? 0.0f : expf(r.p[j] - off);
        r.p[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    r.inv = sum > 0.0f ? 1.0f / sum : 0.0f;
    r.mx = mx;
    const float invT = 1.0f / (float)d.T;
#pragma unroll
    for (int j = 0; j < W_MAXK; ++j) {
        const int k = lane + 64 * j;
        r.p[j] = r.uniform ? ((k < d.T) ? invT : 0.0f) : r.p[j] * r.inv;   // modules.py:227-244
    }
}

__global__ __launch_bounds__(256) void k_attn_wide_fwd(cr_attn_desc d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    const int head = blockIdx.y / d.B, n = blockIdx.y % d.B;
    if (q >= d.T) return;
    const int base_row = n * d.T, hoff = head * d.d;
    const size_t row = (size_t)(base_row + q);
    const bool is_dead = d.dead_ids && d.dead_ids[row] == 0;
    const float isd = (float)(1.0 / sqrt((double)d.d));
    WideRow r;
    wide_softmax(d, d.Q + row * d.ld + hoff, d.K + (size_t)base_row * d.ld + hoff, d.k_valid + base_row, q, is_dead, isd, r);
    const float qv = d.q_valid[row];
    const DropCtx dc = drop_ctx(d.drop);
    const uint32_t ridx = attn_row_idx(d, head, n, q);
#pragma unroll
    for (int j = 0; j < W_MAXK; ++j) {
        const int k = lane + 64 * j;
        float p = r.p[j] * qv;                                                           // modules.py:248-253
        if (dc.on) p *= drop_factor(dc, ridx + (uint32_t)k);                              // modules.py:256-257
        r.p[j] = p;
        if (d.attn_weights && k < d.T) d.attn_weights[((size_t)blockIdx.y * d.T + q) * d.T + k] = p;   // modules.py:259
    }
    // out[c] = sum_k p[k] V[k][c] + residual   (modules.py:262-269)
    float acc[W_MAXC] = {0.f, 0.f, 0.f, 0.f};
    const int kend = r.uniform ? d.T : q + 1;
    const float* Vb = d.V + (size_t)base_row * d.ld + hoff;
#pragma unroll
    for (int j = 0; j < W_MAXK; ++j) {
        if (64 * j >= kend) break;
        for (int kk = 0; kk < 64 && 64 * j + kk < kend; ++kk) {
            const float pk = __shfl(r.p[j], kk, 64);
            if (pk != 0.0f) {
                const float* Vr = Vb + (size_t)(64 * j + kk) * d.ld;
#pragma unroll
                for (int i = 0; i < W_MAXC; ++i) {
                    const int c = lane + 64 * i;
                    if (c < d.d) acc[i] = fmaf(pk, Vr[c], acc[i]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < W_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < d.d) d.out[row * d.ldo + hoff + c] = acc[i] + d.residual[row * d.ldr + hoff + c];
    }
}

__global__ __launch_bounds__(256) void k_attn_wide_bwd(cr_attn_bwd_desc bd) {
    const cr_attn_desc& d = bd.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    const int head = blockIdx.y / d.B, n = blockIdx.y % d.B;
    if (q >= d.T) return;
    const int base_row = n * d.T, hoff = head * d.d;
    const size_t row = (size_t)(base_row + q);
    const bool is_dead = d.dead_ids && d.dead_ids[row] == 0;
    const float isd = (float)(1.0 / sqrt((double)d.d));
    const float* Qrow = d.Q + row * d.ld + hoff;
    const float* dOrow = bd.dout + row * bd.lddo + hoff;
    const float* Kb = d.K + (size_t)base_row * d.ld + hoff;
    const float* Vb = d.V + (size_t)base_row * d.ld + hoff;
    WideRow r;
    wide_softmax(d, Qrow, Kb, d.k_valid + base_row, q, is_dead, isd, r);
    const float qv = d.q_valid[row];
    const DropCtx dc = drop_ctx(d.drop);
    const uint32_t ridx = attn_row_idx(d, head, n, q);
    const int kend = r.uniform ? d.T : q + 1;
    const bool live = !r.uniform && !is_dead;
    // per key: w = query mask * dropout factor; dP = dO . V[k]; delta = sum_k dP*w*p
    float w[W_MAXK], dp[W_MAXK];
    float delta = 0.0f;
#pragma unroll
    for (int j = 0; j < W_MAXK; ++j) {
        const int k = lane + 64 * j;
        w[j] = 0.0f; dp[j] = 0.0f;
        if (k < kend && r.p[j] != 0.0f) {
            w[j] = dc.on ? qv * drop_factor(dc, ridx + (uint32_t)k) : qv;
            const float* Vr = Vb + (size_t)k * d.ld;
            float acc = 0.0f;
            for (int c = 0; c < d.d; ++c) acc = fmaf(dOrow[c], Vr[c], acc);
            dp[j] = acc * w[j];
            delta += dp[j] * r.p[j];
        }
    }
    delta = wave_sum(delta);
    // dV[k][c] += p[k] w[k] dO[c] ; dS[k] = p (dp - delta)/sqrt(d) ; dK[k][c] += dS[k] Q[c] ; dQ[c] = sum_k dS[k] K[k][c]
    float dq[W_MAXC] = {0.f, 0.f, 0.f, 0.f};
    float qreg[W_MAXC], doreg[W_MAXC];
#pragma unroll
    for (int i = 0; i < W_MAXC; ++i) {
        const int c = lane + 64 * i;
        qreg[i] = (c < d.d) ? Qrow[c] : 0.0f;
        doreg[i] = (c < d.d) ? dOrow[c] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < W_MAXK; ++j) {
        if (64 * j >= kend) break;
        const float a_mine = r.p[j] * w[j];                                           // A after mask + dropout
        const float ds_mine = live ? r.p[j] * (dp[j] - delta) * isd : 0.0f;           // dS / sqrt(d)
        for (int kk = 0; kk < 64 && 64 * j + kk < kend; ++kk) {
            const float a = __shfl(a_mine, kk, 64), ds = __shfl(ds_mine, kk, 64);
            if (a == 0.0f && ds == 0.0f) continue;
            const size_t krow = (size_t)(base_row + 64 * j + kk);
#pragma unroll
            for (int i = 0; i < W_MAXC; ++i) {
                const int c = lane + 64 * i;
                if (c < d.d) {
                    if (a != 0.0f) atomicAdd(bd.dV + krow * bd.ldg + hoff + c, a * doreg[i]);
                    if (ds != 0.0f) {
                        atomicAdd(bd.dK + krow * bd.ldg + hoff + c, ds * qreg[i]);
                        dq[i] = fmaf(ds, Kb[(size_t)(64 * j + kk) * d.ld + c], dq[i]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < W_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < d.d) bd.dQ[row * bd.ldg + hoff + c] = dq[i];
    }
}

// zero the [hoff, hoff + H*d) columns of dK / dV rows before the atomic accumulation
__global__ __launch_bounds__(256) void k_attn_wide_zero(float* dK, float* dV, int ldg, int M, int C) {
    const long long total = (long long)M * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / C), c = (int)(i % C);
        dK[(size_t)m * ldg + c] = 0.0f;
        dV[(size_t)m * ldg + c] = 0.0f;
    }
}

int cr_attn_wide_supported(const cr_attn_desc* d) { return d->T <= 64 * W_MAXK && d->d <= 64 * W_MAXC; }

int cr_attn_wide_fwd_launch(const cr_attn_desc* d, hipStream_t s) {
    hipLaunchKernelGGL(k_attn_wide_fwd, dim3(cr_ceil_div(d->T, 4), d->B * d->H), dim3(256), 0, s, *d);
    return cr_check_launch("cr_attn_fwd(wide)");
}

int cr_attn_wide_bwd_launch(const cr_attn_bwd_desc* bd, hipStream_t s) {
    const cr_attn_desc* d = &bd->f;
    const int M = d->B * d->T, C = d->H * d->d;
    int grid = cr_ceil_div(M * C, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_attn_wide_zero, dim3(grid), dim3(256), 0, s, bd->dK, bd->dV, bd->ldg, M, C);
    hipLaunchKernelGGL(k_attn_wide_bwd, dim3(cr_ceil_div(d->T, 4), d->B * d->H), dim3(256), 0, s, *bd);
    return cr_check_launch("cr_attn_bwd(wide)");
}
