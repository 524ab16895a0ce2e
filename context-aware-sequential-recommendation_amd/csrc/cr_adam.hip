// Adam, TensorFlow-1.15 formulation (tf.train.AdamOptimizer(lr, beta2=0.98), sasrec.py:120):
//   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= lr_t * m / (sqrt(v) + eps)
// applied densely to EVERY variable (the reference's zero-padded lookup goes through tf.concat, so
// TF produces dense table gradients and moves every row every step).  Fused here: gradient
// normalisation by n_target, reduction of the dense-parameter slabs, zeroing of the table gradient.
#include <math.h>

#include "cr_common.hpp"

__global__ __launch_bounds__(256) void k_adam(cr_adam_desc d) {
    const uint32_t t = *reinterpret_cast<const uint32_t*>(d.state + 4);
    const float n = d.state[2];
    const float inv_n = n > 0.0f ? 1.0f / n : 0.0f;
    const float b1t = powf(d.beta1, (float)t), b2t = powf(d.beta2, (float)t);
    const float lr_t = d.lr * sqrtf(1.0f - b2t) / (1.0f - b1t);
    const int total = d.n_table + d.n_dense;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        float g;
        if (i < d.n_table) {
            g = d.table_grad[i];
            d.table_grad[i] = 0.0f;
        } else {
            const int j = i - d.n_table;
            // slab sum with 8 loads in flight per thread (fixed association: bitwise reproducible)
            float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int s = 0;
            for (; s + 8 <= d.n_slabs; s += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) p8[u] += d.dense_slabs[(size_t)(s + u) * d.n_dense + j];
            }
            for (; s < d.n_slabs; ++s) p8[0] += d.dense_slabs[(size_t)s * d.n_dense + j];
            g = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
        }
        g *= inv_n;
        const float m = d.beta1 * d.m[i] + (1.0f - d.beta1) * g;
        const float v = d.beta2 * d.v[i] + (1.0f - d.beta2) * g * g;
        d.m[i] = m;
        d.v[i] = v;
        d.p[i] -= lr_t * m / (sqrtf(v) + d.eps);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.state[5] = n > 0.0f ? d.state[0] / n : 0.0f;   // loss  (sasrec.py:105-108)
        d.state[6] = n > 0.0f ? d.state[1] / n : 0.0f;   // auc   (sasrec.py:113-115)
    }
}

extern "C" int cr_adam_step(const cr_adam_desc* d, void* stream) {
    CR_REQUIRE(d && d->p && d->m && d->v && d->state, "cr_adam_step: NULL pointer");
    CR_REQUIRE(d->n_table >= 0 && d->n_dense >= 0 && d->n_table + d->n_dense > 0, "cr_adam_step: bad sizes");
    CR_REQUIRE(d->n_table == 0 || d->table_grad, "cr_adam_step: table_grad is NULL");
    CR_REQUIRE(d->n_dense == 0 || (d->dense_slabs && d->n_slabs > 0), "cr_adam_step: dense_slabs missing");
    const int total = d->n_table + d->n_dense;
    int grid = cr_ceil_div(total, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, cr_stream(stream), *d);
    return cr_check_launch("cr_adam_step");
}

__global__ __launch_bounds__(256) void k_reduce_slabs(const float* slabs, int n_slabs, int n_dense, float* out,
                                                      const float* state, float* stats_out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_dense; i += gridDim.x * 256) {
        float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 8 <= n_slabs; s += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p8[u] += slabs[(size_t)(s + u) * n_dense + i];
        }
        for (; s < n_slabs; ++s) p8[0] += slabs[(size_t)s * n_dense + i];
        out[i] = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
    }
    if (blockIdx.x == 0 && threadIdx.x < 3 && stats_out) stats_out[threadIdx.x] = state[threadIdx.x];
}

extern "C" int cr_reduce_slabs(const float* dense_slabs, int n_slabs, int n_dense, float* out, const float* state,
                               float* stats_out, void* stream) {
    CR_REQUIRE(dense_slabs && out && n_slabs > 0 && n_dense > 0, "cr_reduce_slabs: bad arguments");
    CR_REQUIRE(stats_out == nullptr || state != nullptr, "cr_reduce_slabs: state is NULL");
    int grid = cr_ceil_div(n_dense, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_reduce_slabs, dim3(grid), dim3(256), 0, cr_stream(stream), dense_slabs, n_slabs, n_dense, out, state, stats_out);
    return cr_check_launch("cr_reduce_slabs");
}
