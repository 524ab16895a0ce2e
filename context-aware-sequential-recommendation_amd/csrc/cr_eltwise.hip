// Element-wise helpers on [M,N] views with leading dimensions (concat / split are views).
// Reference call sites: tf.concat + tf.layers.dropout in cast_2.py:89-92, cast_4.py:115-124 ...;
// ReLU gradients of modules.py:300,333-334.
#include "cr_common.hpp"

__global__ __launch_bounds__(256) void k_eltwise(cr_elt_desc d) {
    const bool regen = !(d.op == CR_ELT_GRADPREP && d.aux);
    DropCtx dc;
    dc.on = false; dc.key = 0; dc.thresh = 0; dc.scale = 1.0f;
    if (regen) dc = drop_ctx(d.drop);
    const float gate_scale = (d.drop.rate > 0.0f) ? 1.0f / (1.0f - d.drop.rate) : 1.0f;
    const long long total = (long long)d.M * d.N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / d.N), n = (int)(i % d.N);
        float v = d.x[(size_t)m * d.ldx + n];
        switch (d.op) {
            case CR_ELT_ADD: v += d.aux[(size_t)m * d.ldaux + n]; break;
            case CR_ELT_DROPOUT:
                v = drop_apply(dc, (d.drop.row_offset + (uint32_t)m) * (uint32_t)d.N + (uint32_t)n, v);
                break;
            case CR_ELT_RELU_BWD: v = (d.aux[(size_t)m * d.ldaux + n] > 0.0f) ? v : 0.0f; break;
            case CR_ELT_GRADPREP:
                if (d.aux) v = (d.aux[(size_t)m * d.ldaux + n] > 0.0f) ? v * gate_scale : 0.0f;
                else v = drop_apply(dc, (d.drop.row_offset + (uint32_t)m) * (uint32_t)d.N + (uint32_t)n, v);
                break;
            default: break;
        }
        if (d.mask_ids && d.mask_ids[m] == 0) v = 0.0f;
        float* p = d.y + (size_t)m * d.ldy + n;
        *p = d.accumulate ? (*p + v) : v;
    }
}

extern "C" int cr_eltwise(const cr_elt_desc* d, void* stream) {
    CR_REQUIRE(d && d->x && d->y, "cr_eltwise: NULL pointer");
    CR_REQUIRE(d->M > 0 && d->N > 0 && d->op >= 0 && d->op <= CR_ELT_GRADPREP, "cr_eltwise: bad shape/op");
    CR_REQUIRE(!(d->op == CR_ELT_ADD || d->op == CR_ELT_RELU_BWD) || d->aux, "cr_eltwise: aux required");
    const long long total = (long long)d->M * d->N;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_eltwise, dim3(grid), dim3(256), 0, cr_stream(stream), *d);
    return cr_check_launch("cr_eltwise");
}
