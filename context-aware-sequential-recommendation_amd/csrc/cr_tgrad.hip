// cr_table_grad: the table section's gradient of a step WRITTEN from the batch's occurrence index (include/castrec.h, "occurrence
// index"; device code: cr_tgrad.hpp) -- the data-parallel path, whose gradient bucket is all-reduced before Adam runs.  The
// one-GPU step never materialises the table gradient: cr_adam_step sums each listed row and updates it in place (cr_adam.hip).
#include <string.h>

#include <stdlib.h>
#include "cr_tgrad.hpp"

template <int LPR, int VEC, int ENT>
__global__ __launch_bounds__(TG_NT) void k_table_grad(cr_tgrad_desc g, float* table_grad) {
    __shared__ float part[TG_NT * VEC + 4];
    const int32_t* ix = tg_index(g, g.step ? *g.step : 0u);
    tg_unit_blocks<LPR, VEC, ENT>(g, ix, blockIdx.x, gridDim.x, part, reinterpret_cast<int*>(part + TG_NT * VEC),
                                  [&](int row, int col0, const float (&acc)[VEC]) { tg_store<VEC>(table_grad + (size_t)row * g.D + col0, acc); });
}

extern "C" int cr_tgrad_geometry(int D, int* ng, int* ent) {
    int lpr = 0, vec = 0, e = 0;
    if (!tg_shape(D, &lpr, &vec, &e)) return 0;
    if (ng) *ng = TG_NT / lpr;
    if (ent) *ent = e;
    return 1;
}

// unit workgroups of a launch: one wave of them where the chip holds it (a batch's plan: a few hundred at the headline shape)
int tg_unit_grid(const cr_tgrad_desc* d) {
    static const int cap = getenv("CASTREC_TG_UNITS") ? atoi(getenv("CASTREC_TG_UNITS")) : 384;      // (measurement override)
    return d->lay.cap_blocks < cap ? d->lay.cap_blocks : cap;
}

extern "C" int cr_table_grad(const cr_tgrad_desc* d, float* table_grad, void* stream) {
    const char* why = tg_unsupported(d);
    CR_REQUIRE(why == nullptr, "cr_table_grad: %s", why ? why : "");
    CR_REQUIRE(table_grad != nullptr, "cr_table_grad: table_grad is NULL");
    int lpr = 0, vec = 0, ent = 0;
    tg_shape(d->D, &lpr, &vec, &ent);
    const dim3 grid((unsigned)tg_unit_grid(d));
#define TG_LAUNCH(L, V, E) hipLaunchKernelGGL((k_table_grad<L, V, E>), grid, dim3(TG_NT), 0, cr_stream(stream), *d, table_grad)
    if (vec == 4 && lpr == 64) TG_LAUNCH(64, 4, 8);
    else if (vec == 4 && lpr == 32) TG_LAUNCH(32, 4, 8);
    else if (vec == 4) TG_LAUNCH(16, 4, 8);
    else if (vec == 2 && lpr == 64) TG_LAUNCH(64, 2, 16);
    else if (vec == 2 && lpr == 32) TG_LAUNCH(32, 2, 16);
    else if (vec == 2) TG_LAUNCH(16, 2, 16);
    else if (lpr == 64) TG_LAUNCH(64, 1, 16);
    else if (lpr == 32) TG_LAUNCH(32, 1, 16);
    else TG_LAUNCH(16, 1, 16);
#undef TG_LAUNCH
    return cr_check_launch("cr_table_grad");
}
