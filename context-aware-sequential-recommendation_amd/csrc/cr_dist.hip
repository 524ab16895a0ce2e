// Data-parallel exchange of a row-sparse table gradient (the reference has no multi-GPU path).
//
// A step touches at most 3 B T rows of the item table per rank (seq / pos / neg ids).  For large tables the ranks exchange
// (row id, gradient row) pairs instead of the dense table: cr_rows_pack packs this rank's touched rows into FIXED slots -- slot i
// belongs to ids[i]; a row is packed by the first slot that claims it (a row-flag table and one atomic exchange per slot: no
// sort), later duplicates and the zero-pad id carry id 0 and a zero row -- and zeroes them in the table (this rank's share then
// travels in the packed buffer like everyone's); after the all-gather cr_rows_add adds one rank's slots into the table.  It is
// launched once per rank, in rank order: within a rank a row occurs once, so a launch has no conflicting writes, and a row's
// partial sums are added in the same order on every replica: bit-identical tables.
#include "cr_common.hpp"

// one 16-lane group per slot; packed row = [id bits | D floats]
__global__ __launch_bounds__(256) void k_rows_pack(float* table, const int32_t* ids, int n, int D, int V, uint32_t* flags, const uint32_t* tag_ptr,
                                                   float* packed, int zero_rows) {
    const uint32_t tag = *tag_ptr;
    const int grp = (blockIdx.x * 256 + threadIdx.x) >> 4, l = threadIdx.x & 15;
    if (grp >= n) return;
    int id = ids[grp];
    if (id < 0 || id >= V) id = 0;
    int first = 0;
    if (l == 0 && id != 0) first = atomicExch(&flags[id], tag) != tag;
    first = __shfl(first, (threadIdx.x & 63) & ~15, 64);
    float* out = packed + (size_t)grp * (D + 1);
    if (l == 0) out[0] = __int_as_float(first ? id : 0);
    float* row = table + (size_t)id * D;
    for (int c = l; c < D; c += 16) {
        out[1 + c] = first ? row[c] : 0.0f;
        if (first && zero_rows) row[c] = 0.0f;
    }
}
__global__ __launch_bounds__(256) void k_rows_add(float* table, const float* packed, int n, int D, int V) {
    const int grp = (blockIdx.x * 256 + threadIdx.x) >> 4, l = threadIdx.x & 15;
    if (grp >= n) return;
    const float* in = packed + (size_t)grp * (D + 1);
    const int id = __float_as_int(in[0]);
    if (id <= 0 || id >= V) return;
    float* row = table + (size_t)id * D;
    for (int c = l; c < D; c += 16) row[c] += in[1 + c];
}

extern "C" int cr_rows_pack(float* table, const int32_t* ids, int n, int D, int V, uint32_t* flags, const uint32_t* tag, float* packed,
                            int zero_rows, void* stream) {
    CR_REQUIRE(table && ids && flags && tag && packed, "cr_rows_pack: NULL pointer");
    CR_REQUIRE(n > 0 && D > 0 && V > 0, "cr_rows_pack: bad shape");
    hipLaunchKernelGGL(k_rows_pack, dim3(cr_ceil_div(n, 16)), dim3(256), 0, cr_stream(stream), table, ids, n, D, V, flags, tag, packed, zero_rows);
    return cr_check_launch("cr_rows_pack");
}
extern "C" int cr_rows_add(float* table, const float* packed, int n, int D, int V, void* stream) {
    CR_REQUIRE(table && packed, "cr_rows_add: NULL pointer");
    CR_REQUIRE(n > 0 && D > 0 && V > 0, "cr_rows_add: bad shape");
    hipLaunchKernelGGL(k_rows_add, dim3(cr_ceil_div(n, 16)), dim3(256), 0, cr_stream(stream), table, packed, n, D, V);
    return cr_check_launch("cr_rows_add");
}
