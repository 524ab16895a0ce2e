// Adam, TensorFlow-1.15 formulation (tf.train.AdamOptimizer(lr, beta2=0.98), sasrec.py:120):
//   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= lr_t * m / (sqrt(v) + eps)
// applied densely to EVERY variable (the reference's zero-padded lookup goes through tf.concat, so
// TF produces dense table gradients and moves every row every step).  Fused here: gradient
// normalisation by n_target, reduction of the dense-parameter slabs, zeroing of the table gradient.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cr_common.hpp"
#include "cr_tgrad.hpp"

int tg_unit_grid(const cr_tgrad_desc* d);     // cr_tgrad.hip

// Sum of the gradient slabs for ADAM_COLS = 256 consecutive dense parameters [j0, j0 + 256): wave w of the block's
// ADAM_WAVES waves adds slabs w, w + 16, w + 32, ... -- each read is 1 KiB contiguous (16 bytes per lane; the first version
// read 256-byte pieces of each slab from 16-lane groups and reached 2.5 TB/s) -- and the wave partials are added in a
// fixed order through LDS: bitwise reproducible.  Result: thread t < 256 returns the sum of column j0 + t.
#define ADAM_COLS 256
#define ADAM_WAVES 16
typedef float f4a __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float slab_sum256(const float* slabs, int n_slabs, int n_dense, int j0, float (*part)[ADAM_COLS]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = j0 + 4 * lane;
    f4a acc = (f4a){0.f, 0.f, 0.f, 0.f};
    if (j + 3 < n_dense) {
        // every load of the wave's share is issued before the first add (n_slabs <= 256: at most 16 per wave): one
        // memory round trip instead of one per batch of 8
        for (int s0 = w; s0 < n_slabs; s0 += 16 * ADAM_WAVES) {
            f4a v[16];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (s0 + ADAM_WAVES * u < n_slabs) { v[u] = *reinterpret_cast<const f4a*>(slabs + (size_t)(s0 + ADAM_WAVES * u) * n_dense + j); cnt = u + 1; }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u < cnt) acc += v[u];
        }
    } else {
        for (int s = w; s < n_slabs; s += ADAM_WAVES)
            for (int u = 0; u < 4; ++u)
                if (j + u < n_dense) acc[u] += slabs[(size_t)s * n_dense + j + u];
    }
    part[w][4 * lane + 0] = acc.x; part[w][4 * lane + 1] = acc.y; part[w][4 * lane + 2] = acc.z; part[w][4 * lane + 3] = acc.w;
    __syncthreads();
    float g = 0.0f;
    if (threadIdx.x < ADAM_COLS) {
#pragma unroll
        for (int k = 0; k < ADAM_WAVES; ++k) g += part[k][threadIdx.x];
    }
    return g;
}

// blocks [0, nb_dense): 256 dense parameters each (slab reduction + update); then nb_lazy blocks (lazy item-table rows,
// one wave per listed id); the rest: table entries, grid-stride
// STREAM (a swept table section of the Infinity Cache's size or more: nothing of it is read again before 7 x its size has passed):
// the sweep's accesses are streaming loads / stores, and four groups instead of two are in flight per thread.  The 10 M-item table
// of config C5 (71.7 GB per step): 14.2 -> 13.6 ms.  (Four groups cost 76 bytes of scratch at this kernel's 128 registers -- the four
// arrays' 64-bit addresses; a contiguous region per workgroup, which needs one 32-bit offset only, ran the sweep 20 % SLOWER: 256
// separate streams per array instead of one moving window.)
// LPR > 0 (with VEC): the table section's gradient comes from the batch's occurrence index `g` (cr_tgrad.hpp): nb_units blocks sum
// the listed rows and update them in place, the sweep blocks give every row WITHOUT a unit the zero-gradient update -- no table_grad
// array is read or zeroed (two of the sweep's eight streams: config C5's dense step moves 53.8 instead of 71.7 GB).
template <bool STREAM, int LPR, int VEC>
__global__ __launch_bounds__(64 * ADAM_WAVES) void k_adam(cr_adam_desc d, cr_tgrad_desc g, int nb_dense, int nb_lazy, int nb_units, int nb_ring) {
    __shared__ float part_s[ADAM_WAVES * ADAM_COLS + 4];                         // (+ one word: "this slice arrived last", cr_tgrad.hpp)
    float (*part)[ADAM_COLS] = reinterpret_cast<float (*)[ADAM_COLS]>(part_s);
    constexpr bool TG = LPR > 0;
    cr_kernarg_touch<sizeof(cr_adam_desc) + (TG ? sizeof(cr_tgrad_desc) : 0)>();
    const uint32_t t = d.step_snapshot ? *d.step_snapshot : *reinterpret_cast<const uint32_t*>(d.state + 4);
    // (Two other dispatch orders were measured and lost: the unit workgroups FIRST -- 29.7 against 25.8 us at the time: the dense blocks'
    //  slab sums are the other long chain and want the early start -- and the short sections, id ring and sweep, first -- 21.8 against
    //  20.6 us.  The unit grid holds a batch's plan in ONE pass -- 384 workgroups: 260-280 are in use at the headline shape; at 256 a few
    //  workgroups ran two passes: 20.6 against 19.8 us.)
    const int bid = (int)blockIdx.x;
    if (bid >= (int)gridDim.x - nb_ring) {
        // the last nb_ring blocks: the next step's ids out of the resident ring (nothing else in this launch reads the static id
        // buffers; the occurrence index of the RUNNING step is read from its ring slot, which these blocks do not write)
        constexpr int NT = 64 * ADAM_WAVES;
        const long long n_copy = d.ids_copy_elems ? d.ids_copy_elems : d.ids_slot_elems;
        const int32_t* src1 = d.ids_ring + (long long)((t + 1u) % (uint32_t)d.ids_ring_slots) * d.ids_slot_elems;
        const long long first = (long long)(bid - ((int)gridDim.x - nb_ring)) * NT + threadIdx.x, stride = (long long)nb_ring * NT;
        if ((d.ids_slot_elems & 3) == 0 && (n_copy & 3) == 0 && (((uintptr_t)d.ids_ring | (uintptr_t)d.ids_dst) & 15) == 0) {
            const int4* src = reinterpret_cast<const int4*>(src1);
            int4* dst = reinterpret_cast<int4*>(d.ids_dst);
            for (long long i = first; i < (n_copy >> 2); i += stride) dst[i] = src[i];
        } else {                                         // (a slot that is no multiple of 16 bytes: B * T odd)
            for (long long i = first; i < n_copy; i += stride) d.ids_dst[i] = src1[i];
        }
        return;
    }
    const float* st = d.stats ? d.stats : d.state;
    const float n = st[2];
    // the step's scalars (two dependent loads, two powf) are formed where a section first needs them: BEHIND its first loads, which
    // do not depend on them -- one memory round trip of this short kernel's critical path (tools/adam_probe.py)
    float inv_n = 0.0f, lr_t = 0.0f;
    auto scalars = [&]() {
        inv_n = n > 0.0f ? 1.0f / n : 0.0f;
        const float b1t = powf(d.beta1, (float)t), b2t = powf(d.beta2, (float)t);
        lr_t = d.lr * sqrtf(1.0f - b2t) / (1.0f - b1t);
    };
    auto update = [&](long long i, float g) {
        g *= inv_n;
        if (i < d.n_l2) g = fmaf(d.l2, d.p[i], g);       // d/dp of l2 * sum(p^2) / 2 on the lookup tables (modules.py:153)
        const float m = d.beta1 * d.m[i] + (1.0f - d.beta1) * g;
        const float v = d.beta2 * d.v[i] + (1.0f - d.beta2) * g * g;
        d.m[i] = m;
        d.v[i] = v;
        d.p[i] -= lr_t * m / (sqrtf(v) + d.eps);
    };
    if (bid < nb_dense) {
        const int j0 = bid * ADAM_COLS;
        const int ns = d.slab_counts ? min(d.slab_counts[bid], d.n_slabs) : d.n_slabs;
        // the parameter and its moments are requested BEFORE the slab sum (they do not depend on it): one memory round trip less on
        // the critical path of this short kernel
        const bool mine = threadIdx.x < ADAM_COLS && j0 + (int)threadIdx.x < d.n_dense;
        const long long i = d.n_table + j0 + (mine ? (int)threadIdx.x : 0);
        const float p0 = mine ? d.p[i] : 0.0f, m0 = mine ? d.m[i] : 0.0f, v0 = mine ? d.v[i] : 0.0f;
        float g = slab_sum256(d.dense_slabs, ns, d.n_dense, j0, part);
        scalars();
        if (mine) {
            g *= inv_n;
            if (i < d.n_l2) g = fmaf(d.l2, p0, g);
            const float m = d.beta1 * m0 + (1.0f - d.beta1) * g;
            const float v = d.beta2 * v0 + (1.0f - d.beta2) * g * g;
            d.m[i] = m;
            d.v[i] = v;
            d.p[i] = p0 - lr_t * m / (sqrtf(v) + d.eps);
        }
    } else if (bid < nb_dense + nb_lazy) {
        // lazy rows: wave w of the lazy blocks walks ids w, w + W, ...; the first wave to swap the step number into a row's
        // flag owns the row (every other occurrence of the id finds it there and moves on)
        const int lane = threadIdx.x & 63;
        const int W = nb_lazy * ADAM_WAVES;
        scalars();
        for (int k = (bid - nb_dense) * ADAM_WAVES + (threadIdx.x >> 6); k < d.n_lazy_ids; k += W) {
            const int id = d.lazy_ids[k];
            if (id <= 0 || id >= d.lazy_rows) continue;                       // row 0: the zero-pad row never has a gradient
            int mine = 0;
            if (lane == 0) mine = atomicExch(&d.lazy_flags[id], t) != t;
            mine = __shfl(mine, 0, 64);
            if (!mine) continue;
            for (int c = lane; c < d.lazy_D; c += 64) {
                const long long i = (long long)id * d.lazy_D + c;
                const float g = d.table_grad[i];
                d.table_grad[i] = 0.0f;
                update(i, g);
            }
        }
    } else if (TG && bid < nb_dense + nb_lazy + nb_units) {
        // the rows the batch looked up: gradient = ordered sum over the row's occurrences, update in place.  (p / m / v requested in
        // front of the batch's loads: 20 bytes of scratch per lane at 16 occurrences in flight, and no faster -- 20.1 against 20.3 us.)
        const int32_t* ix = tg_index(g, t);
        constexpr int W = TG ? VEC : 1;
        scalars();
        tg_unit_blocks<TG ? LPR : 16, W, (TG && VEC == 4) ? 8 : 16>(g, ix, bid - nb_dense - nb_lazy, nb_units, part_s,
            reinterpret_cast<int*>(part_s + ADAM_WAVES * ADAM_COLS),
            [&](int row, int col0, const float (&acc)[W]) {
                const long long i = (long long)row * g.D + col0;
                float p0[W], m0[W], v0[W];
                tg_load<W>(p0, d.p + i); tg_load<W>(m0, d.m + i); tg_load<W>(v0, d.v + i);
#pragma unroll
                for (int u = 0; u < W; ++u) {
                    float gu = acc[u] * inv_n;
                    if (i + u < d.n_l2) gu = fmaf(d.l2, p0[u], gu);
                    m0[u] = d.beta1 * m0[u] + (1.0f - d.beta1) * gu;
                    v0[u] = d.beta2 * v0[u] + (1.0f - d.beta2) * gu * gu;
                    p0[u] -= lr_t * m0[u] / (sqrtf(v0[u]) + d.eps);
                }
                tg_store<W>(d.m + i, m0); tg_store<W>(d.v + i, v0); tg_store<W>(d.p + i, p0);
            });
    } else if (TG) {
        // the rows WITHOUT a unit: the zero-gradient update of TensorFlow's dense Adam (m, v decay, p moves on its momentum)
        const int32_t* ix = tg_index(g, t);
        const uint32_t* bits = reinterpret_cast<const uint32_t*>(ix + ix[6]);
        const int nb_table = gridDim.x - nb_dense - nb_lazy - nb_units - nb_ring;
        const int tb = bid - nb_dense - nb_lazy - nb_units;
        constexpr int NT = 64 * ADAM_WAVES;
        const long long stride = (long long)nb_table * NT;
        auto touched = [&](uint32_t row) { return ((bits[row >> 5] >> (row & 31)) & 1u) != 0u; };
        scalars();
        if ((g.D & 3) == 0) {
            const float b1 = d.beta1, b2 = d.beta2, c1 = 1.0f - d.beta1, c2 = 1.0f - d.beta2;
            auto ld4 = [](const float* q) { return STREAM ? __builtin_nontemporal_load(reinterpret_cast<const f4a*>(q)) : *reinterpret_cast<const f4a*>(q); };
            auto st4 = [](float* q, const f4a x) {
                if (STREAM) __builtin_nontemporal_store(x, reinterpret_cast<f4a*>(q));
                else *reinterpret_cast<f4a*>(q) = x;
            };
            const long long n4 = d.n_table >> 2;
            const unsigned D4 = (unsigned)g.D >> 2;
            const int sh = (D4 & (D4 - 1u)) == 0u ? __builtin_ctz(D4) : -1;      // (128 / 256 columns: a shift instead of a 64-bit division per group)
            constexpr int U = 4;
            // p / m / v of a group are requested WITHOUT waiting for the bitmap word that says whether the row is listed (a dependent
            // round trip in front of every batch of loads: the first version swept config C5's table no faster than the sweep that also
            // read and zeroed a gradient array); a listed row's values are read and dropped (2 % of the rows at C5), only its STORES
            // are skipped -- the unit workgroups write those elements
            for (long long q = (long long)tb * NT + threadIdx.x; q < n4; q += U * stride) {
                bool in[U];
                uint32_t bw[U];
                uint32_t row[U];
                f4a p[U], m[U], v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long long qq = q + u * stride;
                    in[u] = qq < n4;
                    const long long qc = in[u] ? qq : q;
                    row[u] = sh >= 0 ? (uint32_t)(qc >> sh) : (uint32_t)((unsigned long long)qc / D4);
                    bw[u] = bits[row[u] >> 5];
                    p[u] = ld4(d.p + 4 * qc); m[u] = ld4(d.m + 4 * qc); v[u] = ld4(d.v + 4 * qc);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!in[u] || ((bw[u] >> (row[u] & 31)) & 1u)) continue;
                    const long long i = 4 * (q + u * stride);
                    f4a po;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float gu = 0.0f;
                        if (i + e < d.n_l2) gu = fmaf(d.l2, p[u][e], gu);
                        m[u][e] = b1 * m[u][e] + c1 * gu;
                        v[u][e] = b2 * v[u][e] + c2 * gu * gu;
                        po[e] = p[u][e] - lr_t * m[u][e] / (sqrtf(v[u][e]) + d.eps);
                    }
                    st4(d.m + i, m[u]); st4(d.v + i, v[u]); st4(d.p + i, po);
                }
            }
        } else {
            for (long long i = (long long)tb * NT + threadIdx.x; i < d.n_table; i += stride)
                if (!touched((uint32_t)((unsigned long long)i / (unsigned)g.D))) update(i, 0.0f);
        }
    } else {
        // table section: 16 bytes per lane per array (4-byte accesses moved 3.2 TB/s on a 188 MB table), two
        // groups of four in flight per thread; the scalar head / tail (a start or an end that is not a multiple of 4) goes to
        // the first table block
        const int nb_table = gridDim.x - nb_dense - nb_lazy - nb_ring;
        const int tb = bid - nb_dense - nb_lazy;
        const long long first = nb_lazy > 0 ? (long long)d.lazy_rows * d.lazy_D : 0;   // the lazy part of the table section is not swept
        constexpr int NT = 64 * ADAM_WAVES;
        const long long a0 = min((first + 3) & ~3ll, d.n_table);
        const long long n4 = (d.n_table - a0) >> 2;
        const float b1 = d.beta1, b2 = d.beta2, c1 = 1.0f - d.beta1, c2 = 1.0f - d.beta2;
        auto ld4 = [](const float* q) { return STREAM ? __builtin_nontemporal_load(reinterpret_cast<const f4a*>(q)) : *reinterpret_cast<const f4a*>(q); };
        auto st4 = [](float* q, const f4a v) {
            if (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<f4a*>(q));
            else *reinterpret_cast<f4a*>(q) = v;
        };
        auto finish4 = [&](long long i, f4a g, const f4a p, f4a m, f4a v) {
            st4(d.table_grad + i, (f4a){0.f, 0.f, 0.f, 0.f});
            g *= inv_n;
            f4a po;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float gu = g[u];
                if (i + u < d.n_l2) gu = fmaf(d.l2, p[u], gu);
                m[u] = b1 * m[u] + c1 * gu;
                v[u] = b2 * v[u] + c2 * gu * gu;
                po[u] = p[u] - lr_t * m[u] / (sqrtf(v[u]) + d.eps);
            }
            st4(d.m + i, m);
            st4(d.v + i, v);
            st4(d.p + i, po);
        };
        auto update4 = [&](long long q) {
            const long long i = a0 + 4 * q;
            const f4a g = ld4(d.table_grad + i);
            const f4a p = ld4(d.p + i);
            const f4a m = ld4(d.m + i), v = ld4(d.v + i);
            finish4(i, g, p, m, v);
        };
        const long long stride = (long long)nb_table * NT;
        long long q = (long long)tb * NT + threadIdx.x;
        {
            // the thread's first group (its only one on a table of a few hundred thousand entries): requested, THEN the scalars
            const bool have = q < n4;
            const long long i = a0 + 4 * (have ? q : 0);
            f4a g = (f4a){0.f, 0.f, 0.f, 0.f}, p = g, m = g, v = g;
            if (have) {
                g = ld4(d.table_grad + i);
                p = ld4(d.p + i);
                m = ld4(d.m + i);
                v = ld4(d.v + i);
            }
            scalars();
            if (have) {
                finish4(i, g, p, m, v);
                q += stride;
            }
        }
        constexpr int U = STREAM ? 4 : 2;                 // groups of the four arrays requested before the first update
        for (; q + (U - 1) * stride < n4; q += U * stride) {
            f4a g[U], p[U], m[U], v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long i = a0 + 4 * (q + u * stride);
                g[u] = ld4(d.table_grad + i); p[u] = ld4(d.p + i); m[u] = ld4(d.m + i); v[u] = ld4(d.v + i);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) finish4(a0 + 4 * (q + u * stride), g[u], p[u], m[u], v[u]);
        }
        for (; q < n4; q += stride) update4(q);
        if (tb == 0) {
            for (long long i = first + threadIdx.x; i < a0; i += NT) {
                const float g = d.table_grad[i];
                d.table_grad[i] = 0.0f;
                update(i, g);
            }
            for (long long i = a0 + 4 * n4 + threadIdx.x; i < d.n_table; i += NT) {
                const float g = d.table_grad[i];
                d.table_grad[i] = 0.0f;
                update(i, g);
            }
        }
    }
    if (bid == 0 && threadIdx.x == 0) {
        d.state[5] = (n > 0.0f ? st[0] / n : 0.0f) + (d.n_l2 > 0 ? d.state[7] : 0.0f);   // loss (sasrec.py:105-110)
        d.state[6] = n > 0.0f ? st[1] / n : 0.0f;        // auc   (sasrec.py:113-115)
        if (d.step_snapshot) {                           // end of the step: no block of this kernel reads state[0..4]
            d.state[0] = 0.0f; d.state[1] = 0.0f; d.state[2] = 0.0f; d.state[3] = 0.0f;
            *reinterpret_cast<uint32_t*>(d.state + 4) = t + 1u;
        }
    }
}

extern "C" int cr_adam_step(const cr_adam_desc* d, void* stream) {
    CR_REQUIRE(d && d->p && d->m && d->v && d->state, "cr_adam_step: NULL pointer");
    CR_REQUIRE(d->n_table >= 0 && d->n_dense >= 0 && d->n_table + d->n_dense > 0, "cr_adam_step: bad sizes");
    CR_REQUIRE(d->n_table == 0 || d->table_grad || d->tg, "cr_adam_step: table_grad is NULL");
    CR_REQUIRE(d->n_dense == 0 || (d->dense_slabs && d->n_slabs > 0), "cr_adam_step: dense_slabs missing");
    CR_REQUIRE(d->step_snapshot == nullptr || d->stats != nullptr, "cr_adam_step: step_snapshot needs stats (a copy of the sums that does not alias state[0..2])");
    const int nb_dense = cr_ceil_div(d->n_dense, ADAM_COLS);
    int nb_lazy = 0;
    long long n_swept = d->n_table;
    if (d->lazy_ids) {
        CR_REQUIRE(d->lazy_flags && d->n_lazy_ids > 0 && d->lazy_rows > 0 && d->lazy_D > 0 &&
                   (long long)d->lazy_rows * d->lazy_D <= d->n_table, "cr_adam_step: bad lazy-row arguments");
        nb_lazy = cr_ceil_div(d->n_lazy_ids, ADAM_WAVES * 8);             // ~8 ids per wave
        if (nb_lazy > 4096) nb_lazy = 4096;
        n_swept = d->n_table - (long long)d->lazy_rows * d->lazy_D;
    }
    constexpr int NT = 64 * ADAM_WAVES;
    const long long groups = n_swept / 4 + 1;                                 // 16-byte groups (+ one block's worth of head / tail)
    int nb_table = (int)((groups + NT - 1) / NT > 1024 ? 1024 : (groups + NT - 1) / NT);
    if (nb_table < 1) nb_table = 1;
    int nb_ring = 0;
    if (d->ids_ring) {
        CR_REQUIRE(d->lazy_ids == nullptr, "cr_adam_step: ids_ring and lazy_ids exclude each other (row-sparse Adam reads the step's ids)");
        CR_REQUIRE(d->ids_dst && d->ids_ring_slots > 0 && d->ids_slot_elems > 0, "cr_adam_step: bad id-ring arguments");
        CR_REQUIRE(d->ids_copy_elems >= 0 && d->ids_copy_elems <= d->ids_slot_elems, "cr_adam_step: ids_copy_elems exceeds the slot");
        const long long slot4 = ((d->ids_copy_elems ? d->ids_copy_elems : d->ids_slot_elems) + 3) / 4;
        nb_ring = (int)((slot4 + NT - 1) / NT > 256 ? 256 : (slot4 + NT - 1) / NT);
    }
    static const char* nt_env = getenv("CASTREC_ADAM_STREAM");
    const bool stream_sweep = nt_env ? atoi(nt_env) != 0 : n_swept * 4 >= (256ll << 20);
    cr_tgrad_desc g;
    memset(&g, 0, sizeof(g));
    int nb_units = 0, lpr = 0, vec = 0;
    if (d->tg) {
        g = *d->tg;
        const char* why = tg_unsupported(&g);
        CR_REQUIRE(why == nullptr, "cr_adam_step: tg: %s", why ? why : "");
        CR_REQUIRE(d->lazy_ids == nullptr, "cr_adam_step: tg and lazy_ids exclude each other");
        CR_REQUIRE(d->n_table == ((int64_t)g.lay.V + g.lay.T_pos) * g.D, "cr_adam_step: tg: n_table must be (V + T_pos) * D");
        CR_REQUIRE((g.ring != nullptr) == (d->ids_ring != nullptr) && (!g.ring || (g.ring == d->ids_ring && g.ring_slots == d->ids_ring_slots && g.slot_words == d->ids_slot_elems)),
                   "cr_adam_step: tg: the index ring must be the id ring of this launch (or both absent)");
        int ent = 0;
        tg_shape(g.D, &lpr, &vec, &ent);
        nb_units = tg_unit_grid(&g);
    }
    const dim3 grid(nb_dense + nb_lazy + nb_units + nb_table + nb_ring);
#define ADAM_LAUNCH(S, L, V) hipLaunchKernelGGL((k_adam<S, L, V>), grid, dim3(NT), 0, cr_stream(stream), *d, g, nb_dense, nb_lazy, nb_units, nb_ring)
    if (!d->tg) {
        if (stream_sweep) ADAM_LAUNCH(true, 0, 1);
        else ADAM_LAUNCH(false, 0, 1);
    } else if (vec == 4 && lpr == 64) {
        if (stream_sweep) ADAM_LAUNCH(true, 64, 4);
        else ADAM_LAUNCH(false, 64, 4);
    } else if (vec == 4 && lpr == 32) {
        if (stream_sweep) ADAM_LAUNCH(true, 32, 4);
        else ADAM_LAUNCH(false, 32, 4);
    } else if (vec == 4) ADAM_LAUNCH(false, 16, 4);
    else if (vec == 2 && lpr == 64) ADAM_LAUNCH(false, 64, 2);
    else if (vec == 2 && lpr == 32) ADAM_LAUNCH(false, 32, 2);
    else if (vec == 2) ADAM_LAUNCH(false, 16, 2);
    else if (lpr == 64) ADAM_LAUNCH(false, 64, 1);
    else if (lpr == 32) ADAM_LAUNCH(false, 32, 1);
    else ADAM_LAUNCH(false, 16, 1);
#undef ADAM_LAUNCH
    return cr_check_launch("cr_adam_step");
}

// l2 * sum(p^2) / 2 over the lookup tables: ONE workgroup, per-thread strided partial sums, then a fixed-order tree
__global__ __launch_bounds__(1024) void k_l2_penalty(const float* p, long long n, float scale, float* state) {
    __shared__ float red[1024];
    float acc = 0.0f;
    for (long long i = threadIdx.x; i < n; i += 1024) acc = fmaf(p[i], p[i], acc);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) state[7] = scale * red[0];
}

extern "C" int cr_l2_penalty(const float* p, int64_t n, float scale, float* state, void* stream) {
    CR_REQUIRE(p && state && n > 0, "cr_l2_penalty: bad arguments");
    hipLaunchKernelGGL(k_l2_penalty, dim3(1), dim3(1024), 0, cr_stream(stream), p, n, scale, state);
    return cr_check_launch("cr_l2_penalty");
}

__global__ __launch_bounds__(64 * ADAM_WAVES) void k_reduce_slabs(const float* slabs, int n_slabs, int n_dense, float* out,
                                                      const float* state, float* stats_out, const int32_t* slab_counts) {
    __shared__ float part[ADAM_WAVES][ADAM_COLS];
    const int j0 = blockIdx.x * ADAM_COLS;
    const float g = slab_sum256(slabs, slab_counts ? min(slab_counts[blockIdx.x], n_slabs) : n_slabs, n_dense, j0, part);
    if (threadIdx.x < ADAM_COLS && j0 + (int)threadIdx.x < n_dense) out[j0 + threadIdx.x] = g;
    if (blockIdx.x == 0 && threadIdx.x < 3 && stats_out) stats_out[threadIdx.x] = state[threadIdx.x];
}

extern "C" int cr_reduce_slabs(const float* dense_slabs, int n_slabs, int n_dense, float* out, const float* state,
                               float* stats_out, const int32_t* slab_counts, void* stream) {
    CR_REQUIRE(dense_slabs && out && n_slabs > 0 && n_dense > 0, "cr_reduce_slabs: bad arguments");
    CR_REQUIRE(stats_out == nullptr || state != nullptr, "cr_reduce_slabs: state is NULL");
    const int grid = cr_ceil_div(n_dense, ADAM_COLS);
    hipLaunchKernelGGL(k_reduce_slabs, dim3(grid), dim3(64 * ADAM_WAVES), 0, cr_stream(stream), dense_slabs, n_slabs, n_dense, out, state, stats_out, slab_counts);
    return cr_check_launch("cr_reduce_slabs");
}
