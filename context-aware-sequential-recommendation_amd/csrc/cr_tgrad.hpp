// The table section's gradient as a GATHER over the batch's occurrence index (include/castrec.h, "occurrence index"; host side:
// cr_index.cpp): device code shared by cr_table_grad (cr_tgrad.hip: writes the rows, the data-parallel path) and cr_adam_step
// (cr_adam.hip: sums a row and applies its update in place).
//
// Work split.  A workgroup of NT threads is NT / LPR lane groups; a lane group owns LPR * VEC >= D columns of one table row
// (VEC floats per lane: 16-, 8- or 4-byte loads -- D % 4 == 0, D even, any D).  A LIGHT unit (<= CR_INDEX_HEAVY occurrences) is one
// lane group's: its occurrences are summed in list order, BATCH of them in flight.  A HEAVY unit takes the whole workgroup: the
// groups sum contiguous slices of the list, the partials meet in LDS and are added in slice order -- a fixed order either way, so
// the table gradient holds the same bits on every run (the float atomics it replaces did not).
// Every load of a batch is unconditional (absent entries read row 0 with a zero coefficient, a pos / neg entry reads its row twice
// instead of a second partial): no branch stands between two loads.
#pragma once
#include "cr_common.hpp"

typedef float tg_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float tg_f2 __attribute__((ext_vector_type(2), aligned(4)));

template <int VEC>
__device__ __forceinline__ void tg_load(float (&v)[VEC], const float* p) {
    if constexpr (VEC == 4) {
        const tg_f4 t = *reinterpret_cast<const tg_f4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if constexpr (VEC == 2) {
        const tg_f2 t = *reinterpret_cast<const tg_f2*>(p);
        v[0] = t.x; v[1] = t.y;
    } else {
        v[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void tg_store(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) *reinterpret_cast<tg_f4*>(p) = (tg_f4){v[0], v[1], v[2], v[3]};
    else if constexpr (VEC == 2) *reinterpret_cast<tg_f2*>(p) = (tg_f2){v[0], v[1]};
    else *p = v[0];
}

// the index of the step this launch runs: a static buffer, or the slot of the id ring that holds the step's batch
__device__ __forceinline__ const int32_t* tg_index(const cr_tgrad_desc& g, uint32_t step) {
    return g.ring ? g.ring + (size_t)(step % (uint32_t)g.ring_slots) * (size_t)g.slot_words + (size_t)g.index_off : g.index;
}

// acc += sum over occurrences [s, e) of the list, in order; `col`: the lane's first column (clamped into the row by the caller)
template <int VEC, int BATCH>
__device__ __forceinline__ void tg_slice(const cr_tgrad_desc& g, const int32_t* occ, int s, int e, int col, float (&acc)[VEC]) {
    const size_t M = (size_t)g.lay.M;
    const float* rows2 = g.rows2;
    for (int b0 = s; b0 < e; b0 += BATCH) {
        uint32_t w[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) w[j] = (uint32_t)occ[min(b0 + j, e - 1)];
        float c1[BATCH], c2[BATCH], v1[BATCH][VEC], v2[BATCH][VEC];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const bool ok = b0 + j < e;
            const int kind = (int)(w[j] >> 30);
            const size_t m = (size_t)(w[j] & 0x3fffffffu);
            const bool isrow = kind == 0 || kind == 3;
            const float cf = g.coef[isrow ? 0 : (size_t)(kind - 1) * M + m];                 // (row kinds: a dummy read, not used)
            const float c = !ok ? 0.0f : (kind == 0 ? g.scale : (kind == 3 ? 1.0f : cf));
            const float* p1 = isrow ? g.rows + m * (size_t)g.ld_rows : g.seq_emb + m * (size_t)g.ld_emb;
            const float* p2 = (isrow && rows2) ? rows2 + m * (size_t)g.ld_rows : p1;
            c1[j] = c;
            c2[j] = (isrow && rows2) ? c : 0.0f;
            tg_load<VEC>(v1[j], p1 + col);
            tg_load<VEC>(v2[j], p2 + col);
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
#pragma unroll
            for (int u = 0; u < VEC; ++u) acc[u] = fmaf(c2[j], v2[j][u], fmaf(c1[j], v1[j][u], acc[u]));
    }
}

// Unit blocks of a launch: block `ub` of `nub`.  consume(flat row, first column of the lane, acc) is called by the lanes whose
// columns exist, once per listed row.  part: LDS, NT * VEC floats.
template <int LPR, int VEC, int NT, typename F>
__device__ __forceinline__ void tg_unit_blocks(const cr_tgrad_desc& g, const int32_t* ix, int ub, int nub, float* part, F&& consume) {
    constexpr int NG = NT / LPR;                          // lane groups per workgroup
    const int n_light = ix[0], n_heavy = ix[1];
    const int32_t* light = ix + g.lay.off_light;
    const int32_t* heavy = ix + g.lay.off_heavy;
    const int32_t* occ = ix + g.lay.off_occ;
    const int grp = threadIdx.x / LPR, l = threadIdx.x % LPR;
    const int col0 = VEC * l;
    const bool colok = col0 < g.D;                        // (D is a multiple of VEC by the launcher's choice of VEC)
    const int col = colok ? col0 : 0;
    const int n_lblocks = (n_light + NG - 1) / NG;
    for (int u = ub; u < n_heavy + n_lblocks; u += nub) {
        if (u < n_heavy) {
            // heavy: the groups sum contiguous slices; partials through LDS, added in slice order by group 0
            const int4 rec = *reinterpret_cast<const int4*>(heavy + 4 * (size_t)u);
            const int per = (rec.z + NG - 1) / NG;
            const int s = rec.y + min(grp * per, rec.z), e = rec.y + min((grp + 1) * per, rec.z);
            float acc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
            tg_slice<VEC, 8>(g, occ, s, e, col, acc);
            __syncthreads();                              // (the previous iteration's readers of `part` are through)
#pragma unroll
            for (int k = 0; k < VEC; ++k) part[(grp * LPR + l) * VEC + k] = acc[k];
            __syncthreads();
            if (grp == 0) {
                float t[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) t[k] = 0.0f;
                const int used = min(NG, (rec.z + per - 1) / per);
                for (int q = 0; q < used; ++q)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) t[k] += part[(q * LPR + l) * VEC + k];
                if (colok) consume(rec.x, col0, t);
            }
        } else {
            const int r = (u - n_heavy) * NG + grp;
            if (r < n_light) {
                const int4 rec = *reinterpret_cast<const int4*>(light + 4 * (size_t)r);
                float acc[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
                tg_slice<VEC, 8>(g, occ, rec.y, rec.y + rec.z, col, acc);
                if (colok) consume(rec.x, col0, acc);
            }
        }
    }
}

// (lanes per row, floats per lane) for a hidden size; 0 = not covered (D > 256, or an odd D above 64)
static inline bool tg_shape(int D, int* lpr, int* vec) {
    if (D % 4 == 0 && D <= 256) { *vec = 4; *lpr = D <= 64 ? 16 : (D <= 128 ? 32 : 64); return true; }
    if (D % 2 == 0 && D <= 128) { *vec = 2; *lpr = D <= 32 ? 16 : (D <= 64 ? 32 : 64); return true; }
    if (D <= 64) { *vec = 1; *lpr = D <= 16 ? 16 : (D <= 32 ? 32 : 64); return true; }
    return false;
}
static inline const char* tg_unsupported(const cr_tgrad_desc* g) {
    if (!g) return "NULL description";
    int lpr, vec;
    if (g->D < 1 || !tg_shape(g->D, &lpr, &vec)) return "hidden size: a multiple of 4 up to 256, even up to 128, or any up to 64";
    if (!g->rows || !g->seq_emb || !g->coef || g->ld_rows < g->D || g->ld_emb < g->D) return "rows / seq_emb / coef";
    if (g->lay.M < 1 || g->lay.V < 2 || g->lay.total_words < 8) return "layout (cr_batch_index_layout)";
    if (vec > 1 && ((g->ld_rows % vec) || (g->ld_emb % vec))) return "leading dimensions must be multiples of the vector width";
    if (g->ring) {
        if (g->ring_slots < 1 || g->slot_words < g->index_off + g->lay.total_words || !g->step) return "ring arguments";
    } else if (!g->index) {
        return "index";
    }
    return nullptr;
}
