// The table section's gradient as a GATHER over the batch's occurrence index (include/castrec.h, "occurrence index"; host side:
// cr_index.cpp): device code shared by cr_table_grad (cr_tgrad.hip: writes the rows, the data-parallel path) and cr_adam_step
// (cr_adam.hip: sums a row and applies its update in place).
//
// Geometry.  A workgroup of 1024 threads is NG = 1024 / LPR lane groups; a lane group owns LPR * VEC >= D columns of one table row
// (VEC floats per lane: 16-, 8- or 4-byte loads -- D % 4 == 0, D even, any D) and sums at most ENT occurrences: ONE batch of loads,
// all in flight together, no loop.  The index carries the plan (which group sums which occurrences of which row; a row's groups are
// consecutive in one workgroup), so a launch is one wave of workgroups of equal, short work: first version (a group per row for up to 64
// occurrences, a workgroup for more) ran the headline's Adam launch 27 us longer -- the hottest item of the Zipf corpus holds 3 000 of a
// batch's 76 800 lookups, and ONE CU issues a load instruction per ~16 clocks whatever is in flight.
// Partials of a row's groups meet in LDS and are added in group order; a row cut into slices (more occurrences than a workgroup
// takes) leaves one partial row per slice in memory, and the slice that finishes LAST adds them in slice order: plain stores ->
// every wave's s_waitcnt vmcnt(0) -> barrier -> one agent-scope release fence -> ticket; last arriver: one agent-scope acquire
// fence -> barrier -> plain loads (cdna_hip_programming.md, in-launch split-K reduction: correct for any placement of the slices).
// Fixed orders everywhere: the table gradient holds the same bits on every run (the float atomics it replaces did not).
// Every load of a batch is unconditional (absent entries read row 0 with a zero coefficient, a pos / neg entry reads its row twice
// instead of a second partial): no branch stands between two loads.
#pragma once
#include "cr_common.hpp"

#define TG_NT 1024

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

// acc = sum over occurrences [s, s + n) of the list (n <= ENT <= LPR), in order; `col`: the lane's first column (clamped into the row).
// Lane j of the group fetches occurrence j and its coefficient ONCE (two load instructions per wave for all of its groups' entries);
// the group then takes them from that lane, entry after entry (cross-lane reads, no memory).  The launch is bound by the CU's load
// ISSUE (one wave instruction per ~16 clocks: a row load of this shape carries 400-1 024 bytes): first version -- every lane loading
// every occurrence word and coefficient itself, and a second row load per entry whether it had a second partial or not -- issued 64
// loads per wave and batch where this issues ~25: 12.2 -> ... us for the headline batch (rocprofv3, cr_table_grad alone).
template <int LPR, int VEC, int ENT>
__device__ __forceinline__ void tg_batch(const cr_tgrad_desc& g, const int32_t* occ, int s, int n, int col, float (&acc)[VEC]) {
    static_assert(ENT <= LPR, "an occurrence per lane of the group");
    const size_t M = (size_t)g.lay.M;
    const float* rows2 = g.rows2;
    const int lane = threadIdx.x & 63, lg = lane % LPR, gbase = lane - lg;
    // my occurrence (lanes >= n repeat the last one with a zero coefficient; an idle group -- n == 0 -- reads word s = 0 of the list)
    const uint32_t wm = (uint32_t)occ[s + (lg < n ? lg : (n > 0 ? n - 1 : 0))];
    const int kind_m = (int)(wm >> 30);
    const bool isrow_m = kind_m == 0 || kind_m == 3;
    const float cf = g.coef[isrow_m ? 0 : (size_t)(kind_m - 1) * M + (size_t)(wm & 0x3fffffffu)];     // (row kinds: a dummy read)
    const float cm = lg < n ? (kind_m == 0 ? g.scale : (kind_m == 3 ? 1.0f : cf)) : 0.0f;
    float c1[ENT], v1[ENT][VEC], v2[ENT][VEC];
    bool two[ENT];
#pragma unroll
    for (int j = 0; j < ENT; ++j) {
        const uint32_t w = (uint32_t)__shfl((int)wm, gbase + j, 64);
        c1[j] = __shfl(cm, gbase + j, 64);
        const int kind = (int)(w >> 30);
        const size_t m = (size_t)(w & 0x3fffffffu);
        const bool isrow = kind == 0 || kind == 3;
        const float* p1 = isrow ? g.rows + m * (size_t)g.ld_rows : g.seq_emb + m * (size_t)g.ld_emb;
        tg_load<VEC>(v1[j], p1 + col);
        two[j] = isrow && rows2 != nullptr && j < n;
#pragma unroll
        for (int u = 0; u < VEC; ++u) v2[j][u] = 0.0f;
        if (two[j]) tg_load<VEC>(v2[j], rows2 + m * (size_t)g.ld_rows + col);        // (uniform per lane group)
    }
#pragma unroll
    for (int u = 0; u < VEC; ++u) acc[u] = 0.0f;
#pragma unroll
    for (int j = 0; j < ENT; ++j)
#pragma unroll
        for (int u = 0; u < VEC; ++u) acc[u] = fmaf(c1[j], v2[j][u], fmaf(c1[j], v1[j][u], acc[u]));
}

// Workgroup `ub` of `nub` unit workgroups.  consume(flat row, first column of the lane, acc) is called by the lanes whose columns
// exist, once per listed row.  part: LDS, TG_NT * VEC floats; flag: one LDS word.
template <int LPR, int VEC, int ENT, typename F>
__device__ __forceinline__ void tg_unit_blocks(const cr_tgrad_desc& g, const int32_t* ix, int ub, int nub, float* part, int* flag, F&& consume) {
    constexpr int NG = TG_NT / LPR;
    const int32_t* recs = ix + g.lay.off_recs;
    const int grp = threadIdx.x / LPR, l = threadIdx.x % LPR;
    const int col0 = VEC * l;
    const bool colok = col0 < g.D;                        // (D is a multiple of VEC by the launcher's choice of VEC)
    const int col = colok ? col0 : 0;
    // the header and this workgroup's first record are requested together (a record slot inside the capacity always exists): the
    // chain header -> record -> occurrences -> rows is this launch's critical path, four dependent round trips as first written
    int4 rec = *reinterpret_cast<const int4*>(recs + 4 * ((size_t)min(ub, g.lay.cap_blocks - 1) * NG + grp));
    const int n_blocks = ix[0];
    const int32_t* occ = ix + ((g.lay.off_recs + 4 * NG * n_blocks + 3) & ~3);          // (= ix[5], without waiting for it)
    for (int u = ub; u < n_blocks; u += nub) {
        if (u != ub) rec = *reinterpret_cast<const int4*>(recs + 4 * ((size_t)u * NG + grp));
        const uint32_t info = (uint32_t)rec.w;
        const int q = info & 63, k = (info >> 6) & 127, sidx = (info >> 13) & 511, nsl = info >> 22;
        float acc[VEC];
        tg_batch<LPR, VEC, ENT>(g, occ, rec.y, rec.z, col, acc);     // (an idle group: count 0, every coefficient 0)
#pragma unroll
        for (int e = 0; e < VEC; ++e) part[(grp * LPR + l) * VEC + e] = acc[e];
        __syncthreads();
        float t[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) t[e] = 0.0f;
        const bool head = rec.z > 0 && q == 0;            // the first group of a row (of a slice) adds its k partials, in group order
        if (head) {
            for (int j = 0; j < k; ++j)
#pragma unroll
                for (int e = 0; e < VEC; ++e) t[e] += part[((grp + j) * LPR + l) * VEC + e];
        }
        // slices: the workgroup holds ONE row's slice (groups 0 .. k-1): workgroup-uniform branch on group 0's record
        const uint32_t info0 = (uint32_t)recs[4 * ((size_t)u * NG) + 3];
        if ((info0 >> 22) > 1u) {
            const int n_sl = (int)(info0 >> 22), s0 = (int)((info0 >> 13) & 511u);
            const int first = u - s0;                     // the row's first slice = its ticket and its first partial row
            const int pitch = (g.D + 3) & ~3;
            if (head && colok) tg_store<VEC>(g.part_rows + (size_t)u * pitch + col0, t);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned prev = __hip_atomic_fetch_add(g.tickets + first, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = prev == (unsigned)(n_sl - 1) ? 1 : 0;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    g.tickets[first] = 0u;                // (every slice has arrived: nobody adds to it again in this launch)
                }
                *flag = last;
            }
            __syncthreads();
            if (*flag && grp == 0) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) t[e] = 0.0f;
                for (int j = 0; j < n_sl; ++j) {
                    float pv[VEC];
                    tg_load<VEC>(pv, g.part_rows + (size_t)(first + j) * pitch + col);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) t[e] += pv[e];
                }
                if (colok) consume(rec.x, col0, t);
            }
            (void)sidx; (void)nsl;
        } else if (head && colok) {
            consume(rec.x, col0, t);
        }
        __syncthreads();                                  // `part` / `flag` are rewritten by the next workgroup's worth of records
    }
}

// (lanes per row, floats per lane, occurrences per lane group) for a hidden size; false = not covered
static inline bool tg_shape(int D, int* lpr, int* vec, int* ent) {
    if (D >= 1 && D % 4 == 0 && D <= 256) { *vec = 4; *lpr = D <= 64 ? 16 : (D <= 128 ? 32 : 64); *ent = 8; return true; }
    if (D >= 1 && D % 2 == 0 && D <= 128) { *vec = 2; *lpr = D <= 32 ? 16 : (D <= 64 ? 32 : 64); *ent = 16; return true; }
    if (D >= 1 && D <= 64) { *vec = 1; *lpr = D <= 16 ? 16 : (D <= 32 ? 32 : 64); *ent = 16; return true; }
    return false;
}
static inline const char* tg_unsupported(const cr_tgrad_desc* g) {
    if (!g) return "NULL description";
    int lpr, vec, ent;
    if (!tg_shape(g->D, &lpr, &vec, &ent)) return "hidden size: a multiple of 4 up to 256, even up to 128, or any up to 64";
    if (!g->rows || !g->seq_emb || !g->coef || g->ld_rows < g->D || g->ld_emb < g->D) return "rows / seq_emb / coef";
    if (g->lay.M < 1 || g->lay.V < 2 || g->lay.total_words < 8 || g->lay.off_recs != 8) return "layout (cr_batch_index_layout)";
    if (g->lay.ng != TG_NT / lpr || g->lay.ent != ent) return "the index was planned for another geometry (cr_tgrad_geometry)";
    if (vec > 1 && ((g->ld_rows % vec) || (g->ld_emb % vec))) return "leading dimensions must be multiples of the vector width";
    if (!g->part_rows || !g->tickets) return "part_rows / tickets (the workspace of sliced rows)";
    if (g->ring) {
        if (g->ring_slots < 1 || g->slot_words < g->index_off + g->lay.total_words || !g->step) return "ring arguments";
    } else if (!g->index) {
        return "index";
    }
    return nullptr;
}
