// Occurrence index of a batch (include/castrec.h, "occurrence index"): table row -> the (kind, batch row) pairs that looked it
// up, so that the device forms a table row's gradient as an ordered sum over its occurrences instead of scattering float
// atomics (autodiff of modules.py:157 through sasrec.py:27, of sasrec.py:89-90, and of the positional lookup sasrec.py:40-50).
// Host code only (no HIP).
//   1. A stable counting sort in the order "seq ids, pos ids, neg ids, each by ascending row", O(rows of the batch): the two work
//      arrays are as long as the table and persist in the builder (a bin is reset through the list of rows the batch touched), so a
//      build never walks the table -- config C5's has 10^7 rows, a batch touches 2 % of them.
//   2. The work plan of the gather: a lane group of the device sums at most `ent` occurrences (ONE batch of loads: the sums of a Zipf
//      corpus are bound by dependent round trips and by a CU's load issue, not by bytes), a workgroup holds `ng` lane groups.  A
//      row with c occurrences takes ceil(c / ent) consecutive groups of ONE workgroup (their partials meet in LDS, added in group
//      order); a row that needs more than a workgroup's groups takes whole workgroups (slices), whose partials meet through memory
//      and are added in slice order by the last one to arrive.  Rows are packed into workgroups largest first (by group count, a
//      counting sort), a workgroup is filled with the largest row that still fits: every closed workgroup is more than half full,
//      and the single-group rows -- the bulk -- fill the gaps exactly.
#include <stdint.h>
#include <string.h>

#include <new>
#include <vector>

#include "castrec.h"

int cr_set_error(int code, const char* fmt, ...);

static inline int64_t align4(int64_t w) { return (w + 3) & ~(int64_t)3; }

extern "C" int cr_batch_index_layout(int M, int V, int T_pos, int ng, int ent, cr_index_layout* out) {
    if (!out || M < 1 || V < 2 || T_pos < 0 || (T_pos > 0 && M % T_pos != 0) || (int64_t)M >= ((int64_t)1 << 30))
        return cr_set_error(CR_ERR_INVALID, "cr_batch_index_layout: need M in [1, 2^30), V >= 2, T_pos >= 0 dividing M");
    if (ng < 1 || ng > 64 || ent < 1 || ent > 16) return cr_set_error(CR_ERR_INVALID, "cr_batch_index_layout: 1 <= ng <= 64 lane groups, 1 <= ent <= 16 occurrences per group");
    cr_index_layout L;
    memset(&L, 0, sizeof(L));
    L.M = M; L.V = V; L.T_pos = T_pos; L.ng = ng; L.ent = ent;
    const int64_t n_occ = (int64_t)3 * M + (T_pos ? M : 0);
    if (n_occ > 0x7fffffff) return cr_set_error(CR_ERR_INVALID, "cr_batch_index_layout: batch too large");
    // capacity.  The packer fills a workgroup with the largest row that still fits, so single-group rows (<= n_occ of them) close
    // every gap while they last: those workgroups are full.  What follows holds only rows of k >= 2 groups (c > ent occurrences, k <=
    // c / ent + 1 <= 2 c / ent: at most 2 n_occ / ent groups in all) and is more than half full.  Slices are whole workgroups.
    const int64_t blocks = (n_occ + ng - 1) / ng + (4 * n_occ) / ((int64_t)ent * ng) + n_occ / ((int64_t)ent * ng) + 4;
    if (blocks > 0x3fffff) return cr_set_error(CR_ERR_INVALID, "cr_batch_index_layout: batch too large");
    L.cap_blocks = (int)blocks;
    L.cap_occ = (int)n_occ;
    L.bitmap_words = (int)(((int64_t)V + T_pos + 31) / 32);
    L.off_recs = 8;
    L.total_words = align4(L.off_recs + (int64_t)4 * ng * L.cap_blocks) + align4(L.cap_occ) + align4(L.bitmap_words);
    *out = L;
    return CR_OK;
}

struct cr_index_builder {
    cr_index_layout L;
    std::vector<int32_t> count;       // per flat row: occurrences in this batch (0 between builds)
    std::vector<int32_t> cursor;      // per flat row: next free slot of its occurrence range
    std::vector<int32_t> uniq;        // flat rows in order of first appearance
    std::vector<int32_t> start;       // per entry of uniq: first occurrence
    std::vector<int32_t> by_size;     // entries of uniq ordered by group count (descending), stable
};

extern "C" cr_index_builder* cr_index_builder_create(int M, int V, int T_pos, int ng, int ent) {
    cr_index_layout L;
    if (cr_batch_index_layout(M, V, T_pos, ng, ent, &L) != CR_OK) return nullptr;
    cr_index_builder* b = new (std::nothrow) cr_index_builder();
    if (!b) return nullptr;
    b->L = L;
    try {
        b->count.assign((size_t)V + T_pos, 0);
        b->cursor.assign((size_t)V + T_pos, 0);
        const size_t distinct = (size_t)((int64_t)V - 1 < (int64_t)3 * M ? (int64_t)V - 1 : (int64_t)3 * M) + T_pos;
        b->uniq.reserve(distinct);
        b->start.reserve(distinct);
        b->by_size.reserve(distinct);
    } catch (...) {
        delete b;
        return nullptr;
    }
    return b;
}

extern "C" void cr_index_builder_destroy(cr_index_builder* b) { delete b; }

extern "C" int cr_index_build(cr_index_builder* b, const int32_t* seq, const int32_t* pos, const int32_t* neg, int32_t* out) {
    if (!b || !seq || !pos || !neg || !out) return cr_set_error(CR_ERR_INVALID, "cr_index_build: NULL pointer");
    const cr_index_layout& L = b->L;
    const int M = L.M, V = L.V, T = L.T_pos, NG = L.ng, ENT = L.ent;
    int32_t* cnt = b->count.data();
    int32_t* cur = b->cursor.data();
    std::vector<int32_t>& uniq = b->uniq;
    uniq.clear();
    const int32_t* lists[3] = {seq, pos, neg};
    // pass 1: counts, rows in order of first appearance (ids outside the table are the caller's to refuse; here they are skipped
    // so that no write leaves the work arrays)
    int bad = 0;
    for (int k = 0; k < 3; ++k) {
        const int32_t* ids = lists[k];
        for (int m = 0; m < M; ++m) {
            const int32_t id = ids[m];
            if (id == 0) continue;
            if (id < 0 || id >= V) { bad = 1; continue; }
            if (cnt[id]++ == 0) uniq.push_back(id);
        }
    }
    if (T > 0) {
        const int B = M / T;
        for (int t = 0; t < T; ++t) { cnt[V + t] = B; uniq.push_back(V + t); }
    }
    const size_t nu = uniq.size();
    // occurrence ranges in order of first appearance
    b->start.resize(nu);
    int32_t at = 0;
    for (size_t u = 0; u < nu; ++u) {
        const int32_t row = uniq[u];
        b->start[u] = at;
        cur[row] = at;
        at += cnt[row];
    }
    const int32_t n_occ = at;
    // the plan.  Group count of a row: k = ceil(c / ENT); rows beyond a workgroup (k > NG) are cut into slices first.
    int32_t* hdr = out;
    int32_t* recs = out + L.off_recs;
    int64_t n_blocks = 0;
    auto rec_at = [&](int64_t blk, int g) { return recs + 4 * (blk * NG + g); };
    auto fill_row = [&](int64_t blk, int g0, int32_t row, int32_t st, int32_t c, int k, int sidx, int nsl) {
        const int32_t per = (c + k - 1) / k;                 // occurrences per group, evenly (<= ENT)
        for (int q = 0; q < k; ++q) {
            int32_t* r = rec_at(blk, g0 + q);
            const int32_t s = st + (q * per < c ? q * per : c), e = st + ((q + 1) * per < c ? (q + 1) * per : c);
            r[0] = row; r[1] = s; r[2] = e - s;
            r[3] = (int32_t)((uint32_t)q | ((uint32_t)k << 6) | ((uint32_t)sidx << 13) | ((uint32_t)nsl << 22));
        }
    };
    auto pad_block = [&](int64_t blk, int g0) {
        for (int g = g0; g < NG; ++g) { int32_t* r = rec_at(blk, g); r[0] = 0; r[1] = 0; r[2] = 0; r[3] = 0; }
    };
    int rc = CR_OK;
    // (a) rows of more than a workgroup: slices of whole workgroups, slice j of n at workgroup first + j
    for (size_t u = 0; u < nu && rc == CR_OK; ++u) {
        const int32_t row = uniq[u], c = cnt[row];
        const int64_t k = ((int64_t)c + ENT - 1) / ENT;
        if (k <= NG) continue;
        const int64_t per_blk = (int64_t)ENT * NG;
        const int64_t nsl = (c + per_blk - 1) / per_blk;
        if (nsl > 511 || n_blocks + nsl > L.cap_blocks) { rc = cr_set_error(CR_ERR_INVALID, "cr_index_build: a row with %d occurrences exceeds the plan's capacity", c); break; }
        const int32_t per = (int32_t)((c + nsl - 1) / nsl);
        for (int64_t j = 0; j < nsl; ++j) {
            const int32_t s = (int32_t)(j * per < c ? j * per : c), e = (int32_t)((j + 1) * per < c ? (j + 1) * per : c);
            const int kk = (int)((e - s + ENT - 1) / ENT);
            fill_row(n_blocks, 0, row, b->start[u] + s, e - s, kk, (int)j, (int)nsl);
            pad_block(n_blocks, kk);
            ++n_blocks;
        }
    }
    // (b) the rest: counting sort by group count, then workgroups filled with the largest row that still fits
    std::vector<int32_t>& by = b->by_size;
    by.resize(nu);
    std::vector<int32_t> first((size_t)NG + 2, 0), left((size_t)NG + 2, 0);
    for (size_t u = 0; u < nu; ++u) {
        const int64_t k = ((int64_t)cnt[uniq[u]] + ENT - 1) / ENT;
        if (k <= NG) ++left[(size_t)k];
    }
    {
        int32_t p = 0;
        for (int k = NG; k >= 1; --k) { first[(size_t)k] = p; p += left[(size_t)k]; }
        std::vector<int32_t> fillp(first);
        for (size_t u = 0; u < nu; ++u) {
            const int64_t k = ((int64_t)cnt[uniq[u]] + ENT - 1) / ENT;
            if (k <= NG) by[(size_t)fillp[(size_t)k]++] = (int32_t)u;
        }
    }
    {
        int64_t remaining = 0;
        for (int k = 1; k <= NG; ++k) remaining += left[(size_t)k];
        int kmax = NG;
        while (remaining > 0 && rc == CR_OK) {
            if (n_blocks >= L.cap_blocks) { rc = cr_set_error(CR_ERR_INVALID, "cr_index_build: the plan exceeds its capacity (%d workgroups)", L.cap_blocks); break; }
            int g = 0;
            while (kmax >= 1 && left[(size_t)kmax] == 0) --kmax;
            int k = kmax;
            while (g < NG && k >= 1) {
                if (k > NG - g) k = NG - g;
                while (k >= 1 && left[(size_t)k] == 0) --k;
                if (k < 1) break;
                const int32_t u = by[(size_t)first[(size_t)k]++];
                --left[(size_t)k];
                --remaining;
                fill_row(n_blocks, g, uniq[(size_t)u], b->start[(size_t)u], cnt[uniq[(size_t)u]], k, 0, 1);
                g += k;
            }
            pad_block(n_blocks, g);
            ++n_blocks;
        }
    }
    // occurrences: the same traversal as pass 1 fills the ranges -- stable: within a row kind 0 by ascending m, then 1, then 2
    const int64_t off_occ = align4(L.off_recs + (int64_t)4 * NG * n_blocks);
    const int64_t off_bitmap = off_occ + align4(n_occ);
    const int64_t used = off_bitmap + align4(L.bitmap_words);
    int32_t* occ = out + off_occ;
    for (int k = 0; k < 3; ++k) {
        const int32_t* ids = lists[k];
        const int32_t tag = (int32_t)((uint32_t)k << 30);
        for (int m = 0; m < M; ++m) {
            const int32_t id = ids[m];
            if (id <= 0 || id >= V) continue;
            occ[cur[id]++] = tag | m;
        }
    }
    if (T > 0) {
        const int B = M / T;
        for (int t = 0; t < T; ++t)
            for (int bb = 0; bb < B; ++bb) occ[cur[V + t]++] = (int32_t)(3u << 30) | (bb * T + t);
    }
    for (int64_t i = n_occ; i < align4(n_occ); ++i) occ[i] = 0;
    uint32_t* bits = reinterpret_cast<uint32_t*>(out + off_bitmap);
    memset(bits, 0, (size_t)align4(L.bitmap_words) * 4);
    for (size_t u = 0; u < nu; ++u) {
        const int32_t row = uniq[u];
        bits[(uint32_t)row >> 5] |= 1u << (row & 31);
        cnt[row] = 0;                                       // the work array is all zero again
    }
    for (int64_t i = L.off_recs + (int64_t)4 * NG * n_blocks; i < off_occ; ++i) out[i] = 0;
    hdr[0] = (int32_t)n_blocks; hdr[1] = (int32_t)nu; hdr[2] = n_occ; hdr[3] = CR_INDEX_MAGIC;
    hdr[4] = (int32_t)used; hdr[5] = (int32_t)off_occ; hdr[6] = (int32_t)off_bitmap; hdr[7] = 0;
    if (rc != CR_OK) return rc;
    if (bad) return cr_set_error(CR_ERR_INVALID, "cr_index_build: an id lies outside [0, %d)", V);
    return CR_OK;
}
