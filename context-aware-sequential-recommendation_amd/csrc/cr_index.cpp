// Occurrence index of a batch (include/castrec.h, "occurrence index"): table row -> the (kind, batch row) pairs that looked it
// up, so that the device forms a table row's gradient as an ordered sum over its occurrences instead of scattering float
// atomics (autodiff of modules.py:157 through sasrec.py:27, of sasrec.py:89-90, and of the positional lookup sasrec.py:40-50).
// Host code only (no HIP).  A stable counting sort in the order "seq ids, pos ids, neg ids, each by ascending row", O(rows of the
// batch): the two work arrays are as long as the table and persist in the builder (a bin is reset through the list of rows the
// batch touched), so a build never walks the table -- config C5's has 10^7 rows, a batch touches 2 % of them.
#include <stdint.h>
#include <string.h>

#include <new>
#include <vector>

#include "castrec.h"

int cr_set_error(int code, const char* fmt, ...);

static inline int64_t align4(int64_t w) { return (w + 3) & ~(int64_t)3; }

extern "C" int cr_batch_index_layout(int M, int V, int T_pos, cr_index_layout* out) {
    if (!out || M < 1 || V < 2 || T_pos < 0 || (T_pos > 0 && M % T_pos != 0) || (int64_t)M >= ((int64_t)1 << 30))
        return cr_set_error(CR_ERR_INVALID, "cr_batch_index_layout: need M in [1, 2^30), V >= 2, T_pos >= 0 dividing M");
    cr_index_layout L;
    memset(&L, 0, sizeof(L));
    L.M = M; L.V = V; L.T_pos = T_pos;
    const int64_t n_occ = (int64_t)3 * M + (T_pos ? M : 0);
    const int64_t distinct = ((int64_t)V - 1 < (int64_t)3 * M ? (int64_t)V - 1 : (int64_t)3 * M) + T_pos;
    if (n_occ > 0x7fffffff) return cr_set_error(CR_ERR_INVALID, "cr_batch_index_layout: batch too large");
    L.cap_light = (int)distinct;
    L.cap_heavy = (int)(n_occ / (CR_INDEX_HEAVY + 1)) + 1;
    L.cap_occ = (int)n_occ;
    L.bitmap_words = (int)(((int64_t)V + T_pos + 31) / 32);
    L.off_light = 8;
    L.off_heavy = align4(L.off_light + (int64_t)4 * L.cap_light);
    L.off_occ = align4(L.off_heavy + (int64_t)4 * L.cap_heavy);
    L.off_bitmap = align4(L.off_occ + L.cap_occ);
    L.total_words = align4(L.off_bitmap + L.bitmap_words);
    *out = L;
    return CR_OK;
}

struct cr_index_builder {
    cr_index_layout L;
    std::vector<int32_t> count;       // per flat row: occurrences in this batch (0 between builds)
    std::vector<int32_t> cursor;      // per flat row: next free slot of its occurrence range
    std::vector<int32_t> uniq;        // flat rows in order of first appearance
};

extern "C" cr_index_builder* cr_index_builder_create(int M, int V, int T_pos) {
    cr_index_layout L;
    if (cr_batch_index_layout(M, V, T_pos, &L) != CR_OK) return nullptr;
    cr_index_builder* b = new (std::nothrow) cr_index_builder();
    if (!b) return nullptr;
    b->L = L;
    try {
        b->count.assign((size_t)V + T_pos, 0);
        b->cursor.assign((size_t)V + T_pos, 0);
        b->uniq.reserve((size_t)L.cap_light + L.cap_heavy);
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
    const int M = L.M, V = L.V, T = L.T_pos;
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
    // units and occurrence ranges
    int32_t* hdr = out;
    int32_t* light = out + L.off_light;
    int32_t* heavy = out + L.off_heavy;
    int32_t* occ = out + L.off_occ;
    uint32_t* bits = reinterpret_cast<uint32_t*>(out + L.off_bitmap);
    memset(bits, 0, (size_t)(L.total_words - L.off_bitmap) * 4);
    int n_light = 0, n_heavy = 0;
    int32_t at = 0;
    for (size_t u = 0; u < uniq.size(); ++u) {
        const int32_t row = uniq[u], c = cnt[row];
        int32_t* rec = (c > CR_INDEX_HEAVY) ? heavy + 4 * (size_t)n_heavy++ : light + 4 * (size_t)n_light++;
        rec[0] = row; rec[1] = at; rec[2] = c; rec[3] = 0;
        cur[row] = at;
        at += c;
        bits[(uint32_t)row >> 5] |= 1u << (row & 31);
    }
    // pass 2: the same traversal fills the ranges -- stable: within a row kind 0 by ascending m, then kind 1, then kind 2
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
    for (size_t u = 0; u < uniq.size(); ++u) cnt[uniq[u]] = 0;              // the work array is all zero again
    // unused tails: zero (the buffer travels to the device as it is; nothing reads beyond the counts, but keep it deterministic)
    memset(light + 4 * (size_t)n_light, 0, (size_t)(L.cap_light - n_light) * 16);
    memset(heavy + 4 * (size_t)n_heavy, 0, (size_t)(L.off_occ - L.off_heavy - 4 * (int64_t)n_heavy) * 4);
    memset(occ + at, 0, (size_t)(L.off_bitmap - L.off_occ - at) * 4);
    memset(light + 4 * (size_t)L.cap_light, 0, (size_t)(L.off_heavy - L.off_light - 4 * (int64_t)L.cap_light) * 4);
    hdr[0] = n_light; hdr[1] = n_heavy; hdr[2] = at; hdr[3] = CR_INDEX_MAGIC;
    hdr[4] = hdr[5] = hdr[6] = hdr[7] = 0;
    if (bad) return cr_set_error(CR_ERR_INVALID, "cr_index_build: an id lies outside [0, %d)", V);
    return CR_OK;
}
