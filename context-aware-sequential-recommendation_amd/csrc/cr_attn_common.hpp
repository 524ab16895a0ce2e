// Shared pieces of the attention kernels (cr_attn_fwd.hip, cr_attn_bwd.hip).
//
// Causal multi-head self-attention core of modules.py:208-269 (scores, key / causal / query masks,
// softmax, dropout, weighted sum, residual); scores never leave the chip.
//
// Work decomposition: grid = (H*B, nsplit); a workgroup (up to 8 waves) stages the K and V rows of ONE
// (head, sample) in LDS; each wave owns 16-query tiles and keeps the whole score row block
// (NKT key tiles x 4 registers) in VGPRs, so softmax is a register + 2-shuffle reduction.
//
// MFMA trick (v_mfma_f32_16x16x4_f32): scores are computed TRANSPOSED, St[key][query] = K Q^T, so a
// lane holds, for ITS query (lane & 15), keys {16*kt + 4*(lane>>4) + r}.  That accumulator layout is
// exactly the A-operand layout of the next product (P V, dS K) when the k-dimension of MFMA step r
// is taken as key 4*lg + r -- no LDS round trip, no cross-lane movement between the two GEMMs.
// The key-owner backward kernel uses the mirrored form (S[query][key]) for dK / dV.
//
// Code shape (second rewrite, see DESIGN.md "attention"): NKT (key tiles) and NDS (head-dim k-steps of
// 4) are template parameters so every MFMA group is preceded by ONE batch of LDS operand reads (the
// per-MFMA read->wait->issue pattern of the first versions exposed the LDS latency 200 times per tile)
// and the per-wave Q / dOut / K / V fragments live in registers.
//
// Exactness notes (tests/test_ops_gpu.py): masked entries are -2^32+1 in the reference, so
//  * a row with >= 1 valid key: masked probabilities are exactly 0;
//  * a row with NO valid key ("uniform"): probability 1/T on ALL T keys, future ones included
//    (modules.py:227-244) -- handled explicitly, contributes to out and to dV, no score gradient.
#pragma once
#include <math.h>

#include "cr_common.hpp"

#define A_MAX_WAVES 8        // two waves per SIMD hide each other's LDS / MFMA latencies
#define A_TAIL 64            // floats of slack after a B-pattern-read LDS array (reads of padded columns)

struct AttnGeom {
    int T16, nkt;            // padded T, number of 16-tiles
    int nds, ndt;            // template values in use: k-steps (pitch = 4*nds + 2), 16-column output tiles
    int PA, PB;              // LDS pitches: 4*nds+2 (A-pattern, % 4 == 2); B-pattern (% 8 == 4)
    float isd;               // 1/sqrt(d)   (modules.py:219)
    float isd_log2e;         // isd * log2(e): softmax exponent in base 2 (v_exp_f32)
    float invT;
    unsigned long long* ts;  // debug: per-wave phase stamps [waves][16] (tools/attn_ts.py); NULL in production
};
// debug-only phase stamps: slots 0 / 15 = wall clock (100 MHz), others = s_memtime; first tile of a wave only
#define AT_TS(slot)                                                                                          \
    do {                                                                                                     \
        if (g.ts && (threadIdx.x & 63) == 0)                                                                 \
            g.ts[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] = \
                ((slot) == 0 || (slot) == 15) ? wall_clock64() : clock64();                                  \
    } while (0)
extern unsigned long long* g_attn_ts;
extern int g_attn_ts_which;   // 0 = forward, 1 = backward (query-owner), 2 = backward (key-owner)

__device__ __forceinline__ float grp_max(float v) {   // over the 4 lanes that share (lane & 15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float grp_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// dropout element index of attention_weights[(j*Bglobal + n), q, 0]
__device__ __forceinline__ uint32_t attn_row_idx(const cr_attn_desc& d, int head, int n, int q) {
    const uint32_t ng = d.drop.row_offset / (uint32_t)d.T + (uint32_t)n;
    return (((uint32_t)head * (uint32_t)d.batch_global + ng) * (uint32_t)d.T + (uint32_t)q) * (uint32_t)d.T;
}

// branch-free dropout factor: keep ? scale : 0  (all lanes hash; no divergent control flow)
__device__ __forceinline__ float drop_factor(const DropCtx& c, uint32_t idx) {
    const uint32_t h = cr_fmix32(idx * 0x9E3779B1u + c.key);
    return (h >= c.thresh) ? c.scale : 0.0f;
}

// D[i][j] = sum_k A[i][k] * frag[k][j]: A = 16-row LDS tile read in the A-pattern (row li, column 4s+lg),
// the other operand is a register fragment (lane holds element [4s+lg] of ITS row li).  The NDS reads
// are issued as one batch ahead of the MFMA chain.
template <int NDS>
__device__ __forceinline__ f32x4 mma_tile_frag(const float* a_tile, int pa, const float (&frag)[NDS]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const float* ap = a_tile + li * pa + lg;
    float a[NDS];
#pragma unroll
    for (int s = 0; s < NDS; ++s) a[s] = ap[4 * s];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NDS; ++s) acc = mfma16(a[s], frag[s], acc);
    return acc;
}

// register fragment of a 16-row LDS tile: element s = tile[li][4s + lg]
template <int NDS>
__device__ __forceinline__ void load_frag(const float* tile, int pa, float (&frag)[NDS]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int s = 0; s < NDS; ++s) frag[s] = tile[li * pa + 4 * s + lg];
}

// Tile schedule.  Tiles are ranked by weight (rank 0 = heaviest: the causal triangle makes query tile qt cost
// ~qt + 1 key tiles) and dealt in serpentine rounds over P = nsplit * nw/2 SIMD slots: waves w and w + nw/2 of
// a workgroup share a SIMD, the first takes the even rounds, the second the odd ones (reversed), so every SIMD
// gets a heavy and a light tile.  T = 200 (13 tiles, 2 x 8 waves): worst SIMD 13 units instead of 18.
struct TileSched { int P, p, r, R; };
__device__ __forceinline__ TileSched sched_init(int nw, int wave) {
    TileSched s;
    s.R = nw >= 2 ? 2 : 1;
    const int half = nw / s.R;
    s.P = (int)gridDim.y * half;
    s.p = (wave % half) * (int)gridDim.y + (int)blockIdx.y;
    s.r = wave / half;
    return s;
}
__device__ __forceinline__ int sched_rank_at(const TileSched& s, int r) { return r * s.P + ((r & 1) ? s.P - 1 - s.p : s.p); }
__device__ __forceinline__ int sched_rank(const TileSched& s) { return sched_rank_at(s, s.r); }
__device__ __forceinline__ int sched_peek(const TileSched& s) { return sched_rank_at(s, s.r + s.R); }
__device__ __forceinline__ int sched_next(TileSched& s) { s.r += s.R; return sched_rank(s); }

// Register fragment straight from global memory, in two halves so the loads stay in flight behind other
// work: frag_issue starts the NDS loads of element s = M[row0 + li][hoff + 4s + lg] (addresses clamped
// instead of predicated: no branches; a wave touches 16 rows x 16 B per load, L2-resident activations);
// frag_finish zeroes what lies outside the valid rows / columns (first use = the wait).
template <int NDS>
__device__ __forceinline__ void frag_issue(const float* src, int ld, int row0, int hoff, int nrows_valid, int d, float (&raw)[NDS]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const float* p = src + (size_t)(row0 + (li < nrows_valid ? li : 0)) * ld + hoff;
#pragma unroll
    for (int s = 0; s < NDS; ++s) {
        const int c = 4 * s + lg;
        raw[s] = p[c < d ? c : 0];
    }
}
template <int NDS>
__device__ __forceinline__ void frag_finish(const float (&raw)[NDS], int nrows_valid, int d, float (&frag)[NDS]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const bool rok = li < nrows_valid;
#pragma unroll
    for (int s = 0; s < NDS; ++s) frag[s] = (rok && 4 * s + lg < d) ? raw[s] : 0.0f;
}

// stage rows [0,nrows_valid) x [hoff, hoff+d) of a [M, ld] matrix into LDS with pitch P <= 68; everything
// outside (rows >= nrows_valid, columns >= d up to the pitch) is zero so padded k-steps contribute 0.
// `nw` waves take part, `tr` is the calling wave's index among them.
// Fast path (single head, ld == d, 16-byte aligned): the rows are one contiguous stream, read as float4
// with 4 loads in flight per lane and scattered into the pitched rows.  Otherwise a wave copies one row
// per pass (4*d-byte segments), 8 rows in flight.
__device__ __forceinline__ void stage_rows(float* dst, int P, const float* src, int ld, int row0, int hoff,
                                           int nrows_valid, int d, int nrows, int tr, int nw) {
    const int tc = threadIdx.x & 63;
    const float* base = src + (size_t)row0 * ld + hoff;
    if (ld == d && ((reinterpret_cast<uintptr_t>(base) & 15) == 0)) {
        const int tid = tr * 64 + tc, nth = nw * 64;
        const int nv = nrows_valid < 0 ? 0 : (nrows_valid < nrows ? nrows_valid : nrows);
        const int total = nv * d, n4 = total >> 2;
        const float inv_d = 1.0f / (float)d;
        const float4* b4 = reinterpret_cast<const float4*>(base);
        constexpr int U4 = 4;
        for (int i0 = tid; i0 < n4; i0 += nth * U4) {
            float4 v[U4];
#pragma unroll
            for (int u = 0; u < U4; ++u) {
                const int i = i0 + u * nth;
                v[u] = b4[i < n4 ? i : n4 - 1];
            }
#pragma unroll
            for (int u = 0; u < U4; ++u) {
                const int i = i0 + u * nth;
                if (i < n4) {
                    const int idx = 4 * i;
                    int r = (int)(((float)idx + 0.5f) * inv_d);      // exact: idx < 2^16, |(idx+.5)/d - integer| >= 1/(2d)
                    int c = idx - r * d;
                    const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        dst[r * P + c] = e[k];
                        if (++c == d) { c = 0; ++r; }
                    }
                }
            }
        }
        for (int i = 4 * n4 + tid; i < total; i += nth) {                // at most 3 elements
            const int r = i / d;
            dst[r * P + (i - r * d)] = base[i];
        }
        const int npad = P - d;
        for (int i = tid; i < nv * npad; i += nth) {
            const int r = i / npad;
            dst[r * P + d + (i - r * npad)] = 0.0f;
        }
        for (int i = nv * P + tid; i < nrows * P; i += nth) dst[i] = 0.0f;
        return;
    }
    constexpr int U = 8;
    for (int t0 = tr; t0 < nrows; t0 += nw * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + nw * u;
            v[u] = (t < nrows_valid && tc < d) ? src[(size_t)(row0 + t) * ld + hoff + tc] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + nw * u;
            if (t < nrows) {
                if (tc < P) dst[t * P + tc] = v[u];
                if (tc + 64 < P) dst[t * P + tc + 64] = 0.0f;   // d <= 64: always padding
            }
        }
    }
}

// Two matrices of the same shape at once (K and V, or Q and dOut): when both take the flat float4 path all
// their loads are issued before the first LDS write, so the staging costs one memory latency instead of one
// per matrix and per round (per-wave timelines, tools/attn_ts.py: 5-6 us -> the dominant start-up cost).
__device__ __forceinline__ void stage_rows2(float* dstA, int PA_, const float* srcA, int ldA, float* dstB, int PB_,
                                            const float* srcB, int ldB, int row0, int hoff, int nrows_valid, int d,
                                            int nrows, int tr, int nw) {
    const float* baseA = srcA + (size_t)row0 * ldA + hoff;
    const float* baseB = srcB + (size_t)row0 * ldB + hoff;
    const bool flat = ldA == d && ldB == d && ((reinterpret_cast<uintptr_t>(baseA) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(baseB) & 15) == 0);
    if (!flat) {
        stage_rows(dstA, PA_, srcA, ldA, row0, hoff, nrows_valid, d, nrows, tr, nw);
        stage_rows(dstB, PB_, srcB, ldB, row0, hoff, nrows_valid, d, nrows, tr, nw);
        return;
    }
    const int tc = threadIdx.x & 63, tid = tr * 64 + tc, nth = nw * 64;
    const int nv = nrows_valid < 0 ? 0 : (nrows_valid < nrows ? nrows_valid : nrows);
    const int total = nv * d, n4 = total >> 2;
    const float inv_d = 1.0f / (float)d;
    const float4* a4 = reinterpret_cast<const float4*>(baseA);
    const float4* b4 = reinterpret_cast<const float4*>(baseB);
    constexpr int U4 = 8;
    for (int i0 = tid; i0 < n4; i0 += nth * U4) {
        float4 va[U4], vb[U4];
#pragma unroll
        for (int u = 0; u < U4; ++u) {
            const int i = i0 + u * nth;
            va[u] = a4[i < n4 ? i : n4 - 1];
        }
#pragma unroll
        for (int u = 0; u < U4; ++u) {
            const int i = i0 + u * nth;
            vb[u] = b4[i < n4 ? i : n4 - 1];
        }
#pragma unroll
        for (int u = 0; u < U4; ++u) {
            const int i = i0 + u * nth;
            if (i < n4) {
                const int idx = 4 * i;
                int r = (int)(((float)idx + 0.5f) * inv_d);
                int c = idx - r * d;
                const float ea[4] = {va[u].x, va[u].y, va[u].z, va[u].w};
                const float eb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    dstA[r * PA_ + c] = ea[k];
                    dstB[r * PB_ + c] = eb[k];
                    if (++c == d) { c = 0; ++r; }
                }
            }
        }
    }
    for (int i = 4 * n4 + tid; i < total; i += nth) {                // at most 3 elements
        const int r = i / d;
        dstA[r * PA_ + (i - r * d)] = baseA[i];
        dstB[r * PB_ + (i - r * d)] = baseB[i];
    }
    for (int i = tid; i < nv * (PA_ - d); i += nth) {
        const int r = i / (PA_ - d);
        dstA[r * PA_ + d + (i - r * (PA_ - d))] = 0.0f;
    }
    for (int i = tid; i < nv * (PB_ - d); i += nth) {
        const int r = i / (PB_ - d);
        dstB[r * PB_ + d + (i - r * (PB_ - d))] = 0.0f;
    }
    for (int i = nv * PA_ + tid; i < nrows * PA_; i += nth) dstA[i] = 0.0f;
    for (int i = nv * PB_ + tid; i < nrows * PB_; i += nth) dstB[i] = 0.0f;
}

// per-lane key-validity bits: bit (4*kt + r) <=> key 16*kt + 4*lg + r is a valid key (< T, k_valid != 0)
template <int NKT>
__device__ __forceinline__ uint64_t key_bits(const float* kv, int T) {
    const int lg = (threadIdx.x & 63) >> 4;
    uint64_t bits = 0;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * lg + r;
            if (key < T && kv[key] != 0.0f) bits |= (1ull << (4 * kt + r));
        }
    }
    return bits;
}

// first 16-key tile that holds a valid key (wave-uniform); NKT if none.  Left-padded sequences make the
// first tiles all-invalid: their probabilities are exactly 0, so their MFMAs are skipped.
template <int NKT>
__device__ __forceinline__ int first_valid_tile(uint64_t kbits) {
    // a tile's 16 keys are spread over the 4 lane groups: OR the per-lane bits across lg with two shuffles
    uint32_t lo = (uint32_t)kbits, hi = (uint32_t)(kbits >> 32);
    lo |= __shfl_xor(lo, 16, 64); lo |= __shfl_xor(lo, 32, 64);
    hi |= __shfl_xor(hi, 16, 64); hi |= __shfl_xor(hi, 32, 64);
    const uint64_t all = ((uint64_t)hi << 32) | lo;
    int t = NKT;
#pragma unroll
    for (int kt = NKT - 1; kt >= 0; --kt)
        if ((all >> (4 * kt)) & 0xFull) t = kt;
    return __builtin_amdgcn_readfirstlane(t);
}

// Score row block of one 16-query tile.  On return st[kt][r] = SOFTMAX probability (before query mask /
// dropout) of key 16*kt + 4*lg + r for query q0 + li; m2 = row max in base-2 exponent units,
// inv = 1/sum; `uniform` marks rows with no valid key.
template <int NKT, int NDS>
__device__ __forceinline__ void score_rows(const AttnGeom& g, const float* Ks, const float (&qf)[NDS], uint64_t kbits,
                                           int kt_lo, int qt, int T, bool is_dead, bool q_in_range, f32x4 (&st)[NKT],
                                           float& m2, float& inv, bool& uniform) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    float mx = -INFINITY;
    const float c2 = is_dead ? 0.0f : g.isd_log2e;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        f32x4 acc = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (kt >= kt_lo && kt <= qt) {                                           // wave-uniform; tiles below kt_lo hold no valid key
            acc = mma_tile_frag<NDS>(Ks + 16 * kt * g.PA, g.PA, qf);             // St tile (modules.py:216)
            const bool below_diag = kt < qt;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = (((kbits >> (4 * kt + r)) & 1ull) != 0) && !is_dead &&
                                   (below_diag || (4 * lg + r <= li));           // key mask + causal (modules.py:222-241)
                const float sv = valid ? acc[r] * c2 : -INFINITY;               // scale (modules.py:219), base 2
                acc[r] = sv;
                mx = fmaxf(mx, sv);
            }
        }
        st[kt] = acc;
    }
    mx = grp_max(mx);
    uniform = (mx == -INFINITY) && !is_dead && q_in_range;
    const float off = (mx == -INFINITY) ? 0.0f : mx;
    float sum = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt >= kt_lo && kt <= qt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[kt][r] - off);   // v_exp_f32; exp2(-inf) = 0 for masked entries
                st[kt][r] = p;
                sum += p;
            }
        } else {
            st[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    sum = grp_sum(sum);
    inv = sum > 0.0f ? 1.0f / sum : 0.0f;
    m2 = mx;
    const float uni = uniform ? g.invT : 0.0f;                                   // modules.py:227-244
    const float sc = uniform ? 0.0f : inv;                                       // modules.py:244
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float u = (16 * kt + 4 * lg + r < T) ? uni : 0.0f;
            st[kt][r] = st[kt][r] * sc + u;
        }
    }
}

// out[q][16 cols] tiles: acc[jt] += sum over keys of st[kt][r] * Bs[key][16*jt + li]  (B-pattern reads,
// one batch of 4*NDT operands per key tile ahead of its 4*NDT MFMAs)
template <int NKT, int NDT>
__device__ __forceinline__ void mma_prob_rows(const f32x4 (&st)[NKT], const float* Bs, int pb, int kt_lo, int kt_end, f32x4 (&acc)[NDT]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const float* bp = Bs + (4 * lg) * pb + li;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt >= kt_lo && kt < kt_end) {
            float b[4][NDT];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jt = 0; jt < NDT; ++jt) b[r][jt] = bp[(16 * kt + r) * pb + 16 * jt];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jt = 0; jt < NDT; ++jt) acc[jt] = mfma16(st[kt][r], b[r][jt], acc[jt]);
        }
    }
}

// geometry shared by host launchers ---------------------------------------------------------------
static inline int attn_pick_nds(int d) { return d <= 32 ? 8 : (d <= 52 ? 13 : 16); }
static inline int attn_pick_nkt(int nkt) { return nkt <= 4 ? 4 : (nkt <= 13 ? 13 : 16); }

// general-shape fallback (cr_attn_wide.hip) for shapes outside the LDS-resident envelope
int cr_attn_wide_supported(const cr_attn_desc* d);
int cr_attn_wide_fwd_launch(const cr_attn_desc* d, hipStream_t s);
int cr_attn_wide_bwd_launch(const cr_attn_bwd_desc* bd, hipStream_t s);

static int attn_validate(const cr_attn_desc* d, const char* who) {
    CR_REQUIRE(d->Q && d->K && d->V && d->k_valid && d->q_valid, "%s: NULL pointer", who);
    CR_REQUIRE(d->B > 0 && d->T > 0 && d->H > 0 && d->d > 0, "%s: bad shape B=%d T=%d H=%d d=%d", who, d->B, d->T, d->H, d->d);
    if (!cr_attn_wide_supported(d))
        return cr_set_error(CR_ERR_UNSUPPORTED, "%s: T=%d (max 1024) / head dim %d (max 256) not supported", who, d->T, d->d);
    CR_REQUIRE(d->ld >= d->H * d->d, "%s: ld too small", who);
    CR_REQUIRE(d->batch_global >= d->B, "%s: batch_global < B", who);
    return CR_OK;
}
// the MFMA kernels keep K and V of one (sample, head) resident in LDS: T <= 256, head dim <= 64
static inline bool attn_lds_envelope(const cr_attn_desc* d) { return d->T <= 256 && d->d <= 64; }

static int attn_geom(const cr_attn_desc* d, AttnGeom* g, const char* who) {
    int rc = attn_validate(d, who);
    if (rc) return rc;
    g->T16 = (d->T + 15) / 16 * 16;
    g->nkt = g->T16 / 16;
    g->nds = attn_pick_nds(d->d);
    g->ndt = g->nds == 8 ? 2 : 4;
    g->PA = 4 * g->nds + 2;                                   // % 4 == 2: conflict-free A-pattern reads
    g->PB = 4 * g->nds + (((4 * g->nds) % 8 == 4) ? 0 : 4);   // % 8 == 4: conflict-free B-pattern reads
    g->isd = (float)(1.0 / sqrt((double)d->d));
    g->isd_log2e = (float)(1.4426950408889634 / sqrt((double)d->d));
    g->invT = 1.0f / (float)d->T;
    g->ts = g_attn_ts;
    return CR_OK;
}

static inline int attn_nsplit(const cr_attn_desc* d, const AttnGeom& g, int waves) {
    int want = (256 + d->B * d->H - 1) / (d->B * d->H);
    int maxs = (g.nkt + waves - 1) / waves;
    if (want > maxs) want = maxs;
    return want < 1 ? 1 : want;
}

static inline int attn_set_lds_attr(const void* fn) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    return CR_OK;
}

// largest wave count (8, 4, 2, 1) whose LDS footprint fits the 160 KiB of a CU; 0 if none does
template <class F>
static int attn_pick_waves(const AttnGeom& g, F lds) {
    for (int w = A_MAX_WAVES; w >= 1; w >>= 1)
        if (lds(g, w) <= 160 * 1024) return w;
    return 0;
}
