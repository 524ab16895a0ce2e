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
    int rot[16];             // single-pass backward: query tile at which key tile kt starts its rotated walk
    unsigned long long* ts;  // debug: per-wave phase stamps [waves][16] (tools/attn_ts.py); NULL in production
};
#ifdef CR_TIMELINE
// debug-only phase stamps: slots 0 / 15 = wall clock (100 MHz), others = s_memtime; first tile of a wave only
#define AT_TS(slot)                                                                                          \
    do {                                                                                                     \
        if (g.ts && (threadIdx.x & 63) == 0)                                                                 \
            g.ts[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] = \
                ((slot) == 0 || (slot) == 15) ? wall_clock64() : clock64();                                  \
    } while (0)
extern unsigned long long* g_attn_ts;
extern int g_attn_ts_which;   // 0 = forward, 1 = backward (query-owner), 2 = backward (key-owner), 3 = single-pass backward
#else
#define AT_TS(slot) do { } while (0)
static unsigned long long* const g_attn_ts = nullptr;
static const int g_attn_ts_which = 0;
#endif

// Reductions over the 4 lanes that share (lane & 15): v_permlane16_swap / v_permlane32_swap (gfx950) exchange the odd
// rows of 16 lanes with the even ones and the upper half of the wave with the lower -- two register-file instructions
// instead of the LDS round trip (address arithmetic + ds_bpermute) __shfl_xor compiles to.
typedef unsigned cr_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float grp_max(float v) {
    cr_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float grp_sum(float v) {
#ifdef CR_NO_PERMLANE_SWAP
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
#else
    cr_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
#endif
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// index of the first valid key of the sample (T if none): rows before it that are not known-dead are the
// "uniform" rows of modules.py:227-244
__device__ __forceinline__ int first_valid_key(const float* k_valid, int base_row, int T) {
    const int lane = threadIdx.x & 63;
    int f = T;
    for (int t = lane; t < T; t += 64)
        if (k_valid[base_row + t] != 0.0f) f = min(f, t);
    return wave_min_i(f);
}

// the same from the additive key bias of a staged chunk (0 = valid), rows [0, n16): one 16-byte LDS read per lane
__device__ __forceinline__ int first_valid_key_lds(const float* kb, int n16, int T) {
    const int lane = threadIdx.x & 63;
    int f = T;
    if (4 * lane < n16) {
        const float4 b = *reinterpret_cast<const float4*>(kb + 4 * lane);
        const int i = b.x == 0.0f ? 0 : (b.y == 0.0f ? 1 : (b.z == 0.0f ? 2 : (b.w == 0.0f ? 3 : 1 << 20)));
        f = min(T, 4 * lane + i);
    }
    return wave_min_i(f);
}

// dropout element index of attention_weights[(j*Bglobal + n), q, 0]
__device__ __forceinline__ uint32_t attn_row_idx(const cr_attn_desc& d, int head, int n, int q) {
    const uint32_t ng = d.drop.row_offset / (uint32_t)d.T + (uint32_t)n;
    return (((uint32_t)head * (uint32_t)d.batch_global + ng) * (uint32_t)d.T + (uint32_t)q) * (uint32_t)d.T;
}

// branch-free dropout factor: keep ? scale : 0  (all lanes hash; no divergent control flow)
__device__ __forceinline__ float drop_factor(const DropCtx& c, uint32_t idx) {
    return drop_factor_x(c, idx * CR_PHI + c.key);
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
    __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);      // keep the operand reads one batch ahead of the chain
    __builtin_amdgcn_sched_group_barrier(0x008, NDS, 0);
    return acc;
}

// Two independent products of that kind with their MFMA chains interleaved: a dependent accumulate chain of
// v_mfma_f32_16x16x4_f32 issues every 40 cycles, two alternating chains every 32 (the pipe rate), and the
// 2 * NDS LDS operand reads go out as one batch.
template <int NDS>
__device__ __forceinline__ void mma_tile_frag2(const float* a_tile0, const float* a_tile1, int pa, const float (&frag0)[NDS],
                                               const float (&frag1)[NDS], f32x4& acc0, f32x4& acc1) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const float* ap0 = a_tile0 + li * pa + lg;
    const float* ap1 = a_tile1 + li * pa + lg;
    float a0[NDS], a1[NDS];
#pragma unroll
    for (int s = 0; s < NDS; ++s) { a0[s] = ap0[4 * s]; a1[s] = ap1[4 * s]; }
    acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NDS; ++s) {
        acc0 = mfma16(a0[s], frag0[s], acc0);
        acc1 = mfma16(a1[s], frag1[s], acc1);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * NDS, 0);  // all operand reads first, then the two interleaved chains
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NDS, 0);
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

// Row-chunk staging of a PAIR of [T, d] matrices (K and V, or Q and dOut) into pitched LDS tiles.  An item is
// (row r, 4-column chunk q), NDS chunks per row; its 16 bytes are one dword-aligned global load (gfx950 needs
// no more) and two 8-byte LDS writes, so the address arithmetic is a couple of multiply-adds per 4 elements --
// the element-wise scatter of a flat stream cost ~40 VALU instructions per float4 and made the staging the
// most expensive phase of every attention kernel (ISA census: 3200 VALU instructions per wave).
// All loads of a batch (U per matrix) are issued before the first LDS write.  The chunk that crosses column d
// is read shifted back to [d-4, d) and rotated, so nothing outside the matrices is touched (needs d >= 4).
typedef float f4v __attribute__((ext_vector_type(4), aligned(4)));
template <int NDS>
__device__ __forceinline__ void stage_pair(float* dstA, int PA_, const float* srcA, int ldA, float* dstB, int PB_,
                                           const float* srcB, int ldB, int row0, int hoff, int T, int d, int T16) {
    const int tid = threadIdx.x, nth = blockDim.x;
    const int total = T16 * NDS;
    constexpr int U = 6;
    for (int i0 = tid; i0 < total; i0 += nth * U) {
        f4v va[U], vb[U];
        int rr[U], qq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int item = min(i0 + u * nth, total - 1);
            const int r = item / NDS, q = item - r * NDS;
            rr[u] = r; qq[u] = q;
            const int rc = r < T ? r : T - 1;
            const int col0 = min(4 * q, d - 4);
            va[u] = *reinterpret_cast<const f4v*>(srcA + (size_t)(row0 + rc) * ldA + hoff + col0);
            vb[u] = *reinterpret_cast<const f4v*>(srcB + (size_t)(row0 + rc) * ldB + hoff + col0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * nth < total) {
                const int r = rr[u], q = qq[u];
                const int shift = 4 * q - min(4 * q, d - 4);
                const bool rok = r < T;
                float ea[4], eb[4];
                const float a0 = va[u].x, a1 = va[u].y, a2 = va[u].z, a3 = va[u].w;
                const float b0 = vb[u].x, b1 = vb[u].y, b2 = vb[u].z, b3 = vb[u].w;
                ea[0] = shift == 0 ? a0 : (shift == 1 ? a1 : (shift == 2 ? a2 : a3));
                ea[1] = shift == 0 ? a1 : (shift == 1 ? a2 : (shift == 2 ? a3 : 0.0f));
                ea[2] = shift == 0 ? a2 : (shift == 1 ? a3 : 0.0f);
                ea[3] = shift == 0 ? a3 : 0.0f;
                eb[0] = shift == 0 ? b0 : (shift == 1 ? b1 : (shift == 2 ? b2 : b3));
                eb[1] = shift == 0 ? b1 : (shift == 1 ? b2 : (shift == 2 ? b3 : 0.0f));
                eb[2] = shift == 0 ? b2 : (shift == 1 ? b3 : 0.0f);
                eb[3] = shift == 0 ? b3 : 0.0f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool ok = rok && (4 * q + t < d);
                    ea[t] = ok ? ea[t] : 0.0f;
                    eb[t] = ok ? eb[t] : 0.0f;
                }
                float2* pa = reinterpret_cast<float2*>(dstA + r * PA_ + 4 * q);
                float2* pb = reinterpret_cast<float2*>(dstB + r * PB_ + 4 * q);
                pa[0] = make_float2(ea[0], ea[1]); pa[1] = make_float2(ea[2], ea[3]);
                pb[0] = make_float2(eb[0], eb[1]); pb[1] = make_float2(eb[2], eb[3]);
            }
        }
    }
}

// First 16-key tile that holds a valid key (wave-uniform); NKT if none.  Left-padded sequences make the first
// tiles all-invalid: their probabilities are exactly 0, so their MFMAs are skipped.  `kb` is the additive key
// bias in LDS (0 = valid).  Lane kt inspects tile kt with four 16-byte reads and the wave takes the lowest set
// bit of the ballot.  (The first version built a 64-bit validity mask per lane with 52 guarded LDS reads; the
// compiler serialised them -- read, wait, branch -- and the per-wave timeline showed 7 us between the staging
// barrier and the first MFMA.)
template <int NKT>
__device__ __forceinline__ int first_valid_tile(const float* kb, int nkt) {
    const int lane = threadIdx.x & 63;
    const int kt = lane < nkt ? lane : nkt - 1;
    const float4* p = reinterpret_cast<const float4*>(kb + 16 * kt);
    const float4 a = p[0], b = p[1], c = p[2], e = p[3];
    const float m = fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
    const float n = fmaxf(fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)), fmaxf(fmaxf(e.x, e.y), fmaxf(e.z, e.w)));
    const bool has = (fmaxf(m, n) == 0.0f) && lane < nkt;                        // max over {0, -inf} biases
    const unsigned long long mask = __ballot(has ? 1 : 0);
    return mask ? (int)__builtin_ctzll(mask) : NKT;
}

// Score row block of one 16-query tile.  On return st[kt][r] = SOFTMAX probability (before query mask /
// dropout) of key 16*kt + 4*lg + r for query q0 + li; m2 = row max in base-2 exponent units,
// inv = 1/sum; `uniform` marks rows with no valid key.
// `kb` is the additive key bias in LDS (0 = valid key, -inf = masked or beyond T), read 4 keys at a time: the
// mask costs one fused multiply-add per score; only the diagonal tile adds the causal compare.  (The first
// version derived every element's validity from bit tests and boolean logic: 34 VALU + 22 SALU instructions per
// score, more issue time than the MFMAs.)  Rows whose item id is 0 ("dead") are computed like any other and
// forced to probability 0 at the end.
template <int NKT, int NDS>
__device__ __forceinline__ void score_rows(const AttnGeom& g, const float* Ks, const float (&qf)[NDS], const float* kb,
                                           int kt_lo, int qt, int T, bool is_dead, bool q_in_range, f32x4 (&st)[NKT],
                                           float& m2, float& inv, bool& uniform, bool fine = false) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    float mx = -INFINITY;
    const float c2 = g.isd_log2e;                                                // scale (modules.py:219), base 2
    auto finish = [&](int kt, f32x4 acc) {                                       // mask + running max of one key tile
        const float4 b4 = *reinterpret_cast<const float4*>(kb + 16 * kt + 4 * lg);   // key mask (modules.py:222-229)
        acc[0] = fmaf(acc[0], c2, b4.x); acc[1] = fmaf(acc[1], c2, b4.y);
        acc[2] = fmaf(acc[2], c2, b4.z); acc[3] = fmaf(acc[3], c2, b4.w);
        if (kt == qt) {                                                          // causal mask on the diagonal tile (modules.py:232-241)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = (4 * lg + r <= li) ? acc[r] : -INFINITY;
        }
        mx = fmaxf(fmaxf(mx, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
        return acc;
    };
    const f32x4 ninf = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int kt = 0; kt < NKT; kt += 2) {                                        // St tiles (modules.py:216), two chains at a time
        const bool c0 = kt >= kt_lo && kt <= qt;                                 // wave-uniform; tiles below kt_lo hold no valid key
        const bool c1 = (kt + 1 < NKT) && kt + 1 >= kt_lo && kt + 1 <= qt;
        f32x4 acc0 = ninf, acc1 = ninf;
        if (c0 && c1) {
            mma_tile_frag2<NDS>(Ks + 16 * kt * g.PA, Ks + 16 * (kt + 1) * g.PA, g.PA, qf, qf, acc0, acc1);
            acc0 = finish(kt, acc0);
            acc1 = finish(kt + 1, acc1);
        } else if (c0) {
            acc0 = finish(kt, mma_tile_frag<NDS>(Ks + 16 * kt * g.PA, g.PA, qf));
        } else if (c1) {
            acc1 = finish(kt + 1, mma_tile_frag<NDS>(Ks + 16 * (kt + 1) * g.PA, g.PA, qf));
        }
        st[kt] = acc0;
        if (kt + 1 < NKT) st[kt + 1] = acc1;
#ifdef CR_TIMELINE
        if (fine) AT_TS(8 + kt / 2);
#endif
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
    if (is_dead) inv = 0.0f;                                                     // dead rows: every probability 0
    if (__any(uniform ? 1 : 0)) {                                                // rare: a row with no valid key at all
        const float uni = uniform ? g.invT : 0.0f;                               // modules.py:227-244
        const float sc = uniform ? 0.0f : inv;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u = (16 * kt + 4 * lg + r < T) ? uni : 0.0f;
                st[kt][r] = st[kt][r] * sc + u;
            }
        }
    } else {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) st[kt] *= inv;                          // modules.py:244
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
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * NDT, 0);   // the key tile's operand reads as one batch,
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NDT, 0);   // then its MFMAs (NDT independent chains)
        }
    }
}

// geometry shared by host launchers ---------------------------------------------------------------
static inline int attn_pick_nds(int d) { return d <= 32 ? 8 : (d <= 52 ? 13 : 16); }
static inline int attn_pick_nkt(int nkt) { return nkt <= 4 ? 4 : (nkt <= 13 ? 13 : 16); }

// single-pass backward (cr_attn_bwd1.hip): 1 = launched, 0 = shape does not fit (fall back to two passes), < 0 = error
struct AttnGeom;
int cr_attn_bwd_single_pass(const cr_attn_bwd_desc* bd, const AttnGeom& g, hipStream_t s);
// general-shape fallback (cr_attn_wide.hip) for shapes outside the LDS-resident envelope
int cr_attn_wide_supported(const cr_attn_desc* d);
int cr_attn_wide_fwd_launch(const cr_attn_desc* d, hipStream_t s);
int cr_attn_wide_bwd_launch(const cr_attn_bwd_desc* bd, hipStream_t s);

// bf16-MFMA kernels (cr_attn_bf.hip): CR_PREC_BF16X3 / CR_PREC_BF16
bool cr_attn_bf_supported_fwd(const cr_attn_desc* d);
bool cr_attn_bf_supported_bwd(const cr_attn_desc* d);
int cr_attn_bf_fwd_launch(const cr_attn_desc* d, hipStream_t s);
int cr_attn_bf_bwd_launch(const cr_attn_bwd_desc* bd, hipStream_t s);
int cr_attn_zero_cols_launch(float* p, int ld, int M, int C, hipStream_t s);

static int attn_validate(const cr_attn_desc* d, const char* who) {
    CR_REQUIRE(d->Q && d->K && d->V && d->k_valid && d->q_valid, "%s: NULL pointer", who);
    CR_REQUIRE(d->B > 0 && d->T > 0 && d->H > 0 && d->d > 0, "%s: bad shape B=%d T=%d H=%d d=%d", who, d->B, d->T, d->H, d->d);
    if (!cr_attn_wide_supported(d))
        return cr_set_error(CR_ERR_UNSUPPORTED, "%s: T=%d (max 1024) / head dim %d (max 256) not supported", who, d->T, d->d);
    CR_REQUIRE(d->ld >= d->H * d->d, "%s: ld too small", who);
    CR_REQUIRE(d->batch_global >= d->B, "%s: batch_global < B", who);
    CR_REQUIRE(d->precision >= CR_PREC_F32 && d->precision <= CR_PREC_BF16, "%s: unknown precision %d", who, d->precision);
    return CR_OK;
}
// the MFMA kernels keep K and V of one (sample, head) resident in LDS: T <= 256, head dim <= 64
static inline bool attn_lds_envelope(const cr_attn_desc* d) { return d->T <= 256 && d->d <= 64 && d->d >= 4; }

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

// largest wave count (8, 4, 2, 1) whose LDS footprint fits the 160 KiB of a CU; 0 if none does
template <class F>
static int attn_pick_waves(const AttnGeom& g, F lds) {
    for (int w = A_MAX_WAVES; w >= 1; w >>= 1)
        if (lds(g, w) <= 160 * 1024) return w;
    return 0;
}

// (shared by cr_stack_bwd1.hip and cr_attn_bf.hip: the deal of an attention pass's tiles over eight waves, at most two per wave, nkt <= 16;
//  wave_of[t] = the wave of tile t, order[] = the tiles heaviest first -- see cr_stack_bwd1.hip b1_deal_tiles for the objective)
inline void cr_deal_search(int nkt, bool query_pass, int (&best_wave)[16], int (&order)[16]) {
    int cost[16];
    for (int t = 0; t < nkt; ++t) { cost[t] = ((query_pass ? t + 1 : nkt - t) + 1) / 2 + 3; order[t] = t; }
    for (int i = 1; i < nkt; ++i)                                          // heaviest first (stable insertion sort)
        for (int j = i; j > 0 && cost[order[j]] > cost[order[j - 1]]; --j) { const int x = order[j]; order[j] = order[j - 1]; order[j - 1] = x; }
    int simd_of[16], sums[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
    for (int t = 0; t < 16; ++t) simd_of[t] = -1;
    long best_key[3] = {1L << 40, 1L << 40, 1L << 40};
    bool have = false;
    // the best split of one SIMD's tiles over its two waves (at most two each): returns the larger wave's load
    auto split = [&](int s, int* wave_of) {
        int ts[4], k = 0;
        for (int t = 0; t < nkt; ++t) if (simd_of[t] == s) ts[k++] = t;
        if (k == 1) { wave_of[ts[0]] = s; return cost[ts[0]]; }           // a lone tile: the lower wave, as ever
        int best_m = 1 << 30, best_mask = 0;
        for (int mask = 0; mask < (1 << k); ++mask) {
            const int na = __builtin_popcount(mask);
            if (na > 2 || k - na > 2) continue;
            int la = 0, lb = 0;
            for (int i = 0; i < k; ++i) ((mask >> i) & 1 ? la : lb) += cost[ts[i]];
            if (la > lb) continue;                                         // the lighter wave is the lower one (ties: either)
            if (lb < best_m) { best_m = lb; best_mask = mask; }
        }
        if (best_m == (1 << 30)) return -1;
        for (int i = 0; i < k; ++i) wave_of[ts[i]] = ((best_mask >> i) & 1) ? s : s + 4;
        return best_m;
    };
    // iterative depth-first search
    int choice[16];
    int i = 0;
    choice[0] = -1;
    while (i >= 0) {
        if (i == nkt) {
            int wave_of[16], wmax = 0;
            bool ok = true;
            for (int s = 0; s < 4 && ok; ++s) {
                if (cnt[s] == 0) continue;
                const int m = split(s, wave_of);
                if (m < 0) ok = false; else if (m > wmax) wmax = m;
            }
            if (ok) {
                long smax = 0, sq = 0;
                for (int s = 0; s < 4; ++s) { if (sums[s] > smax) smax = sums[s]; sq += (long)sums[s] * sums[s]; }
                const long key[3] = {smax, wmax, sq};
                if (!have || key[0] < best_key[0] || (key[0] == best_key[0] && (key[1] < best_key[1] || (key[1] == best_key[1] && key[2] < best_key[2])))) {
                    have = true;
                    for (int k = 0; k < 3; ++k) best_key[k] = key[k];
                    for (int t = 0; t < nkt; ++t) best_wave[t] = wave_of[t];
                }
            }
            --i;
            continue;
        }
        const int t = order[i];
        if (choice[i] >= 0) { const int s = choice[i]; --cnt[s]; sums[s] -= cost[t]; simd_of[t] = -1; }     // undo the last choice at this depth
        int s = choice[i] + 1;
        for (; s < 4; ++s) {
            if (cnt[s] >= 4) continue;
            bool dup = false;                                              // an equal partial SIMD was tried already at this depth
            for (int p = 0; p < s; ++p) if (sums[p] == sums[s] && cnt[p] == cnt[s]) { dup = true; break; }
            if (dup) continue;
            if (have && sums[s] + cost[t] > best_key[0]) continue;
            break;
        }
        if (s >= 4) { choice[i] = -1; --i; continue; }
        choice[i] = s; ++cnt[s]; sums[s] += cost[t]; simd_of[t] = s;
        ++i;
        if (i < 16) choice[i] = -1;
    }
}
