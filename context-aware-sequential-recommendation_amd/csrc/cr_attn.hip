// Causal multi-head self-attention core of modules.py:208-269 (scores, key / causal / query masks,
// softmax, dropout, weighted sum, residual) -- forward and backward, scores never leave the chip.
//
// Work decomposition: grid = (H*B, nsplit); a workgroup (8 waves) stages the K and V rows of ONE
// (head, sample) in LDS; each wave owns 16-query tiles, stages its Q (and dOut) tile in a private LDS
// slot and keeps the whole score row block (<= 16 key tiles x 4 registers) in VGPRs, so softmax is a
// register + 2-shuffle reduction.
//
// MFMA trick (v_mfma_f32_16x16x4_f32): scores are computed TRANSPOSED, St[key][query] = K Q^T, so a
// lane holds, for ITS query (lane & 15), keys {16*kt + 4*(lane>>4) + r}.  That accumulator layout is
// exactly the A-operand layout of the next product (P V, dS K) when the k-dimension of MFMA step r
// is taken as key 4*lg + r -- no LDS round trip, no cross-lane movement between the two GEMMs.
// The key-owner backward kernel uses the mirrored form (S[query][key]) for dK / dV.
//
// Code-shape rule learnt from the first version (256 VGPRs + scratch, 750 branches): every MFMA loop
// is ROLLED with both operands read from LDS; only the loop over the score-register array (kt) is
// statically unrolled.
//
// Exactness notes (tests/test_ops_gpu.py): masked entries are -2^32+1 in the reference, so
//  * a row with >= 1 valid key: masked probabilities are exactly 0;
//  * a row with NO valid key ("uniform"): probability 1/T on ALL T keys, future ones included
//    (modules.py:227-244) -- handled explicitly, contributes to out and to dV, no score gradient.
#include <math.h>

#include "cr_common.hpp"

#define A_THREADS 512        // 8 waves: two per SIMD hide each other's LDS / MFMA latencies
#define A_WAVES (A_THREADS / 64)
#define A_TAIL 64            // floats of slack after a B-pattern-read LDS array (reads of padded columns)

struct AttnGeom {
    int T16, nkt;            // padded T, number of 16-tiles
    int dp;                  // head dim rounded up to 4
    int nds, ndt;            // dp/4, ceil(d/16)
    int PA, PB;              // LDS pitches: dp+2 (A-pattern, % 4 == 2) and dp or dp+4 (B-pattern, % 8 == 4)
    float isd;               // 1/sqrt(d)   (modules.py:219)
    float isd_log2e;         // isd * log2(e): softmax exponent in base 2 (v_exp_f32)
    float invT;
};

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

// D[i][j] += sum_k A[i][k] * B[j][k] over nds k-steps of 4: both operands are 16-row LDS tiles read in
// the A-pattern (lane: row li, column 4s + lg).  Rolled on purpose.
__device__ __forceinline__ f32x4 mma_rows_t(const float* a_tile, int pa, const float* b_tile, int pb, int nds) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const float* ap = a_tile + li * pa + lg;
    const float* bp = b_tile + li * pb + lg;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int s = 0; s < nds; ++s) acc = mfma16(ap[4 * s], bp[4 * s], acc);
    return acc;
}

// stage rows [0,T) x [hoff, hoff+d) of a [M, ld] matrix into LDS with pitch P <= 68; everything
// outside (rows >= T, columns >= d up to the pitch) is zero so padded k-steps contribute 0.
// A wave copies one row per pass (coalesced 4*d-byte segments), 8 rows in flight.
__device__ __forceinline__ void stage_rows(float* dst, int P, const float* src, int ld, int row0, int hoff,
                                           int nrows_valid, int d, int nrows, int tr, int nw) {
    const int tc = threadIdx.x & 63;
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

// Score row block of one 16-query tile.  On return st[kt][r] = SOFTMAX probability (before query mask /
// dropout) of key 16*kt + 4*lg + r for query q0 + li; m2 = row max * log2(e) (base-2 exponent offset),
// inv = 1/sum; `uniform` marks rows with no valid key.
template <int NKT>
__device__ __forceinline__ void score_rows(const AttnGeom& g, const float* Ks, const float* Qw, uint64_t kbits, int qt,
                                           int T, bool is_dead, bool q_in_range, f32x4 (&st)[NKT], float& m2, float& inv,
                                           bool& uniform) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kt <= qt) {
            acc = mma_rows_t(Ks + 16 * kt * g.PA, g.PA, Qw, g.PA, g.nds);          // St tile (modules.py:216)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bool valid = ((kbits >> (4 * kt + r)) & 1ull) != 0 && !is_dead;
                if (kt == qt) valid = valid && (4 * lg + r <= li);               // causal (modules.py:232-241)
                const float sv = valid ? acc[r] * g.isd_log2e : -INFINITY;      // scale (modules.py:219), base 2
                acc[r] = sv;
                mx = fmaxf(mx, sv);
            }
        } else {
            acc = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
        st[kt] = acc;
    }
    mx = grp_max(mx);
    uniform = (mx == -INFINITY) && !is_dead && q_in_range;
    const float off = (mx == -INFINITY) ? 0.0f : mx;
    float sum = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt <= qt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(st[kt][r] - off);        // exp2(-inf) = 0 for masked entries
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
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * lg + r;
            st[kt][r] = uniform ? ((key < T) ? g.invT : 0.0f) : st[kt][r] * inv;   // modules.py:244
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(A_THREADS) void k_attn_fwd(cr_attn_desc d, AttnGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                                   // [T16][PA]  A-pattern reads
    float* Vs = Ks + g.T16 * g.PA;                      // [T16][PB]  B-pattern reads (+ tail)
    float* Qs = Vs + g.T16 * g.PB + A_TAIL;             // [A_WAVES][16][PA] per-wave query tiles
    float* kv = Qs + (blockDim.x >> 6) * 16 * g.PA;       // [T16]
    float* qv = kv + g.T16;                             // [T16]
    float* dead = qv + g.T16;                           // [T16]
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    stage_rows(Ks, g.PA, d.K, d.ld, base_row, hoff, T, d.d, g.T16, wave, (int)(blockDim.x >> 6));
    stage_rows(Vs, g.PB, d.V, d.ld, base_row, hoff, T, d.d, g.T16, wave, (int)(blockDim.x >> 6));
    for (int t = threadIdx.x; t < g.T16; t += blockDim.x) {
        kv[t] = (t < T) ? d.k_valid[base_row + t] : 0.0f;
        qv[t] = (t < T) ? d.q_valid[base_row + t] : 0.0f;
        dead[t] = (t >= T || (d.dead_ids && d.dead_ids[base_row + t] == 0)) ? 1.0f : 0.0f;
    }
    __syncthreads();
    const uint64_t kbits = key_bits<NKT>(kv, T);
    const DropCtx dc = drop_ctx(d.drop);
    float* Qw = Qs + wave * 16 * g.PA;
    const int wpb = blockDim.x >> 6, nwaves = gridDim.y * wpb;
    for (int qi = blockIdx.y * wpb + wave; qi < g.nkt; qi += nwaves) {
        const int qt = g.nkt - 1 - qi;                  // heaviest tiles first
        const int q0 = 16 * qt, q = q0 + li;
        const bool is_dead = dead[q] != 0.0f;
        if (__all(is_dead ? 1 : 0) && d.attn_weights == nullptr) {
            // the whole tile is padding: A = 0 -> out = residual (known dead downstream, sasrec.py:83)
            for (int rr = 0; rr < 16; ++rr) {
                const int qq = q0 + rr;
                if (qq < T && lane < d.d) {
                    const size_t row = (size_t)(base_row + qq);
                    d.out[row * d.ldo + hoff + lane] = d.residual[row * d.ldr + hoff + lane];
                }
            }
            continue;
        }
        stage_rows(Qw, g.PA, d.Q, d.ld, base_row + q0, hoff, T - q0, d.d, 16, 0, 1);
        f32x4 st[NKT];
        float m2, inv;
        bool uniform;
        score_rows<NKT>(g, Ks, Qw, kbits, qt, T, is_dead, q < T, st, m2, inv, uniform);
        const float qvq = qv[q];
        const bool any_uni = __any(uniform ? 1 : 0) != 0;
        const uint32_t ridx = attn_row_idx(d, head, n, q);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * lg + r;
                float p = st[kt][r] * qvq;                                          // modules.py:248-253
                if (p != 0.0f) p = drop_apply(dc, ridx + (uint32_t)key, p);          // modules.py:256-257
                st[kt][r] = p;
                if (d.attn_weights && q < T && key < T)
                    d.attn_weights[((size_t)blockIdx.x * T + q) * T + key] = p;      // modules.py:259
            }
        }
        const int kt_end = any_uni ? g.nkt : qt + 1;
#pragma unroll 1
        for (int jt = 0; jt < g.ndt; ++jt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* vp = Vs + (4 * lg) * g.PB + 16 * jt + li;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt < kt_end) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = mfma16(st[kt][r], vp[(16 * kt + r) * g.PB], acc);   // modules.py:262
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + 4 * lg + r, c = 16 * jt + li;
                if (qq < T && c < d.d) {
                    const size_t row = (size_t)(base_row + qq);
                    d.out[row * d.ldo + hoff + c] = acc[r] + d.residual[row * d.ldr + hoff + c];          // modules.py:265-269
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, query-owner pass: dQ + per-row statistics (max*log2e, 1/sum, delta, flag) for the key pass.
// flag: 0 = normal row, 1 = uniform row with a non-zero incoming gradient, 2 = contributes nothing.
// ------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(A_THREADS) void k_attn_bwd_q(cr_attn_bwd_desc bd, AttnGeom g) {
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                                   // [T16][PA]  A-pattern (scores) and B-pattern (dQ) reads
    float* Vs = Ks + g.T16 * g.PA + A_TAIL;             // [T16][PA]  A-pattern reads (dP^T)
    float* Qs = Vs + g.T16 * g.PA;                      // [A_WAVES][2][16][PA]: Q tile, dOut tile
    float* kv = Qs + (blockDim.x >> 6) * 32 * g.PA;
    float* qv = kv + g.T16;
    float* dead = qv + g.T16;
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    stage_rows(Ks, g.PA, d.K, d.ld, base_row, hoff, T, d.d, g.T16, wave, (int)(blockDim.x >> 6));
    stage_rows(Vs, g.PA, d.V, d.ld, base_row, hoff, T, d.d, g.T16, wave, (int)(blockDim.x >> 6));
    for (int t = threadIdx.x; t < g.T16; t += blockDim.x) {
        kv[t] = (t < T) ? d.k_valid[base_row + t] : 0.0f;
        qv[t] = (t < T) ? d.q_valid[base_row + t] : 0.0f;
        dead[t] = (t >= T || (d.dead_ids && d.dead_ids[base_row + t] == 0)) ? 1.0f : 0.0f;
    }
    __syncthreads();
    const uint64_t kbits = key_bits<NKT>(kv, T);
    const DropCtx dc = drop_ctx(d.drop);
    float* Qw = Qs + wave * 32 * g.PA;
    float* Ow = Qw + 16 * g.PA;
    const int wpb = blockDim.x >> 6, nwaves = gridDim.y * wpb;
    for (int qi = blockIdx.y * wpb + wave; qi < g.nkt; qi += nwaves) {
        const int qt = g.nkt - 1 - qi;
        const int q0 = 16 * qt, q = q0 + li;
        const bool is_dead = dead[q] != 0.0f;
        if (__all(is_dead ? 1 : 0)) {                  // whole tile dead: dQ = 0, flag 2
            for (int rr = 0; rr < 16; ++rr) {
                const int qq = q0 + rr;
                if (qq < T && lane < d.d) bd.dQ[(size_t)(base_row + qq) * bd.ldg + hoff + lane] = 0.0f;
            }
            if (lg == 0 && q < T) {
                float* sp = bd.stats + ((size_t)blockIdx.x * T + q) * 4;
                sp[0] = 0.0f; sp[1] = 0.0f; sp[2] = 0.0f; sp[3] = 2.0f;
            }
            continue;
        }
        stage_rows(Qw, g.PA, d.Q, d.ld, base_row + q0, hoff, T - q0, d.d, 16, 0, 1);
        stage_rows(Ow, g.PA, bd.dout, bd.lddo, base_row + q0, hoff, T - q0, d.d, 16, 0, 1);
        // does this query's incoming gradient row vanish? (lanes li, all lg, own 1/4 of the columns each)
        float nz = 0.0f;
        for (int s = 0; s < g.nds; ++s)
            if (Ow[li * g.PA + 4 * s + lg] != 0.0f) nz = 1.0f;
        nz = grp_max(nz);
        f32x4 st[NKT];
        float m2, inv;
        bool uniform;
        score_rows<NKT>(g, Ks, Qw, kbits, qt, T, is_dead, q < T, st, m2, inv, uniform);
        const float qvq = qv[q];
        const bool live = !uniform && !is_dead && (q < T);
        const uint32_t ridx = attn_row_idx(d, head, n, q);
        // dP^T[key][q] = V dO^T, then softmax backward (delta = sum_k dPsm * Psm)
        f32x4 dps[NKT];
        float delta = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (kt <= qt) {
                acc = mma_rows_t(Vs + 16 * kt * g.PA, g.PA, Ow, g.PA, g.nds);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * lg + r;
                    const float psm = st[kt][r];
                    float w = 0.0f;
                    if (live && psm != 0.0f) w = drop_apply(dc, ridx + (uint32_t)key, qvq);
                    const float dpsm = acc[r] * w;
                    delta += dpsm * psm;
                    acc[r] = dpsm;
                }
            }
            dps[kt] = acc;
        }
        delta = grp_sum(delta);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st[kt][r] = live ? st[kt][r] * (dps[kt][r] - delta) * g.isd : 0.0f;   // dS / sqrt(d)
        }
        // dQ[q][dim] = sum_key dS[q][key] K[key][dim]
#pragma unroll 1
        for (int jt = 0; jt < g.ndt; ++jt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + (4 * lg) * g.PA + 16 * jt + li;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt <= qt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = mfma16(st[kt][r], kp[(16 * kt + r) * g.PA], acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + 4 * lg + r, c = 16 * jt + li;
                if (qq < T && c < d.d) bd.dQ[(size_t)(base_row + qq) * bd.ldg + hoff + c] = acc[r];
            }
        }
        if (lg == 0 && q < T) {
            float* sp = bd.stats + ((size_t)blockIdx.x * T + q) * 4;
            float flag = 0.0f;
            if (is_dead || (uniform && nz == 0.0f)) flag = 2.0f;
            else if (uniform) flag = 1.0f;
            sp[0] = m2; sp[1] = inv; sp[2] = delta; sp[3] = flag;
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, key-owner pass: dK and dV of the wave's 16 keys, summed over queries in registers.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(A_THREADS) void k_attn_bwd_kv(cr_attn_bwd_desc bd, AttnGeom g) {
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                                   // [T16][PA]  A- and B-pattern reads
    float* Os = Qs + g.T16 * g.PA + A_TAIL;             // [T16][PA]  dOut
    float* KVs = Os + g.T16 * g.PA + A_TAIL;            // [A_WAVES][2][16][PA]: the wave's K tile, V tile
    float* smx = KVs + (blockDim.x >> 6) * 32 * g.PA;     // [T16] each
    float* sinv = smx + g.T16;
    float* sdel = sinv + g.T16;
    float* sflag = sdel + g.T16;
    float* qv = sflag + g.T16;
    float* tile_uni = qv + g.T16;                       // [nkt]: tile holds a flag==1 row
    float* tile_live = tile_uni + g.nkt;                // [nkt]: tile holds a row with flag != 2
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    stage_rows(Qs, g.PA, d.Q, d.ld, base_row, hoff, T, d.d, g.T16, wave, (int)(blockDim.x >> 6));
    stage_rows(Os, g.PA, bd.dout, bd.lddo, base_row, hoff, T, d.d, g.T16, wave, (int)(blockDim.x >> 6));
    for (int t = threadIdx.x; t < g.T16; t += blockDim.x) {
        const float* sp = bd.stats + ((size_t)blockIdx.x * T + (t < T ? t : 0)) * 4;
        smx[t] = (t < T) ? sp[0] : 0.0f;
        sinv[t] = (t < T) ? sp[1] : 0.0f;
        sdel[t] = (t < T) ? sp[2] : 0.0f;
        sflag[t] = (t < T) ? sp[3] : 2.0f;
        qv[t] = (t < T) ? d.q_valid[base_row + t] : 0.0f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < g.nkt; t += blockDim.x) {
        float u = 0.0f, lv = 0.0f;
        for (int i = 0; i < 16; ++i) {
            if (sflag[16 * t + i] == 1.0f) u = 1.0f;
            if (sflag[16 * t + i] != 2.0f) lv = 1.0f;
        }
        tile_uni[t] = u;
        tile_live[t] = lv;
    }
    __syncthreads();
    const DropCtx dc = drop_ctx(d.drop);
    float* Kw = KVs + wave * 32 * g.PA;
    float* Vw = Kw + 16 * g.PA;
    const int wpb = blockDim.x >> 6, nwaves = gridDim.y * wpb;
    for (int kt = blockIdx.y * wpb + wave; kt < g.nkt; kt += nwaves) {
        const int key = 16 * kt + li;
        stage_rows(Kw, g.PA, d.K, d.ld, base_row + 16 * kt, hoff, T - 16 * kt, d.d, 16, 0, 1);
        stage_rows(Vw, g.PA, d.V, d.ld, base_row + 16 * kt, hoff, T - 16 * kt, d.d, 16, 0, 1);
        const bool kvk = (key < T) && (d.k_valid[base_row + (key < T ? key : 0)] != 0.0f);
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            dk[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int qt = 0; qt < g.nkt; ++qt) {
            if (tile_live[qt] == 0.0f) continue;                          // nothing flows through dead query tiles
            if (qt < kt && tile_uni[qt] == 0.0f) continue;                // causal skip (uniform rows see all keys)
            const f32x4 s_acc = mma_rows_t(Qs + 16 * qt * g.PA, g.PA, Kw, g.PA, g.nds);   // S[q][key]
            const f32x4 p_acc = mma_rows_t(Os + 16 * qt * g.PA, g.PA, Vw, g.PA, g.nds);   // dP[q][key]
            float pa[4], pd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * qt + 4 * lg + r;
                const float flag = sflag[q];
                float psm = 0.0f;
                if (flag == 0.0f) {
                    const bool valid = (key <= q) && kvk;
                    psm = valid ? exp2f(s_acc[r] * g.isd_log2e - smx[q]) * sinv[q] : 0.0f;
                } else if (flag == 1.0f) {
                    psm = (key < T) ? g.invT : 0.0f;
                }
                float w = 0.0f;
                if (psm != 0.0f) w = drop_apply(dc, attn_row_idx(d, head, n, q) + (uint32_t)key, qv[q]);
                pa[r] = psm * w;                                                          // A after mask+dropout
                pd[r] = (flag == 0.0f) ? psm * (p_acc[r] * w - sdel[q]) * g.isd : 0.0f;   // dS / sqrt(d)
            }
            const float* op = Os + (16 * qt + 4 * lg) * g.PA + li;
            const float* qp = Qs + (16 * qt + 4 * lg) * g.PA + li;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                if (jt < g.ndt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dv[jt] = mfma16(pa[r], op[r * g.PA + 16 * jt], dv[jt]);
                        dk[jt] = mfma16(pd[r], qp[r * g.PA + 16 * jt], dk[jt]);
                    }
                }
            }
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            if (jt < g.ndt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = 16 * kt + 4 * lg + r, c = 16 * jt + li;
                    if (kk < T && c < d.d) {
                        bd.dK[(size_t)(base_row + kk) * bd.ldg + hoff + c] = dk[jt][r];
                        bd.dV[(size_t)(base_row + kk) * bd.ldg + hoff + c] = dv[jt][r];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int attn_geom(const cr_attn_desc* d, AttnGeom* g, const char* who) {
    CR_REQUIRE(d->Q && d->K && d->V && d->k_valid && d->q_valid, "%s: NULL pointer", who);
    CR_REQUIRE(d->B > 0 && d->T > 0 && d->H > 0 && d->d > 0, "%s: bad shape B=%d T=%d H=%d d=%d", who, d->B, d->T, d->H, d->d);
    if (d->d > 64) return cr_set_error(CR_ERR_UNSUPPORTED, "%s: head dim %d > 64", who, d->d);
    if (d->T > 256) return cr_set_error(CR_ERR_UNSUPPORTED, "%s: T=%d > 256 (LDS-resident K/V design)", who, d->T);
    CR_REQUIRE(d->ld >= d->H * d->d, "%s: ld too small", who);
    CR_REQUIRE(d->batch_global >= d->B, "%s: batch_global < B", who);
    g->T16 = (d->T + 15) / 16 * 16;
    g->nkt = g->T16 / 16;
    g->dp = (d->d + 3) / 4 * 4;
    g->nds = g->dp / 4;
    g->ndt = (d->d + 15) / 16;
    g->PA = g->dp + 2;                                  // % 4 == 2: conflict-free A-pattern reads
    g->PB = g->dp + ((g->dp % 8 == 4) ? 0 : 4);         // % 8 == 4: conflict-free B-pattern reads
    g->isd = (float)(1.0 / sqrt((double)d->d));
    g->isd_log2e = (float)(1.4426950408889634 / sqrt((double)d->d));
    g->invT = 1.0f / (float)d->T;
    return CR_OK;
}

static int attn_nsplit(const cr_attn_desc* d, const AttnGeom& g, int waves) {
    int want = (256 + d->B * d->H - 1) / (d->B * d->H);
    int maxs = (g.nkt + waves - 1) / waves;
    if (want > maxs) want = maxs;
    return want < 1 ? 1 : want;
}

// LDS bytes of the three kernels for a workgroup of `w` waves
static size_t lds_fwd(const AttnGeom& g, int w) {
    return sizeof(float) * ((size_t)g.T16 * (g.PA + g.PB) + A_TAIL + (size_t)w * 16 * g.PA + 3 * g.T16);
}
static size_t lds_bwd_q(const AttnGeom& g, int w) {
    return sizeof(float) * ((size_t)g.T16 * (2 * g.PA) + A_TAIL + (size_t)w * 32 * g.PA + 3 * g.T16);
}
static size_t lds_bwd_kv(const AttnGeom& g, int w) {
    return sizeof(float) * ((size_t)g.T16 * (2 * g.PA) + 2 * A_TAIL + (size_t)w * 32 * g.PA + 5 * g.T16 + 2 * g.nkt);
}
// largest wave count (8, 4, 2, 1) whose LDS footprint fits the 160 KiB of a CU; 0 if none does
template <class F>
static int pick_waves(const AttnGeom& g, F lds) {
    for (int w = A_WAVES; w >= 1; w >>= 1)
        if (lds(g, w) <= 160 * 1024) return w;
    return 0;
}

static int set_lds_attr(const void* fn) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    return CR_OK;
}

template <int NKT>
static int launch_fwd(const cr_attn_desc* d, const AttnGeom& g, int waves, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        int rc = set_lds_attr(reinterpret_cast<const void*>(&k_attn_fwd<NKT>));
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_attn_fwd<NKT>, dim3(d->B * d->H, attn_nsplit(d, g, waves)), dim3(64 * waves), lds_fwd(g, waves), s, *d, g);
    return cr_check_launch("cr_attn_fwd");
}

template <int NKT>
static int launch_bwd_q(const cr_attn_bwd_desc* bd, const AttnGeom& g, int waves, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        int rc = set_lds_attr(reinterpret_cast<const void*>(&k_attn_bwd_q<NKT>));
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_attn_bwd_q<NKT>, dim3(bd->f.B * bd->f.H, attn_nsplit(&bd->f, g, waves)), dim3(64 * waves), lds_bwd_q(g, waves), s, *bd, g);
    return cr_check_launch("cr_attn_bwd(q)");
}

extern "C" int cr_attn_fwd(const cr_attn_desc* d, void* stream) {
    CR_REQUIRE(d != nullptr, "cr_attn_fwd: NULL desc");
    AttnGeom g;
    int rc = attn_geom(d, &g, "cr_attn_fwd");
    if (rc) return rc;
    CR_REQUIRE(d->out && d->residual, "cr_attn_fwd: NULL out/residual");
    const int waves = pick_waves(g, lds_fwd);
    if (!waves) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_attn_fwd: T=%d d=%d needs %zu B of LDS", d->T, d->d, lds_fwd(g, 1));
    hipStream_t s = cr_stream(stream);
    if (g.nkt <= 4) return launch_fwd<4>(d, g, waves, s);
    if (g.nkt <= 8) return launch_fwd<8>(d, g, waves, s);
    if (g.nkt <= 13) return launch_fwd<13>(d, g, waves, s);
    return launch_fwd<16>(d, g, waves, s);
}

extern "C" int cr_attn_bwd(const cr_attn_bwd_desc* bd, void* stream) {
    CR_REQUIRE(bd != nullptr, "cr_attn_bwd: NULL desc");
    const cr_attn_desc* d = &bd->f;
    AttnGeom g;
    int rc = attn_geom(d, &g, "cr_attn_bwd");
    if (rc) return rc;
    CR_REQUIRE(bd->dout && bd->dQ && bd->dK && bd->dV && bd->stats, "cr_attn_bwd: NULL pointer");
    const int wq = pick_waves(g, lds_bwd_q), wkv = pick_waves(g, lds_bwd_kv);
    if (!wq || !wkv)
        return cr_set_error(CR_ERR_UNSUPPORTED, "cr_attn_bwd: T=%d d=%d needs %zu B of LDS", d->T, d->d, lds_bwd_kv(g, 1));
    hipStream_t s = cr_stream(stream);
    if (g.nkt <= 4) rc = launch_bwd_q<4>(bd, g, wq, s);
    else if (g.nkt <= 8) rc = launch_bwd_q<8>(bd, g, wq, s);
    else if (g.nkt <= 13) rc = launch_bwd_q<13>(bd, g, wq, s);
    else rc = launch_bwd_q<16>(bd, g, wq, s);
    if (rc) return rc;
    static bool attr_set = false;
    if (!attr_set) {
        rc = set_lds_attr(reinterpret_cast<const void*>(&k_attn_bwd_kv));
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_attn_bwd_kv, dim3(d->B * d->H, attn_nsplit(d, g, wkv)), dim3(64 * wkv), lds_bwd_kv(g, wkv), s, *bd, g);
    return cr_check_launch("cr_attn_bwd(kv)");
}
