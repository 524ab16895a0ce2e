// Causal multi-head self-attention core of modules.py:208-269 (scores, key / causal / query masks,
// softmax, dropout, weighted sum, residual) -- forward and backward, scores never leave the chip.
//
// Work decomposition: grid = (H*B, nsplit); a workgroup (4 waves) stages the K and V rows of ONE
// (head, sample) in LDS; each wave owns 16-query tiles and keeps the whole score row block
// (<= 16 key tiles x 4 registers) in VGPRs, so softmax is a register + 2-shuffle reduction.
//
// MFMA trick (v_mfma_f32_16x16x4_f32): scores are computed TRANSPOSED, St[key][query] = K Q^T, so a
// lane holds, for ITS query (lane & 15), keys {16*kt + 4*(lane>>4) + r}.  That accumulator layout is
// exactly the A-operand layout of the next product (P V, dS K) when the k-dimension of MFMA step r
// is taken as key 4*lg + r -- no LDS round trip, no cross-lane movement between the two GEMMs.
// The key-owner backward kernel uses the mirrored form (S[query][key]) for dK / dV.
//
// Exactness notes (tests/test_attn_gpu.py): masked entries are -2^32+1 in the reference, so
//  * a row with >= 1 valid key: masked probabilities are exactly 0;
//  * a row with NO valid key ("uniform"): probability 1/T on ALL T keys, future ones included
//    (modules.py:227-244) -- handled explicitly, contributes to out and to dV, no score gradient.
#include <math.h>

#include "cr_common.hpp"

#define A_MAX_DS 16          // head dim <= 64  (k-steps of 4)
#define A_MAX_DT 4           // head dim <= 64  (16-column output tiles)

struct AttnGeom {
    int T16, nkt;            // padded T, number of 16-tiles
    int dp;                  // head dim rounded up to 4
    int nds, ndt;            // dp/4, ceil(d/16)
    int PA, PB;              // LDS pitches: A-pattern (dp+2) and B-pattern (dp rounded so pitch%8==4)
    float isd;               // 1/sqrt(d)   (modules.py:219)
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

// dropout element index of attention_weights[(j*Bglobal + n), q, k]
__device__ __forceinline__ uint32_t attn_idx(const cr_attn_desc& d, int head, int n, int q, int k) {
    const uint32_t ng = d.drop.row_offset / (uint32_t)d.T + (uint32_t)n;
    return (((uint32_t)head * (uint32_t)d.batch_global + ng) * (uint32_t)d.T + (uint32_t)q) * (uint32_t)d.T + (uint32_t)k;
}

// ------------------------------------------------------------------------------------------
// Shared by forward and backward-Q: score row block of one 16-query tile.
// On return st[kt][r] holds the SOFTMAX probability (before query mask / dropout) of
// key 16*kt + 4*lg + r for query q0 + li; mx / inv are the row max and 1/sum; uniform / dead flags.
// ------------------------------------------------------------------------------------------
template <int NKT>
__device__ __forceinline__ void score_rows(const AttnGeom& g, const float* __restrict__ Ks, const float* __restrict__ kv,
                                           const float (&qf)[A_MAX_DS], int qt, int q, int T, bool is_dead,
                                           f32x4 (&st)[NKT], float& mx, float& inv, bool& uniform) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kt <= qt) {
#pragma unroll
            for (int s = 0; s < A_MAX_DS; ++s)
                if (s < g.nds) acc = mfma16(Ks[(16 * kt + li) * g.PA + 4 * s + lg], qf[s], acc);
        }
        st[kt] = acc;
    }
    mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * lg + r;
            const bool valid = (kt <= qt) && (key <= q) && (key < T) && !is_dead && (kv[key < g.T16 ? key : 0] != 0.0f);
            const float sv = valid ? st[kt][r] * g.isd : -INFINITY;
            st[kt][r] = sv;
            mx = fmaxf(mx, sv);
        }
    }
    mx = grp_max(mx);
    uniform = (mx == -INFINITY) && !is_dead && (q < T);
    float sum = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sv = st[kt][r];
            const float p = (sv == -INFINITY) ? 0.0f : expf(sv - mx);
            st[kt][r] = p;
            sum += p;
        }
    }
    sum = grp_sum(sum);
    inv = sum > 0.0f ? 1.0f / sum : 0.0f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * lg + r;
            st[kt][r] = uniform ? ((key < T) ? g.invT : 0.0f) : st[kt][r] * inv;
        }
    }
}

// stage rows [0,T) x [hoff, hoff+d) of a [M, ld] matrix into LDS with pitch P; everything outside
// (rows >= T, columns >= d up to the pitch) is zero so padded k-steps / column tiles contribute 0.
__device__ __forceinline__ void stage_rows(float* dst, int P, const float* src, int ld, int base_row, int hoff,
                                           int T, int d, int T16, int dp) {
    (void)dp;
    for (int e = threadIdx.x; e < T16 * P; e += 256) {
        const int t = e / P, c = e % P;
        dst[e] = (t < T && c < d) ? src[(size_t)(base_row + t) * ld + hoff + c] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(256) void k_attn_fwd(cr_attn_desc d, AttnGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                       // [T16][PA]  A-pattern reads
    float* Vs = Ks + g.T16 * g.PA;          // [T16][PB]  B-pattern reads
    float* kv = Vs + g.T16 * g.PB;          // [T16]
    float* qv = kv + g.T16;                 // [T16]
    float* dead = qv + g.T16;               // [T16]
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    stage_rows(Ks, g.PA, d.K, d.ld, base_row, hoff, T, d.d, g.T16, g.dp);
    stage_rows(Vs, g.PB, d.V, d.ld, base_row, hoff, T, d.d, g.T16, g.dp);
    for (int t = threadIdx.x; t < g.T16; t += 256) {
        kv[t] = (t < T) ? d.k_valid[base_row + t] : 0.0f;
        qv[t] = (t < T) ? d.q_valid[base_row + t] : 0.0f;
        dead[t] = (t >= T || (d.dead_ids && d.dead_ids[base_row + t] == 0)) ? 1.0f : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const DropCtx dc = drop_ctx(d.drop);
    const int nwaves = gridDim.y * 4;
    for (int qi = blockIdx.y * 4 + wave; qi < g.nkt; qi += nwaves) {
        const int qt = g.nkt - 1 - qi;       // heaviest tiles first
        const int q0 = 16 * qt, q = q0 + li;
        float qf[A_MAX_DS];
#pragma unroll
        for (int s = 0; s < A_MAX_DS; ++s) {
            const int c = 4 * s + lg;
            qf[s] = (s < g.nds && q < T && c < d.d) ? d.Q[(size_t)(base_row + q) * d.ld + hoff + c] : 0.0f;
        }
        const bool is_dead = dead[q] != 0.0f;
        f32x4 st[NKT];
        float mx, inv;
        bool uniform;
        score_rows<NKT>(g, Ks, kv, qf, qt, q, T, is_dead, st, mx, inv, uniform);
        const float qvq = qv[q];
        const bool any_uni = __any(uniform ? 1 : 0) != 0;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * lg + r;
                float p = st[kt][r] * qvq;                                   // modules.py:248-253
                if (p != 0.0f) p = drop_apply(dc, attn_idx(d, head, n, q, key), p);   // modules.py:256-257
                st[kt][r] = p;
                if (d.attn_weights && q < T && key < T && kt < g.nkt)
                    d.attn_weights[((size_t)blockIdx.x * T + q) * T + key] = p;        // modules.py:259
            }
        }
        const int kt_end = any_uni ? g.nkt : qt + 1;
#pragma unroll
        for (int jt = 0; jt < A_MAX_DT; ++jt) {
            if (jt >= g.ndt) break;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt < kt_end) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc = mfma16(st[kt][r], Vs[(16 * kt + 4 * lg + r) * g.PB + 16 * jt + li], acc);   // modules.py:262
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
// backward, query-owner pass: dQ + per-row statistics (max, 1/sum, delta, flag) for the key pass.
// flag: 0 = normal row, 1 = uniform row with a non-zero incoming gradient, 2 = contributes nothing.
// ------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(256) void k_attn_bwd_q(cr_attn_bwd_desc bd, AttnGeom g) {
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                       // [T16][PA]  A-pattern (scores) and B-pattern (dQ) reads
    float* Vs = Ks + g.T16 * g.PA;          // [T16][PA]  A-pattern reads (dP^T)
    float* kv = Vs + g.T16 * g.PA;
    float* qv = kv + g.T16;
    float* dead = qv + g.T16;
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    stage_rows(Ks, g.PA, d.K, d.ld, base_row, hoff, T, d.d, g.T16, g.dp);
    stage_rows(Vs, g.PA, d.V, d.ld, base_row, hoff, T, d.d, g.T16, g.dp);
    for (int t = threadIdx.x; t < g.T16; t += 256) {
        kv[t] = (t < T) ? d.k_valid[base_row + t] : 0.0f;
        qv[t] = (t < T) ? d.q_valid[base_row + t] : 0.0f;
        dead[t] = (t >= T || (d.dead_ids && d.dead_ids[base_row + t] == 0)) ? 1.0f : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const DropCtx dc = drop_ctx(d.drop);
    const int nwaves = gridDim.y * 4;
    for (int qi = blockIdx.y * 4 + wave; qi < g.nkt; qi += nwaves) {
        const int qt = g.nkt - 1 - qi;
        const int q0 = 16 * qt, q = q0 + li;
        float qf[A_MAX_DS], dof[A_MAX_DS];
        float nz = 0.0f;
#pragma unroll
        for (int s = 0; s < A_MAX_DS; ++s) {
            const int c = 4 * s + lg;
            const bool ok = (s < g.nds && q < T && c < d.d);
            qf[s] = ok ? d.Q[(size_t)(base_row + q) * d.ld + hoff + c] : 0.0f;
            dof[s] = ok ? bd.dout[(size_t)(base_row + q) * bd.lddo + hoff + c] : 0.0f;
            if (dof[s] != 0.0f) nz = 1.0f;
        }
        nz = grp_max(nz);
        const bool is_dead = dead[q] != 0.0f;
        f32x4 st[NKT];
        float mx, inv;
        bool uniform;
        score_rows<NKT>(g, Ks, kv, qf, qt, q, T, is_dead, st, mx, inv, uniform);
        const float qvq = qv[q];
        const bool live = !uniform && !is_dead && (q < T);
        // dP^T[key][q] = V dO^T, then softmax backward (delta = sum_k dPsm * Psm)
        f32x4 dps[NKT];
        float delta = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (kt <= qt) {
#pragma unroll
                for (int s = 0; s < A_MAX_DS; ++s)
                    if (s < g.nds) acc = mfma16(Vs[(16 * kt + li) * g.PA + 4 * s + lg], dof[s], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * lg + r;
                const float psm = st[kt][r];
                float w = 0.0f;
                if (live && psm != 0.0f) w = drop_apply(dc, attn_idx(d, head, n, q, key), qvq);
                const float dpsm = acc[r] * w;
                delta += dpsm * psm;
                dps[kt][r] = dpsm;
            }
        }
        delta = grp_sum(delta);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st[kt][r] = live ? st[kt][r] * (dps[kt][r] - delta) * g.isd : 0.0f;   // dS / sqrt(d)
        }
        // dQ[q][dim] = sum_key dS[q][key] K[key][dim]
#pragma unroll
        for (int jt = 0; jt < A_MAX_DT; ++jt) {
            if (jt >= g.ndt) break;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt <= qt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc = mfma16(st[kt][r], Ks[(16 * kt + 4 * lg + r) * g.PA + 16 * jt + li], acc);
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
            sp[0] = mx; sp[1] = inv; sp[2] = delta; sp[3] = flag;
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, key-owner pass: dK and dV of the wave's 16 keys, summed over queries in registers.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_attn_bwd_kv(cr_attn_bwd_desc bd, AttnGeom g) {
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                        // [T16][PA]
    float* Os = Qs + g.T16 * g.PA;           // [T16][PA]  dOut
    float* smx = Os + g.T16 * g.PA;          // [T16] each
    float* sinv = smx + g.T16;
    float* sdel = sinv + g.T16;
    float* sflag = sdel + g.T16;
    float* qv = sflag + g.T16;
    float* tile_uni = qv + g.T16;            // [nkt]: tile holds a flag==1 row
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    stage_rows(Qs, g.PA, d.Q, d.ld, base_row, hoff, T, d.d, g.T16, g.dp);
    stage_rows(Os, g.PA, bd.dout, bd.lddo, base_row, hoff, T, d.d, g.T16, g.dp);
    for (int t = threadIdx.x; t < g.T16; t += 256) {
        const float* sp = bd.stats + ((size_t)blockIdx.x * T + t) * 4;
        smx[t] = (t < T) ? sp[0] : 0.0f;
        sinv[t] = (t < T) ? sp[1] : 0.0f;
        sdel[t] = (t < T) ? sp[2] : 0.0f;
        sflag[t] = (t < T) ? sp[3] : 2.0f;
        qv[t] = (t < T) ? d.q_valid[base_row + t] : 0.0f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < g.nkt; t += 256) {
        float u = 0.0f;
        for (int i = 0; i < 16; ++i)
            if (sflag[16 * t + i] == 1.0f) u = 1.0f;
        tile_uni[t] = u;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const DropCtx dc = drop_ctx(d.drop);
    const int nwaves = gridDim.y * 4;
    for (int kt = blockIdx.y * 4 + wave; kt < g.nkt; kt += nwaves) {
        const int key = 16 * kt + li;
        float kf[A_MAX_DS], vf[A_MAX_DS];
#pragma unroll
        for (int s = 0; s < A_MAX_DS; ++s) {
            const int c = 4 * s + lg;
            const bool ok = (s < g.nds && key < T && c < d.d);
            kf[s] = ok ? d.K[(size_t)(base_row + key) * d.ld + hoff + c] : 0.0f;
            vf[s] = ok ? d.V[(size_t)(base_row + key) * d.ld + hoff + c] : 0.0f;
        }
        const bool kvk = (key < T) && (d.k_valid[base_row + (key < T ? key : 0)] != 0.0f);
        f32x4 dk[A_MAX_DT], dv[A_MAX_DT];
#pragma unroll
        for (int jt = 0; jt < A_MAX_DT; ++jt) {
            dk[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int qt = 0; qt < g.nkt; ++qt) {
            if (qt < kt && tile_uni[qt] == 0.0f) continue;      // causal skip (uniform rows see all keys)
            f32x4 s_acc = (f32x4){0.f, 0.f, 0.f, 0.f}, p_acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < A_MAX_DS; ++s) {
                if (s < g.nds) {
                    s_acc = mfma16(Qs[(16 * qt + li) * g.PA + 4 * s + lg], kf[s], s_acc);   // S[q][key]
                    p_acc = mfma16(Os[(16 * qt + li) * g.PA + 4 * s + lg], vf[s], p_acc);   // dP[q][key]
                }
            }
            float pa[4], pd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * qt + 4 * lg + r;
                const float flag = sflag[q];
                float psm = 0.0f;
                if (flag == 0.0f) {
                    const bool valid = (key <= q) && kvk;
                    psm = valid ? expf(s_acc[r] * g.isd - smx[q]) * sinv[q] : 0.0f;
                } else if (flag == 1.0f) {
                    psm = (key < T) ? g.invT : 0.0f;
                }
                float w = 0.0f;
                if (psm != 0.0f) w = drop_apply(dc, attn_idx(d, head, n, q, key), qv[q]);
                pa[r] = psm * w;                                                          // A after mask+dropout
                pd[r] = (flag == 0.0f) ? psm * (p_acc[r] * w - sdel[q]) * g.isd : 0.0f;   // dS / sqrt(d)
            }
#pragma unroll
            for (int jt = 0; jt < A_MAX_DT; ++jt) {
                if (jt >= g.ndt) break;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dv[jt] = mfma16(pa[r], Os[(16 * qt + 4 * lg + r) * g.PA + 16 * jt + li], dv[jt]);
                    dk[jt] = mfma16(pd[r], Qs[(16 * qt + 4 * lg + r) * g.PA + 16 * jt + li], dk[jt]);
                }
            }
        }
#pragma unroll
        for (int jt = 0; jt < A_MAX_DT; ++jt) {
            if (jt >= g.ndt) break;
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
    g->PA = 16 * g->ndt + 2;                       // >= dp, covers the 16-column B-pattern reads, % 4 == 2
    g->PB = 16 * g->ndt + ((16 * g->ndt) % 8 == 4 ? 0 : 4);   // % 8 == 4
    g->isd = (float)(1.0 / sqrt((double)d->d));
    g->invT = 1.0f / (float)d->T;
    return CR_OK;
}

static int attn_nsplit(const cr_attn_desc* d, const AttnGeom& g) {
    int want = (256 + d->B * d->H - 1) / (d->B * d->H);
    int maxs = (g.nkt + 3) / 4;
    if (want > maxs) want = maxs;
    return want < 1 ? 1 : want;
}

template <int NKT>
static int launch_fwd(const cr_attn_desc* d, const AttnGeom& g, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd<NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_attn_fwd<NKT>, dim3(d->B * d->H, attn_nsplit(d, g)), dim3(256), lds, s, *d, g);
    return cr_check_launch("cr_attn_fwd");
}

template <int NKT>
static int launch_bwd_q(const cr_attn_bwd_desc* bd, const AttnGeom& g, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd_q<NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_attn_bwd_q<NKT>, dim3(bd->f.B * bd->f.H, attn_nsplit(&bd->f, g)), dim3(256), lds, s, *bd, g);
    return cr_check_launch("cr_attn_bwd(q)");
}

extern "C" int cr_attn_fwd(const cr_attn_desc* d, void* stream) {
    CR_REQUIRE(d != nullptr, "cr_attn_fwd: NULL desc");
    AttnGeom g;
    int rc = attn_geom(d, &g, "cr_attn_fwd");
    if (rc) return rc;
    CR_REQUIRE(d->out && d->residual, "cr_attn_fwd: NULL out/residual");
    const size_t lds = sizeof(float) * ((size_t)g.T16 * (g.PA + g.PB) + 3 * g.T16);
    if (lds > 160 * 1024) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_attn_fwd: T=%d d=%d needs %zu B of LDS", d->T, d->d, lds);
    hipStream_t s = cr_stream(stream);
    if (g.nkt <= 4) return launch_fwd<4>(d, g, lds, s);
    if (g.nkt <= 8) return launch_fwd<8>(d, g, lds, s);
    if (g.nkt <= 13) return launch_fwd<13>(d, g, lds, s);
    return launch_fwd<16>(d, g, lds, s);
}

extern "C" int cr_attn_bwd(const cr_attn_bwd_desc* bd, void* stream) {
    CR_REQUIRE(bd != nullptr, "cr_attn_bwd: NULL desc");
    const cr_attn_desc* d = &bd->f;
    AttnGeom g;
    int rc = attn_geom(d, &g, "cr_attn_bwd");
    if (rc) return rc;
    CR_REQUIRE(bd->dout && bd->dQ && bd->dK && bd->dV && bd->stats, "cr_attn_bwd: NULL pointer");
    const size_t lds_q = sizeof(float) * ((size_t)g.T16 * (2 * g.PA) + 3 * g.T16);
    const size_t lds_kv = sizeof(float) * ((size_t)g.T16 * (2 * g.PA) + 5 * g.T16 + g.nkt);
    if (lds_kv > 160 * 1024) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_attn_bwd: T=%d d=%d needs %zu B of LDS", d->T, d->d, lds_kv);
    hipStream_t s = cr_stream(stream);
    if (g.nkt <= 4) rc = launch_bwd_q<4>(bd, g, lds_q, s);
    else if (g.nkt <= 8) rc = launch_bwd_q<8>(bd, g, lds_q, s);
    else if (g.nkt <= 13) rc = launch_bwd_q<13>(bd, g, lds_q, s);
    else rc = launch_bwd_q<16>(bd, g, lds_q, s);
    if (rc) return rc;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd_kv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_attn_bwd_kv, dim3(d->B * d->H, attn_nsplit(d, g)), dim3(256), lds_kv, s, *bd, g);
    return cr_check_launch("cr_attn_bwd(kv)");
}
