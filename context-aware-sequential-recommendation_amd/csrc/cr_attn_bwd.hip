// Attention backward (gradient of modules.py:208-269); see cr_attn_common.hpp for the design.
//   pass 1 (query-owner): dQ + per-row statistics (max, 1/sum, delta, flag) for pass 2
//   pass 2 (key-owner):   dK, dV of the wave's 16 keys, summed over queries in registers
// Two passes recompute the scores in the orientation each one needs (7 MFMA products instead of the
// textbook 5) but need no LDS transposes, no cross-wave reductions and no atomics: bitwise reproducible.
#include "cr_attn_common.hpp"

// flag: 0 = normal row, 1 = uniform row with a non-zero incoming gradient, 2 = contributes nothing.
template <int NKT, int NDS, int NDT>
__global__ __launch_bounds__(64 * A_MAX_WAVES) void k_attn_bwd_q(cr_attn_bwd_desc bd, AttnGeom g) {
    constexpr int KPA = 4 * NDS + 2;                     // LDS pitches as compile-time constants: operand offsets fold into
    constexpr int KPB = 4 * NDS + (((4 * NDS) % 8 == 4) ? 0 : 4);   // the ds_read immediates (the runtime pitch cost a multiply-add per access)
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nw = blockDim.x >> 6;
    float* Ks = smem;                                   // [T16][PA]  A-pattern (scores) and B-pattern (dQ) reads
    float* Vs = Ks + g.T16 * KPA + A_TAIL;             // [T16][PA]  A-pattern reads (dP^T)
    float* kv = Vs + g.T16 * KPA;
    float* qv = kv + g.T16;
    float* dead = qv + g.T16;
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    AT_TS(0); AT_TS(1);
    const DropCtx dc = drop_ctx(d.drop);                // reads the step counter: requested first, needed late
    TileSched sch = sched_init(nw, wave);
    int qi = sched_rank(sch);                           // rank of the wave's first tile (0 = heaviest)
    float qn[NDS], don[NDS];                            // Q / dOut fragments of the wave's next tile (in flight during the staging)
    if (qi < g.nkt) {
        const int q0n = 16 * (g.nkt - 1 - qi);
        frag_issue<NDS>(d.Q, d.ld, base_row + q0n, hoff, T - q0n, d.d, qn);
        frag_issue<NDS>(bd.dout, bd.lddo, base_row + q0n, hoff, T - q0n, d.d, don);
    }
    const int t0 = threadIdx.x;
    const int t0c = (t0 < T) ? base_row + t0 : base_row;
    const float kv0 = d.k_valid[t0c], qv0 = d.q_valid[t0c];
    const int id0 = d.dead_ids ? d.dead_ids[t0c] : 1;
    stage_pair<NDS>(Ks, KPA, d.K, d.ld, Vs, KPA, d.V, d.ld, base_row, hoff, T, d.d, g.T16);
    if (t0 < g.T16) {
        kv[t0] = (t0 < T && kv0 != 0.0f) ? 0.0f : -INFINITY;      // additive key bias
        qv[t0] = (t0 < T) ? qv0 : 0.0f;
        dead[t0] = (t0 >= T || id0 == 0) ? 1.0f : 0.0f;
    }
    for (int t = t0 + blockDim.x; t < g.T16; t += blockDim.x) {
        kv[t] = (t < T && d.k_valid[base_row + t] != 0.0f) ? 0.0f : -INFINITY;
        qv[t] = (t < T) ? d.q_valid[base_row + t] : 0.0f;
        dead[t] = (t >= T || (d.dead_ids && d.dead_ids[base_row + t] == 0)) ? 1.0f : 0.0f;
    }
    AT_TS(2);
    __syncthreads();
    AT_TS(3);
    const int qi_first = qi;
    const int kt_first = first_valid_tile<NKT>(kv, g.nkt);
    for (; qi < g.nkt; qi = sched_next(sch)) {
        const int qt = g.nkt - 1 - qi;
        const int q0 = 16 * qt, q = q0 + li;
        float qf[NDS], dof[NDS];
        frag_finish<NDS>(qn, T - q0, d.d, qf);
        frag_finish<NDS>(don, T - q0, d.d, dof);
        if (sched_peek(sch) < g.nkt) {                  // prefetch the next tile's fragments behind this tile's work
            const int q0n = 16 * (g.nkt - 1 - sched_peek(sch));
            frag_issue<NDS>(d.Q, d.ld, base_row + q0n, hoff, T - q0n, d.d, qn);
            frag_issue<NDS>(bd.dout, bd.lddo, base_row + q0n, hoff, T - q0n, d.d, don);
        }
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
        float nz = 0.0f;                               // does this query's incoming gradient row vanish?
#pragma unroll
        for (int s = 0; s < NDS; ++s)
            if (dof[s] != 0.0f) nz = 1.0f;
        nz = grp_max(nz);
        f32x4 st[NKT];
        float m2, inv;
        bool uniform;
        score_rows<NKT, NDS>(g, Ks, qf, kv, kt_first, qt, T, is_dead, q < T, st, m2, inv, uniform);
        if (qi == qi_first) AT_TS(4);
        const float qvq = qv[q];
        const bool live = !uniform && !is_dead && (q < T);
        const float wq = live ? qvq : 0.0f;
        const uint32_t ridx = attn_row_idx(d, head, n, q);
        const uint32_t xrow = (ridx + (uint32_t)(4 * lg)) * CR_PHI + dc.key;     // dropout counter of key 4*lg
        // dP^T[key][q] = V dO^T, then softmax backward (delta = sum_k dPsm * Psm)
        f32x4 dps[NKT];
        float delta = 0.0f;
        auto dp_finish = [&](int kt, f32x4 acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float psm = st[kt][r];
                float w = wq;                                                 // query mask (* dropout keep / (1-rate))
                if (dc.on) w *= drop_factor_x(dc, xrow + (uint32_t)(16 * kt + r) * CR_PHI);
                const float dpsm = acc[r] * w;
                delta += dpsm * psm;
                acc[r] = dpsm;
            }
            return acc;
        };
        const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; kt += 2) {                                 // two interleaved MFMA chains at a time
            const bool c0 = kt >= kt_first && kt <= qt;                       // below kt_first every probability is 0 (uniform rows have dS = 0)
            const bool c1 = (kt + 1 < NKT) && kt + 1 >= kt_first && kt + 1 <= qt;
            f32x4 acc0 = zero4, acc1 = zero4;
            if (c0 && c1) {
                mma_tile_frag2<NDS>(Vs + 16 * kt * KPA, Vs + 16 * (kt + 1) * KPA, KPA, dof, dof, acc0, acc1);
                acc0 = dp_finish(kt, acc0);
                acc1 = dp_finish(kt + 1, acc1);
            } else if (c0) {
                acc0 = dp_finish(kt, mma_tile_frag<NDS>(Vs + 16 * kt * KPA, KPA, dof));
            } else if (c1) {
                acc1 = dp_finish(kt + 1, mma_tile_frag<NDS>(Vs + 16 * (kt + 1) * KPA, KPA, dof));
            }
            dps[kt] = acc0;
            if (kt + 1 < NKT) dps[kt + 1] = acc1;
        }
        delta = grp_sum(delta);
        if (qi == qi_first) AT_TS(5);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st[kt][r] = live ? st[kt][r] * (dps[kt][r] - delta) * g.isd : 0.0f;   // dS / sqrt(d)
        }
        // dQ[q][dim] = sum_key dS[q][key] K[key][dim]
        f32x4 acc[NDT];
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) acc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_prob_rows<NKT, NDT>(st, Ks, KPA, kt_first, qt + 1, acc);
        if (qi == qi_first) AT_TS(6);
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + 4 * lg + r, c = 16 * jt + li;
                if (qq < T && c < d.d) bd.dQ[(size_t)(base_row + qq) * bd.ldg + hoff + c] = acc[jt][r];
            }
        }
        if (lg == 0 && q < T) {
            float* sp = bd.stats + ((size_t)blockIdx.x * T + q) * 4;
            float flag = 0.0f;
            if (is_dead || (uniform && nz == 0.0f)) flag = 2.0f;
            else if (uniform) flag = 1.0f;
            sp[0] = m2; sp[1] = inv; sp[2] = delta; sp[3] = flag;
        }
        if (qi == qi_first) AT_TS(7);
    }
    AT_TS(15);
}

template <int NDS, int NDT>
__global__ __launch_bounds__(64 * A_MAX_WAVES) void k_attn_bwd_kv(cr_attn_bwd_desc bd, AttnGeom g) {
    constexpr int KPA = 4 * NDS + 2;                     // LDS pitches as compile-time constants: operand offsets fold into
    constexpr int KPB = 4 * NDS + (((4 * NDS) % 8 == 4) ? 0 : 4);   // the ds_read immediates (the runtime pitch cost a multiply-add per access)
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nw = blockDim.x >> 6;
    float* Qs = smem;                                   // [T16][PA]  A- and B-pattern reads
    float* Os = Qs + g.T16 * KPA + A_TAIL;             // [T16][PA]  dOut
    // per-row statistics, stored so that the inner loop is branch-free (16-byte reads of 4 consecutive rows):
    //   P[q][key] = valid * exp2(s*c - smx) * sinv + (key < T ? suni : 0)
    // normal row: smx = max, sinv = 1/sum, suni = 0; uniform row: sinv = 0, suni = 1/T; dead row: both 0
    // (smx = +1e30 wherever sinv = 0, so the exponential is exactly 0 instead of a possible inf * 0)
    float* smx = Os + g.T16 * KPA + A_TAIL;            // [T16] each
    float* sinv = smx + g.T16;
    float* sdel = sinv + g.T16;
    float* sflag = sdel + g.T16;
    float* qv = sflag + g.T16;
    float* suni = qv + g.T16;
    float* tile_uni = suni + g.T16;                     // [nkt]: tile holds a flag==1 row
    float* tile_live = tile_uni + g.nkt;                // [nkt]: tile holds a row with flag != 2
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    AT_TS(0); AT_TS(1);
    const DropCtx dc = drop_ctx(d.drop);                // reads the step counter: requested first, needed late
    TileSched sch = sched_init(nw, wave);
    int kt = sched_rank(sch);                           // key tile 0 meets every query tile: rank == kt
    const int kt_first_ = kt;
    float kn[NDS], vn[NDS];                             // K / V fragments of the wave's next key tile (in flight during the staging)
    if (kt < g.nkt) {
        frag_issue<NDS>(d.K, d.ld, base_row + 16 * kt, hoff, T - 16 * kt, d.d, kn);
        frag_issue<NDS>(d.V, d.ld, base_row + 16 * kt, hoff, T - 16 * kt, d.d, vn);
    }
    // per-row statistics of pass 1: requested before the Q / dOut streams, written to LDS after them
    typedef float f4s __attribute__((ext_vector_type(4), aligned(4)));
    const int t0 = threadIdx.x;
    const f4s st0 = *reinterpret_cast<const f4s*>(bd.stats + ((size_t)blockIdx.x * T + (t0 < T ? t0 : 0)) * 4);
    const float qv0 = d.q_valid[base_row + (t0 < T ? t0 : 0)];
    stage_pair<NDS>(Qs, KPA, d.Q, d.ld, Os, KPA, bd.dout, bd.lddo, base_row, hoff, T, d.d, g.T16);
    auto put_stats = [&](int t, float mx_, float inv_, float del_, float flag_, float qv_) {
        const float flag = (t < T) ? flag_ : 2.0f;
        const bool normal = flag == 0.0f;
        smx[t] = normal ? mx_ : 1e30f;
        sinv[t] = normal ? inv_ : 0.0f;
        sdel[t] = normal ? del_ : 0.0f;
        suni[t] = (flag == 1.0f) ? g.invT : 0.0f;
        sflag[t] = flag;
        qv[t] = (t < T) ? qv_ : 0.0f;
    };
    if (t0 < g.T16) put_stats(t0, st0.x, st0.y, st0.z, st0.w, qv0);
    for (int t = t0 + blockDim.x; t < g.T16; t += blockDim.x) {
        const float* sp = bd.stats + ((size_t)blockIdx.x * T + (t < T ? t : 0)) * 4;
        put_stats(t, sp[0], sp[1], sp[2], sp[3], d.q_valid[base_row + (t < T ? t : 0)]);
    }
    AT_TS(2);
    __syncthreads();
    AT_TS(3);
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
    AT_TS(4);
    for (; kt < g.nkt; kt = sched_next(sch)) {
        const int key = 16 * kt + li;
        const float key_in_T = key < T ? 1.0f : 0.0f;
        const uint32_t drop_base = attn_row_idx(d, head, n, 0) + (uint32_t)key;
        float kf[NDS], vf[NDS];
        frag_finish<NDS>(kn, T - 16 * kt, d.d, kf);
        frag_finish<NDS>(vn, T - 16 * kt, d.d, vf);
        if (sched_peek(sch) < g.nkt) {
            const int ktn = sched_peek(sch);
            frag_issue<NDS>(d.K, d.ld, base_row + 16 * ktn, hoff, T - 16 * ktn, d.d, kn);
            frag_issue<NDS>(d.V, d.ld, base_row + 16 * ktn, hoff, T - 16 * ktn, d.d, vn);
        }
        const bool kvk = (key < T) && (d.k_valid[base_row + (key < T ? key : 0)] != 0.0f);
        const bool tile_has_key = __any(kvk ? 1 : 0) != 0;      // all-padding key tile: only uniform rows reach it
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) {
            dk[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int qt = 0; qt < g.nkt; ++qt) {
            if (tile_live[qt] == 0.0f) continue;                          // nothing flows through dead query tiles
            if ((qt < kt || !tile_has_key) && tile_uni[qt] == 0.0f) continue;   // causal / padding skip (uniform rows see all keys)
            f32x4 s_acc, p_acc;                                                      // S[q][key], dP[q][key]: two interleaved chains
            mma_tile_frag2<NDS>(Qs + 16 * qt * KPA, Os + 16 * qt * KPA, KPA, kf, vf, s_acc, p_acc);
            float pa[4], pd[4];
            {
                const int q4 = 16 * qt + 4 * lg;                                          // this lane's 4 query rows
                const float4 m4 = *reinterpret_cast<const float4*>(smx + q4), i4 = *reinterpret_cast<const float4*>(sinv + q4);
                const float4 d4 = *reinterpret_cast<const float4*>(sdel + q4), u4 = *reinterpret_cast<const float4*>(suni + q4);
                const float4 w4 = *reinterpret_cast<const float4*>(qv + q4);
                const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w};
                const float dd[4] = {d4.x, d4.y, d4.z, d4.w}, uu[4] = {u4.x, u4.y, u4.z, u4.w};
                const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
                const uint32_t x0 = (drop_base + (uint32_t)(q4 * T)) * CR_PHI + dc.key;    // counter of attention_weights[(j*B+n), q4, key]
                const uint32_t xT = (uint32_t)T * CR_PHI;                                 // next query row
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = (key <= q4 + r) && kvk;                            // causal + key mask
                    const float e = __builtin_amdgcn_exp2f(fmaf(s_acc[r], g.isd_log2e, -mm[r])) * ii[r];
                    const float pn = valid ? e : 0.0f;                                    // softmax probability of a normal row
                    float w = ww[r];                                                      // query mask (* dropout keep / (1-rate))
                    if (dc.on) w *= drop_factor_x(dc, x0 + (uint32_t)r * xT);
                    pa[r] = (pn + key_in_T * uu[r]) * w;                                  // A after mask + dropout
                    pd[r] = pn * (p_acc[r] * w - dd[r]) * g.isd;                          // dS / sqrt(d)
                }
            }
            const float* op = Os + (16 * qt + 4 * lg) * KPA + li;
            const float* qp = Qs + (16 * qt + 4 * lg) * KPA + li;
            float bo[4][NDT], bq[4][NDT];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jt = 0; jt < NDT; ++jt) {
                    bo[r][jt] = op[r * KPA + 16 * jt];
                    bq[r][jt] = qp[r * KPA + 16 * jt];
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jt = 0; jt < NDT; ++jt) {
                    dv[jt] = mfma16(pa[r], bo[r][jt], dv[jt]);
                    dk[jt] = mfma16(pd[r], bq[r][jt], dk[jt]);
                }
        }
        if (kt == kt_first_) AT_TS(5);
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * kt + 4 * lg + r, c = 16 * jt + li;
                if (kk < T && c < d.d) {
                    bd.dK[(size_t)(base_row + kk) * bd.ldg + hoff + c] = dk[jt][r];
                    bd.dV[(size_t)(base_row + kk) * bd.ldg + hoff + c] = dv[jt][r];
                }
            }
        }
        if (kt == kt_first_) AT_TS(6);
    }
    AT_TS(15);
}

static size_t lds_bwd_q(const AttnGeom& g, int w) {
    return sizeof(float) * ((size_t)g.T16 * (2 * g.PA) + A_TAIL + 3 * g.T16) + 0 * (size_t)w;
}
static size_t lds_bwd_kv(const AttnGeom& g, int w) {
    return sizeof(float) * ((size_t)g.T16 * (2 * g.PA) + 2 * A_TAIL + 6 * g.T16 + 2 * g.nkt) + 0 * (size_t)w;
}

template <int NKT, int NDS, int NDT>
static int launch_bwd_q(const cr_attn_bwd_desc* bd, const AttnGeom& g, int waves, hipStream_t s) {
    static cr_devmask attr_set = 0;
    {
        int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_attn_bwd_q<NKT, NDS, NDT>), &attr_set);
        if (rc) return rc;
    }
    AttnGeom gg = g;
    if (g_attn_ts_which != 1) gg.ts = nullptr;
    hipLaunchKernelGGL((k_attn_bwd_q<NKT, NDS, NDT>), dim3(bd->f.B * bd->f.H, attn_nsplit(&bd->f, g, waves)), dim3(64 * waves),
                       lds_bwd_q(g, waves), s, *bd, gg);
    return cr_check_launch("cr_attn_bwd(q)");
}

template <int NKT>
static int dispatch_bwd_q(const cr_attn_bwd_desc* bd, const AttnGeom& g, int waves, hipStream_t s) {
    if (g.nds == 8) return launch_bwd_q<NKT, 8, 2>(bd, g, waves, s);
    if (g.nds == 13) return launch_bwd_q<NKT, 13, 4>(bd, g, waves, s);
    return launch_bwd_q<NKT, 16, 4>(bd, g, waves, s);
}

template <int NDS, int NDT>
static int launch_bwd_kv(const cr_attn_bwd_desc* bd, const AttnGeom& g, int waves, hipStream_t s) {
    static cr_devmask attr_set = 0;
    {
        int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_attn_bwd_kv<NDS, NDT>), &attr_set);
        if (rc) return rc;
    }
    AttnGeom gg = g;
    if (g_attn_ts_which != 2) gg.ts = nullptr;
    hipLaunchKernelGGL((k_attn_bwd_kv<NDS, NDT>), dim3(bd->f.B * bd->f.H, attn_nsplit(&bd->f, g, waves)), dim3(64 * waves),
                       lds_bwd_kv(g, waves), s, *bd, gg);
    return cr_check_launch("cr_attn_bwd(kv)");
}

__global__ __launch_bounds__(256) void k_attn_zero_cols(float* p, int ld, int M, int C) {
    const long long total = (long long)M * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        p[(size_t)(i / C) * ld + (i % C)] = 0.0f;
}

int cr_attn_zero_cols_launch(float* p, int ld, int M, int C, hipStream_t s) {
    int grid = cr_ceil_div(M * C, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_attn_zero_cols, dim3(grid), dim3(256), 0, s, p, ld, M, C);
    return cr_check_launch("cr_attn_bwd(zero dQ_part)");
}

extern "C" int cr_attn_bwd(const cr_attn_bwd_desc* bd, void* stream) {
    CR_REQUIRE(bd != nullptr, "cr_attn_bwd: NULL desc");
    const cr_attn_desc* d = &bd->f;
    AttnGeom g;
    int rc = attn_validate(d, "cr_attn_bwd");
    if (rc) return rc;
    CR_REQUIRE(bd->dout && bd->dQ && bd->dK && bd->dV && bd->stats, "cr_attn_bwd: NULL pointer");
    CR_REQUIRE(bd->ldg >= d->H * d->d && bd->lddo >= d->H * d->d, "cr_attn_bwd: ldg / lddo too small");
    hipStream_t s = cr_stream(stream);
    if (d->precision != CR_PREC_F32 && cr_attn_bf_supported_bwd(d)) {
        if (bd->dQ_part) {                              // the bf16 kernels return the whole dQ: the second partial is zero
            rc = cr_attn_zero_cols_launch(bd->dQ_part, bd->ldg, d->B * d->T, d->H * d->d, s);
            if (rc) return rc;
        }
        return cr_attn_bf_bwd_launch(bd, s);
    }
    if (!attn_lds_envelope(d)) {
        if (bd->dQ_part) {
            const int M = d->B * d->T, Cc = d->H * d->d;
            int grid = cr_ceil_div(M * Cc, 256);
            if (grid > 2048) grid = 2048;
            hipLaunchKernelGGL(k_attn_zero_cols, dim3(grid), dim3(256), 0, s, bd->dQ_part, bd->ldg, M, Cc);
        }
        return cr_attn_wide_bwd_launch(bd, s);
    }
    rc = attn_geom(d, &g, "cr_attn_bwd");
    if (rc) return rc;
    if (d->row_stats && bd->delta && bd->dQ_part) {     // single pass: forward statistics + delta supplied
        rc = cr_attn_bwd_single_pass(bd, g, s);
        if (rc != 0) return rc < 0 ? rc : CR_OK;
    }
    if (bd->dQ_part) {                                  // two passes return the whole dQ: the second partial is zero
        const int M = d->B * d->T, Cc = d->H * d->d;
        int grid = cr_ceil_div(M * Cc, 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(k_attn_zero_cols, dim3(grid), dim3(256), 0, s, bd->dQ_part, bd->ldg, M, Cc);
    }
    const int wq = attn_pick_waves(g, lds_bwd_q), wkv = attn_pick_waves(g, lds_bwd_kv);
    if (!wq || !wkv) return cr_attn_wide_bwd_launch(bd, s);
    const int nkt = attn_pick_nkt(g.nkt);
    if (nkt == 4) rc = dispatch_bwd_q<4>(bd, g, wq, s);
    else if (nkt == 13) rc = dispatch_bwd_q<13>(bd, g, wq, s);
    else rc = dispatch_bwd_q<16>(bd, g, wq, s);
    if (rc) return rc;
    if (g.nds == 8) return launch_bwd_kv<8, 2>(bd, g, wkv, s);
    if (g.nds == 13) return launch_bwd_kv<13, 4>(bd, g, wkv, s);
    return launch_bwd_kv<16, 4>(bd, g, wkv, s);
}
