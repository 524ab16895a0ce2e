// Attention forward (modules.py:208-269); see cr_attn_common.hpp for the design.
#include "cr_attn_common.hpp"

#ifdef CR_TIMELINE
unsigned long long* g_attn_ts = nullptr;
int g_attn_ts_which = 0;
extern "C" void cr_debug_attn_ts(void* p, int which) { g_attn_ts = static_cast<unsigned long long*>(p); g_attn_ts_which = which; }
#endif

template <int NKT, int NDS, int NDT>
__global__ __launch_bounds__(64 * A_MAX_WAVES) void k_attn_fwd(cr_attn_desc d, AttnGeom g) {
    constexpr int KPA = 4 * NDS + 2;                     // LDS pitches as compile-time constants: operand offsets fold into
    constexpr int KPB = 4 * NDS + (((4 * NDS) % 8 == 4) ? 0 : 4);   // the ds_read immediates (the runtime pitch cost a multiply-add per access)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nw = blockDim.x >> 6;
    float* Ks = smem;                                   // [T16][PA]  A-pattern reads
    float* Vs = Ks + g.T16 * KPA;                      // [T16][PB]  B-pattern reads (+ tail)
    float* kv = Vs + g.T16 * KPB + A_TAIL;             // [T16]
    float* qv = kv + g.T16;                             // [T16]
    float* dead = qv + g.T16;                           // [T16]
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    AT_TS(0); AT_TS(1);
    const DropCtx dc = drop_ctx(d.drop);                // reads the step counter: requested first, needed late
    TileSched sch = sched_init(nw, wave);
    int qi = sched_rank(sch);                           // rank of the wave's first tile (0 = heaviest)
    float qn[NDS];                                      // Q fragment of the wave's next tile, in flight during the staging
    if (qi < g.nkt) frag_issue<NDS>(d.Q, d.ld, base_row + 16 * (g.nkt - 1 - qi), hoff, T - 16 * (g.nkt - 1 - qi), d.d, qn);
    // per-row flags: requested before the K/V streams, written to LDS after them (one latency for everything)
    const int t0 = threadIdx.x;
    const int t0c = (t0 < T) ? base_row + t0 : base_row;
    const float kv0 = d.k_valid[t0c], qv0 = d.q_valid[t0c];
    const int id0 = d.dead_ids ? d.dead_ids[t0c] : 1;
    stage_pair<NDS>(Ks, KPA, d.K, d.ld, Vs, KPB, d.V, d.ld, base_row, hoff, T, d.d, g.T16);
    if (t0 < g.T16) {
        kv[t0] = (t0 < T && kv0 != 0.0f) ? 0.0f : -INFINITY;      // additive key bias
        qv[t0] = (t0 < T) ? qv0 : 0.0f;
        dead[t0] = (t0 >= T || id0 == 0) ? 1.0f : 0.0f;
    }
    for (int t = t0 + blockDim.x; t < g.T16; t += blockDim.x) {     // fewer threads than rows (small workgroups)
        kv[t] = (t < T && d.k_valid[base_row + t] != 0.0f) ? 0.0f : -INFINITY;
        qv[t] = (t < T) ? d.q_valid[base_row + t] : 0.0f;
        dead[t] = (t >= T || (d.dead_ids && d.dead_ids[base_row + t] == 0)) ? 1.0f : 0.0f;
    }
    AT_TS(2);
    __syncthreads();
    AT_TS(3);
    const int kt_first = first_valid_tile<NKT>(kv, g.nkt);
    const int qi_first = qi;
    for (; qi < g.nkt; qi = sched_next(sch)) {
        const int qt = g.nkt - 1 - qi;                  // heaviest tiles first
        const int q0 = 16 * qt, q = q0 + li;
        float qf[NDS];
        frag_finish<NDS>(qn, T - q0, d.d, qf);
        if (sched_peek(sch) < g.nkt) {                  // prefetch the next tile's fragment behind this tile's work
            const int qtn = g.nkt - 1 - sched_peek(sch);
            frag_issue<NDS>(d.Q, d.ld, base_row + 16 * qtn, hoff, T - 16 * qtn, d.d, qn);
        }
        const bool is_dead = dead[q] != 0.0f;
        if (__all(is_dead ? 1 : 0) && d.attn_weights == nullptr) {
            // the whole tile is padding: A = 0 -> out = residual (known dead downstream, sasrec.py:83)
            if (d.row_stats && lg == 0 && q < T) {
                float* sp = d.row_stats + ((size_t)blockIdx.x * T + q) * 4;
                sp[0] = 0.0f; sp[1] = 0.0f; sp[2] = 2.0f; sp[3] = 0.0f;
            }
            for (int rr = 0; rr < 16; ++rr) {
                const int qq = q0 + rr;
                if (qq < T && lane < d.d) {
                    const size_t row = (size_t)(base_row + qq);
                    d.out[row * d.ldo + hoff + lane] = d.residual[row * d.ldr + hoff + lane];
                }
            }
            continue;
        }
        f32x4 st[NKT];
        float m2, inv;
        bool uniform;
        score_rows<NKT, NDS>(g, Ks, qf, kv, kt_first, qt, T, is_dead, q < T, st, m2, inv, uniform, qi == qi_first);
        if (qi == qi_first) AT_TS(4);
        if (d.row_stats && lg == 0 && q < T) {                                       // for the single-pass backward
            float* sp = d.row_stats + ((size_t)blockIdx.x * T + q) * 4;
            sp[0] = m2; sp[1] = inv; sp[2] = is_dead ? 2.0f : (uniform ? 1.0f : 0.0f); sp[3] = 0.0f;
        }
        const float qvq = qv[q];
        const bool any_uni = __any(uniform ? 1 : 0) != 0;
        const uint32_t ridx = attn_row_idx(d, head, n, q);
        const uint32_t xrow = (ridx + (uint32_t)(4 * lg)) * CR_PHI + dc.key;         // counter of key 4*lg; + (16kt + r) * PHI per element
        if (dc.on) {                                                                 // wave-uniform
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r)                                          // modules.py:248-257
                    st[kt][r] *= qvq * drop_factor_x(dc, xrow + (uint32_t)(16 * kt + r) * CR_PHI);
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) st[kt] *= qvq;                          // modules.py:248-253
        }
        if (d.attn_weights) {                                                        // modules.py:259 (on request only)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * lg + r;
                    if (q < T && key < T) d.attn_weights[((size_t)blockIdx.x * T + q) * T + key] = st[kt][r];
                }
            }
        }
        if (qi == qi_first) AT_TS(5);
        // residual (modules.py:265-269) requested ahead of the P V MFMAs that hide its latency
        float resid[NDT][4];
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + 4 * lg + r, c = 16 * jt + li;
                const bool ok = qq < T && c < d.d;
                resid[jt][r] = d.residual[ok ? (size_t)(base_row + qq) * d.ldr + hoff + c : (size_t)base_row * d.ldr + hoff];
            }
        f32x4 acc[NDT];
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) acc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_prob_rows<NKT, NDT>(st, Vs, KPB, any_uni ? 0 : kt_first, any_uni ? g.nkt : qt + 1, acc);   // modules.py:262
        if (qi == qi_first) AT_TS(6);
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + 4 * lg + r, c = 16 * jt + li;
                if (qq < T && c < d.d) {
                    const size_t row = (size_t)(base_row + qq);
                    d.out[row * d.ldo + hoff + c] = acc[jt][r] + resid[jt][r];   // modules.py:265-269
                }
            }
        }
        if (qi == qi_first) AT_TS(7);
    }
    AT_TS(15);
}

static size_t lds_fwd(const AttnGeom& g, int w) {
    return sizeof(float) * ((size_t)g.T16 * (g.PA + g.PB) + A_TAIL + 3 * g.T16) + 0 * (size_t)w;
}

template <int NKT, int NDS, int NDT>
static int launch_fwd(const cr_attn_desc* d, const AttnGeom& g, int waves, hipStream_t s) {
    static cr_devmask attr_set = 0;
    {
        int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_attn_fwd<NKT, NDS, NDT>), &attr_set);
        if (rc) return rc;
    }
    AttnGeom gg = g;
    if (g_attn_ts_which != 0) gg.ts = nullptr;
    hipLaunchKernelGGL((k_attn_fwd<NKT, NDS, NDT>), dim3(d->B * d->H, attn_nsplit(d, g, waves)), dim3(64 * waves),
                       lds_fwd(g, waves), s, *d, gg);
    return cr_check_launch("cr_attn_fwd");
}

template <int NKT>
static int dispatch_fwd(const cr_attn_desc* d, const AttnGeom& g, int waves, hipStream_t s) {
    if (g.nds == 8) return launch_fwd<NKT, 8, 2>(d, g, waves, s);
    if (g.nds == 13) return launch_fwd<NKT, 13, 4>(d, g, waves, s);
    return launch_fwd<NKT, 16, 4>(d, g, waves, s);
}

extern "C" int cr_attn_fwd(const cr_attn_desc* d, void* stream) {
    CR_REQUIRE(d != nullptr, "cr_attn_fwd: NULL desc");
    AttnGeom g;
    int rc = attn_validate(d, "cr_attn_fwd");
    if (rc) return rc;
    CR_REQUIRE(d->out && d->residual, "cr_attn_fwd: NULL out/residual");
    hipStream_t s = cr_stream(stream);
    if (d->precision != CR_PREC_F32 && cr_attn_bf_supported_fwd(d)) return cr_attn_bf_fwd_launch(d, s);
    if (!attn_lds_envelope(d)) return cr_attn_wide_fwd_launch(d, s);
    rc = attn_geom(d, &g, "cr_attn_fwd");
    if (rc) return rc;
    const int waves = attn_pick_waves(g, lds_fwd);
    if (!waves) return cr_attn_wide_fwd_launch(d, s);
    const int nkt = attn_pick_nkt(g.nkt);
    if (nkt == 4) return dispatch_fwd<4>(d, g, waves, s);
    if (nkt == 13) return dispatch_fwd<13>(d, g, waves, s);
    return dispatch_fwd<16>(d, g, waves, s);
}
