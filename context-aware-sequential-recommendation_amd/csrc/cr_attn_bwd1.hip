// Single-pass attention backward (gradient of modules.py:208-269), key-owner form.
//
// The two-pass backward (cr_attn_bwd.hip) recomputes S and dP in both passes: 7 MFMA products and two launches.
// With the forward's per-row softmax statistics saved (cr_attn_desc.row_stats) and the softmax-backward row
// term delta[q] = sum_c dO[q][c] * (O[q][c] - residual[q][c]) supplied by the caller (identical to
// sum_k dP[q][k] P[q][k], dropout and query mask included), one pass is enough: 5 products, one launch.
//
// A wave owns 16 keys (K / V fragments in registers, dK / dV accumulators in registers) and walks the query
// tiles at or below the diagonal, exactly like the key-owner pass of the two-pass version.  New here: it also
// produces this key tile's contribution to dQ,  dQ[q][:] += dS[q][key-tile] K[key-tile][:].  The score tile
// comes out of the MFMA as S[q = 4 lg + r][key = li]; dQ needs dS with the QUERY on the lane index, so the
// 16 x 16 dS tile makes one round trip through a per-wave LDS slot (4 writes + 4 reads per lane), and the
// product is accumulated into an LDS-resident dQ tile [T16][PA] shared by the waves of the workgroup.
// LDS float atomics are far too slow for that (measured: 91 us, ds_add_f32 runs at a few lanes per clock), so
// the walk is ROTATED instead: in step j the wave that owns key tile kt works on query tile (kt + j) mod nkt, and
// a workgroup barrier closes every step.  Distinct key tiles meet distinct query tiles in every step, so the
// accumulation is a plain LDS read-add-write with a fixed order (bitwise reproducible), at the price of nkt
// barriers per round.  The start of each key tile's walk (AttnGeom.rot, chosen on the host) is shifted so that the
// two waves sharing a SIMD have their causally live steps at different times (a light tile's few live steps fit
// into the heavy tile's dead ones): every step then costs one pair, not two.  When a sample is split over two workgroups (gridDim.y = 2) each holds a partial dQ:
// workgroup 0 writes `dQ`, workgroup 1 writes `dQ_part`, the caller adds.
#include "cr_attn_common.hpp"

#define B1_TP 17       // pitch of the per-wave 16 x 16 transpose slot

template <int NDS, int NDT>
__global__ __launch_bounds__(64 * A_MAX_WAVES) void k_attn_bwd_one(cr_attn_bwd_desc bd, AttnGeom g) {
    constexpr int KPA = 4 * NDS + 2;                     // LDS pitches as compile-time constants: operand offsets fold into
    constexpr int KPB = 4 * NDS + (((4 * NDS) % 8 == 4) ? 0 : 4);   // the ds_read immediates (the runtime pitch cost a multiply-add per access)
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nw = blockDim.x >> 6;
    float* Qs = smem;                                   // [T16][PA]  A- and B-pattern reads
    float* Os = Qs + g.T16 * KPA + A_TAIL;             // [T16][PA]  dOut
    float* dQs = Os + g.T16 * KPA + A_TAIL;            // [T16][PA]  dQ accumulator (this workgroup's key tiles)
    float* Tw = dQs + g.T16 * KPA;                     // [nw][16][B1_TP] dS transpose slots
    float* smx = Tw + nw * 16 * B1_TP;                  // per-row statistics, as in the two-pass key-owner kernel (16-byte
                                                        // aligned: every block above is a multiple of 4 floats)
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
    const DropCtx dc = drop_ctx(d.drop);                // reads the step counter: requested first, needed late
    AT_TS(0); AT_TS(1);
    TileSched sch = sched_init(nw, wave);
    int kt = sched_rank(sch);                           // key tile 0 meets every query tile: rank == kt
    float kn[NDS], vn[NDS];                             // K / V fragments of the wave's next key tile
    float kbn[4][NDT];                                  // ... and its K rows as a B operand: K[key = 4 s + lg][c = 16 jt + li]
    auto issue_kb = [&](int ktile) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int key = 16 * ktile + 4 * s + lg;
            const float* p = d.K + (size_t)(base_row + (key < T ? key : 0)) * d.ld + hoff;
#pragma unroll
            for (int jt = 0; jt < NDT; ++jt) kbn[s][jt] = p[(16 * jt + li < d.d) ? 16 * jt + li : 0];
        }
    };
    if (kt < g.nkt) {
        frag_issue<NDS>(d.K, d.ld, base_row + 16 * kt, hoff, T - 16 * kt, d.d, kn);
        frag_issue<NDS>(d.V, d.ld, base_row + 16 * kt, hoff, T - 16 * kt, d.d, vn);
        issue_kb(kt);
    }
    // per-row statistics of the forward + delta: requested before the Q / dOut streams, written to LDS after them
    typedef float f4s __attribute__((ext_vector_type(4), aligned(4)));
    const int t0 = threadIdx.x;
    const int t0c = (t0 < T) ? t0 : 0;
    const f4s st0 = *reinterpret_cast<const f4s*>(d.row_stats + ((size_t)blockIdx.x * T + t0c) * 4);
    const float qv0 = d.q_valid[base_row + t0c];
    const float dl0 = bd.delta[base_row + t0c];
    // dQ accumulator cleared while the fragment / statistics requests above are in flight (16-byte LDS stores; the
    // block is a multiple of 4 floats and 16-byte aligned), then the Q / dOut staging
    for (int i = threadIdx.x; i < (g.T16 * KPA) >> 2; i += blockDim.x) reinterpret_cast<float4*>(dQs)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    stage_pair<NDS>(Qs, KPA, d.Q, d.ld, Os, KPA, bd.dout, bd.lddo, base_row, hoff, T, d.d, g.T16);
    auto put_stats = [&](int t, float mx_, float inv_, float flag_, float del_, float qv_) {
        const float flag = (t < T) ? flag_ : 2.0f;
        const bool normal = flag == 0.0f;
        smx[t] = normal ? mx_ : 1e30f;
        sinv[t] = normal ? inv_ : 0.0f;
        sdel[t] = normal ? del_ : 0.0f;
        suni[t] = (flag == 1.0f) ? g.invT : 0.0f;
        sflag[t] = flag;
        qv[t] = (t < T) ? qv_ : 0.0f;
    };
    if (t0 < g.T16) put_stats(t0, st0.x, st0.y, st0.z, dl0, qv0);
    for (int t = t0 + blockDim.x; t < g.T16; t += blockDim.x) {
        const int tc = t < T ? t : 0;
        const float* sp = d.row_stats + ((size_t)blockIdx.x * T + tc) * 4;
        put_stats(t, sp[0], sp[1], sp[2], bd.delta[base_row + tc], d.q_valid[base_row + tc]);
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
    float* tw = Tw + wave * 16 * B1_TP;
    const int nrounds = (g.nkt + (int)gridDim.y * nw - 1) / ((int)gridDim.y * nw);   // same for every wave: barriers inside
    for (int round = 0; round < nrounds; ++round, kt = sched_next(sch)) {
        const bool have = kt < g.nkt;                       // this wave owns a key tile in this round
        const int key = 16 * kt + li;
        const float key_in_T = key < T ? 1.0f : 0.0f;
        const uint32_t drop_base = attn_row_idx(d, head, n, 0) + (uint32_t)key;
        float kf[NDS], vf[NDS], kb[4][NDT];
        frag_finish<NDS>(kn, have ? T - 16 * kt : 0, d.d, kf);
        frag_finish<NDS>(vn, have ? T - 16 * kt : 0, d.d, vf);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int jt = 0; jt < NDT; ++jt)
                kb[s][jt] = (have && 16 * kt + 4 * s + lg < T && 16 * jt + li < d.d) ? kbn[s][jt] : 0.0f;
        if (sched_peek(sch) < g.nkt) {
            const int ktn = sched_peek(sch);
            frag_issue<NDS>(d.K, d.ld, base_row + 16 * ktn, hoff, T - 16 * ktn, d.d, kn);
            frag_issue<NDS>(d.V, d.ld, base_row + 16 * ktn, hoff, T - 16 * ktn, d.d, vn);
            issue_kb(ktn);
        }
        const bool kvk = have && (key < T) && (d.k_valid[base_row + (key < T ? key : 0)] != 0.0f);
        const bool tile_has_key = __any(kvk ? 1 : 0) != 0;      // all-padding key tile: only uniform rows reach it
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) {
            dk[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int step = 0; step < g.nkt; ++step) {
            int qt = (have ? g.rot[kt] : 0) + step;                                  // rotated walk: see the header
            if (qt >= g.nkt) qt -= g.nkt;
            const bool causal_live = have && (qt >= kt) && tile_has_key;             // normal rows of this tile see these keys
            if (have && tile_live[qt] != 0.0f && (causal_live || tile_uni[qt] != 0.0f)) {
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
            // dS tile -> the wave's transpose slot (row = query, column = key); read back below with the query on li
#pragma unroll
            for (int r = 0; r < 4; ++r) tw[(4 * lg + r) * B1_TP + li] = pd[r];
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
            float at[4];                                                                  // dS[q = li][key = 4 s + lg]
#pragma unroll
            for (int s = 0; s < 4; ++s) at[s] = tw[li * B1_TP + 4 * s + lg];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jt = 0; jt < NDT; ++jt) {
                    dv[jt] = mfma16(pa[r], bo[r][jt], dv[jt]);
                    dk[jt] = mfma16(pd[r], bq[r][jt], dk[jt]);
                }
            if (causal_live) {                                                            // uniform-only visits carry no dS
                f32x4 dq[NDT];
#pragma unroll
                for (int jt = 0; jt < NDT; ++jt) dq[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int jt = 0; jt < NDT; ++jt) dq[jt] = mfma16(at[s], kb[s][jt], dq[jt]);
                float* dqp = dQs + (16 * qt + 4 * lg) * KPA + li;   // no other wave touches this query tile in this step
                // Columns >= 4*NDS of the 16*NDT computed ones are padding (exact zeros) and, the pitch being < 64, would
                // alias the first columns of the NEXT row: two lanes of one wave would read-add-write the same word in
                // different instructions, which is only safe if those stay strictly sequential.  Skip them.
#pragma unroll
                for (int jt = 0; jt < NDT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * jt + li < 4 * NDS) dqp[r * KPA + 16 * jt] += dq[jt][r];
            }
            }
            __syncthreads();
            if (round == 0 && step == 0) AT_TS(5);
            if (round == 0 && step == 5) AT_TS(6);
        }
        if (round == 0) AT_TS(7);
        if (!have) continue;
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
    }
    AT_TS(8);
    __syncthreads();
    AT_TS(9);
    // this workgroup's partial dQ: row-chunk stores (16 bytes per lane)
    float* gq = (blockIdx.y == 0) ? bd.dQ : bd.dQ_part;
    typedef float f4q __attribute__((ext_vector_type(4), aligned(4)));
    for (int item = threadIdx.x; item < T * NDS; item += blockDim.x) {
        const int r = item / NDS, q = item - r * NDS;
        if (4 * q >= d.d) continue;
        const float2* pt = reinterpret_cast<const float2*>(dQs + r * KPA + 4 * q);
        const float2 a = pt[0], b = pt[1];
        float* gp = gq + (size_t)(base_row + r) * bd.ldg + hoff + 4 * q;
        if (4 * q + 3 < d.d) {
            *reinterpret_cast<f4q*>(gp) = (f4q){a.x, a.y, b.x, b.y};
        } else {
            gp[0] = a.x;
            if (4 * q + 1 < d.d) gp[1] = a.y;
            if (4 * q + 2 < d.d) gp[2] = b.x;
        }
    }
    if (gridDim.y == 1 && bd.dQ_part) {                 // unsplit sample: the second partial is all zeros
        for (int item = threadIdx.x; item < T * d.d; item += blockDim.x) {
            const int r = item / d.d, c = item - r * d.d;
            bd.dQ_part[(size_t)(base_row + r) * bd.ldg + hoff + c] = 0.0f;
        }
    }
    AT_TS(15);
}

static size_t lds_bwd_one(const AttnGeom& g, int w) {
    return sizeof(float) * ((size_t)g.T16 * (3 * g.PA) + 2 * A_TAIL + (size_t)w * 16 * B1_TP + 32 + 6 * g.T16 + 2 * g.nkt);
}

template <int NDS, int NDT>
static int launch_bwd_one(const cr_attn_bwd_desc* bd, const AttnGeom& g, int waves, int nsplit, hipStream_t s) {
    static cr_devmask attr_set = 0;
    {
        int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_attn_bwd_one<NDS, NDT>), &attr_set);
        if (rc) return rc;
    }
    AttnGeom gg = g;
    if (g_attn_ts_which != 3) gg.ts = nullptr;
    hipLaunchKernelGGL((k_attn_bwd_one<NDS, NDT>), dim3(bd->f.B * bd->f.H, nsplit), dim3(64 * waves), lds_bwd_one(g, waves), s, *bd, gg);
    return cr_check_launch("cr_attn_bwd(single pass)");
}

// Rotation starts.  Key tile kt is causally live on query tiles kt .. nkt-1; walking from start o it is live in the
// cyclic step window [kt - o, nkt - o).  Per workgroup and round: starts must be distinct (that is what makes the LDS
// accumulation conflict-free) and the windows of the two tiles on one SIMD should not overlap.  Small backtracking
// search (<= 8 tiles); if no perfect assignment exists (e.g. tile 0 shares a SIMD), keep the plain start o = kt.
static bool rot_search(int i, int n, const int* kt, const int* simd, int nkt, int* off, unsigned* used, unsigned* busy) {
    if (i == n) return true;
    const int L = nkt - kt[i];
    for (int j0 = 0; j0 < nkt; ++j0) {
        const int o = ((kt[i] - j0) % nkt + nkt) % nkt;
        if (*used & (1u << o)) continue;
        unsigned win = 0;
        for (int s = 0; s < L; ++s) win |= 1u << ((j0 + s) % nkt);
        if (busy[simd[i]] & win) continue;
        *used |= 1u << o; busy[simd[i]] |= win; off[kt[i]] = o;
        if (rot_search(i + 1, n, kt, simd, nkt, off, used, busy)) return true;
        *used &= ~(1u << o); busy[simd[i]] &= ~win;
    }
    return false;
}

static void rot_table(AttnGeom* g, int nw, int nsplit) {
    for (int k = 0; k < 16; ++k) g->rot[k] = k;
    const int R = nw >= 2 ? 2 : 1, half = nw / R, P = nsplit * half;
    const int nrounds = (g->nkt + nsplit * nw - 1) / (nsplit * nw);
    for (int y = 0; y < nsplit; ++y)
        for (int round = 0; round < nrounds; ++round) {
            int kt[16], simd[16], n = 0;
            for (int w = 0; w < nw; ++w) {                        // mirrors sched_init / sched_rank_at on the device
                const int p = (w % half) * nsplit + y, r = w / half + R * round;
                const int rank = r * P + ((r & 1) ? P - 1 - p : p);
                if (rank < g->nkt) { kt[n] = rank; simd[n] = w % half; ++n; }
            }
            for (int a = 0; a < n; ++a)                           // heaviest (smallest kt) first
                for (int b = a + 1; b < n; ++b)
                    if (kt[b] < kt[a]) { int t = kt[a]; kt[a] = kt[b]; kt[b] = t; t = simd[a]; simd[a] = simd[b]; simd[b] = t; }
            int off[16];
            unsigned used = 0, busy[16] = {0};
            if (rot_search(0, n, kt, simd, g->nkt, off, &used, busy))
                for (int a = 0; a < n; ++a) g->rot[kt[a]] = off[kt[a]];
        }
}

// returns 1 if the single-pass kernel was launched, 0 if the shape does not fit it (caller falls back), < 0 on error
int cr_attn_bwd_single_pass(const cr_attn_bwd_desc* bd, const AttnGeom& g0, hipStream_t s) {
    const cr_attn_desc* d = &bd->f;
    if (d->H != 1) return 0;                            // dQ partial layout assumes one head per row block
    AttnGeom g = g0;
    const int waves = attn_pick_waves(g, lds_bwd_one);
    if (waves < A_MAX_WAVES) return 0;                  // LDS-resident dQ tile does not fit next to Q and dOut
    int nsplit = attn_nsplit(d, g, waves);
    if (nsplit > 2) nsplit = 2;
    rot_table(&g, waves, nsplit);
    int rc;
    if (g.nds == 8) rc = launch_bwd_one<8, 2>(bd, g, waves, nsplit, s);
    else if (g.nds == 13) rc = launch_bwd_one<13, 4>(bd, g, waves, nsplit, s);
    else rc = launch_bwd_one<16, 4>(bd, g, waves, nsplit, s);
    return rc ? rc : 1;
}
