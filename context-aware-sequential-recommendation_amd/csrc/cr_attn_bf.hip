// Attention core (modules.py:208-269) on the bf16 matrix pipe: v_mfma_f32_16x16x32_bf16, fp32 accumulation.
//
// Two arithmetic forms, selected by cr_attn_desc.precision:
//   CR_PREC_BF16X3  every fp32 operand x is split into hi = bf16(x), lo = bf16(x - hi) and a product is the three
//                   MFMAs hi*hi + hi*lo + lo*hi (lo*lo, 2^-18 relative, is dropped): ~1e-5 relative per product,
//                   inside the 1e-3 fp32 logit bound of the north star, at 3/16 of the fp32-MFMA issue time;
//   CR_PREC_BF16    hi only: plain bf16 operands (BASELINE.json configs[1] names bf16), tolerance stated in the tests.
// The fp32 kernels (cr_attn_fwd/bwd/bwd1.hip, v_mfma_f32_16x16x4_f32) stay the exact path (CR_PREC_F32).
//
// Why a new structure and not a port of those kernels.  With K = 32 per instruction a 16 x 16 x 64 score tile is 2
// (6 split) MFMAs fed by 16-byte LDS reads instead of 13 + 13 dword reads, so the matrix pipe is no longer what a
// key/query tile pair costs; the per-pair cost is LDS latency and vector work.  The backward is therefore two
// barrier-free passes that share nothing (query-owner: dQ; key-owner: dK, dV) instead of the rotated,
// barrier-per-step single pass: each recomputes S and dP on the (now cheap) matrix pipe, no cross-wave reduction,
// no LDS accumulators, no atomics, bitwise reproducible, and the same two kernels stream K/V (or Q/dOut) through
// LDS in 256-row chunks for T up to 1024 (config C5's maxlen 512) -- shapes that had no MFMA kernel at all.
//
// Data layout.  A [rows][64] bf16 image per operand matrix and half (hi, lo), 128 bytes per row, the 16-byte chunk
// index XOR-ed with (row & 6): conflict-free both for the row reads (ds_read_b128, A/B operand with k = head dim) and
// for the transposed reads (ds_read_b64_tr_b16, B operand with k = row) -- checked by brute force over all
// XOR-linear swizzles against the bank rules of MI355X_MICROARCH.md (tools/lds_banks.py).  One image serves both
// kinds of read, so nothing is stored twice.
//
// MFMA operand maps (v_mfma_f32_16x16x32_bf16, lane l, li = l & 15, lg = l >> 4):
//   A[i = li][k = 8 lg + j], B[k = 8 lg + j][col = li], D[row = 4 lg + r][col = li].
// Scores are computed transposed in the query-owner kernels (St[key][query] = K Q^T): a lane then holds, for ITS
// query li, keys 4 lg + r of a key tile; two key tiles give the 8 k-elements of the next product's A operand
// (k index 8 lg + j  <->  key tile j >> 2, key 4 lg + (j & 3)), and the matching B operand (V or K rows in that
// order, one output column per lane) is exactly what two ds_read_b64_tr_b16 deliver.  The key-owner kernel uses the
// mirrored form (S[query][key] = Q K^T, lane = key, two query tiles per k-step).
#include <stdlib.h>

#include "cr_attn_common.hpp"
#include "cr_bf16.hpp"

#define BF_CH 256                 // rows of one LDS chunk (K/V rows in the query-owner kernels, Q/dOut rows in the key-owner one)
#define BF_IMG (BF_CH * 32 * NKS) // bf16 elements of one image (inside a kernel templated on NKS)

struct BfGeom {
    int T16, nkt;                 // padded T, 16-row tiles
    int nch, ch_rows;             // chunks, rows per chunk (T16 when nch == 1, else 256)
    int M;                        // rows of the operand matrices (B * T): the vector loads may run into the NEXT row, never past the last
    float isd, isd_log2e, invT;
    unsigned long long* ts;       // debug: per-wave phase stamps [waves][16] (tools/attn_bf_ts.py); NULL in production
    // fused launch (one workgroup of 8 waves runs a whole pass of its sample): the tiles of wave w, heaviest first, up to three
    // 5-bit tile numbers (31 = none) -- dealt on the host so that the two waves of a SIMD (w and w + 4) together get an equal
    // share of the pair iterations (bf_deal_tiles)
    unsigned qpk[8], kpk[8];
};
#ifdef CR_TIMELINE
#define BT_TS(slot)                                                                                          \
    do {                                                                                                     \
        if (g.ts && (threadIdx.x & 63) == 0)                                                                 \
            g.ts[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] = \
                ((slot) == 0 || (slot) == 15) ? wall_clock64() : clock64();                                  \
    } while (0)
#else
#define BT_TS(slot) do { } while (0)
#endif

// Stage rows [crow0, crow0 + nrows) of two [T, d] head blocks into their LDS images (chunk-relative rows).
// An item is (row, 16-byte chunk); the 4 x U loads of a batch are issued before the first conversion
// (U = 4: the 208 x 8 items of the headline shape are ONE batch for 512 threads, one memory latency).
template <bool SPLIT, int NKS>
__device__ __forceinline__ void stage_pair_bf(__bf16* ah, __bf16* al, const float* srcA, int ldA, __bf16* bh, __bf16* bl,
                                              const float* srcB, int ldB, int base_row, int crow0, int nrows, int T,
                                              int hoff, int d, int M) {
    constexpr int CPR = 4 * NKS;                         // chunks per row that MFMAs read (columns < 32 NKS)
    const int total = nrows * CPR;
    constexpr int U = 4;
    const bool wg_has_last = base_row + T == M;          // only the last sample's workgroups can meet the matrix's last row
    for (int i0 = threadIdx.x; i0 < total; i0 += blockDim.x * U) {
        float va[U][8], vb[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int item = min(i0 + u * (int)blockDim.x, total - 1);
            const int r = item / CPR, ch = item - r * CPR;
            const int t = crow0 + r;
            const bool rok = t < T;
            const int grow = base_row + (rok ? t : 0);
            const bool fix = item_fix(rok, grow == M - 1, 8 * ch, d);
            item_issue(va[u], srcA + (size_t)grow * ldA + hoff, 8 * ch, d, fix);
            item_issue(vb[u], srcB + (size_t)grow * ldB + hoff, 8 * ch, d, fix);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * (int)blockDim.x < total) {
                const int item = i0 + u * (int)blockDim.x;
                const int r = item / CPR, ch = item - r * CPR;
                const int t = crow0 + r;
                const bool rok = t < T;
                const bool fix = wg_has_last && item_fix(rok, t == T - 1, 8 * ch, d);
                const int o = img_off<NKS>(r, ch);
                bf8 h, l;
                item_mask(va[u], 8 * ch, d, rok, fix);
                item_mask(vb[u], 8 * ch, d, rok, fix);
                if (__builtin_expect(wg_has_last && fix, 0)) {           // one thread of the grid
                    item_refill(va[u], srcA + (size_t)(M - 1) * ldA + hoff, 8 * ch, d);
                    item_refill(vb[u], srcB + (size_t)(M - 1) * ldB + hoff, 8 * ch, d);
                }
                split8<SPLIT>(va[u], h, l);
                *reinterpret_cast<bf8*>(ah + o) = h;
                if (SPLIT) *reinterpret_cast<bf8*>(al + o) = l;
                split8<SPLIT>(vb[u], h, l);
                *reinterpret_cast<bf8*>(bh + o) = h;
                if (SPLIT) *reinterpret_cast<bf8*>(bl + o) = l;
            }
        }
    }
}

// =====================================================================================================
// forward, T <= 256 (one chunk): the whole score row block of a query tile lives in registers
// =====================================================================================================
template <int NKT, int NKS, bool SPLIT>
__global__ __launch_bounds__(512) void k_bf_fwd(cr_attn_desc d, BfGeom g) {
    constexpr int NDT = 2 * NKS;                         // 16-column output tiles
    constexpr int JB = SPLIT ? 2 : NDT;                  // column tiles per batch of transposed reads (register budget)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* Kh = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Kl = Kh + (SPLIT ? g.T16 * (32 * NKS) : 0);
    __bf16* Vh = Kl + g.T16 * (32 * NKS);
    __bf16* Vl = Vh + (SPLIT ? g.T16 * (32 * NKS) : 0);
    float* kb = reinterpret_cast<float*>(Vl + g.T16 * (32 * NKS));   // [T16] additive key bias
    const int nw = blockDim.x >> 6;
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    BT_TS(0); BT_TS(1);
    const DropCtx dc = drop_ctx(d.drop);
    TileSched sch = sched_init(nw, wave);
    int qi = sched_rank(sch);
    // the wave's next tile: Q fragment and row flags, requested ahead (during the staging / the previous tile)
    GFrag<NKS> qn;
    float qv_n = 0.0f;
    int id_n = 1;
    auto issue_tile = [&](int rank) {
        const int q0n = 16 * (g.nkt - 1 - rank);
        gfrag_issue<NKS>(qn, d.Q, d.ld, base_row + q0n, hoff, T - q0n, d.d, g.M);
        const int qc = min(q0n + li, T - 1);
        qv_n = d.q_valid[base_row + qc];
        id_n = d.dead_ids ? d.dead_ids[base_row + qc] : 1;
    };
    const int qi_first = qi;
    if (qi < g.nkt) issue_tile(qi);
    const int t0 = threadIdx.x;
    const float kv0 = d.k_valid[base_row + min(t0, T - 1)];
    stage_pair_bf<SPLIT, NKS>(Kh, Kl, d.K, d.ld, Vh, Vl, d.V, d.ld, base_row, 0, g.T16, T, hoff, d.d, g.M);
    if (t0 < g.T16) kb[t0] = (t0 < T && kv0 != 0.0f) ? 0.0f : -INFINITY;
    BT_TS(2);
    __syncthreads();
    BT_TS(3);
    const int fvk = first_valid_key_lds(kb, g.T16, T);
    const int kt_first = min(fvk >> 4, NKT - 1);         // tiles below hold no valid key: probabilities exactly 0
    const float c2 = g.isd_log2e;
    // one tile per wave: the host sizes the grid so (a second round of tiles kept the fragments of both rounds live and
    // spilled: 170 -> 256 VGPRs + scratch, and a spilled value is a ~1 us scratch round trip here)
    for (bool once = qi < g.nkt; once; once = false) {
        const int qt = g.nkt - 1 - qi;                   // rank 0 = heaviest tile
        const int q0 = 16 * qt, q = q0 + li;
        bf8 qh[NKS], ql[NKS];
        gfrag_finish<SPLIT, NKS>(qn, d.Q, d.ld, base_row + q0, hoff, T - q0, d.d, g.M, qh, ql);
        const bool is_dead = q >= T || id_n == 0;
        const float qvq = q < T ? qv_n : 0.0f;
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
        // ---- scores St[key][query] (modules.py:216-241), kept for the whole row block; two key tiles per batch of reads
        f32x4 st[NKT];
        float mx = -INFINITY;
        auto finish = [&](int kt, f32x4 acc) {
            const float4 b4 = *reinterpret_cast<const float4*>(kb + 16 * kt + 4 * lg);   // key mask (modules.py:222-229)
            acc[0] = fmaf(acc[0], c2, b4.x); acc[1] = fmaf(acc[1], c2, b4.y);
            acc[2] = fmaf(acc[2], c2, b4.z); acc[3] = fmaf(acc[3], c2, b4.w);
            if (kt == qt) {                              // causal mask on the diagonal tile (modules.py:232-241)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = (4 * lg + r <= li) ? acc[r] : -INFINITY;
            }
            mx = fmaxf(fmaxf(mx, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
            return acc;
        };
        const f32x4 ninf = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kt = 0; kt < NKT; kt += 2) {
            const bool c0 = kt >= kt_first && kt <= qt;                         // wave-uniform
            const bool c1 = (kt + 1 < NKT) && kt + 1 >= kt_first && kt + 1 <= qt;
            f32x4 a0 = ninf, a1 = ninf;
            if (c0 || c1) {
                const int r0 = 16 * (c0 ? kt : kt + 1), r1 = 16 * (c1 ? kt + 1 : kt);
                bf8 k0h[NKS], k0l[NKS], k1h[NKS], k1l[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    k0h[ks] = row_frag<NKS>(Kh, r0, ks); k1h[ks] = row_frag<NKS>(Kh, r1, ks);
                    k0l[ks] = SPLIT ? row_frag<NKS>(Kl, r0, ks) : k0h[ks]; k1l[ks] = SPLIT ? row_frag<NKS>(Kl, r1, ks) : k1h[ks];
                }
                f32x4 x0 = (f32x4){0.f, 0.f, 0.f, 0.f}, x1 = x0;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    x0 = mma<SPLIT>(k0h[ks], k0l[ks], qh[ks], ql[ks], x0);
                    x1 = mma<SPLIT>(k1h[ks], k1l[ks], qh[ks], ql[ks], x1);
                }
                BF_SGB(0x100, (SPLIT ? 4 : 2) * NKS, 0);   // the pair's operand reads as one batch,
                BF_SGB(0x008, (SPLIT ? 6 : 2) * NKS, 0);   // then its MFMAs
                if (c0) a0 = finish(kt, c0 ? x0 : x1);
                if (c1) a1 = finish(kt + 1, x1);
            }
            st[kt] = a0;
            if (kt + 1 < NKT) st[kt + 1] = a1;
        }
        mx = grp_max(mx);
        if (qi == qi_first) BT_TS(4);
        const bool uniform = (mx == -INFINITY) && !is_dead && q < T;
        const float off = (mx == -INFINITY) ? 0.0f : mx;
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt >= kt_first && kt <= qt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(st[kt][r] - off);     // exp2(-inf) = 0 for masked entries
                    st[kt][r] = p;
                    sum += p;
                }
            } else {
                st[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        sum = grp_sum(sum);
        float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
        if (is_dead) inv = 0.0f;
        const bool any_uni = __any(uniform ? 1 : 0) != 0;
        if (d.row_stats && lg == 0 && q < T) {           // for the backward kernels
            float* sp = d.row_stats + ((size_t)blockIdx.x * T + q) * 4;
            sp[0] = mx; sp[1] = inv; sp[2] = is_dead ? 2.0f : (uniform ? 1.0f : 0.0f); sp[3] = 0.0f;
        }
        // ---- softmax scale, query mask, dropout (modules.py:244-257) folded into one factor per element
        const float wq = inv * qvq;
        const uint32_t ridx = attn_row_idx(d, head, n, q);
        const uint32_t xrow = (ridx + (uint32_t)(4 * lg)) * CR_PHI + dc.key;
        if (any_uni) {                                   // rare: a row with no valid key at all (modules.py:227-244)
            const float uni = uniform ? g.invT * qvq : 0.0f;
            const float sc = uniform ? 0.0f : wq;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float w = 1.0f;
                    if (dc.on) w = drop_factor_x(dc, xrow + (uint32_t)(16 * kt + r) * CR_PHI);
                    st[kt][r] = (st[kt][r] * sc + ((16 * kt + 4 * lg + r < T) ? uni : 0.0f)) * w;
                }
        } else if (dc.on) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][r] *= wq * drop_factor_x(dc, xrow + (uint32_t)(16 * kt + r) * CR_PHI);
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) st[kt] *= wq;
        }
        if (d.attn_weights) {                            // modules.py:259 (on request only)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * lg + r;
                    if (q < T && key < T) d.attn_weights[((size_t)blockIdx.x * T + q) * T + key] = st[kt][r];
                }
        }
        if (qi == qi_first) BT_TS(5);
        // ---- out = A V + residual (modules.py:262-269): two key tiles per k-step, V through transposed reads
        // residual requested ahead of the MFMAs that hide its latency
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
        const int kt_lo = any_uni ? 0 : kt_first, kt_end = any_uni ? g.nkt : qt + 1;
#pragma unroll
        for (int kp = 0; kp < (NKT + 1) / 2; ++kp) {
            const int k0 = 2 * kp, k1 = 2 * kp + 1;
            if (k1 >= kt_lo && k0 < kt_end) {            // wave-uniform; tiles outside the live range hold zeros
                float x[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    x[r] = st[k0][r];
                    x[4 + r] = (k1 < NKT) ? st[k1 < NKT ? k1 : k0][r] : 0.0f;
                }
                bf8 ph, pl;
                split8<SPLIT>(x, ph, pl);
                const int ra = 16 * k0, rb = 16 * (k1 < g.nkt ? k1 : k0);   // tiles beyond T16 are not staged (their A is 0)
#pragma unroll
                for (int j0 = 0; j0 < NDT; j0 += JB) {                   // JB column tiles per batch of transposed reads
                    bf8 bh[JB], bl[JB];
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) {
                        bh[jt] = tr_frag<NKS>(Vh, ra, rb, j0 + jt);
                        bl[jt] = SPLIT ? tr_frag<NKS>(Vl, ra, rb, j0 + jt) : bh[jt];
                    }
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) acc[j0 + jt] = mma<SPLIT>(ph, pl, bh[jt], bl[jt], acc[j0 + jt]);
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * JB, 0);
                    BF_SGB(0x008, (SPLIT ? 3 : 1) * JB, 0);
                }
            }
        }
        if (qi == qi_first) BT_TS(6);
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + 4 * lg + r, c = 16 * jt + li;
                if (qq < T && c < d.d) d.out[(size_t)(base_row + qq) * d.ldo + hoff + c] = acc[jt][r] + resid[jt][r];
            }
        if (qi == qi_first) BT_TS(7);
    }
    BT_TS(15);
}

// =====================================================================================================
// forward, 256 < T <= 1024: K / V stream through LDS in 256-key chunks, online softmax across the chunks
// (running row maximum m and sum l; the output accumulator is rescaled when m grows; probabilities are used
// un-normalised and the row is divided by l once at the end).  Workgroup y owns query tiles 8y .. 8y+7, one per
// wave.  Rows without any valid key (modules.py:227-244) are known up front from the sample's first valid key:
// they take 1/T on EVERY key of EVERY chunk, so a workgroup that holds such a row walks all chunks.
// attention_weights are not produced here (a caller that wants them at T > 256 gets the general kernels).
// =====================================================================================================
template <int NKS, bool SPLIT>
__global__ __launch_bounds__(512) void k_bf_fwd_long(cr_attn_desc d, BfGeom g) {
    constexpr int NDT = 2 * NKS;
    constexpr int JB = SPLIT ? 2 : NDT;
    constexpr int NKT = BF_CH / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* Kh = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Kl = Kh + (SPLIT ? BF_IMG : 0);
    __bf16* Vh = Kl + BF_IMG;
    __bf16* Vl = Vh + (SPLIT ? BF_IMG : 0);
    float* kb = reinterpret_cast<float*>(Vl + BF_IMG);   // [256] additive key bias of the staged chunk
    const int nw = blockDim.x >> 6;
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const DropCtx dc = drop_ctx(d.drop);
    // (the workgroups of the late, heavy query tiles get the low block indices: they are dispatched first and the light ones
    //  fill the tail of the launch)
    const int yq = (int)gridDim.y - 1 - (int)blockIdx.y;
    const int qt = yq * nw + wave;
    const bool have = qt < g.nkt;
    const int q0 = 16 * (have ? qt : 0), q = q0 + li;
    const int qc = min(q, T - 1);
    GFrag<NKS> qn;
    if (have) gfrag_issue<NKS>(qn, d.Q, d.ld, base_row + q0, hoff, T - q0, d.d, g.M);
    const float qv_ = d.q_valid[base_row + qc];
    const int id_ = d.dead_ids ? d.dead_ids[base_row + qc] : 1;
    const int fvk = first_valid_key(d.k_valid, base_row, T);
    const int kt_first = fvk >> 4;
    const bool is_dead = !have || q >= T || id_ == 0;
    const float qvq = (have && q < T) ? qv_ : 0.0f;
    const bool uniform = have && q < T && !is_dead && q < fvk;             // no valid key at or before q
    const bool any_uni = __any(uniform ? 1 : 0) != 0;
    const bool tile_dead = __all(is_dead ? 1 : 0) != 0;
    bf8 qh[NKS], ql[NKS];
    if (have) gfrag_finish<SPLIT, NKS>(qn, d.Q, d.ld, base_row + q0, hoff, T - q0, d.d, g.M, qh, ql);
    const float c2 = g.isd_log2e;
    const uint32_t ridx = attn_row_idx(d, head, n, q);
    const uint32_t xrow = (ridx + (uint32_t)(4 * lg)) * CR_PHI + dc.key;
    float m_run = -INFINITY, l_run = 0.0f;
    f32x4 acc[NDT];
#pragma unroll
    for (int jt = 0; jt < NDT; ++jt) acc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // chunks this workgroup walks: up to the chunk of its last query row; all of them when it may hold a uniform row
    const int blk_q0 = 16 * yq * nw;
    const int c_hi = (blk_q0 < fvk) ? g.nch - 1 : min(g.nch - 1, (blk_q0 + 16 * nw - 1) / BF_CH);
    for (int c = 0; c <= c_hi; ++c) {
        __syncthreads();
        {
            const int crow0 = c * BF_CH, t0 = threadIdx.x;
            const float kv0 = d.k_valid[base_row + min(crow0 + t0, T - 1)];
            stage_pair_bf<SPLIT, NKS>(Kh, Kl, d.K, d.ld, Vh, Vl, d.V, d.ld, base_row, crow0, BF_CH, T, hoff, d.d, g.M);
            if (t0 < BF_CH) kb[t0] = (crow0 + t0 < T && kv0 != 0.0f) ? 0.0f : -INFINITY;
        }
        __syncthreads();
        if (!have || tile_dead) continue;
        const int kt_c0 = c * NKT;
        // live score tiles of this chunk (global tile index in [kt_first, qt]); uniform rows need every tile below T16
        const int lo = max(kt_first, kt_c0) - kt_c0, hi = min(qt, kt_c0 + NKT - 1) - kt_c0;      // chunk-local, may be empty
        const int nt_c = min(NKT, g.nkt - kt_c0);                                                 // tiles of the chunk that exist
        if (lo > hi && !any_uni) continue;
        f32x4 st[NKT];
        float mx = -INFINITY;
        const f32x4 ninf = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kt = 0; kt < NKT; kt += 2) {
            const bool c0 = kt >= lo && kt <= hi;
            const bool c1 = kt + 1 >= lo && kt + 1 <= hi;
            f32x4 a0 = ninf, a1 = ninf;
            if (c0 || c1) {
                const int r0 = 16 * (c0 ? kt : kt + 1), r1 = 16 * (c1 ? kt + 1 : kt);
                bf8 k0h[NKS], k0l[NKS], k1h[NKS], k1l[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    k0h[ks] = row_frag<NKS>(Kh, r0, ks); k1h[ks] = row_frag<NKS>(Kh, r1, ks);
                    k0l[ks] = SPLIT ? row_frag<NKS>(Kl, r0, ks) : k0h[ks]; k1l[ks] = SPLIT ? row_frag<NKS>(Kl, r1, ks) : k1h[ks];
                }
                f32x4 x0 = (f32x4){0.f, 0.f, 0.f, 0.f}, x1 = x0;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    x0 = mma<SPLIT>(k0h[ks], k0l[ks], qh[ks], ql[ks], x0);
                    x1 = mma<SPLIT>(k1h[ks], k1l[ks], qh[ks], ql[ks], x1);
                }
                BF_SGB(0x100, (SPLIT ? 4 : 2) * NKS, 0);
                BF_SGB(0x008, (SPLIT ? 6 : 2) * NKS, 0);
                auto finish = [&](int ktl, f32x4 a) {
                    const float4 b4 = *reinterpret_cast<const float4*>(kb + 16 * ktl + 4 * lg);
                    a[0] = fmaf(a[0], c2, b4.x); a[1] = fmaf(a[1], c2, b4.y);
                    a[2] = fmaf(a[2], c2, b4.z); a[3] = fmaf(a[3], c2, b4.w);
                    if (kt_c0 + ktl == qt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[r] = (4 * lg + r <= li) ? a[r] : -INFINITY;
                    }
                    mx = fmaxf(fmaxf(mx, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
                    return a;
                };
                if (c0) a0 = finish(kt, c0 ? x0 : x1);
                if (c1) a1 = finish(kt + 1, x1);
            }
            st[kt] = a0;
            st[kt + 1] = a1;
        }
        mx = grp_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float off = (m_new == -INFINITY) ? 0.0f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(m_run - off);   // rescale of what is accumulated so far
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[kt][r] - off);    // exp2(-inf) = 0: masked entries and tiles not computed
                sum += p;
                float w = qvq;
                if (dc.on) w *= drop_factor_x(dc, xrow + (uint32_t)(16 * (kt_c0 + kt) + r) * CR_PHI);
                const float pu = (16 * (kt_c0 + kt) + 4 * lg + r < T) ? g.invT : 0.0f;              // uniform rows: 1/T on every key
                st[kt][r] = (uniform ? pu : p) * w;
            }
        sum = grp_sum(sum);
        l_run = l_run * alpha + sum;
        m_run = m_new;
        // accumulator rows are queries 4 lg + r, alpha lives on the lane of ITS query: one cross-lane read per row
        // (uniform rows keep their sum: their factor is 1)
        const float a_eff = uniform ? 1.0f : alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(a_eff, 4 * lg + r, 64);
#pragma unroll
            for (int jt = 0; jt < NDT; ++jt) acc[jt][r] *= ar;
        }
        const int p_lo = any_uni ? 0 : lo, p_end = any_uni ? nt_c : hi + 1;
#pragma unroll
        for (int kp = 0; kp < NKT / 2; ++kp) {
            const int k0 = 2 * kp, k1 = 2 * kp + 1;
            if (k1 >= p_lo && k0 < p_end) {
                float x[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { x[r] = st[k0][r]; x[4 + r] = st[k1][r]; }
                bf8 ph, pl;
                split8<SPLIT>(x, ph, pl);
                const int ra = 16 * k0, rb = 16 * k1;                        // all 16 tiles of a chunk are staged (zeros beyond T)
#pragma unroll
                for (int j0 = 0; j0 < NDT; j0 += JB) {
                    bf8 bh[JB], bl[JB];
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) {
                        bh[jt] = tr_frag<NKS>(Vh, ra, rb, j0 + jt);
                        bl[jt] = SPLIT ? tr_frag<NKS>(Vl, ra, rb, j0 + jt) : bh[jt];
                    }
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) acc[j0 + jt] = mma<SPLIT>(ph, pl, bh[jt], bl[jt], acc[j0 + jt]);
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * JB, 0);
                    BF_SGB(0x008, (SPLIT ? 3 : 1) * JB, 0);
                }
            }
        }
    }
    if (!have) return;
    float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
    if (is_dead) inv = 0.0f;
    if (d.row_stats && lg == 0 && q < T) {
        float* sp = d.row_stats + ((size_t)blockIdx.x * T + q) * 4;
        sp[0] = m_run; sp[1] = uniform ? 0.0f : inv; sp[2] = is_dead ? 2.0f : (uniform ? 1.0f : 0.0f); sp[3] = 0.0f;
    }
    const float fin = uniform ? 1.0f : inv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float fr = __shfl(fin, 4 * lg + r, 64);
        const int qq = q0 + 4 * lg + r;
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) {
            const int cidx = 16 * jt + li;
            if (qq < T && cidx < d.d) {
                const size_t row = (size_t)(base_row + qq);
                d.out[row * d.ldo + hoff + cidx] = acc[jt][r] * fr + d.residual[row * d.ldr + hoff + cidx];
            }
        }
    }
}

// =====================================================================================================
// backward, query-owner pass: dQ (and delta, when the caller did not supply it)
// =====================================================================================================
template <int NKS, bool SPLIT, bool MULTI, bool PAIRED>
__device__ __forceinline__ void bf_bwd_q_pass(const cr_attn_bwd_desc& bd, const BfGeom& g, float* delta_out) {
    constexpr int NDT = 2 * NKS;
    constexpr int JB = SPLIT ? 2 : NDT;                  // column tiles per batch of transposed reads (register budget)
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // FAST: images at a compile-time stride, operand reads = (per-lane base + pair offset) + immediate (see bf_bwd_k_pass)
    constexpr bool FAST = PAIRED && !MULTI;
    constexpr int FSTR = 256 * 32 * NKS;                  // elements between images (FAST): 256 rows
    constexpr int TILEB = 1024 * NKS;                     // bytes of a 16-row tile of an image
    __bf16* Kh = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Kl = Kh + (FAST ? FSTR : (SPLIT ? g.ch_rows * (32 * NKS) : 0));
    __bf16* Vh = Kl + (FAST ? FSTR : g.ch_rows * (32 * NKS));
    __bf16* Vl = Vh + (FAST ? FSTR : (SPLIT ? g.ch_rows * (32 * NKS) : 0));
    float* kb = reinterpret_cast<float*>(Vl + (FAST ? FSTR : g.ch_rows * (32 * NKS)));   // [ch_rows] (FAST: + 16, the absent tile of an odd count)
    const int nw = blockDim.x >> 6;
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int frk0 = 2 * img_off<NKS>(li, lg), frk1 = 2 * img_off<NKS>(li, lg + 4 * (NKS - 1));
    int ftr[4];
    {
        const int q_ = li >> 2, p_ = li & 3;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) ftr[jt] = 2 * (img_off<NKS>(4 * lg + q_, (2 * jt + (p_ >> 1)) & (4 * NKS - 1)) + 4 * (p_ & 1));
    }
    BT_TS(0); BT_TS(1);
    const DropCtx dc = drop_ctx(d.drop);
    TileSched sch = sched_init(nw, wave);
    constexpr bool multi = MULTI;                        // several 256-row chunks (T > 256): compiled as its own kernel
    auto stage = [&](int c) {
        const int crow0 = c * BF_CH;
        const int t0 = threadIdx.x;
        const float kv0 = d.k_valid[base_row + min(crow0 + t0, T - 1)];
        stage_pair_bf<SPLIT, NKS>(Kh, Kl, d.K, d.ld, Vh, Vl, d.V, d.ld, base_row, crow0, g.ch_rows, T, hoff, d.d, g.M);
        if (t0 < g.ch_rows) kb[t0] = (crow0 + t0 < T && kv0 != 0.0f) ? 0.0f : -INFINITY;
    };
    // Tiles of this wave.  Separate kernels: ONE tile (serpentine rank over the sample's workgroups, or block y's tile
    // `wave` when T > 256).  PAIRED (the fused kernel: this workgroup runs the whole pass of its sample): tiles w and
    // nkt-1-w, a heavy and a light one -- nkt + 1 tile pairs for every wave.
    const int rank = sched_rank(sch);
    unsigned tpk = 0x7FFFu;                              // up to three 5-bit tile numbers, 31 = none
    int ntile = 1;
    if (PAIRED) {
        tpk = g.qpk[wave & 7];                                            // dealt on the host (bf_deal_tiles), heaviest first
        ntile = ((tpk & 31u) != 31u) + (((tpk >> 5) & 31u) != 31u) + (((tpk >> 10) & 31u) != 31u);
    }
    const int yq = (int)gridDim.y - 1 - (int)blockIdx.y;                  // T > 256: heavy (late) query tiles in the workgroups dispatched first
    const int t_single = multi ? yq * nw + wave : rank;                   // separate kernels: ONE tile (up to 64 of them at T = 1024)
    auto tile_at = [&](int ti) {
        if (!PAIRED) return t_single;
        const int t = (int)((tpk >> (5 * ti)) & 31u);
        return t == 31 ? -1 : t;
    };
    // the first tile's fragments and row constants, requested ahead of the staging
    GFrag<NKS> qn, on, un, rn;                           // Q, dOut and (delta formed here) out, residual
    typedef float f4s __attribute__((ext_vector_type(4), aligned(4)));
    f4s st_n = (f4s){0.f, 0.f, 2.0f, 0.f};
    float qv_n = 0.0f, dl_n = 0.0f;
    auto issue_tile = [&](int qt) {
        const int q0n = 16 * qt;
        gfrag_issue<NKS>(qn, d.Q, d.ld, base_row + q0n, hoff, T - q0n, d.d, g.M);
        gfrag_issue<NKS>(on, bd.dout, bd.lddo, base_row + q0n, hoff, T - q0n, d.d, g.M);
        const int qc = min(q0n + li, T - 1);
        st_n = *reinterpret_cast<const f4s*>(d.row_stats + ((size_t)blockIdx.x * T + qc) * 4);
        qv_n = d.q_valid[base_row + qc];
        if (bd.delta) {
            dl_n = bd.delta[(size_t)blockIdx.x * T + qc];
        } else {
            gfrag_issue<NKS>(un, d.out, d.ldo, base_row + q0n, hoff, T - q0n, d.d, g.M);
            gfrag_issue<NKS>(rn, d.residual, d.ldr, base_row + q0n, hoff, T - q0n, d.d, g.M);
        }
    };
    if (!PAIRED && tile_at(0) >= 0 && tile_at(0) < g.nkt) issue_tile(tile_at(0));
    int kt_first = 0;
    if (!multi) {
        stage(0);
        if (FAST && (g.nkt & 1)) {                       // the absent second tile of the last pair: finite (zero) rows, masked keys
            const int t = threadIdx.x, im = t / (TILEB / 16), o16 = t % (TILEB / 16);
            if (im < 4) *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(Kh + im * FSTR) + (size_t)g.ch_rows * (64 * NKS) + 16 * o16) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < 16) kb[g.ch_rows + t] = -INFINITY;
        }
        BT_TS(2);
        __syncthreads();
        BT_TS(3);
        kt_first = first_valid_key_lds(kb, g.ch_rows, T) >> 4;
    } else {
        kt_first = first_valid_key(d.k_valid, base_row, T) >> 4;
    }
#pragma unroll 1
    for (int ti = 0; ti < (PAIRED ? ntile : 1); ++ti) {
        const int round = ti;
        const int qt = tile_at(ti);
        const bool have = (PAIRED || ntile > 0) && qt >= 0 && qt < g.nkt;
        const int q0 = 16 * (have ? qt : 0), q = q0 + li;
        if (PAIRED && have) issue_tile(qt);             // paired tiles are not prefetched: a fragment live across the loop spills
        // forward statistics of this lane's query row
        float mrow = 1e30f, inv = 0.0f;
        bf8 qh[NKS], ql[NKS], oh[NKS], ol[NKS];
        float delta = 0.0f, qvq = 0.0f;
        bool normal = false;
        if (have) {
            normal = q < T && st_n.z == 0.0f;
            if (normal) { mrow = st_n.x; inv = st_n.y; }
            qvq = q < T ? qv_n : 0.0f;
            gfrag_finish<SPLIT, NKS>(qn, d.Q, d.ld, base_row + q0, hoff, T - q0, d.d, g.M, qh, ql);
            if (bd.delta) {
                delta = dl_n;
                gfrag_finish<SPLIT, NKS>(on, bd.dout, bd.lddo, base_row + q0, hoff, T - q0, d.d, g.M, oh, ol);
            } else {
                // delta[q] = sum_c dO[q][c] (O[q][c] - residual[q][c])  ==  sum_k dA[q][k] A[q][k] (mask and dropout included)
                gfrag_mask<NKS>(on, bd.dout, bd.lddo, base_row + q0, hoff, T - q0, d.d, g.M);
                gfrag_mask<NKS>(un, d.out, d.ldo, base_row + q0, hoff, T - q0, d.d, g.M);
                gfrag_mask<NKS>(rn, d.residual, d.ldr, base_row + q0, hoff, T - q0, d.d, g.M);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) delta = fmaf(on.v[ks][j], un.v[ks][j] - rn.v[ks][j], delta);
                delta = grp_sum(delta);
                if (lg == 0 && q < T) delta_out[(size_t)blockIdx.x * T + q] = delta;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) split8<SPLIT>(on.v[ks], oh[ks], ol[ks]);
            }
        }
        const bool tile_live = __any(normal ? 1 : 0) != 0;              // uniform and dead rows carry no score gradient
        if (round == 0) BT_TS(4);
        const uint32_t ridx = attn_row_idx(d, head, n, q);
        const uint32_t xrow = (ridx + (uint32_t)(4 * lg)) * CR_PHI + dc.key;
        f32x4 dq[NDT];
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) dq[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // chunk range: causal -- the chunk of the workgroup's last query tile bounds the loop (workgroup-uniform)
        const int c_hi = multi ? min(g.nch - 1, (16 * (yq * nw + nw - 1) + 15) / BF_CH) : 0;
        for (int c = 0; c <= c_hi; ++c) {
            if (multi) {
                __syncthreads();
                stage(c);
                __syncthreads();
            }
            if (!have || !tile_live) continue;
            const int kt_c0 = c * (BF_CH / 16);                          // global index of the chunk's first key tile
            const int lo = max(kt_first, kt_c0), hi = min(qt, kt_c0 + g.ch_rows / 16 - 1);
            for (int kp = lo >> 1; 2 * kp <= hi; ++kp) {                 // pairs of key tiles (global indices 2kp, 2kp+1)
                const int k0 = 2 * kp, k1 = 2 * kp + 1;
                // chunk-local tiles (l1 clamped: zeros below; FAST: always k0 + 1 -- real or zeroed rows, coefficients 0)
                const int l0 = k0 - kt_c0, l1 = FAST ? k1 : (k1 <= hi ? k1 : k0) - kt_c0;
                f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
                const int po = 2 * TILEB * kp;
                auto rfF = [&](int base, int ks, int second, int lo_) {
                    return *reinterpret_cast<const bf8*>(smem_raw + (base + po + (ks ? frk1 : frk0)) + TILEB * second + 2 * FSTR * lo_);
                };
                auto trF = [&](int base, int jt, int lo_) {
                    const unsigned char* pa = smem_raw + (base + po + ftr[jt]) + 2 * FSTR * lo_;
                    const bf4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(pa));
                    const bf4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(pa + TILEB));
                    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                {
                    bf8 a0h[NKS], a0l[NKS], a1h[NKS], a1l[NKS];
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        if (FAST) {
                            a0h[ks] = rfF(0, ks, 0, 0); a1h[ks] = rfF(0, ks, 1, 0);
                            a0l[ks] = SPLIT ? rfF(0, ks, 0, 1) : a0h[ks]; a1l[ks] = SPLIT ? rfF(0, ks, 1, 1) : a1h[ks];
                        } else {
                            a0h[ks] = row_frag<NKS>(Kh, 16 * l0, ks); a1h[ks] = row_frag<NKS>(Kh, 16 * l1, ks);
                            a0l[ks] = SPLIT ? row_frag<NKS>(Kl, 16 * l0, ks) : a0h[ks]; a1l[ks] = SPLIT ? row_frag<NKS>(Kl, 16 * l1, ks) : a1h[ks];
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        s0 = mma<SPLIT>(a0h[ks], a0l[ks], qh[ks], ql[ks], s0);
                        s1 = mma<SPLIT>(a1h[ks], a1l[ks], qh[ks], ql[ks], s1);
                    }
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * NKS, 0);
                    BF_SGB(0x008, (SPLIT ? 6 : 2) * NKS, 0);
                }
                {
                    bf8 v0h[NKS], v0l[NKS], v1h[NKS], v1l[NKS];
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        if (FAST) {
                            v0h[ks] = rfF(4 * FSTR, ks, 0, 0); v1h[ks] = rfF(4 * FSTR, ks, 1, 0);
                            v0l[ks] = SPLIT ? rfF(4 * FSTR, ks, 0, 1) : v0h[ks]; v1l[ks] = SPLIT ? rfF(4 * FSTR, ks, 1, 1) : v1h[ks];
                        } else {
                            v0h[ks] = row_frag<NKS>(Vh, 16 * l0, ks); v1h[ks] = row_frag<NKS>(Vh, 16 * l1, ks);
                            v0l[ks] = SPLIT ? row_frag<NKS>(Vl, 16 * l0, ks) : v0h[ks]; v1l[ks] = SPLIT ? row_frag<NKS>(Vl, 16 * l1, ks) : v1h[ks];
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        p0 = mma<SPLIT>(v0h[ks], v0l[ks], oh[ks], ol[ks], p0);      // dA^T[key][q] = V dO^T
                        p1 = mma<SPLIT>(v1h[ks], v1l[ks], oh[ks], ol[ks], p1);
                    }
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * NKS, 0);
                    BF_SGB(0x008, (SPLIT ? 6 : 2) * NKS, 0);
                }
                // the dQ product's K operand (transposed reads): first batch requested before the element-wise phase that hides it
                bf8 bh[JB], bl[JB];
#pragma unroll
                for (int jt = 0; jt < JB; ++jt) {
                    if (FAST) {
                        bh[jt] = trF(0, jt, 0);
                        bl[jt] = SPLIT ? trF(0, jt, 1) : bh[jt];
                    } else {
                        bh[jt] = tr_frag<NKS>(Kh, 16 * l0, 16 * l1, jt);
                        bl[jt] = SPLIT ? tr_frag<NKS>(Kl, 16 * l0, 16 * l1, jt) : bh[jt];
                    }
                }
                float x[8];
                auto finish = [&](int kt, int lt, const f32x4& s, const f32x4& p, bool on_, int xo) {
                    const float4 b4 = *reinterpret_cast<const float4*>(kb + 16 * lt + 4 * lg);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = 16 * kt + 4 * lg + r;
                        const bool valid = on_ && key <= q && bb[r] == 0.0f;             // causal + key mask
                        const float e = __builtin_amdgcn_exp2f(fmaf(s[r], g.isd_log2e, -mrow)) * inv;
                        const float pn = valid ? e : 0.0f;
                        float w = qvq;
                        if (dc.on) w *= drop_factor_x(dc, xrow + (uint32_t)(16 * kt + r) * CR_PHI);
                        x[xo + r] = pn * (p[r] * w - delta) * g.isd;                     // dS / sqrt(d)
                    }
                };
                finish(k0, l0, s0, p0, k0 >= lo, 0);
                finish(k1, l1, s1, p1, k1 <= hi, 4);
                bf8 ah, al;
                split8<SPLIT>(x, ah, al);
#pragma unroll
                for (int jt = 0; jt < JB; ++jt) dq[jt] = mma<SPLIT>(ah, al, bh[jt], bl[jt], dq[jt]);    // dQ += dS K
#pragma unroll
                for (int j0 = JB; j0 < NDT; j0 += JB) {
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) {
                        if (FAST) {
                            bh[jt] = trF(0, j0 + jt, 0);
                            bl[jt] = SPLIT ? trF(0, j0 + jt, 1) : bh[jt];
                        } else {
                            bh[jt] = tr_frag<NKS>(Kh, 16 * l0, 16 * l1, j0 + jt);
                            bl[jt] = SPLIT ? tr_frag<NKS>(Kl, 16 * l0, 16 * l1, j0 + jt) : bh[jt];
                        }
                    }
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) dq[j0 + jt] = mma<SPLIT>(ah, al, bh[jt], bl[jt], dq[j0 + jt]);
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * JB, 0);
                    BF_SGB(0x008, (SPLIT ? 3 : 1) * JB, 0);
                }
            }
        }
        if (round == 0) BT_TS(5);
        if (have) {
#pragma unroll
            for (int jt = 0; jt < NDT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = q0 + 4 * lg + r, cidx = 16 * jt + li;
                    if (qq < T && cidx < d.d) bd.dQ[(size_t)(base_row + qq) * bd.ldg + hoff + cidx] = dq[jt][r];
                }
        }
        if (round == 0) BT_TS(6);
    }
}

// =====================================================================================================
// backward, key-owner pass: dK, dV of the wave's 16 keys, summed over queries in registers
// =====================================================================================================
template <int NKS, bool SPLIT, bool MULTI, bool PAIRED>
__device__ __forceinline__ void bf_bwd_k_pass(const cr_attn_bwd_desc& bd, const BfGeom& g, const float* delta_in) {
    constexpr int NDT = 2 * NKS;
    constexpr int JB = SPLIT ? 2 : NDT;                  // column tiles per batch of transposed reads (register budget)
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // FAST (the fused launch at head dims 33..64): the four images sit at a compile-time stride of 256 rows, so that inside the
    // pair loop an operand read is (per-lane base + pair offset) + immediate: 12 address adds per iteration instead of one per
    // read (the run-time image pointers and tile indices made every one of the 60 reads of an iteration compute its own
    // address: 99 of the 313 vector instructions of the loop)
    constexpr bool FAST = PAIRED && !MULTI;
    constexpr int FSTR = 256 * 32 * NKS;                  // elements between images (FAST): 256 rows
    constexpr int TILEB = 1024 * NKS;                     // bytes of a 16-row tile of an image
    __bf16* Qh = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Ql = Qh + (FAST ? FSTR : (SPLIT ? g.ch_rows * (32 * NKS) : 0));
    __bf16* Oh = Ql + (FAST ? FSTR : g.ch_rows * (32 * NKS));
    __bf16* Ol = Oh + (FAST ? FSTR : (SPLIT ? g.ch_rows * (32 * NKS) : 0));
    // per-row statistics of the staged query chunk, stored so that the inner loop is branch-free:
    //   A[q][key] = valid * exp2(s c - smx) * sinv + (key < T ? suni : 0); normal row: suni = 0; uniform row: sinv = 0,
    //   suni = 1/T; dead row: both 0 (smx = 1e30 wherever sinv = 0: the exponential is exactly 0, never inf * 0)
    float* smx = reinterpret_cast<float*>(Ol + (FAST ? FSTR : g.ch_rows * (32 * NKS)));          // [ch_rows] each
    float* sinv = smx + g.ch_rows;
    float* sdel = sinv + g.ch_rows;
    float* suni = sdel + g.ch_rows;
    float* sqv = suni + g.ch_rows;
    float* tile_flag = sqv + g.ch_rows;                                  // [ch_rows/16]: 0 nothing flows, 1 normal rows only, 2 has a uniform row
    const int nw = blockDim.x >> 6;
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    // FAST: per-lane byte offsets of the operand reads at tile 0 of an image (see img_off: the swizzle term (row & 6) does not
    // depend on the tile, so a tile adds 2048 bytes)
    const int frk0 = 2 * img_off<NKS>(li, lg), frk1 = 2 * img_off<NKS>(li, lg + 4 * (NKS - 1));
    int ftr[4];
    {
        const int q = li >> 2, p = li & 3;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) ftr[jt] = 2 * (img_off<NKS>(4 * lg + q, (2 * jt + (p >> 1)) & (4 * NKS - 1)) + 4 * (p & 1));
    }
    BT_TS(8);
    const DropCtx dc = drop_ctx(d.drop);
    TileSched sch = sched_init(nw, wave);
    constexpr bool multi = MULTI;                        // several 256-row chunks (T > 256): compiled as its own kernel
    typedef float f4s __attribute__((ext_vector_type(4), aligned(4)));
    auto stage = [&](int c) {
        const int crow0 = c * BF_CH;
        const int t = threadIdx.x, tq = crow0 + t;                       // ch_rows <= 256 < blockDim: one row per thread
        const int tc = min(tq, T - 1);
        const f4s st = *reinterpret_cast<const f4s*>(d.row_stats + ((size_t)blockIdx.x * T + tc) * 4);
        const float del = delta_in[(size_t)blockIdx.x * T + tc];
        const float qv = d.q_valid[base_row + tc];
        stage_pair_bf<SPLIT, NKS>(Qh, Ql, d.Q, d.ld, Oh, Ol, bd.dout, bd.lddo, base_row, crow0, g.ch_rows, T, hoff, d.d, g.M);
        if (t < g.ch_rows) {
            const float flag = tq < T ? st.z : 2.0f;
            const bool normal = flag == 0.0f;
            smx[t] = normal ? st.x : 1e30f;
            sinv[t] = normal ? st.y : 0.0f;
            sdel[t] = normal ? del : 0.0f;
            suni[t] = (flag == 1.0f) ? g.invT : 0.0f;
            sqv[t] = tq < T ? qv : 0.0f;
        }
    };
    auto stage_flags = [&]() {                                           // after a barrier: per query tile of the chunk
        // one lane per ROW, the 16 rows of a tile folded by ballots (a thread per tile read its 32 values one after the other:
        // 0.8 us between two barriers of the key-owner prologue)
        const int t = threadIdx.x;                                       // ch_rows <= 256 < blockDim: whole waves take part
        if ((t & ~63) < g.ch_rows) {
            const bool live = t < g.ch_rows;
            const unsigned long long bn = __ballot(live && sinv[live ? t : 0] != 0.0f);
            const unsigned long long bu = __ballot(live && suni[live ? t : 0] != 0.0f);
            if ((t & 15) == 0 && live) {
                const int sh = t & 48;                                   // this tile's 16 bits of the wave's ballot
                const bool anyu = ((bu >> sh) & 0xFFFFull) != 0, anyn = ((bn >> sh) & 0xFFFFull) != 0;
                tile_flag[t >> 4] = anyu ? 2.0f : (anyn ? 1.0f : 0.0f);
            }
        }
    };
    const int rank = sched_rank(sch);
    unsigned tpk = 0x7FFFu;                              // see bf_bwd_q_pass
    int ntile = 1;
    if (PAIRED) {
        tpk = g.kpk[wave & 7];
        ntile = ((tpk & 31u) != 31u) + (((tpk >> 5) & 31u) != 31u) + (((tpk >> 10) & 31u) != 31u);
    }
    const int t_single = multi ? (int)blockIdx.y * nw + wave : rank;      // key tile 0 meets every query tile: rank == kt
    auto tile_at = [&](int ti) {
        if (!PAIRED) return t_single;
        const int t = (int)((tpk >> (5 * ti)) & 31u);
        return t == 31 ? -1 : t;
    };
    GFrag<NKS> kn, vn;                                   // K / V fragments of the wave's next key tile, requested ahead
    float kvn = 0.0f;
    auto issue_tile = [&](int kt) {
        gfrag_issue<NKS>(kn, d.K, d.ld, base_row + 16 * kt, hoff, T - 16 * kt, d.d, g.M);
        gfrag_issue<NKS>(vn, d.V, d.ld, base_row + 16 * kt, hoff, T - 16 * kt, d.d, g.M);
        kvn = d.k_valid[base_row + min(16 * kt + li, T - 1)];
    };
    if (!PAIRED && tile_at(0) >= 0 && tile_at(0) < g.nkt) issue_tile(tile_at(0));
    int fvk = 0;
    if (!multi) {
        stage(0);
        if (FAST && (g.nkt & 1)) {
            // an odd tile count: the last pair's second tile does not exist; its rows are read all the same (with zero
            // coefficients) and must hold finite values: zero them (16 rows x 128 bytes x 4 images, 16 bytes per thread)
            const int t = threadIdx.x, im = t / (TILEB / 16), o16 = t % (TILEB / 16);
            if (im < 4) *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(Qh + im * FSTR) + (size_t)g.ch_rows * (64 * NKS) + 16 * o16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        BT_TS(9);
        __syncthreads();
        stage_flags();
        __syncthreads();
        BT_TS(10);
    } else {
        fvk = first_valid_key(d.k_valid, base_row, T);
    }
#pragma unroll 1
    for (int ti = 0; ti < (PAIRED ? ntile : 1); ++ti) {
        const int round = ti;
        const int kt = tile_at(ti);
        const bool have = (PAIRED || ntile > 0) && kt >= 0 && kt < g.nkt;
        const int key0 = 16 * (have ? kt : 0), key = key0 + li;
        const float key_in_T = (have && key < T) ? 1.0f : 0.0f;
        const uint32_t drop_base = attn_row_idx(d, head, n, 0) + (uint32_t)key;
        bf8 kh[NKS], kl[NKS], vh[NKS], vl[NKS];
        bool kvk = false;
        if (PAIRED && have) issue_tile(kt);             // paired tiles are not prefetched (see bf_bwd_q_pass)
        if (have) {
            gfrag_finish<SPLIT, NKS>(kn, d.K, d.ld, base_row + key0, hoff, T - key0, d.d, g.M, kh, kl);
            gfrag_finish<SPLIT, NKS>(vn, d.V, d.ld, base_row + key0, hoff, T - key0, d.d, g.M, vh, vl);
            kvk = key < T && kvn != 0.0f;
        }
        if (round == 0) BT_TS(11);
        const bool tile_has_key = __any(kvk ? 1 : 0) != 0;              // all-padding key tile: only uniform rows reach it
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) {
            dk[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // query chunks: from the workgroup's first key tile on; from 0 when rows without a valid key may exist (they see ALL keys)
        const int c_lo = multi ? ((fvk > 0) ? 0 : (16 * (int)blockIdx.y * nw) / BF_CH) : 0;
        for (int c = c_lo; c < g.nch; ++c) {
            if (multi) {
                __syncthreads();
                stage(c);
                __syncthreads();
                stage_flags();
                __syncthreads();
            }
            if (!have) continue;
            const int qt_c0 = c * (BF_CH / 16);
            const int ntile = g.ch_rows / 16;
            for (int qp = 0; 2 * qp < ntile; ++qp) {                     // pairs of query tiles (chunk-local 2qp, 2qp+1)
                const int l0 = 2 * qp, l1 = (2 * qp + 1 < ntile) ? 2 * qp + 1 : 2 * qp;
                const bool two = 2 * qp + 1 < ntile;
                auto wanted = [&](int lt) {
                    const float f = tile_flag[lt];
                    if (f == 0.0f) return false;                                 // nothing flows through dead query tiles
                    if (f == 2.0f) return true;                                  // uniform rows see every key
                    return (qt_c0 + lt >= kt) && tile_has_key;                   // causal / padding skip
                };
                const bool w0 = wanted(l0), w1 = two && wanted(l1);
                if (!w0 && !w1) continue;
                f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
                // FAST: byte addresses of this pair's tiles: (lane base + 4096 * qp); second tile + 2048, lo image + 32768
                const int po = 2 * TILEB * qp;
                auto rfF = [&](int base, int ks, int second, int lo) {
                    return *reinterpret_cast<const bf8*>(smem_raw + (base + po + (ks ? frk1 : frk0)) + TILEB * second + 2 * FSTR * lo);
                };
                auto trF = [&](int base, int jt, int lo) {
                    const unsigned char* pa = smem_raw + (base + po + ftr[jt]) + 2 * FSTR * lo;
                    const bf4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(pa));
                    const bf4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(pa + TILEB));
                    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                {
                    bf8 a0h[NKS], a0l[NKS], a1h[NKS], a1l[NKS];
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        if (FAST) {
                            a0h[ks] = rfF(0, ks, 0, 0); a1h[ks] = rfF(0, ks, 1, 0);
                            a0l[ks] = SPLIT ? rfF(0, ks, 0, 1) : a0h[ks]; a1l[ks] = SPLIT ? rfF(0, ks, 1, 1) : a1h[ks];
                        } else {
                            a0h[ks] = row_frag<NKS>(Qh, 16 * l0, ks); a1h[ks] = row_frag<NKS>(Qh, 16 * l1, ks);
                            a0l[ks] = SPLIT ? row_frag<NKS>(Ql, 16 * l0, ks) : a0h[ks]; a1l[ks] = SPLIT ? row_frag<NKS>(Ql, 16 * l1, ks) : a1h[ks];
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        s0 = mma<SPLIT>(a0h[ks], a0l[ks], kh[ks], kl[ks], s0);      // S[q][key]
                        s1 = mma<SPLIT>(a1h[ks], a1l[ks], kh[ks], kl[ks], s1);
                    }
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * NKS, 0);
                    BF_SGB(0x008, (SPLIT ? 6 : 2) * NKS, 0);
                }
                {
                    bf8 o0h[NKS], o0l[NKS], o1h[NKS], o1l[NKS];
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        if (FAST) {
                            o0h[ks] = rfF(4 * FSTR, ks, 0, 0); o1h[ks] = rfF(4 * FSTR, ks, 1, 0);
                            o0l[ks] = SPLIT ? rfF(4 * FSTR, ks, 0, 1) : o0h[ks]; o1l[ks] = SPLIT ? rfF(4 * FSTR, ks, 1, 1) : o1h[ks];
                        } else {
                            o0h[ks] = row_frag<NKS>(Oh, 16 * l0, ks); o1h[ks] = row_frag<NKS>(Oh, 16 * l1, ks);
                            o0l[ks] = SPLIT ? row_frag<NKS>(Ol, 16 * l0, ks) : o0h[ks]; o1l[ks] = SPLIT ? row_frag<NKS>(Ol, 16 * l1, ks) : o1h[ks];
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        p0 = mma<SPLIT>(o0h[ks], o0l[ks], vh[ks], vl[ks], p0);      // dA[q][key] = dO V^T
                        p1 = mma<SPLIT>(o1h[ks], o1l[ks], vh[ks], vl[ks], p1);
                    }
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * NKS, 0);
                    BF_SGB(0x008, (SPLIT ? 6 : 2) * NKS, 0);
                }
                // dOut columns for the dV product: first batch requested before the element-wise phase that hides it
                bf8 oth[JB], otl[JB];
#pragma unroll
                for (int jt = 0; jt < JB; ++jt) {
                    if (FAST) {
                        oth[jt] = trF(4 * FSTR, jt, 0);
                        otl[jt] = SPLIT ? trF(4 * FSTR, jt, 1) : oth[jt];
                    } else {
                        oth[jt] = tr_frag<NKS>(Oh, 16 * l0, 16 * l1, jt);
                        otl[jt] = SPLIT ? tr_frag<NKS>(Ol, 16 * l0, 16 * l1, jt) : oth[jt];
                    }
                }
                float xa[8], xd[8];
                auto finish = [&](int lt, const f32x4& s, const f32x4& p, bool on_, int xo) {
                    const int ql4 = 16 * lt + 4 * lg;                            // chunk-local index of this lane's 4 query rows
                    const int q4 = 16 * qt_c0 + ql4;
                    const float4 m4 = *reinterpret_cast<const float4*>(smx + ql4), i4 = *reinterpret_cast<const float4*>(sinv + ql4);
                    const float4 d4 = *reinterpret_cast<const float4*>(sdel + ql4), u4 = *reinterpret_cast<const float4*>(suni + ql4);
                    const float4 w4 = *reinterpret_cast<const float4*>(sqv + ql4);
                    const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w};
                    const float dd[4] = {d4.x, d4.y, d4.z, d4.w}, uu[4] = {u4.x, u4.y, u4.z, u4.w};
                    const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
                    const uint32_t x0 = (drop_base + (uint32_t)q4 * (uint32_t)T) * CR_PHI + dc.key;   // counter of attention_weights[(j*B+n), q4, key]
                    const uint32_t xT = (uint32_t)T * CR_PHI;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool valid = on_ && (key <= q4 + r) && kvk;         // causal + key mask
                        const float e = __builtin_amdgcn_exp2f(fmaf(s[r], g.isd_log2e, -mm[r])) * ii[r];
                        const float pn = valid ? e : 0.0f;
                        float w = ww[r];
                        if (dc.on) w *= drop_factor_x(dc, x0 + (uint32_t)r * xT);
                        xa[xo + r] = on_ ? (pn + key_in_T * uu[r]) * w : 0.0f;   // A after mask + dropout
                        xd[xo + r] = pn * (p[r] * w - dd[r]) * g.isd;            // dS / sqrt(d)
                    }
                };
                finish(l0, s0, p0, true, 0);
                finish(l1, s1, p1, two, 4);
                bf8 ah, al, dh, dl;
                split8<SPLIT>(xa, ah, al);
                split8<SPLIT>(xd, dh, dl);
#pragma unroll
                for (int jt = 0; jt < JB; ++jt) dv[jt] = mma<SPLIT>(ah, al, oth[jt], otl[jt], dv[jt]);    // dV += A^T dO
#pragma unroll
                for (int j0 = JB; j0 < NDT; j0 += JB) {
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) {
                        if (FAST) {
                            oth[jt] = trF(4 * FSTR, j0 + jt, 0);
                            otl[jt] = SPLIT ? trF(4 * FSTR, j0 + jt, 1) : oth[jt];
                        } else {
                            oth[jt] = tr_frag<NKS>(Oh, 16 * l0, 16 * l1, j0 + jt);
                            otl[jt] = SPLIT ? tr_frag<NKS>(Ol, 16 * l0, 16 * l1, j0 + jt) : oth[jt];
                        }
                    }
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) dv[j0 + jt] = mma<SPLIT>(ah, al, oth[jt], otl[jt], dv[j0 + jt]);
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * JB, 0);
                    BF_SGB(0x008, (SPLIT ? 3 : 1) * JB, 0);
                }
#pragma unroll
                for (int j0 = 0; j0 < NDT; j0 += JB) {
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) {
                        if (FAST) {
                            oth[jt] = trF(0, j0 + jt, 0);
                            otl[jt] = SPLIT ? trF(0, j0 + jt, 1) : oth[jt];
                        } else {
                            oth[jt] = tr_frag<NKS>(Qh, 16 * l0, 16 * l1, j0 + jt);
                            otl[jt] = SPLIT ? tr_frag<NKS>(Ql, 16 * l0, 16 * l1, j0 + jt) : oth[jt];
                        }
                    }
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) dk[j0 + jt] = mma<SPLIT>(dh, dl, oth[jt], otl[jt], dk[j0 + jt]);   // dK += dS^T Q
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * JB, 0);
                    BF_SGB(0x008, (SPLIT ? 3 : 1) * JB, 0);
                }
            }
        }
        if (round == 0) BT_TS(12);
        if (have) {
#pragma unroll
            for (int jt = 0; jt < NDT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = key0 + 4 * lg + r, cidx = 16 * jt + li;
                    if (kk < T && cidx < d.d) {
                        bd.dK[(size_t)(base_row + kk) * bd.ldg + hoff + cidx] = dk[jt][r];
                        bd.dV[(size_t)(base_row + kk) * bd.ldg + hoff + cidx] = dv[jt][r];
                    }
                }
        }
        if (round == 0) BT_TS(13);
    }
}

template <int NKS, bool SPLIT, bool MULTI>
__global__ __launch_bounds__(512) void k_bf_bwd_q(cr_attn_bwd_desc bd, BfGeom g, float* delta_out) {
    bf_bwd_q_pass<NKS, SPLIT, MULTI, false>(bd, g, delta_out);
    BT_TS(15);
}
template <int NKS, bool SPLIT, bool MULTI>
__global__ __launch_bounds__(512) void k_bf_bwd_k(cr_attn_bwd_desc bd, BfGeom g, const float* delta_in) {
    bf_bwd_k_pass<NKS, SPLIT, MULTI, false>(bd, g, delta_in);
    BT_TS(15);
}
// Both passes in ONE launch (T <= 256, delta supplied by the caller): workgroup y = 0 of a sample runs the whole
// query-owner pass, workgroup y = 1 the whole key-owner pass, side by side on two CUs.  Inside a pass the wave w owns
// tiles w and nkt-1-w (a heavy and a light one): nkt + 1 tile pairs for EVERY wave, where the separate kernels --
// one tile per wave, the two passes one after the other -- each end on the wave that owns the heaviest tile (13 pairs
// at T = 200 against an average of 7).  A first fused form (both passes in every workgroup, the K / V images replaced
// by the Q / dOut images in between) gained nothing: the barrier between the passes re-synchronises the waves at the
// heaviest tile of EACH pass (profiles/r02_*: 34.4 against 35.6 us).
template <int NKS, bool SPLIT>
__global__ __launch_bounds__(512) void k_bf_bwd_fused(cr_attn_bwd_desc bd, BfGeom g) {
    if (blockIdx.y == 0) bf_bwd_q_pass<NKS, SPLIT, false, true>(bd, g, nullptr);
    else bf_bwd_k_pass<NKS, SPLIT, false, true>(bd, g, bd.delta);
    BT_TS(15);
}

// =====================================================================================================
// host side
// =====================================================================================================
// Tiles of the fused launch's waves.  A query tile qt meets key tiles 0..qt, a key tile kt query tiles kt..nkt-1, two per loop
// iteration: ceil(count / 2) iterations.  A wave's loop is a serial chain (it issues an instruction every ~5 clocks whatever
// its SIMD neighbour does), so the launch ends with the wave that has the most iterations.  The first version paired tiles w
// and nkt - 1 - w on seven waves: 8 / 7 / 8 / 7 / 8 / 7 / 4 iterations at T = 200 and an idle eighth wave.  Here the tiles go
// heaviest first to the SIMD (waves w and w + 4 share one) with the fewest iterations so far, there to the wave with the fewer
// (at most three tiles per wave): 7 / 6 / 6 / 6 / 6 / 6 / 6 / 6.
static void bf_deal_tiles(int nkt, bool query_pass, unsigned (&pk)[8]) {
    int cost[32], order[32], load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < 8; ++w) pk[w] = 0x7FFFu;                          // three "none" entries
    if (nkt > 24) return;                                                  // (the fused launch takes T <= 256: nkt <= 16)
    for (int t = 0; t < nkt; ++t) {
        cost[t] = ((query_pass ? t + 1 : nkt - t) + 1) / 2;
        order[t] = t;
    }
    for (int i = 1; i < nkt; ++i)                                          // heaviest first (stable insertion sort)
        for (int j = i; j > 0 && cost[order[j]] > cost[order[j - 1]]; --j) { const int x = order[j]; order[j] = order[j - 1]; order[j - 1] = x; }
    for (int i = 0; i < nkt; ++i) {
        const int t = order[i];
        int best = -1;
        for (int s4 = 0; s4 < 4; ++s4) {
            if (cnt[s4] >= 3 && cnt[s4 + 4] >= 3) continue;
            if (best < 0 || load[s4] + load[s4 + 4] < load[best] + load[best + 4]) best = s4;
        }
        if (best < 0) return;                                              // more than 24 tiles: cannot happen (checked above)
        int w = best;
        if (cnt[w] >= 3 || (cnt[w + 4] < 3 && load[w + 4] < load[w])) w = best + 4;
        pk[w] = (pk[w] & ~(31u << (5 * cnt[w]))) | ((unsigned)t << (5 * cnt[w]));
        ++cnt[w];
        load[w] += cost[t];
    }
}

static int bf_geom(const cr_attn_desc* d, BfGeom* g) {
    g->T16 = (d->T + 15) / 16 * 16;
    g->nkt = g->T16 / 16;
    g->nch = (g->T16 + BF_CH - 1) / BF_CH;
    g->ch_rows = g->nch == 1 ? g->T16 : BF_CH;
    g->M = d->B * d->T;
    g->isd = (float)(1.0 / sqrt((double)d->d));
    g->isd_log2e = (float)(1.4426950408889634 / sqrt((double)d->d));
    g->invT = 1.0f / (float)d->T;
    g->ts = nullptr;
    bf_deal_tiles(g->nkt, true, g->qpk);
    bf_deal_tiles(g->nkt, false, g->kpk);
    return CR_OK;
}

bool cr_attn_bf_supported_fwd(const cr_attn_desc* d) {
    return d->d >= 8 && d->d <= 64 && (d->T <= 256 || (d->T <= 1024 && d->attn_weights == nullptr));
}
// (the backward reads the row statistics the bf16 forward saved: same shape conditions as the forward)
bool cr_attn_bf_supported_bwd(const cr_attn_desc* d) { return cr_attn_bf_supported_fwd(d) && d->row_stats != nullptr; }

// workgroups per (sample, head): every wave owns ONE 16-row tile (8 waves per workgroup)
static int bf_nsplit(const cr_attn_desc* d, const BfGeom& g) { (void)d; return (g.nkt + 7) / 8; }

template <int NKT, int NKS, bool SPLIT>
static int launch_bf_fwd(const cr_attn_desc* d, const BfGeom& g, hipStream_t s) {
    static cr_devmask attr_set = 0;
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_bf_fwd<NKT, NKS, SPLIT>), &attr_set);
    if (rc) return rc;
    const size_t lds = (size_t)g.T16 * (64 * NKS) * 2 * (SPLIT ? 2 : 1) + (size_t)g.T16 * 4;
    BfGeom gg = g;
    if (g_attn_ts_which == 4) gg.ts = g_attn_ts;
    hipLaunchKernelGGL((k_bf_fwd<NKT, NKS, SPLIT>), dim3(d->B * d->H, bf_nsplit(d, g)), dim3(512), lds, s, *d, gg);
    return cr_check_launch("cr_attn_fwd(bf16)");
}

template <int NKT>
static int dispatch_bf_fwd(const cr_attn_desc* d, const BfGeom& g, hipStream_t s) {
    const bool split = d->precision == CR_PREC_BF16X3;
    if (d->d <= 32) return split ? launch_bf_fwd<NKT, 1, true>(d, g, s) : launch_bf_fwd<NKT, 1, false>(d, g, s);
    return split ? launch_bf_fwd<NKT, 2, true>(d, g, s) : launch_bf_fwd<NKT, 2, false>(d, g, s);
}

template <int NKS, bool SPLIT>
static int launch_bf_fwd_long(const cr_attn_desc* d, const BfGeom& g, hipStream_t s) {
    static cr_devmask attr_set = 0;
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_bf_fwd_long<NKS, SPLIT>), &attr_set);
    if (rc) return rc;
    const size_t lds = (size_t)BF_CH * (64 * NKS) * 2 * (SPLIT ? 2 : 1) + (size_t)BF_CH * 4;
    hipLaunchKernelGGL((k_bf_fwd_long<NKS, SPLIT>), dim3(d->B * d->H, bf_nsplit(d, g)), dim3(512), lds, s, *d, g);
    return cr_check_launch("cr_attn_fwd(bf16, long)");
}

int cr_attn_bf_fwd_launch(const cr_attn_desc* d, hipStream_t s) {
    BfGeom g;
    bf_geom(d, &g);
    if (g.nch > 1) {
        const bool split = d->precision == CR_PREC_BF16X3;
        if (d->d <= 32) return split ? launch_bf_fwd_long<1, true>(d, g, s) : launch_bf_fwd_long<1, false>(d, g, s);
        return split ? launch_bf_fwd_long<2, true>(d, g, s) : launch_bf_fwd_long<2, false>(d, g, s);
    }
    if (g.nkt <= 4) return dispatch_bf_fwd<4>(d, g, s);
    if (g.nkt <= 13) return dispatch_bf_fwd<13>(d, g, s);
    return dispatch_bf_fwd<16>(d, g, s);
}

static const bool g_bf_two_kernels = getenv("CASTREC_BF_TWO_KERNELS") != nullptr;   // debugging: never fuse the backward passes

template <int NKS, bool SPLIT, bool MULTI>
static int launch_bf_bwd(const cr_attn_bwd_desc* bd, const BfGeom& g, hipStream_t s) {
    static cr_devmask attr_q = 0, attr_k = 0;
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_bf_bwd_q<NKS, SPLIT, MULTI>), &attr_q);
    if (rc) return rc;
    rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_bf_bwd_k<NKS, SPLIT, MULTI>), &attr_k);
    if (rc) return rc;
    const cr_attn_desc* d = &bd->f;
    const size_t img = (size_t)g.ch_rows * (64 * NKS) * 2 * (SPLIT ? 2 : 1);
    const dim3 grid(d->B * d->H, bf_nsplit(d, g));
    float* dws = bd->stats;                                              // delta workspace [H*B*T] when the caller gave none
    if (!MULTI && bd->delta && !g_bf_two_kernels) {
        static cr_devmask attr_f = 0;
        rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_bf_bwd_fused<NKS, SPLIT>), &attr_f);
        if (rc) return rc;
        BfGeom gf = g;
        if (g_attn_ts_which == 5) gf.ts = g_attn_ts;
        // (the key-owner workgroups at NKS == 2 keep their four images at a fixed stride of 256 rows)
        const size_t img_k = (size_t)4 * 256 * (32 * NKS) * 2;               // four images at the fixed stride of 256 rows
        hipLaunchKernelGGL((k_bf_bwd_fused<NKS, SPLIT>), dim3(d->B * d->H, 2), dim3(512), (img_k > img ? img_k : img) + (size_t)g.ch_rows * 4 * 5 + (size_t)(g.ch_rows / 16) * 4 + 64, s, *bd, gf);
        return cr_check_launch("cr_attn_bwd(bf16, fused)");
    }
    BfGeom gq = g, gk = g;
    if (g_attn_ts_which == 5) gq.ts = g_attn_ts;
    if (g_attn_ts_which == 6) gk.ts = g_attn_ts;
    hipLaunchKernelGGL((k_bf_bwd_q<NKS, SPLIT, MULTI>), grid, dim3(512), img + (size_t)g.ch_rows * 4, s, *bd, gq, dws);
    rc = cr_check_launch("cr_attn_bwd(bf16, q)");
    if (rc) return rc;
    hipLaunchKernelGGL((k_bf_bwd_k<NKS, SPLIT, MULTI>), grid, dim3(512), img + (size_t)g.ch_rows * 4 * 5 + (size_t)(g.ch_rows / 16) * 4, s,
                       *bd, gk, bd->delta ? bd->delta : dws);
    return cr_check_launch("cr_attn_bwd(bf16, k)");
}

int cr_attn_bf_bwd_launch(const cr_attn_bwd_desc* bd, hipStream_t s) {
    const cr_attn_desc* d = &bd->f;
    BfGeom g;
    bf_geom(d, &g);
    const bool split = d->precision == CR_PREC_BF16X3;
    if (!bd->delta) CR_REQUIRE(d->out && d->residual, "cr_attn_bwd(bf16): out / residual needed to form delta");
    const bool multi = g.nch > 1;
    if (d->d <= 32) {
        if (multi) return split ? launch_bf_bwd<1, true, true>(bd, g, s) : launch_bf_bwd<1, false, true>(bd, g, s);
        return split ? launch_bf_bwd<1, true, false>(bd, g, s) : launch_bf_bwd<1, false, false>(bd, g, s);
    }
    if (multi) return split ? launch_bf_bwd<2, true, true>(bd, g, s) : launch_bf_bwd<2, false, true>(bd, g, s);
    return split ? launch_bf_bwd<2, true, false>(bd, g, s) : launch_bf_bwd<2, false, false>(bd, g, s);
}
