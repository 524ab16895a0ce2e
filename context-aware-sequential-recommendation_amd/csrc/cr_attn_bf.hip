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
#include "cr_attn_common.hpp"

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf4 lds_bf4;

#define BF_CH 256                 // rows of one LDS chunk (K/V rows in the query-owner kernels, Q/dOut rows in the key-owner one)
#define BF_IMG (BF_CH * 64)       // bf16 elements of one image

struct BfGeom {
    int T16, nkt;                 // padded T, 16-row tiles
    int nch, ch_rows;             // chunks, rows per chunk (T16 when nch == 1, else 256)
    int M;                        // rows of the operand matrices (B * T): the vector loads may run into the NEXT row, never past the last
    float isd, isd_log2e, invT;
};

// element offset of 16-byte chunk `ch` (0..7) of row `row`
__device__ __forceinline__ int img_off(int row, int ch) { return row * 64 + ((ch ^ (row & 6)) << 3); }

// A / B operand with k = head dim: row `row0 + li`, columns 32 ks + 8 lg .. + 7
__device__ __forceinline__ bf8 row_frag(const __bf16* img, int row0, int ks) {
    const int lane = threadIdx.x & 63;
    return *reinterpret_cast<const bf8*>(img + img_off(row0 + (lane & 15), (lane >> 4) + 4 * ks));
}

// B operand with k = row: k index 8 lg + j <-> row (j < 4 ? ra : rb) + 4 lg + (j & 3), output column 16 jt + li
__device__ __forceinline__ bf8 tr_frag(const __bf16* img, int ra, int rb, int jt) {
    const int lane = threadIdx.x & 63, lg = lane >> 4, idx = lane & 15, q = idx >> 2, p = idx & 3;
    const int ch = 2 * jt + (p >> 1), sub = 4 * (p & 1);
    const bf4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + img_off(ra + 4 * lg + q, ch) + sub));
    const bf4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + img_off(rb + 4 * lg + q, ch) + sub));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool SPLIT>
__device__ __forceinline__ f32x4 mma(const bf8& ah, const bf8& al, const bf8& bh, const bf8& bl, f32x4 c) {
    if (SPLIT) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);      // small terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}

template <bool SPLIT>
__device__ __forceinline__ void split8(const float (&x)[8], bf8& hi, bf8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        if (SPLIT) lo[j] = (__bf16)(x[j] - (float)h);
    }
}

// 8 consecutive floats of a row, columns c .. c+7 of a d-column head block.  A chunk that crosses column d is
// read whole (it runs into the next row: inside the matrix for every row but the last) and masked; the last row
// of the matrix takes clamped dword loads.  Chunks beyond d are not read.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void load8(const float* p, int c, int d, bool row_ok, bool not_last, float (&v)[8]) {
    if (c < d) {
        if (c + 8 <= d || not_last) {
            const f4u a = *reinterpret_cast<const f4u*>(p + c), b = *reinterpret_cast<const f4u*>(p + c + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[c + j < d ? c + j : d - 1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (row_ok && c + j < d) ? v[j] : 0.0f;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    }
}

// Register fragment of the wave's own 16-row tile straight from global memory (operand with k = head dim):
// lane (li, lg) holds row row0 + li, columns 32 ks + 8 lg + j.
template <bool SPLIT, int NKS>
__device__ __forceinline__ void gfrag(const float* src, int ld, int grow0, int hoff, int nvalid, int d, int M,
                                      bf8 (&hi)[NKS], bf8 (&lo)[NKS]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const bool rok = li < nvalid;
    const int grow = grow0 + (rok ? li : 0);
    const float* p = src + (size_t)grow * ld + hoff;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        float v[8];
        load8(p, 32 * ks + 8 * lg, d, rok, grow < M - 1, v);
        split8<SPLIT>(v, hi[ks], lo[ks]);
    }
}

// Stage rows [crow0, crow0 + nrows) of two [T, d] head blocks into their LDS images (chunk-relative rows).
// An item is (row, 16-byte chunk); the 2 x U loads of a batch are issued before the first conversion.
template <bool SPLIT, int NKS>
__device__ __forceinline__ void stage_pair_bf(__bf16* ah, __bf16* al, const float* srcA, int ldA, __bf16* bh, __bf16* bl,
                                              const float* srcB, int ldB, int base_row, int crow0, int nrows, int T,
                                              int hoff, int d, int M) {
    constexpr int CPR = 4 * NKS;                         // chunks per row that MFMAs read (columns < 32 NKS)
    const int total = nrows * CPR;
    constexpr int U = 2;
    for (int i0 = threadIdx.x; i0 < total; i0 += blockDim.x * U) {
        float va[U][8], vb[U][8];
        int rr[U], cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int item = min(i0 + u * (int)blockDim.x, total - 1);
            const int r = item / CPR, ch = item - r * CPR;
            rr[u] = r; cc[u] = ch;
            const int t = crow0 + r;
            const bool rok = t < T;
            const int grow = base_row + (rok ? t : 0);
            load8(srcA + (size_t)grow * ldA + hoff, 8 * ch, d, rok, grow < M - 1, va[u]);
            load8(srcB + (size_t)grow * ldB + hoff, 8 * ch, d, rok, grow < M - 1, vb[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * (int)blockDim.x < total) {
                bf8 h, l;
                const int o = img_off(rr[u], cc[u]);
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

// tile owned by a wave in a round: one chunk -> serpentine deal over the sample's workgroups (cr_attn_common.hpp);
// several chunks -> workgroup y owns the 8 consecutive tiles 8y .. 8y+7 (their causal extents are alike, and the
// chunk loop with its barriers is workgroup-wide)
__device__ __forceinline__ int rounds_of(const BfGeom& g, int nw) {
    return g.nch > 1 ? 1 : (g.nkt + (int)gridDim.y * nw - 1) / ((int)gridDim.y * nw);
}

// =====================================================================================================
// forward, T <= 256 (one chunk): the whole score row block of a query tile lives in registers
// =====================================================================================================
template <int NKT, int NKS, bool SPLIT>
__global__ __launch_bounds__(512) void k_bf_fwd(cr_attn_desc d, BfGeom g) {
    constexpr int NDT = 2 * NKS;                         // 16-column output tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* Kh = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Kl = Kh + (SPLIT ? g.T16 * 64 : 0);
    __bf16* Vh = Kl + g.T16 * 64;
    __bf16* Vl = Vh + (SPLIT ? g.T16 * 64 : 0);
    float* kb = reinterpret_cast<float*>(Vl + g.T16 * 64);   // [T16] additive key bias
    const int nw = blockDim.x >> 6;
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const DropCtx dc = drop_ctx(d.drop);
    TileSched sch = sched_init(nw, wave);
    int qi = sched_rank(sch);
    const int fvk = first_valid_key(d.k_valid, base_row, T);
    stage_pair_bf<SPLIT, NKS>(Kh, Kl, d.K, d.ld, Vh, Vl, d.V, d.ld, base_row, 0, g.T16, T, hoff, d.d, g.M);
    for (int t = threadIdx.x; t < g.T16; t += blockDim.x)
        kb[t] = (t < T && d.k_valid[base_row + t] != 0.0f) ? 0.0f : -INFINITY;
    __syncthreads();
    const int kt_first = min(fvk >> 4, NKT - 1);         // tiles below hold no valid key: probabilities exactly 0
    const float c2 = g.isd_log2e;
    for (; qi < g.nkt; qi = sched_next(sch)) {
        const int qt = g.nkt - 1 - qi;                   // heaviest tiles first
        const int q0 = 16 * qt, q = q0 + li;
        const int qc = q < T ? q : T - 1;
        const bool is_dead = q >= T || (d.dead_ids && d.dead_ids[base_row + qc] == 0);
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
        bf8 qh[NKS], ql[NKS];
        gfrag<SPLIT, NKS>(d.Q, d.ld, base_row + q0, hoff, T - q0, d.d, g.M, qh, ql);
        const float qvq = d.q_valid[base_row + qc] * (q < T ? 1.0f : 0.0f);
        // ---- scores St[key][query] (modules.py:216-241), kept for the whole row block
        f32x4 st[NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 acc = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (kt >= kt_first && kt <= qt) {            // wave-uniform
                acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const bf8 ah = row_frag(Kh, 16 * kt, ks);
                    const bf8 al = SPLIT ? row_frag(Kl, 16 * kt, ks) : ah;
                    acc = mma<SPLIT>(ah, al, qh[ks], ql[ks], acc);
                }
                const float4 b4 = *reinterpret_cast<const float4*>(kb + 16 * kt + 4 * lg);   // key mask (modules.py:222-229)
                acc[0] = fmaf(acc[0], c2, b4.x); acc[1] = fmaf(acc[1], c2, b4.y);
                acc[2] = fmaf(acc[2], c2, b4.z); acc[3] = fmaf(acc[3], c2, b4.w);
                if (kt == qt) {                          // causal mask on the diagonal tile (modules.py:232-241)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = (4 * lg + r <= li) ? acc[r] : -INFINITY;
                }
                mx = fmaxf(fmaxf(mx, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
            }
            st[kt] = acc;
        }
        mx = grp_max(mx);
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
        if (any_uni) {                                   // rare: a row with no valid key at all (modules.py:227-244)
            const float uni = uniform ? g.invT : 0.0f;
            const float sc = uniform ? 0.0f : inv;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][r] = st[kt][r] * sc + ((16 * kt + 4 * lg + r < T) ? uni : 0.0f);
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) st[kt] *= inv;
        }
        if (d.row_stats && lg == 0 && q < T) {           // for the backward kernels
            float* sp = d.row_stats + ((size_t)blockIdx.x * T + q) * 4;
            sp[0] = mx; sp[1] = inv; sp[2] = is_dead ? 2.0f : (uniform ? 1.0f : 0.0f); sp[3] = 0.0f;
        }
        // ---- query mask, dropout (modules.py:248-257)
        const uint32_t ridx = attn_row_idx(d, head, n, q);
        const uint32_t xrow = (ridx + (uint32_t)(4 * lg)) * CR_PHI + dc.key;
        if (dc.on) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][r] *= qvq * drop_factor_x(dc, xrow + (uint32_t)(16 * kt + r) * CR_PHI);
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) st[kt] *= qvq;
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
        // ---- out = A V + residual (modules.py:262-269): two key tiles per k-step, V through transposed reads
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
                for (int jt = 0; jt < NDT; ++jt) {
                    const bf8 bh = tr_frag(Vh, ra, rb, jt);
                    const bf8 bl = SPLIT ? tr_frag(Vl, ra, rb, jt) : bh;
                    acc[jt] = mma<SPLIT>(ph, pl, bh, bl, acc[jt]);
                }
            }
        }
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + 4 * lg + r, c = 16 * jt + li;
                if (qq < T && c < d.d) d.out[(size_t)(base_row + qq) * d.ldo + hoff + c] = acc[jt][r] + resid[jt][r];
            }
    }
}

// =====================================================================================================
// backward, query-owner pass: dQ (and delta, when the caller did not supply it)
// =====================================================================================================
template <int NKS, bool SPLIT>
__global__ __launch_bounds__(512) void k_bf_bwd_q(cr_attn_bwd_desc bd, BfGeom g, float* delta_out) {
    constexpr int NDT = 2 * NKS;
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* Kh = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Kl = Kh + (SPLIT ? g.ch_rows * 64 : 0);
    __bf16* Vh = Kl + g.ch_rows * 64;
    __bf16* Vl = Vh + (SPLIT ? g.ch_rows * 64 : 0);
    float* kb = reinterpret_cast<float*>(Vl + g.ch_rows * 64);   // [ch_rows]
    const int nw = blockDim.x >> 6;
    const int head = blockIdx.x / d.B, n = blockIdx.x % d.B;
    const int base_row = n * d.T, hoff = head * d.d;
    const int T = d.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const DropCtx dc = drop_ctx(d.drop);
    TileSched sch = sched_init(nw, wave);
    const int fvk = first_valid_key(d.k_valid, base_row, T);
    const int kt_first = fvk >> 4;
    const bool multi = g.nch > 1;
    auto stage = [&](int c) {
        const int crow0 = c * BF_CH;
        stage_pair_bf<SPLIT, NKS>(Kh, Kl, d.K, d.ld, Vh, Vl, d.V, d.ld, base_row, crow0, g.ch_rows, T, hoff, d.d, g.M);
        for (int t = threadIdx.x; t < g.ch_rows; t += blockDim.x)
            kb[t] = (crow0 + t < T && d.k_valid[base_row + crow0 + t] != 0.0f) ? 0.0f : -INFINITY;
    };
    if (!multi) {
        stage(0);
        __syncthreads();
    }
    const int nrounds = rounds_of(g, nw);
    int rank = sched_rank(sch);
    for (int round = 0; round < nrounds; ++round, rank = sched_next(sch)) {
        const int qt = multi ? (int)blockIdx.y * nw + wave : g.nkt - 1 - rank;
        const bool have = qt >= 0 && qt < g.nkt;
        const int q0 = 16 * (have ? qt : 0), q = q0 + li;
        const int qc = q < T ? q : T - 1;
        // forward statistics of this lane's query row
        float mrow = 1e30f, inv = 0.0f, flag = 2.0f;
        if (have && q < T) {
            const float* sp = d.row_stats + ((size_t)blockIdx.x * T + q) * 4;
            flag = sp[2];
            if (flag == 0.0f) { mrow = sp[0]; inv = sp[1]; }
        }
        const bool tile_live = __any(flag == 0.0f ? 1 : 0) != 0;      // uniform and dead rows carry no score gradient
        bf8 qh[NKS], ql[NKS], oh[NKS], ol[NKS];
        float delta = 0.0f;
        if (have) {
            gfrag<SPLIT, NKS>(d.Q, d.ld, base_row + q0, hoff, T - q0, d.d, g.M, qh, ql);
            gfrag<SPLIT, NKS>(bd.dout, bd.lddo, base_row + q0, hoff, T - q0, d.d, g.M, oh, ol);
            if (bd.delta) {
                delta = bd.delta[(size_t)blockIdx.x * T + qc];
            } else {
                // delta[q] = sum_c dO[q][c] (O[q][c] - residual[q][c])  ==  sum_k dA[q][k] A[q][k] (mask and dropout included)
                const bool rok = q < T;
                const size_t row = (size_t)(base_row + qc);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int c = 32 * ks + 8 * lg + j;
                        const int cc = c < d.d ? c : 0;
                        const float go = bd.dout[row * bd.lddo + hoff + cc];
                        const float oo = d.out[row * d.ldo + hoff + cc] - d.residual[row * d.ldr + hoff + cc];
                        delta += (rok && c < d.d) ? go * oo : 0.0f;
                    }
                delta = grp_sum(delta);
                if (lg == 0 && q < T) delta_out[(size_t)blockIdx.x * T + q] = delta;
            }
        }
        const float qvq = have ? d.q_valid[base_row + qc] * (q < T ? 1.0f : 0.0f) : 0.0f;
        const uint32_t ridx = attn_row_idx(d, head, n, q);
        const uint32_t xrow = (ridx + (uint32_t)(4 * lg)) * CR_PHI + dc.key;
        f32x4 dq[NDT];
#pragma unroll
        for (int jt = 0; jt < NDT; ++jt) dq[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // chunk range: causal -- the chunk of the workgroup's last query tile bounds the loop (workgroup-uniform)
        const int c_hi = multi ? min(g.nch - 1, (16 * ((int)blockIdx.y * nw + nw - 1) + 15) / BF_CH) : 0;
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
                const int l0 = k0 - kt_c0, l1 = (k1 <= hi ? k1 : k0) - kt_c0;   // chunk-local tiles (l1 clamped: zeros below)
                f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const bf8 a0h = row_frag(Kh, 16 * l0, ks), a1h = row_frag(Kh, 16 * l1, ks);
                    const bf8 a0l = SPLIT ? row_frag(Kl, 16 * l0, ks) : a0h, a1l = SPLIT ? row_frag(Kl, 16 * l1, ks) : a1h;
                    s0 = mma<SPLIT>(a0h, a0l, qh[ks], ql[ks], s0);
                    s1 = mma<SPLIT>(a1h, a1l, qh[ks], ql[ks], s1);
                    const bf8 v0h = row_frag(Vh, 16 * l0, ks), v1h = row_frag(Vh, 16 * l1, ks);
                    const bf8 v0l = SPLIT ? row_frag(Vl, 16 * l0, ks) : v0h, v1l = SPLIT ? row_frag(Vl, 16 * l1, ks) : v1h;
                    p0 = mma<SPLIT>(v0h, v0l, oh[ks], ol[ks], p0);      // dA^T[key][q] = V dO^T
                    p1 = mma<SPLIT>(v1h, v1l, oh[ks], ol[ks], p1);
                }
                float x[8];
                auto finish = [&](int kt, int lt, const f32x4& s, const f32x4& p, bool on, int xo) {
                    const float4 b4 = *reinterpret_cast<const float4*>(kb + 16 * lt + 4 * lg);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = 16 * kt + 4 * lg + r;
                        const bool valid = on && key <= q && bb[r] == 0.0f;              // causal + key mask
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
                for (int jt = 0; jt < NDT; ++jt) {
                    const bf8 bh = tr_frag(Kh, 16 * l0, 16 * l1, jt);
                    const bf8 bl = SPLIT ? tr_frag(Kl, 16 * l0, 16 * l1, jt) : bh;
                    dq[jt] = mma<SPLIT>(ah, al, bh, bl, dq[jt]);                         // dQ += dS K
                }
            }
        }
        if (have) {
#pragma unroll
            for (int jt = 0; jt < NDT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = q0 + 4 * lg + r, cidx = 16 * jt + li;
                    if (qq < T && cidx < d.d) bd.dQ[(size_t)(base_row + qq) * bd.ldg + hoff + cidx] = dq[jt][r];
                }
        }
    }
}

// =====================================================================================================
// backward, key-owner pass: dK, dV of the wave's 16 keys, summed over queries in registers
// =====================================================================================================
template <int NKS, bool SPLIT>
__global__ __launch_bounds__(512) void k_bf_bwd_k(cr_attn_bwd_desc bd, BfGeom g, const float* delta_in) {
    constexpr int NDT = 2 * NKS;
    const cr_attn_desc& d = bd.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* Qh = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Ql = Qh + (SPLIT ? g.ch_rows * 64 : 0);
    __bf16* Oh = Ql + g.ch_rows * 64;
    __bf16* Ol = Oh + (SPLIT ? g.ch_rows * 64 : 0);
    // per-row statistics of the staged query chunk, stored so that the inner loop is branch-free:
    //   A[q][key] = valid * exp2(s c - smx) * sinv + (key < T ? suni : 0); normal row: suni = 0; uniform row: sinv = 0,
    //   suni = 1/T; dead row: both 0 (smx = 1e30 wherever sinv = 0: the exponential is exactly 0, never inf * 0)
    float* smx = reinterpret_cast<float*>(Ol + g.ch_rows * 64);          // [ch_rows] each
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
    const DropCtx dc = drop_ctx(d.drop);
    TileSched sch = sched_init(nw, wave);
    const bool multi = g.nch > 1;
    const int fvk = first_valid_key(d.k_valid, base_row, T);
    auto stage = [&](int c) {
        const int crow0 = c * BF_CH;
        stage_pair_bf<SPLIT, NKS>(Qh, Ql, d.Q, d.ld, Oh, Ol, bd.dout, bd.lddo, base_row, crow0, g.ch_rows, T, hoff, d.d, g.M);
        for (int t = threadIdx.x; t < g.ch_rows; t += blockDim.x) {
            const int tq = crow0 + t;
            float flag = 2.0f, mx_ = 0.0f, inv_ = 0.0f, del_ = 0.0f, qv_ = 0.0f;
            if (tq < T) {
                const float* sp = d.row_stats + ((size_t)blockIdx.x * T + tq) * 4;
                mx_ = sp[0]; inv_ = sp[1]; flag = sp[2];
                del_ = delta_in[(size_t)blockIdx.x * T + tq];
                qv_ = d.q_valid[base_row + tq];
            }
            const bool normal = flag == 0.0f;
            smx[t] = normal ? mx_ : 1e30f;
            sinv[t] = normal ? inv_ : 0.0f;
            sdel[t] = normal ? del_ : 0.0f;
            suni[t] = (flag == 1.0f) ? g.invT : 0.0f;
            sqv[t] = qv_;
        }
    };
    auto stage_flags = [&]() {                                           // after a barrier: per query tile of the chunk
        for (int t = threadIdx.x; t < g.ch_rows / 16; t += blockDim.x) {
            float f = 0.0f;
            for (int i = 0; i < 16; ++i) {
                if (sinv[16 * t + i] != 0.0f && f < 1.0f) f = 1.0f;
                if (suni[16 * t + i] != 0.0f) f = 2.0f;
            }
            tile_flag[t] = f;
        }
    };
    if (!multi) {
        stage(0);
        __syncthreads();
        stage_flags();
        __syncthreads();
    }
    const int nrounds = rounds_of(g, nw);
    int rank = sched_rank(sch);
    for (int round = 0; round < nrounds; ++round, rank = sched_next(sch)) {
        const int kt = multi ? (int)blockIdx.y * nw + wave : rank;      // key tile 0 meets every query tile: rank == kt
        const bool have = kt >= 0 && kt < g.nkt;
        const int key0 = 16 * (have ? kt : 0), key = key0 + li;
        const float key_in_T = (have && key < T) ? 1.0f : 0.0f;
        const uint32_t drop_base = attn_row_idx(d, head, n, 0) + (uint32_t)key;
        bf8 kh[NKS], kl[NKS], vh[NKS], vl[NKS];
        if (have) {
            gfrag<SPLIT, NKS>(d.K, d.ld, base_row + key0, hoff, T - key0, d.d, g.M, kh, kl);
            gfrag<SPLIT, NKS>(d.V, d.ld, base_row + key0, hoff, T - key0, d.d, g.M, vh, vl);
        }
        const bool kvk = have && key < T && d.k_valid[base_row + (key < T ? key : 0)] != 0.0f;
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
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const bf8 a0h = row_frag(Qh, 16 * l0, ks), a1h = row_frag(Qh, 16 * l1, ks);
                    const bf8 a0l = SPLIT ? row_frag(Ql, 16 * l0, ks) : a0h, a1l = SPLIT ? row_frag(Ql, 16 * l1, ks) : a1h;
                    s0 = mma<SPLIT>(a0h, a0l, kh[ks], kl[ks], s0);      // S[q][key]
                    s1 = mma<SPLIT>(a1h, a1l, kh[ks], kl[ks], s1);
                    const bf8 o0h = row_frag(Oh, 16 * l0, ks), o1h = row_frag(Oh, 16 * l1, ks);
                    const bf8 o0l = SPLIT ? row_frag(Ol, 16 * l0, ks) : o0h, o1l = SPLIT ? row_frag(Ol, 16 * l1, ks) : o1h;
                    p0 = mma<SPLIT>(o0h, o0l, vh[ks], vl[ks], p0);      // dA[q][key] = dO V^T
                    p1 = mma<SPLIT>(o1h, o1l, vh[ks], vl[ks], p1);
                }
                float xa[8], xd[8];
                auto finish = [&](int lt, const f32x4& s, const f32x4& p, bool on, int xo) {
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
                        const bool valid = on && (key <= q4 + r) && kvk;          // causal + key mask
                        const float e = __builtin_amdgcn_exp2f(fmaf(s[r], g.isd_log2e, -mm[r])) * ii[r];
                        const float pn = valid ? e : 0.0f;
                        float w = ww[r];
                        if (dc.on) w *= drop_factor_x(dc, x0 + (uint32_t)r * xT);
                        xa[xo + r] = on ? (pn + key_in_T * uu[r]) * w : 0.0f;    // A after mask + dropout
                        xd[xo + r] = pn * (p[r] * w - dd[r]) * g.isd;            // dS / sqrt(d)
                    }
                };
                finish(l0, s0, p0, true, 0);
                finish(l1, s1, p1, two, 4);
                bf8 ah, al, dh, dl;
                split8<SPLIT>(xa, ah, al);
                split8<SPLIT>(xd, dh, dl);
#pragma unroll
                for (int jt = 0; jt < NDT; ++jt) {
                    const bf8 oh_ = tr_frag(Oh, 16 * l0, 16 * l1, jt);
                    const bf8 ol_ = SPLIT ? tr_frag(Ol, 16 * l0, 16 * l1, jt) : oh_;
                    dv[jt] = mma<SPLIT>(ah, al, oh_, ol_, dv[jt]);               // dV += A^T dO
                    const bf8 qh_ = tr_frag(Qh, 16 * l0, 16 * l1, jt);
                    const bf8 ql_ = SPLIT ? tr_frag(Ql, 16 * l0, 16 * l1, jt) : qh_;
                    dk[jt] = mma<SPLIT>(dh, dl, qh_, ql_, dk[jt]);               // dK += dS^T Q
                }
            }
        }
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
    }
}

// =====================================================================================================
// host side
// =====================================================================================================
static int bf_geom(const cr_attn_desc* d, BfGeom* g) {
    g->T16 = (d->T + 15) / 16 * 16;
    g->nkt = g->T16 / 16;
    g->nch = (g->T16 + BF_CH - 1) / BF_CH;
    g->ch_rows = g->nch == 1 ? g->T16 : BF_CH;
    g->M = d->B * d->T;
    g->isd = (float)(1.0 / sqrt((double)d->d));
    g->isd_log2e = (float)(1.4426950408889634 / sqrt((double)d->d));
    g->invT = 1.0f / (float)d->T;
    return CR_OK;
}

bool cr_attn_bf_supported_fwd(const cr_attn_desc* d) { return d->T <= 256 && d->d >= 1 && d->d <= 64; }
bool cr_attn_bf_supported_bwd(const cr_attn_desc* d) { return d->T <= 1024 && d->d >= 1 && d->d <= 64 && d->row_stats != nullptr; }

static int bf_nsplit(const cr_attn_desc* d, const BfGeom& g) {
    if (g.nch > 1) return (g.nkt + 7) / 8;                               // one workgroup per block of 8 tiles
    int want = (256 + d->B * d->H - 1) / (d->B * d->H);
    const int maxs = (g.nkt + 7) / 8;
    if (want > maxs) want = maxs;
    return want < 1 ? 1 : want;
}

template <int NKT, int NKS, bool SPLIT>
static int launch_bf_fwd(const cr_attn_desc* d, const BfGeom& g, hipStream_t s) {
    static cr_devmask attr_set = 0;
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_bf_fwd<NKT, NKS, SPLIT>), &attr_set);
    if (rc) return rc;
    const size_t lds = (size_t)g.T16 * 128 * 2 * (SPLIT ? 2 : 1) + (size_t)g.T16 * 4;
    hipLaunchKernelGGL((k_bf_fwd<NKT, NKS, SPLIT>), dim3(d->B * d->H, bf_nsplit(d, g)), dim3(512), lds, s, *d, g);
    return cr_check_launch("cr_attn_fwd(bf16)");
}

template <int NKT>
static int dispatch_bf_fwd(const cr_attn_desc* d, const BfGeom& g, hipStream_t s) {
    const bool split = d->precision == CR_PREC_BF16X3;
    if (d->d <= 32) return split ? launch_bf_fwd<NKT, 1, true>(d, g, s) : launch_bf_fwd<NKT, 1, false>(d, g, s);
    return split ? launch_bf_fwd<NKT, 2, true>(d, g, s) : launch_bf_fwd<NKT, 2, false>(d, g, s);
}

int cr_attn_bf_fwd_launch(const cr_attn_desc* d, hipStream_t s) {
    BfGeom g;
    bf_geom(d, &g);
    if (g.nkt <= 4) return dispatch_bf_fwd<4>(d, g, s);
    if (g.nkt <= 13) return dispatch_bf_fwd<13>(d, g, s);
    return dispatch_bf_fwd<16>(d, g, s);
}

template <int NKS, bool SPLIT>
static int launch_bf_bwd(const cr_attn_bwd_desc* bd, const BfGeom& g, hipStream_t s) {
    static cr_devmask attr_q = 0, attr_k = 0;
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_bf_bwd_q<NKS, SPLIT>), &attr_q);
    if (rc) return rc;
    rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_bf_bwd_k<NKS, SPLIT>), &attr_k);
    if (rc) return rc;
    const cr_attn_desc* d = &bd->f;
    const size_t img = (size_t)g.ch_rows * 128 * 2 * (SPLIT ? 2 : 1);
    const dim3 grid(d->B * d->H, bf_nsplit(d, g));
    float* dws = bd->stats;                                              // delta workspace [H*B*T] when the caller gave none
    hipLaunchKernelGGL((k_bf_bwd_q<NKS, SPLIT>), grid, dim3(512), img + (size_t)g.ch_rows * 4, s, *bd, g, dws);
    rc = cr_check_launch("cr_attn_bwd(bf16, q)");
    if (rc) return rc;
    hipLaunchKernelGGL((k_bf_bwd_k<NKS, SPLIT>), grid, dim3(512), img + (size_t)g.ch_rows * 4 * 5 + (size_t)(g.ch_rows / 16) * 4, s,
                       *bd, g, bd->delta ? bd->delta : dws);
    return cr_check_launch("cr_attn_bwd(bf16, k)");
}

int cr_attn_bf_bwd_launch(const cr_attn_bwd_desc* bd, hipStream_t s) {
    const cr_attn_desc* d = &bd->f;
    BfGeom g;
    bf_geom(d, &g);
    const bool split = d->precision == CR_PREC_BF16X3;
    if (!bd->delta) CR_REQUIRE(d->out && d->residual, "cr_attn_bwd(bf16): out / residual needed to form delta");
    if (d->d <= 32) return split ? launch_bf_bwd<1, true>(bd, g, s) : launch_bf_bwd<1, false>(bd, g, s);
    return split ? launch_bf_bwd<2, true>(bd, g, s) : launch_bf_bwd<2, false>(bd, g, s);
}
