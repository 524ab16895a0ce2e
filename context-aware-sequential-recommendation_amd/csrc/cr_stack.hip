// A whole transformer stack (sasrec.py:65-85: per block LN1 -> Q/K/V -> causal attention -> LN2 -> feed-forward,
// then the final LayerNorm; optionally the embedding gather that composes its input) forward, one workgroup -- or a
// pair -- per sequence: ONE launch for the stack, or one per block when a sequence gets two workgroups (PAIR).
//
// Why.  At the headline shape (T = 200, D = 50, B = 128) a block's row phases are 25 600 x 50 matrices: the separate
// kernels (cr_block_ln_qkv_fwd, cr_attn_fwd, cr_block_ln_ffn_fwd) each run 15-30 us of which launch ramp, weight staging
// and the first HBM round trip are the larger part -- ten launches per step forward.  Everything a sequence needs is
// 200 rows: its K / V images (bf16 hi + lo, 104 KB) and one phase's weights (48 KB) fit the 160 KB LDS of a CU, and
// all row-local work stays in the registers of the wave that owns the 16-row tile (layout R, cr_rlayout.hpp: the chain
// LayerNorm -> projection -> ... -> feed-forward runs through registers with no transposition; the attention core takes
// Q the same way -- B operand of S^T = K Q^T, the K image stored in that k order -- and its output product is formed as
// O^T = V^T P^T, which lands in layout R too).
//
// Every intermediate the backward kernels read (q_in, Q K V, row statistics, o, f_in, hid, y) is written to HBM as
// before; a wave re-reads from there (L2) only rows it wrote itself.
// Arithmetic: bf16 MFMA on split (hi + lo, three products) or plain bf16 operands, fp32 accumulation
// (cr_attn_desc.precision of the blocks); element-wise work in fp32.  Counter-based dropout: same element indices as
// the separate kernels, so the backward kernels regenerate the same masks.
// Shapes: one head with D <= 64, or two heads of 32 columns (D = 64); T <= 208 (split) / 256 (plain).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cr_attn_common.hpp"
#include "cr_bf16.hpp"

#define ST_WAVES 8
#define ST_THREADS (64 * ST_WAVES)
#define ST_NVEC 11                // zero-padded 64-float vectors: ln1 g b, bq bk bv, ln2 g b, b1 b2, lnf g b

struct StackBlk { cr_block_desc bd; cr_attn_desc ad; };
struct StackArgs {
    int nb, T16, nkt;
    unsigned char ptile[2][8];    // PAIR mode: the query tile of wave w of workgroup y (255 = none): stack_deal_pair
    float isd_log2e, invT;
    const float* lnf_g; const float* lnf_b; float* out; int ld_out, col_out;
    unsigned long long* ts;       // debug: per-wave phase stamps [B][8 waves][64] (tools/stack_ts.py); NULL in production
    int gather;                   // blk[0].bd.x is composed here from the embedding recipe `e` (and written: the backward reads it)
    cr_embed_desc e;
    StackBlk blk[CR_STACK_MAX_BLOCKS];
    // HEAD instantiations (round 5): the prediction head (sasrec.py:87-115) and the backward of the final LayerNorm on the rows this
    // launch has just normalised -- cr_head_fwd_bwd_ln's work without its launch; hd.seq_emb / hl.x are not read (registers hold them)
    cr_head_desc hd;
    cr_ln_bwd_desc hl;
};

#ifdef CR_TIMELINE
#define SK_TS(slot)                                                                                          \
    do {                                                                                                     \
        if (a.ts && (threadIdx.x & 63) == 0)                                                                 \
            a.ts[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ST_WAVES + (threadIdx.x >> 6)) * 64 + (slot)] =                       \
                ((slot) == 0 || (slot) == 63) ? wall_clock64() : clock64();                                  \
    } while (0)
#else
#define SK_TS(slot) do { } while (0)
#endif

#include "cr_rbwd.hpp"

// The argument block re-read through a pointer the optimiser cannot trace back to the kernel's parameter: what the head tail needs of
// it (a dozen pointers, some first used under lane predicates) is then fetched from the scalar cache where it is used instead of being
// copied to vector registers at the kernel's top and carried -- spilled -- across the attention phase (cr_stack_bwd1.hip b1_args_again).
__device__ __forceinline__ const StackArgs& stack_args_again() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *reinterpret_cast<const StackArgs*>((const void*)p);
}

// the block's zero-padded vectors (and the final LayerNorm's): ST_NVEC x 64 slots over the workgroup
template <int NT>
__device__ __forceinline__ void vec_issue(float (&vv)[(ST_NVEC * 64 + NT - 1) / NT], const cr_block_desc& d, const StackArgs& a, int D) {
#pragma unroll
    for (int u = 0; u < (ST_NVEC * 64 + NT - 1) / NT; ++u) {
        const int t = tid_now() + NT * u, c = t & 63, which = min(t >> 6, ST_NVEC - 1);
        const float* src = which == 0 ? d.ln1_g : which == 1 ? d.ln1_b : which == 2 ? d.bqkv : which == 3 ? d.bqkv + D
                         : which == 4 ? d.bqkv + 2 * D : which == 5 ? d.ln2_g : which == 6 ? d.ln2_b : which == 7 ? d.b1
                         : which == 8 ? d.b2 : which == 9 ? a.lnf_g : a.lnf_b;
        vv[u] = (src && c < D) ? src[c] : 0.0f;
    }
}
template <int NT>
__device__ __forceinline__ void vec_put(float* vec, const float (&vv)[(ST_NVEC * 64 + NT - 1) / NT]) {
#pragma unroll
    for (int u = 0; u < (ST_NVEC * 64 + NT - 1) / NT; ++u) {
        const int t = tid_now() + NT * u;
        if (t < ST_NVEC * 64) vec[t] = vv[u];
    }
}

// Phases of a block (B0..B3 = workgroup barriers):
//   [Wk Wv + vectors -> LDS] B0  A: per tile  x -> K, V (HBM + LDS images), key mask    B1 [Wq W1 W2 -> LDS] B2
//   B/C: per tile  x -> LN1 -> q_in, Q -> scores, softmax, A V + q_in -> o -> LN2 -> FFN -> y (-> final LN)   B3
// What a phase needs from HBM (weights, the tile's x) is requested before the barrier in front of it.
// DS: the hidden size as a compile-time constant (0 = read it from the description): with it every column-tile
// classification above is resolved by the compiler -- instantiated for the headline D = 50
// PAIR: two workgroups per sequence (grid.y = 2), one block per launch.  The kernel is VALU-issue bound, so when the
// batch leaves half of the CUs idle (B = 128 on 256 CUs) a sequence is given two of them: both workgroups compute
// K / V of ALL tiles (phase A, the smaller phase; workgroup 0 also writes them to HBM), then each takes every other
// tile of the heaviest-first order through phase B / C.  Block i + 1 needs y of both, hence one launch per block.
// HD: heads (1, or 2 with head dim 32: head h is then exactly k-step h of the score product and column tiles 2h, 2h + 1 of
// the output -- config C3: D = 64, two heads)
// HEAD: the last block's launch goes on with the prediction head on its own output rows (cr_stack_fwd_head; PAIR launches only: one
// tile per wave, so the LayerNorm-gradient partials need no registers across an attention phase).  The separate head launch cost
// 14.5 us of a 324 us step (rocprofv3) for 25 600 rows of ~1 KB: its ids -> table rows -> reductions chain is short work behind a
// kernel boundary and a cold start; here the ids and the two table rows of a tile are requested while the tile's feed-forward runs.
template <int NKT, bool SPLIT, int NW, int DS, bool PAIR, int HD, bool HEAD = false>
__global__ __launch_bounds__(64 * NW) void k_stack_fwd(StackArgs a) {
    static_assert(!HEAD || PAIR, "the head tail: PAIR launches");
    constexpr int NT = 64 * NW;
    constexpr int NVV = (ST_NVEC * 64 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cr_kernarg_touch<HEAD ? 1280 : ((sizeof(StackArgs) - sizeof(cr_head_desc) - sizeof(cr_ln_bwd_desc)) < 1280 ? (sizeof(StackArgs) - sizeof(cr_head_desc) - sizeof(cr_ln_bwd_desc)) : 1280)>();      // (the first blocks: one launch per block uses blk[0] only)
    // (DS instantiations run with nkt == NKT: the image size is then a constant and the hi / lo / K / V / weight images are
    //  immediates apart: an operand read is one per-lane base + immediate instead of an address sum per read)
    const int IMG = DS > 0 ? 16 * NKT * 64 : a.T16 * 64;
    __bf16* Kh = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Kl = Kh + (SPLIT ? IMG : 0);
    __bf16* Vh = Kl + IMG;
    __bf16* Vl = Vh + (SPLIT ? IMG : 0);
    __bf16* Wi = Vl + IMG;                               // 3 weight slots x (hi [, lo])
    constexpr int WST = SPLIT ? 2 * ST_WIMG : ST_WIMG;   // elements per weight slot
    float* vec = reinterpret_cast<float*>(Wi + 3 * WST);  // [ST_NVEC][64]
    float* kb = vec + ST_NVEC * 64;                      // [T16] additive key bias
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x;
    const int D = DS > 0 ? DS : a.blk[0].bd.D, T = a.blk[0].ad.T;
    const int base_row = n * T;
    SK_TS(0); SK_TS(1);
    // tiles of this wave, heaviest first: ranks w and 2 W - 1 - w of the order "last tile first" (causal cost = tile + 1)
    auto ptile = [&](int y) { const int t = a.ptile[y][wave]; return t == 255 ? -1 : t; };
    const int tile0 = PAIR ? ptile((int)blockIdx.y) : a.nkt - 1 - wave;
    const int tile1 = PAIR ? -1 : a.nkt - 1 - (2 * NW - 1 - wave);
    const int nB = (tile0 >= 0 ? 1 : 0) + (tile1 >= 0 ? 1 : 0);              // phase B / C tiles (tile1 >= 0 implies tile0 >= 0)
    // phase A tiles: the wave's own B / C tile first, then (PAIR) the tile of the same rank pair that the OTHER workgroup
    // takes through B / C -- so a wave only ever re-reads rows of x / y that it wrote itself
    const int tileP = PAIR ? ptile(1 - (int)blockIdx.y) : tile1;
    const int nA = (tile0 >= 0 ? 1 : 0) + (tileP >= 0 ? 1 : 0);
    auto tile_a = [&](int i) { return (i == 0 && tile0 >= 0) ? tile0 : tileP; };
    auto tile_b = [&](int i) { return i == 0 ? tile0 : tile1; };
    const bool wr_kv = !PAIR || blockIdx.y == 0;                             // this workgroup writes K / V / key flags to HBM
    const float c2 = a.isd_log2e;
    const DCtx dcx = d_ctx<D_NF(DS)>(D);
    auto row_of = [&](int tile) { return (u32)(base_row + min(16 * tile + (lane_now() & 15), T - 1)) * (u32)(4 * D); };   // byte offset of the lane's row

    WRegs<2, NT> wa;
    float vv[NVV];
    RRaw xa, xp, xd;                                      // x rows (or, composing x: table rows), positional rows, addend rows
    // composing x (a.gather): ids of the wave's phase-A rows, loaded once; issue_x requests what a tile's x is made of
    int gid0 = 0, gid1 = 0;
    if (a.gather) {
        gid0 = a.e.ids[base_row + min(16 * max(tile_a(0), 0) + (lane_now() & 15), T - 1)];
        gid1 = a.e.ids[base_row + min(16 * max(tile_a(1), 0) + (lane_now() & 15), T - 1)];
    }
    auto issue_x = [&](const float* xsrc, int tile, int gid, bool gather) {
        if (!gather) {
            r_issue(xa, xsrc, row_of(tile), dcx);
        } else {
            const int t = min(16 * tile + (lane_now() & 15), T - 1);
            r_issue(xa, a.e.table, (u32)gid * (u32)(4 * D), dcx);                               // row 0 exists; zeroed below when zero_pad
            if (a.e.pos_table) r_issue(xp, a.e.pos_table, (u32)((base_row + t) % a.e.T) * (u32)(4 * D), dcx);
            if (a.e.addend) r_issue(xd, a.e.addend, (u32)(base_row + t) * (u32)(4 * a.e.ld_add), dcx);
        }
    };
    // the step counter behind the embedding dropout's key: requested first, used in phase A (see drop_ctx)
    const uint32_t sv_e = a.gather ? cr_step_request(a.e.drop) : 0u;
    {
        const cr_block_desc& d = a.blk[0].bd;
        w_issue<2, NT>(wa, D, d.wqkv, 3 * D, D, d.wqkv, 3 * D, 2 * D, d.wqkv, 3 * D, 2 * D);
        vec_issue<NT>(vv, d, a, D);
        issue_x(d.x, max(tile_a(0), 0), gid0, a.gather != 0);
    }
    // HEAD: the final LayerNorm's dgamma / dbeta partials of the waves, [2][NW][64] floats of their own behind the key bias (nothing else
    // of the LDS is free before every wave has left its feed-forward: 159 296 + 4 192 of 163 840 bytes at 13 tiles)
    float* hpart = kb + 16 * NKT;
    // (HEAD launches hold ONE block -- PAIR launches do -- as a constant: with a run-time block loop around it, everything loop-invariant of
    //  the head tail was formed in front of the loop and carried, spilled, across the attention phase.  NOT for the other PAIR launches:
    //  the constant there removed 32 bytes of scratch from the generic instantiations and made the headline's launches 1 us LONGER each --
    //  28.98 / 26.16 / 28.36 against 27.88 / 25.22 / 27.72 us by position, rocprofv3)
    const int nblk = HEAD ? 1 : a.nb;
#pragma unroll 1
    for (int b = 0; b < nblk; ++b) {
        const cr_block_desc& d = a.blk[b].bd;
        const cr_attn_desc& ad = a.blk[b].ad;
        const bool last = b == nblk - 1;
        // (the block's step counters: requested here, used behind barrier B2)
        const uint32_t sv_a = cr_step_request(ad.drop), sv_1 = cr_step_request(d.drop_ffn1), sv_2 = cr_step_request(d.drop_ffn2);
        {
            w_put<2, NT, SPLIT>(Wi, wa, D, d.wqkv, 3 * D, D, d.wqkv, 3 * D, 2 * D, d.wqkv, 3 * D, 2 * D);
            vec_put<NT>(vec, vv);
        }
        __syncthreads();                                  // B0
        SK_TS(2 + 10 * b);
        // Wq and the feed-forward weights are requested now and land under phase A
        WRegs<3, NT> wb;
        w_issue<3, NT>(wb, D, d.wqkv, 3 * D, 0, d.w1, D, 0, d.w2, D, 0);
        // ---- phase A: K, V of the wave's tiles (HBM + LDS images), key mask ----------------------------
#pragma unroll 1
        for (int i = 0; i < nA; ++i) {
            const int ln = lane_now(), li = ln & 15, lg = ln >> 4;
            const int q0 = 16 * tile_a(i);
            const bool rok = q0 + li < T;
            const int m = base_row + min(q0 + li, T - 1);
            const u32 mo = (u32)m * (u32)(4 * D);
            f32x4 x[4];
            r_finish(x, xa, dcx);
            if (a.gather && b == 0) {
                // x[m] = mask * dropout(table'[id[m]] * scale + pos[m % T] + addend[m])  (cr_embed_fwd: sasrec.py:27-62 / cast_1.py:86-91)
                const cr_embed_desc& e = a.e;
                const int gid = i == 0 ? gid0 : gid1;
                const bool padrow = e.zero_pad && gid == 0;
                const bool dead = e.mask_ids && e.mask_ids[m] == 0;
                f32x4 pv[4], av[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) { pv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; av[ct] = pv[ct]; }
                if (e.pos_table) r_finish(pv, xp, dcx);
                if (e.addend) r_finish(av, xd, dcx);
                const DropCtx dce = drop_ctx(e.drop, (uint32_t)__builtin_amdgcn_readfirstlane(sv_e));
                const uint32_t eb = ((e.drop.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)(4 * lg)) * CR_PHI + dce.key;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = (padrow ? 0.0f : x[ct][r]) * e.scale + pv[ct][r] + av[ct][r];
                        v *= drop_factor_x(dce, eb + (uint32_t)(16 * ct + r) * CR_PHI);       // (rate 0: threshold 0, factor 1.0 -- no branch per element)
                        x[ct][r] = dead ? 0.0f : v;
                    }
                // the wave writes x for the tile it takes through phase B / C (it re-reads those rows there)
                r_store(e.out, mo, x, rok && (!PAIR || tile_a(i) == tile0), dcx);
            }
            if (i + 1 < nA) issue_x(d.x, tile_a(i + 1), gid1, a.gather && b == 0);
            const float xs = r_rowsum(x);
            if (lg == 0) kb[q0 + li] = (rok && xs != 0.0f) ? 0.0f : -INFINITY;   // key mask (modules.py:222)
            if (lg == 0 && rok && wr_kv) d.k_valid[m] = (xs != 0.0f) ? 1.0f : 0.0f;
            bf8 xh[2], xl[2];
            r_split<SPLIT>(x, xh, xl);
            f32x4 acc[4], bias[4];
            // K = x Wk + bk (modules.py:204): image in the k order of the score product's B operand
            r_gemm<SPLIT>(acc, Wi, Wi + ST_WIMG, xh, xl);
            r_vec(bias, vec + 3 * 64);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] += bias[ct];
            r_store(d.qkv + (size_t)d.M * D, mo, acc, rok && wr_kv, dcx);
            {
                bf8 h[2], l[2];                           // (rows beyond T hold copies of row T - 1: finite, and masked as keys)
                r_split<SPLIT>(acc, h, l);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int o = img_off<2>(q0 + li, 4 * ks + lg);
                    *reinterpret_cast<bf8*>(Kh + o) = h[ks];
                    if (SPLIT) *reinterpret_cast<bf8*>(Kl + o) = l[ks];
                }
            }
            // V = x Wv + bv (modules.py:205): image in natural column order (read transposed)
            r_gemm<SPLIT>(acc, Wi + WST, Wi + WST + ST_WIMG, xh, xl);
            r_vec(bias, vec + 4 * 64);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] += bias[ct];
            r_store(d.qkv + (size_t)2 * d.M * D, mo, acc, rok && wr_kv, dcx);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                bf4 h, l;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 v = {acc[ct][r], acc[ct][r + 1]};
                    const bf2 hh = __builtin_convertvector(v, bf2);
                    const bf2 ll = __builtin_convertvector(v - __builtin_convertvector(hh, f32x2), bf2);
                    h[r] = hh[0]; h[r + 1] = hh[1];
                    l[r] = ll[0]; l[r + 1] = ll[1];
                }
                // layout W (cr_rlayout.hpp): this image is only ever read transposed, and a tile written from layout R into the dual-use
                // layout hits four chunk positions with sixteen rows -- 4-way bank conflicts on every write, a fifth of the launch's
                // LDS-active cycles (round 4's counters; the block backward's weight-gradient images had the same and lost it the same way)
                const int o = wimg_off(q0 + li, 4 * ct + lg);
                *reinterpret_cast<bf4*>(Vh + o) = h;
                if (SPLIT) *reinterpret_cast<bf4*>(Vl + o) = l;
            }
            if (i == 0) SK_TS(3 + 10 * b);
        }
        SK_TS(4 + 10 * b);
        // the first B / C tile's x flies across the barrier
        RRaw xb;
        r_issue(xb, d.x, row_of(max(tile_b(0), 0)), dcx);
        __syncthreads();                                  // B1: K / V / kb complete; Wk Wv no longer read
        w_put<3, NT, SPLIT>(Wi, wb, D, d.wqkv, 3 * D, 0, d.w1, D, 0, d.w2, D, 0);
        __syncthreads();                                  // B2
        // (s_setprio 1 for the second-dispatched half of the waves, as in the block backward's attention loops: +4 us per step HERE -- the
        //  heaviest tiles sit in waves 0-3, whose chains set the launch's length; tools/probes/run_ab.sh, round 4)
        SK_TS(5 + 10 * b);
        const int fvk = first_valid_key_lds(kb, a.T16, T);
        const int kt_first = min(fvk >> 4, NKT - 1);      // tiles below hold no valid key: probabilities exactly 0
        const DropCtx dc = drop_ctx(ad.drop, (uint32_t)__builtin_amdgcn_readfirstlane(sv_a));
        const DropCtx d1 = drop_ctx(d.drop_ffn1, (uint32_t)__builtin_amdgcn_readfirstlane(sv_1)), d2 = drop_ctx(d.drop_ffn2, (uint32_t)__builtin_amdgcn_readfirstlane(sv_2));
        // ---- phase B + C per tile: LN1, Q, attention core, then LN2 + feed-forward on the rows in registers -----
#pragma unroll 1
        for (int i = 0; i < nB; ++i) {
            const int ln = lane_now(), li = ln & 15, lg = ln >> 4;
            const int qt = tile_b(i);
            const int q0 = 16 * qt, q = q0 + li;
            const bool rok = q < T;
            const int m = base_row + min(q, T - 1);
            const u32 mo = (u32)m * (u32)(4 * D);
            const int id_n = ad.dead_ids ? ad.dead_ids[m] : 1;
            const int mk = d.mask_ids[m];
            int hpid = 0, hnid = 0;                       // HEAD: the row's pos / neg ids (row 0 of the table reads as zeros)
            if (HEAD && last) { hpid = a.hd.pos[m]; hnid = a.hd.neg[m]; }
            f32x4 o[4];
            bf8 qh[2], ql[2];
            float qvq;
            {
                f32x4 x[4], y[4], bias[4];
                r_finish(x, xb, dcx);
                r_layernorm<true>(y, x, vec, vec + 64, dcx);                       // modules.py:74-78
                const float ys = r_rowsum(y);
                qvq = (rok && ys != 0.0f) ? 1.0f : 0.0f;                         // query mask (modules.py:248-249)
                if (lg == 0 && rok) d.q_valid[m] = qvq;
                r_store(d.q_in, mo, y, rok, dcx);
                bf8 yh[2], yl[2];
                r_split<SPLIT>(y, yh, yl);
                r_gemm<SPLIT>(o, Wi, Wi + ST_WIMG, yh, yl);                      // Q = q_in Wq + bq (modules.py:203)
                r_vec(bias, vec + 2 * 64);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) o[ct] += bias[ct];
                r_store(d.qkv, mo, o, rok, dcx);
                r_split<SPLIT>(o, qh, ql);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) o[ct] = y[ct];                    // residual (modules.py:269) if the tile is dead
            }
            const bool is_dead = !rok || id_n == 0;
#pragma unroll
            for (int h = 0; h < HD; ++h) {
            const int ks_lo = HD == 2 ? h : 0, ks_hi = HD == 2 ? h + 1 : 2;          // k-steps of this head's columns
            const size_t srow = ((size_t)h * gridDim.x + n) * T + q;                // its row statistics
            if (__all(is_dead ? 1 : 0)) {
                // the whole tile is padding: A = 0 -> out = residual (known dead downstream, sasrec.py:83)
                if (ad.row_stats && lg == 0 && rok) {
                    float* sp = ad.row_stats + srow * 4;
                    sp[0] = 0.0f; sp[1] = 0.0f; sp[2] = 2.0f; sp[3] = 0.0f;
                }
            } else {
                // ---- scores St[key][query] (modules.py:216-241), kept for the whole row block
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
                        bf8 k0h[2], k0l[2], k1h[2], k1l[2];
#pragma unroll
                        for (int ks = ks_lo; ks < ks_hi; ++ks) {
                            k0h[ks] = row_frag_l(Kh, r0, ks, ln); k1h[ks] = row_frag_l(Kh, r1, ks, ln);
                            k0l[ks] = SPLIT ? row_frag_l(Kl, r0, ks, ln) : k0h[ks]; k1l[ks] = SPLIT ? row_frag_l(Kl, r1, ks, ln) : k1h[ks];
                        }
                        f32x4 x0 = (f32x4){0.f, 0.f, 0.f, 0.f}, x1 = x0;
#pragma unroll
                        for (int ks = ks_lo; ks < ks_hi; ++ks) {
                            x0 = mma<SPLIT>(k0h[ks], k0l[ks], qh[ks], ql[ks], x0);
                            x1 = mma<SPLIT>(k1h[ks], k1l[ks], qh[ks], ql[ks], x1);
                        }
                        BF_SGB(0x100, (SPLIT ? 8 : 4) / HD, 0);
                        BF_SGB(0x008, (SPLIT ? 12 : 4) / HD, 0);
                        if (c0) a0 = finish(kt, c0 ? x0 : x1);
                        if (c1) a1 = finish(kt + 1, x1);
                    }
                    st[kt] = a0;
                    if (kt + 1 < NKT) st[kt + 1] = a1;
                }
                mx = grp_max(mx);
                const bool uniform = (mx == -INFINITY) && !is_dead && rok;
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
                if (i == 0) SK_TS(6 + 10 * b);
                float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
                if (is_dead) inv = 0.0f;
                const bool any_uni = __any(uniform ? 1 : 0) != 0;
                if (ad.row_stats && lg == 0 && rok) {            // for the backward kernels
                    float* sp = ad.row_stats + srow * 4;
                    sp[0] = mx; sp[1] = inv; sp[2] = is_dead ? 2.0f : (uniform ? 1.0f : 0.0f); sp[3] = 0.0f;
                }
                // ---- softmax scale, query mask, dropout (modules.py:244-257) folded into one factor per element
                const float wq = inv * qvq;
                const uint32_t ridx = attn_row_idx(ad, h, n, q);
                const uint32_t xrow = (ridx + (uint32_t)(4 * lg)) * CR_PHI + dc.key;
                if (any_uni) {                                   // rare: a row with no valid key at all (modules.py:227-244)
                    const float uni = uniform ? a.invT * qvq : 0.0f;
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
                // ---- O^T = V^T A^T (modules.py:262): two key tiles per k-step, V through transposed reads;
                // accumulated on top of the residual
                const int kt_lo = any_uni ? 0 : kt_first, kt_end = any_uni ? a.nkt : qt + 1;
#pragma unroll
                for (int kp = 0; kp < (NKT + 1) / 2; ++kp) {
                    const int k0 = 2 * kp, k1 = 2 * kp + 1;
                    if (k1 >= kt_lo && k0 < kt_end) {            // wave-uniform; tiles outside the live range hold zeros
                        float xx[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            xx[r] = st[k0][r];
                            xx[4 + r] = (k1 < NKT) ? st[k1 < NKT ? k1 : k0][r] : 0.0f;
                        }
                        bf8 ph, pl;
                        split8<SPLIT>(xx, ph, pl);
                        // tiles beyond T16 are not staged (their A is 0); DS instantiations run with nkt == NKT: constant offsets
                        const int ra = 16 * k0, rb = 16 * ((DS > 0 ? k1 < NKT : k1 < a.nkt) ? k1 : k0);
                        constexpr int JB = (SPLIT || HD == 2) ? 2 : 4;
#pragma unroll
                        for (int j0 = (HD == 2 ? 2 * h : 0); j0 < (HD == 2 ? 2 * h + 2 : 4); j0 += JB) {
                            bf8 vh[JB], vl[JB];
#pragma unroll
                            for (int jt = 0; jt < JB; ++jt) {
                                vh[jt] = tr_frag_w(Vh, ra, rb, j0 + jt, ln);
                                vl[jt] = SPLIT ? tr_frag_w(Vl, ra, rb, j0 + jt, ln) : vh[jt];
                            }
#pragma unroll
                            for (int jt = 0; jt < JB; ++jt) o[j0 + jt] = mma<SPLIT>(vh[jt], vl[jt], ph, pl, o[j0 + jt]);
                            BF_SGB(0x100, (SPLIT ? 4 : 2) * JB, 0);
                            BF_SGB(0x008, (SPLIT ? 3 : 1) * JB, 0);
                        }
                    }
                }
            }
            }   // heads
            if (i + 1 < nB) r_issue(xb, d.x, row_of(tile1), dcx);        // the next tile's x flies under the feed-forward
            // HEAD: the two table rows are requested here, behind the attention phase (their ids at the tile's top): they fly under LN2 and
            // the feed-forward, 38 registers that the attention phase does not have
            RRaw hrp, hrn;
            if (HEAD && last) {
                const StackArgs& ah = stack_args_again();
                r_issue(hrp, ah.hd.table, (u32)hpid * (u32)(4 * D), dcx, rok && hpid != 0);
                r_issue(hrn, ah.hd.table, (u32)hnid * (u32)(4 * D), dcx, rok && hnid != 0);
            }
            r_store(d.o, mo, o, rok, dcx);
            if (i == 0) SK_TS(7 + 10 * b);
            // ---- LN2 + point-wise feed-forward (modules.py:300-313), row mask (sasrec.py:83)
            f32x4 fin[4], acc[4], bias[4];
            r_layernorm<true>(fin, o, vec + 5 * 64, vec + 6 * 64, dcx);            // sasrec.py:81
            r_store(d.f_in, mo, fin, rok, dcx);
            bf8 xh[2], xl[2];
            r_split<SPLIT>(fin, xh, xl);
            r_gemm<SPLIT>(acc, Wi + WST, Wi + WST + ST_WIMG, xh, xl);
            r_vec(bias, vec + 7 * 64);
            const uint32_t e1 = ((d.drop_ffn1.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)(4 * lg)) * CR_PHI;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = fmaxf(acc[ct][r] + bias[ct][r], 0.0f);
                    v *= drop_factor_x(d1, e1 + d1.key + (uint32_t)(16 * ct + r) * CR_PHI);                // modules.py:303-304
                    acc[ct][r] = v;
                }
            r_store(d.hid, mo, acc, rok, dcx);
            r_split<SPLIT>(acc, xh, xl);
            r_gemm<SPLIT>(acc, Wi + 2 * WST, Wi + 2 * WST + ST_WIMG, xh, xl);
            r_vec(bias, vec + 8 * 64);
            const float msk = mk != 0 ? 1.0f : 0.0f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[ct][r] + bias[ct][r];
                    v *= drop_factor_x(d2, e1 + d2.key + (uint32_t)(16 * ct + r) * CR_PHI);                // modules.py:309-310
                    acc[ct][r] = (v + fin[ct][r]) * msk;                                                  // modules.py:313, sasrec.py:83
                }
            r_store(d.y, mo, acc, rok, dcx);
            if (i == 0) SK_TS(8 + 10 * b);
            if (last && a.out) {                          // the stack's final LayerNorm (sasrec.py:85)
                r_layernorm<false>(fin, acc, vec + 9 * 64, vec + 10 * 64, dcx);
                r_store(a.out + a.col_out, (u32)m * (u32)(4 * a.ld_out), fin, rok, dcx);
            }
            if (HEAD && last) {
                const StackArgs& ah = stack_args_again();
                // ---- prediction head on the rows in registers (sasrec.py:87-115; cr_head.hip k_head_ln, same arithmetic) ----
                f32x4 ep[4], en[4], pr[4];
                r_finish(ep, hrp, dcx);
                r_finish(en, hrn, dcx);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) pr[ct] = ep[ct] * fin[ct];
                const float pl = r_rowsum(pr);                                             // sasrec.py:100
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) pr[ct] = en[ct] * fin[ct];
                const float nl = r_rowsum(pr);                                             // sasrec.py:101
                const float ist = (rok && hpid != 0) ? 1.0f : 0.0f;                        // sasrec.py:104
                const float sp = 1.0f / (1.0f + expf(-pl)), sn = 1.0f / (1.0f + expf(-nl));
                const float dpl = -ist * sp * (1.0f - sp) / (sp + 1e-24f);
                const float dnl = ist * sn * (1.0f - sn) / (1.0f - sn + 1e-24f);
                float h_loss = 0.0f, h_auc = 0.0f, h_n = 0.0f;
                if (lg == 0 && rok) {
                    h_loss = ist * (-logf(sp + 1e-24f) - logf(1.0f - sn + 1e-24f));        // sasrec.py:105-108
                    const float dlt = pl - nl;
                    const float sgn = (dlt > 0.0f) ? 1.0f : ((dlt < 0.0f) ? -1.0f : 0.0f);
                    h_auc = ist * (sgn + 1.0f) * 0.5f;                                     // sasrec.py:113-115
                    h_n = ist;
                    if (ah.hd.pos_logits) ah.hd.pos_logits[m] = pl;
                    if (ah.hd.neg_logits) ah.hd.neg_logits[m] = nl;
                    if (ah.hd.coef_out) { ah.hd.coef_out[m] = dpl; ah.hd.coef_out[(size_t)ah.hd.M + m] = dnl; }
                }
                {                                         // the wave's three sums (one tile per wave: written, not added to)
                    const float wl = wave_sum(h_loss), wa = wave_sum(h_auc), wn = wave_sum(h_n);
                    float* red = hpart + 2 * NW * 64;
                    if (ln == 0) { red[wave] = wl; red[NW + wave] = wa; red[2 * NW + wave] = wn; }
                }
                // the gradient row dy = dpl E[pos] + dnl E[neg] goes straight through the final LayerNorm's backward (modules.py:74-78):
                // its input is this tile's y (acc), its gain the vector at slot 9
                f32x4 dyh[4], dxh[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dyh[ct][r] = fmaf(dpl, ep[ct][r], dnl * en[ct][r]);     // (element by element: the vector form is a packed multiply + in-place packed fma, the chain build.py's ISA scan refuses)
                if (ah.hd.d_seq_emb) r_store(ah.hd.d_seq_emb, (u32)m * (u32)(4 * ah.hd.ldd), dyh, rok, dcx);
                f32x4 hag[4], hab[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) { hag[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; hab[ct] = hag[ct]; }
                r_ln_bwd(dxh, acc, dyh, vec + 9 * 64, hag, hab, dcx);
                r_store(ah.hl.dx, (u32)m * (u32)(4 * ah.hl.lddx), dxh, rok, dcx);
                // the tile's 16 rows folded per column (DPP row sums), into this wave's slot (one tile per wave: written, not added to)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sg = cr_row16_sum(hag[ct][r]), sb = cr_row16_sum(hab[ct][r]);
                        if (li == 0) {
                            hpart[wave * 64 + 16 * ct + 4 * lg + r] = sg;
                            hpart[(NW + wave) * 64 + 16 * ct + 4 * lg + r] = sb;
                        }
                    }
            }
        }
        SK_TS(10 + 10 * b);
        if (!last) {                                      // the next block's Wk Wv, vectors and first tile (= this block's y)
            const cr_block_desc& dn = a.blk[b + 1].bd;
            w_issue<2, NT>(wa, D, dn.wqkv, 3 * D, D, dn.wqkv, 3 * D, 2 * D, dn.wqkv, 3 * D, 2 * D);
            vec_issue<NT>(vv, dn, a, D);
            issue_x(dn.x, max(tile_a(0), 0), 0, false);
        }
        if (!last) __syncthreads();                       // B3: images, weights and vectors are rewritten by the next block
    }
    if (HEAD) {
        const StackArgs& ah = stack_args_again();
        // dgamma / dbeta of the final LayerNorm: the waves' partials in a fixed order -> this workgroup's slab; the workgroup's loss sums;
        // and the ticket that lets the LAST workgroup of the grid snapshot the step's sums (cr_common.hpp head_snapshot)
        float* red = hpart + 2 * NW * 64;                 // [3][NW]
        if (nB == 0) {                                    // a wave without a tile: its slots read as zeros
            hpart[wave * 64 + (threadIdx.x & 63)] = 0.0f;
            hpart[(NW + wave) * 64 + (threadIdx.x & 63)] = 0.0f;
            if ((threadIdx.x & 63) == 0) { red[wave] = 0.0f; red[NW + wave] = 0.0f; red[2 * NW + wave] = 0.0f; }
        }
        __syncthreads();
        const unsigned slab = blockIdx.y * gridDim.x + blockIdx.x;
        if ((int)threadIdx.x < D) {
            float g = 0.0f, bsum = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { g += hpart[w * 64 + threadIdx.x]; bsum += hpart[(NW + w) * 64 + threadIdx.x]; }
            ah.hl.dgamma[(size_t)slab * ah.hl.slab_stride + threadIdx.x] = g;
            ah.hl.dbeta[(size_t)slab * ah.hl.slab_stride + threadIdx.x] = bsum;
        }
        if (threadIdx.x < 3) {
            float v = 0.0f;
            for (int w = 0; w < NW; ++w) v += red[NW * threadIdx.x + w];
            if (v != 0.0f) atomicAdd(ah.hd.state + threadIdx.x, v);
        }
        head_snapshot_at(ah.hd.state, gridDim.x * gridDim.y, reinterpret_cast<int*>(red + 3 * NW));
    }
    SK_TS(63);
}

// =====================================================================================================
// host side
// =====================================================================================================
static size_t stack_lds_bytes(int T16, bool split, bool head = false, int nkt_template = 0) {
    // (HEAD: the key bias is addressed as [16 NKT] -- the instantiation's tile count -- so that the partials behind it have a constant offset)
    const size_t kb = head ? (size_t)16 * nkt_template * 4 : (size_t)T16 * 4;
    return (size_t)T16 * 64 * 2 * (split ? 4 : 2) + (size_t)3 * ST_WIMG * 2 * (split ? 2 : 1) + (size_t)ST_NVEC * 64 * 4 + kb
           + (head ? (size_t)(2 * ST_WAVES * 64 + 3 * ST_WAVES + 4) * 4 : 0);
}

static const char* stack_unsupported(const cr_stack_desc* s) {
    if (!s || !s->blocks || !s->attn) return "NULL description";
    if (s->n_blocks < 1 || s->n_blocks > CR_STACK_MAX_BLOCKS) return "1..4 blocks";
    const cr_block_desc& b0 = s->blocks[0];
    const cr_attn_desc& a0 = s->attn[0];
    if (b0.D < 8 || b0.D > 64) return "hidden size 8..64";
    if (!((a0.H == 1 && a0.d == b0.D) || (a0.H == 2 && a0.d == 32 && b0.D == 64))) return "one head, or two heads of 32 columns";
    if (a0.precision != CR_PREC_BF16X3 && a0.precision != CR_PREC_BF16) return "bf16 arithmetic (precision) only";
    if (a0.T < 1 || a0.T > 256 || b0.M != a0.B * a0.T) return "T <= 256, M = B T";
    if ((size_t)b0.M * (size_t)(s->out && s->ld_out > b0.D ? s->ld_out : b0.D) * 4 >= ((size_t)1 << 32)) return "activations of 4 GiB or more (32-bit row offsets)";
    const int T16 = (a0.T + 15) / 16 * 16;
    if (stack_lds_bytes(T16, a0.precision == CR_PREC_BF16X3) > 160 * 1024) return "K / V images + weights exceed the LDS";
    // (more than 13 tiles: plain bf16 and one head only -- the split form does not fit the LDS there anyway, and the two-head
    //  16-tile instantiations needed 200+ spilled registers: retired, those shapes take the unfused kernels)
    if (T16 > 208 && (a0.precision == CR_PREC_BF16X3 || a0.H != 1)) return "T <= 208 (bf16x3, or two heads)";
    for (int i = 0; i < s->n_blocks; ++i) {
        const cr_block_desc& b = s->blocks[i];
        const cr_attn_desc& a = s->attn[i];
        if (b.M != b0.M || b.D != b0.D || a.B != a0.B || a.T != a0.T || a.H != a0.H || a.d != a0.d || a.precision != a0.precision) return "blocks differ in shape";
        if (a.attn_weights) return "attention weights are not produced";
        if (a.Q != b.qkv || a.K != b.qkv + (size_t)b.M * b.D || a.V != b.qkv + (size_t)2 * b.M * b.D || a.ld != b.D) return "attn Q/K/V must be the block's qkv";
        if (a.residual != b.q_in || a.ldr != b.D || a.out != b.o || a.ldo != b.D) return "attn residual / out must be the block's q_in / o";
        if (a.k_valid != b.k_valid || a.q_valid != b.q_valid) return "attn masks must be the block's";
        if (!b.x || !b.q_in || !b.qkv || !b.k_valid || !b.q_valid || !b.o || !b.f_in || !b.hid || !b.y || !b.mask_ids) return "NULL buffer";
        if (i > 0 && b.x != s->blocks[i - 1].y) return "blocks must chain (x of block i = y of block i-1)";
    }
    if (s->out && (!s->lnf_gamma || !s->lnf_beta)) return "final LayerNorm parameters";
    if (s->embed) {
        const cr_embed_desc& e = *s->embed;
        if (!e.ids || !e.table || e.M != b0.M || e.D != b0.D || e.T < 1 || e.V < 1) return "embedding recipe: shape";
        if (e.out != b0.x || e.ld_out != b0.D || e.col_off != 0) return "embedding recipe must describe blocks[0].x (dense)";
        if ((size_t)e.V * e.D * 4 >= ((size_t)1 << 32)) return "embedding table of 4 GiB or more (32-bit row offsets)";
        if (e.addend && (size_t)e.M * e.ld_add * 4 >= ((size_t)1 << 32)) return "addend of 4 GiB or more";
    }
    return nullptr;
}

extern "C" int cr_stack_fwd_supported(const cr_stack_desc* s) { return stack_unsupported(s) == nullptr; }

template <int NKT, bool SPLIT, int DS, bool PAIR, int HD, bool HEAD = false>
static int launch_stack_d(const StackArgs& a, int B, hipStream_t s) {
    static cr_devmask attr_set = 0;
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_stack_fwd<NKT, SPLIT, ST_WAVES, DS, PAIR, HD, HEAD>), &attr_set);
    if (rc) return rc;
    hipLaunchKernelGGL((k_stack_fwd<NKT, SPLIT, ST_WAVES, DS, PAIR, HD, HEAD>), dim3(B, PAIR ? 2 : 1), dim3(ST_THREADS), stack_lds_bytes(a.T16, SPLIT, HEAD, NKT), s, a);
    return cr_check_launch(HEAD ? "cr_stack_fwd_head" : "cr_stack_fwd");
}
// head: the launch also runs the prediction head (PAIR launches only; see k_stack_fwd's HEAD)
template <int NKT, bool SPLIT>
static int launch_stack(const StackArgs& a, int B, bool pair, bool head, hipStream_t s) {
    constexpr int DS = (NKT == 4 || NKT == 13) ? 50 : 0;                 // the headline hidden size as a constant
    if constexpr (NKT <= 13) {
        if (a.blk[0].ad.H == 2) {
            if (head) return launch_stack_d<NKT, SPLIT, 0, true, 2, true>(a, B, s);
            return pair ? launch_stack_d<NKT, SPLIT, 0, true, 2>(a, B, s) : launch_stack_d<NKT, SPLIT, 0, false, 2>(a, B, s);
        }
    }
    if (DS && a.blk[0].bd.D == DS && a.nkt == NKT) {
        if (head) return launch_stack_d<NKT, SPLIT, DS, true, 1, true>(a, B, s);
        return pair ? launch_stack_d<NKT, SPLIT, DS, true, 1>(a, B, s) : launch_stack_d<NKT, SPLIT, DS, false, 1>(a, B, s);
    }
    if constexpr (NKT == 13 && SPLIT) {
        // 13 tiles, split arithmetic, another hidden size (the headline's length at --hidden_units 20 / 36 / 44 / 60 ...): the instantiation of
        // the size's FAMILY (whole 16-column tiles as a constant, cr_rlayout.hpp d_ctx<NF>) -- the fully generic one spilled 24-128 bytes per lane
#define ST_FAMILY(DSF)                                                                                                  \
    do {                                                                                                                \
        if (head) return launch_stack_d<NKT, SPLIT, DSF, true, 1, true>(a, B, s);                                       \
        return pair ? launch_stack_d<NKT, SPLIT, DSF, true, 1>(a, B, s) : launch_stack_d<NKT, SPLIT, DSF, false, 1>(a, B, s); \
    } while (0)
        switch (a.blk[0].bd.D / 16) {
        case 0: ST_FAMILY(-1);
        case 1: ST_FAMILY(-2);
        case 2: ST_FAMILY(-3);
        default: ST_FAMILY(-4);
        }
#undef ST_FAMILY
    } else {
        if constexpr (NKT <= 13) {
            if (head) return launch_stack_d<NKT, SPLIT, 0, true, 1, true>(a, B, s);
        }
        return pair ? launch_stack_d<NKT, SPLIT, 0, true, 1>(a, B, s) : launch_stack_d<NKT, SPLIT, 0, false, 1>(a, B, s);
    }
}
static int launch_stack_any(const StackArgs& a, int B, bool split, bool pair, bool head, hipStream_t st) {
    if (a.nkt <= 4) return split ? launch_stack<4, true>(a, B, pair, head, st) : launch_stack<4, false>(a, B, pair, head, st);
    if (a.nkt <= 8) return split ? launch_stack<8, true>(a, B, pair, head, st) : launch_stack<8, false>(a, B, pair, head, st);
    if (a.nkt <= 13) return split ? launch_stack<13, true>(a, B, pair, head, st) : launch_stack<13, false>(a, B, pair, head, st);
    return launch_stack<16, false>(a, B, pair, false, st);       // (stack_unsupported: plain bf16, one head; no head tail at 16 tiles)
}

// batches up to this size run two workgroups per sequence, one launch per block (256 CUs, one workgroup each)
static const int g_stack_pair_max_b = getenv("CASTREC_STACK_PAIR_MAX_B") ? atoi(getenv("CASTREC_STACK_PAIR_MAX_B")) : 160;

static bool stack_pair_mode(const cr_stack_desc* s) {
    const cr_attn_desc& a0 = s->attn[0];
    return a0.B <= g_stack_pair_max_b && (a0.T + 15) / 16 >= 2;
}

// why the prediction head cannot ride on the stack's last launch (nullptr: it can)
static const char* stack_head_unsupported(const cr_stack_desc* s, const cr_head_desc* h, const cr_ln_bwd_desc* n) {
    const char* why = stack_unsupported(s);
    if (why) return why;
    if (!h || !n) return "NULL head / LayerNorm description";
    const cr_block_desc& bl = s->blocks[s->n_blocks - 1];
    const cr_attn_desc& a0 = s->attn[0];
    if (!stack_pair_mode(s)) return "two workgroups per sequence only (B <= 160, more than one tile)";
    if ((a0.T + 15) / 16 > 13) return "at most 13 tiles";
    if (!s->out || s->col_out != 0) return "the stack must end in its final LayerNorm, written dense";
    if (h->seq_emb != s->out || h->ld != s->ld_out || h->M != bl.M || h->D != bl.D || !h->table || !h->pos || !h->neg || !h->state) return "head description";
    if (h->table_grad) return "no table scatter here (cr_head_desc.coef_out: the occurrence index)";
    if ((size_t)h->V * h->D * 4 >= ((size_t)1 << 32)) return "item table of 4 GiB or more (32-bit row offsets)";
    if (n->x != bl.y || n->ldx != bl.D || n->gamma != s->lnf_gamma || !n->dx || n->lddx != bl.D || !n->dgamma || !n->dbeta || n->accumulate || n->M != bl.M || n->D != bl.D)
        return "LayerNorm description (x = the last block's y, gamma = the final LayerNorm's)";
    if (n->n_slabs < 2 * a0.B) return "two slabs per sequence (n_slabs >= 2 B)";
    if (h->d_seq_emb && h->ldd < bl.D) return "d_seq_emb";
    const int T16 = (a0.T + 15) / 16 * 16;
    const int nktt = T16 / 16 <= 4 ? 4 : (T16 / 16 <= 8 ? 8 : 13);
    if (stack_lds_bytes(T16, a0.precision == CR_PREC_BF16X3, true, nktt) > 160 * 1024) return "no LDS left for the head's partials";
    return nullptr;
}
extern "C" int cr_stack_fwd_head_supported(const cr_stack_desc* s, const cr_head_desc* h, const cr_ln_bwd_desc* n) {
    return stack_head_unsupported(s, h, n) == nullptr;
}

// PAIR mode: which query tile wave w of workgroup y carries through the attention and feed-forward phases (each workgroup runs phase A on
// its own tile and on the partner's tile of the same wave, so any deal covers every tile's K / V in both).  Until round 5: tile
// nkt - 1 - (2 w + y) -- workgroup 0's SIMD 0 (waves 0 and 4) carried tiles 12 and 4, SIMD 3 tile 6 alone.  Now: tiles in descending
// order go to the (workgroup, SIMD) with the smallest sum of (key tiles + ST_DEAL_FIXED) so far, upper wave (w + 4) first when free.
#ifndef ST_DEAL_FIXED
#define ST_DEAL_FIXED 6
#endif
static void stack_deal_pair(int nkt, unsigned char (&pt)[2][8]) {
    for (int y = 0; y < 2; ++y) for (int w = 0; w < 8; ++w) pt[y][w] = 255;
    if (nkt > 16) return;                                 // (PAIR mode takes at most 16 tiles: stack_pair_mode)
    static const int fixed = getenv("CASTREC_FWD_DEAL_FIXED") ? atoi(getenv("CASTREC_FWD_DEAL_FIXED")) : ST_DEAL_FIXED;
    if (fixed < 0) {                                      // the deal of rounds 2-4 (measurement switch)
        for (int y = 0; y < 2; ++y) for (int w = 0; w < 8; ++w) { const int t = nkt - 1 - (2 * w + y); pt[y][w] = t >= 0 ? (unsigned char)t : 255; }
        return;
    }
    int sum[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, cnt[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int t = nkt - 1; t >= 0; --t) {
        int by = -1, bs = -1;
        for (int s4 = 0; s4 < 4; ++s4)
            for (int y = 0; y < 2; ++y) {
                if (cnt[y][s4] >= 2) continue;
                if (by < 0 || sum[y][s4] < sum[by][bs]) { by = y; bs = s4; }
            }
        if (by < 0) return;
        const int w = cnt[by][bs] == 0 ? bs : bs + 4;    // a SIMD's first (heavier) tile: the lower wave
        pt[by][w] = (unsigned char)t;
        ++cnt[by][bs];
        sum[by][bs] += t + 1 + fixed;
    }
}

static int stack_fwd_impl(const cr_stack_desc* s, const cr_head_desc* h, const cr_ln_bwd_desc* n, void* stream) {
    const char* why = h ? stack_head_unsupported(s, h, n) : stack_unsupported(s);
    CR_REQUIRE(why == nullptr, "%s: unsupported (%s)", h ? "cr_stack_fwd_head" : "cr_stack_fwd", why ? why : "");
    StackArgs a;
    memset(&a.hd, 0, sizeof(a.hd));
    memset(&a.hl, 0, sizeof(a.hl));
    const cr_attn_desc& a0 = s->attn[0];
    a.T16 = (a0.T + 15) / 16 * 16;
    a.nkt = a.T16 / 16;
    stack_deal_pair(a.nkt, a.ptile);
    a.isd_log2e = (float)(1.4426950408889634 / sqrt((double)a0.d));
    a.invT = 1.0f / (float)a0.T;
    a.ts = g_attn_ts_which == 7 ? g_attn_ts : nullptr;
    a.ld_out = s->ld_out; a.col_out = s->col_out;
    const bool split = a0.precision == CR_PREC_BF16X3;
    const bool pair = stack_pair_mode(s);
    hipStream_t st = cr_stream(stream);
    const int per = pair ? 1 : s->n_blocks;               // blocks per launch
    memset(&a.e, 0, sizeof(a.e));
    for (int i0 = 0; i0 < s->n_blocks; i0 += per) {
        const bool fin = i0 + per >= s->n_blocks && s->out != nullptr;
        a.nb = per;
        a.gather = (i0 == 0 && s->embed) ? 1 : 0;
        if (a.gather) a.e = *s->embed;
        a.lnf_g = fin ? s->lnf_gamma : nullptr; a.lnf_b = fin ? s->lnf_beta : nullptr;
        a.out = fin ? s->out : nullptr;
        for (int i = 0; i < CR_STACK_MAX_BLOCKS; ++i) {
            const int j = i < per ? i0 + i : i0;
            a.blk[i].bd = s->blocks[j];
            a.blk[i].ad = s->attn[j];
        }
        const bool head = h != nullptr && i0 + per >= s->n_blocks;      // the head rides on the LAST launch
        if (head) { a.hd = *h; a.hl = *n; }
        int rc = launch_stack_any(a, a0.B, split, pair, head, st);
        if (rc) return rc;
    }
    return CR_OK;
}

extern "C" int cr_stack_fwd(const cr_stack_desc* s, void* stream) { return stack_fwd_impl(s, nullptr, nullptr, stream); }

extern "C" int cr_stack_fwd_head(const cr_stack_desc* s, const cr_head_desc* h, const cr_ln_bwd_desc* n, void* stream) {
    CR_REQUIRE(h && n, "cr_stack_fwd_head: NULL head / LayerNorm description");
    return stack_fwd_impl(s, h, n, stream);
}
