// Backward of ONE transformer block in ONE launch (round 3): the work of cr_stack_ffn_bwd -> cr_attn_bwd -> cr_stack_qkv_bwd
// per sequence, without leaving the compute unit between them.  (sasrec.py:65-83 and autodiff of it; modules.py:167-318.)
//
// Everything between the gradient of a block's output and the gradient of its input is sequence-local; only the weight
// gradients couple sequences, and they leave as slabs.  A sequence gets a PAIR of workgroups on two compute units:
//
//   Q side (blockIdx.y = 1)                                   K side (blockIdx.y = 0, dispatched first: it is the longer one)
//   1  feed-forward + LN2 backward of ALL 16-row tiles        1  the same chain, data gradients only: d_o goes straight
//      (two rounds of <= 7 tiles): d_o -> global, delta ->       into the dOut IMAGE in LDS (bf16 hi + lo), the tile's Q
//      LDS; images of hid, g2, f_in, g1 -> dW2 db2 dW1 db1       rows into the Q image, delta and the forward's row
//      dgamma2 dbeta2 (+ the stack's final LayerNorm)            statistics into LDS vectors: no staging from memory
//   2  K (natural) / V (column-permuted) images staged;       2  key-owner pass of the attention backward on the images
//      query-owner pass of the attention backward for the        (own K / V tiles from memory); dK^T, dV^T come out of the
//      wave's tiles; dQ^T comes out of the swapped product       swapped products in layout R:  dx_kv = dK Wk^T + dV Wv^T
//      in layout R and goes on through registers:                -> dx2;  dK, dV -> global (for the weight gradients)
//      dq_in = dQ Wq^T + d_o, LN1 backward -> dx
//   3  images of q_in, dQ -> dWq dbq; dgamma1 dbeta1           3  images of x, dK, dV -> dWk dbk dWv dbv
//
// The gradient of the block input is the SUM dx + dx2 of the two sides' partials (no workgroup ever waits for another:
// nothing here depends on dispatch order or co-residency); the consumer -- this kernel for the block below (dy + dy2), the
// embedding scatter applied here by both sides to their partial, cr_embed_bwd (out2), or a cr_eltwise add -- adds them.
// Both sides run the feed-forward chain (2 of the block's 16 products per tile) rather than hand d_o across compute units.
//
// Layout R (cr_rlayout.hpp) carries a tile through every row-local layer.  For the attention products a lane's registers
// are an MFMA operand when the OTHER operand's image has its columns in the k order of layout R (V on the Q side, read by
// rows only: staged permuted like the weights); images that are also read transposed stay in natural column order and
// meet operands that come from memory in natural order (Q on the Q side; K, V on the K side).  A product whose result is
// wanted in layout R is issued with its operands swapped (out^T): the same two fragments, no shuffle.
//
// One gradient slab per SEQUENCE (slab n, the two sides write disjoint parameter ranges of it): B slabs instead of 2 B.
// Fixed summation order, no atomics except the embedding scatter: bitwise reproducible like the kernels it replaces.
// Shapes: one head, 8 <= D < 64 (bias gradients ride the ones column), T <= 224, bf16 arithmetic (split or plain).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <mutex>

#include "cr_rbwd.hpp"

// Global rows written by one wave and read by another wave of the workgroup later on (d_o, dQ / dK / dV, the scatter rows) cross a
// workgroup barrier; __syncthreads() itself only drains the LDS counter on this target (s_waitcnt lgkmcnt(0); s_barrier), so the
// writing wave drains its vector-memory counter explicitly first
#define B1_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// dQ / dK / dV rows (a workspace: written in phase 2, read back once in phase 3, dead behind the launch) leave as streaming stores: they do
// not stay behind as dirty L2 lines for the next launch to wait on (-1.1 us per step, A/B)
#define B1_NT_WS true
#define B1_ROWS 224                       // rows of an attention image (14 tiles of 16)
#define B1_FSTR (B1_ROWS * 64)            // bf16 elements of one image half

// The kernel's hidden-size parameter DS also carries, for the headline shape, the number of 16-row tiles as a CONSTANT: DS = D | nkt << 8
// (0 < D < 256).  Round counts, tile bounds and the guards of the unrolled weight-gradient products then fold, as the column-tile
// classification does with D.  DS < 0: the families of cr_rlayout.hpp d_ctx<NF>; 0: everything at run time.
#define DS_EXACT(DS) ((DS) > 0 ? ((DS) & 255) : (DS))
#define DS_D(DS, run) ((DS) > 0 ? ((DS) & 255) : (run))
#define B1_NKT(DS, a) ((DS) > 255 ? (((DS) >> 8) & 255) : (a).nkt)
#define B1_T(DS, a) ((DS) > 65535 ? ((DS) >> 16) : (a).T)          // ... and the sequence length: DS = D | nkt << 8 | T << 16
#undef D_NF
#define D_NF(DS) ((DS) > 0 ? ((DS) & 255) / 16 : ((DS) < 0 ? -(DS) - 1 : -1))

struct B1Args {
    cr_block_bwd_desc bd;
    cr_attn_desc ad;
    cr_embed_bwd_desc sc;                 // scatter: the embedding backward of the block input, applied to both partials
    cr_ln_bwd_desc ln;                    // has_ln: dy = backward of the stack's final LayerNorm applied to ln.dy (+ ln_dy2)
    const float* dy2;                     // optional second addend of dy
    float* dx2;                           // K side's partial of dx
    const float* ln_dy2;
    float* d_addend2;                     // scatter: K side's partial of d_addend
    float* sbuf; float* sbuf2;            // scatter: where the masked partial rows wait for phase 3 ([M, D] each)
    int B, T, nkt, scatter, has_ln;
    int small;                            // scatter: small-table mode: each side reduces its partial into an LDS image of the table and writes slab (side * gridDim.x + blockIdx.x)
    int n0, add;                          // this launch: sequences n0 .. n0 + gridDim.x - 1; add: the slabs already hold earlier sequences' sums
    float isd, isd_log2e, invT;
    unsigned qpk[8], kpk[8];              // tiles of wave w in the attention passes: two 5-bit tile numbers, 31 = none
    unsigned long long* ts;
};

// The kernel's argument block read again through an opaque pointer: what a late phase needs (output pointers, flags) is fetched from
// the scalar cache where it is used instead of being held in scalar registers across the loops in front of it -- held, the
// pointers were copied to vector registers and spilled to scratch (a reload + full vector-memory wait in front of the row chain)
__device__ __forceinline__ const B1Args& b1_args_again() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *reinterpret_cast<const B1Args*>((const void*)p);
}

#ifdef CR_TIMELINE
#define B1_TS(slot)                                                                                          \
    do {                                                                                                     \
        if (a.ts && (threadIdx.x & 63) == 0) {                                                               \
            unsigned long long* ts_ = a.ts + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SB_WAVES + (threadIdx.x >> 6)) * 32; \
            ts_[(slot)] = ((slot) == 0 || (slot) == 31) ? wall_clock64() : clock64();                        \
            if ((slot) == 0) ts_[30] = clock64();         /* both clocks at the ends: the shader clock's rate */ \
            if ((slot) == 31) ts_[29] = clock64();                                                           \
        }                                                                                                    \
    } while (0)
#else
#define B1_TS(slot) do { } while (0)
#endif

// LDS carve-up (bytes from the start; SPLIT: 158 KB).  The image area comes first: the attention loops address it as
// (per-lane base + pair offset) + immediate.
template <bool SPLIT>
struct B1Lds {
    static constexpr int NIMG = SPLIT ? 4 : 2;                     // image halves in the area
    static constexpr int WST = SPLIT ? 2 * ST_WIMG : ST_WIMG;      // elements of a weight slot
    static constexpr int IST = SPLIT ? 2 * SB_IMG : SB_IMG;        // elements of a weight-gradient image slot (4 slots = the area)
    static constexpr int MATB = (SPLIT ? 2 : 1) * B1_FSTR * 2;     // bytes from the first matrix's images to the second's
    static constexpr int LOB = B1_FSTR * 2;                        // bytes from a hi image to its lo image
    static constexpr int W_OFF = NIMG * B1_FSTR;                   // elements
    static constexpr int F_OFF_BYTES = (W_OFF + 2 * WST) * 2;
    // float vectors behind the weights.  Per-head vectors (delta, m') come twice (two heads of 32 columns at D = 64).  A workgroup is
    // ONE side: the query side's key bias (KB) and the key side's row vectors (SMX .. TFLAG) are the same floats.
    static constexpr int GAM = 0, GAMF = 64, PART = 128, PARTF = PART + 2 * SB_WAVES * 64, SDEL = PARTF + 2 * SB_WAVES * 64,
                         KB = SDEL + 2 * B1_ROWS, SMX = KB, SUNI = SMX + 2 * B1_ROWS, SQV = SUNI + B1_ROWS,
                         TFLAG = SQV + B1_ROWS, NFLOAT = TFLAG + 16;
    static_assert(SUNI >= KB + B1_ROWS + 16, "the key bias (one extra tile) inside the shared floats");
    static constexpr int BYTES = F_OFF_BYTES + NFLOAT * 4;
};

// accumulators D[in = 16 it + 4 lg + r][out = 16 (jt0 + j) + li] -> slab (row pitch ldw), row D = the bias gradient; `add`: the
// workgroup's second and later sequences add to what its first one stored
// BIAS (D == 64: no spare column for the ones trick): the bias gradient comes from accb (row 0 of the all-ones product: wgrad_accum)
template <bool BIAS = false>
__device__ __forceinline__ void b1_wstore(float* dst, int ldw, float* bias_dst, const f32x4 (&acc)[2], const f32x4 (&accb)[2], int D, int it, int jt0, bool add) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = 16 * (jt0 + j) + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * it + 4 * lg + r;
            if (col < D && (BIAS ? k < D : k <= D)) {
                float* p = (k < D) ? dst + (size_t)k * ldw + col : bias_dst + col;
                __builtin_nontemporal_store(add ? *p + acc[j][r] : acc[j][r], p);    // (a streaming store: the slab is read by Adam, launches later)
            }
        }
        if (BIAS && it == 0 && lg == 0 && col < D) bias_dst[col] = add ? bias_dst[col] + accb[j][0] : accb[j][0];
    }
}
// LayerNorm column sums: per-lane partials (the lane's row, 16 columns, two arrays) -> sums over the wave's 16 rows -> the wave's LDS
// slot; b1_ln_flush adds the slots in a fixed order.  The 32 values of a lane are reduced over the 16 lanes of its DPP row by a
// TRANSPOSING butterfly: every step halves the values a lane carries (it keeps one half and adds the partner's copy of that half), so
// the four steps cost 32 + 16 + 12 + 6 vector instructions instead of 32 full row sums of four DPP adds each, and every lane ends
// with two adjacent columns' totals: one 8-byte LDS write per lane, no lane predicate.  (The first form -- 32 row sums, 32 single-lane
// read-modify-writes of the slot -- took 1.4-2.2 us per tile on the query side's row chain: tools/b1_ts.py.)
//   step 1: partner 15 - i (row_mirror), kept by bit 3 of the lane: the sums with xhat (ag) / the plain sums (ab)
//   step 2: partner 7 - i of the half (row_half_mirror), kept by bit 2: column tiles 0, 1 / 2, 3
//   steps 3, 4: partner 3 - i, then i ^ 1 of the quad, kept by bits 1, 0: the column tile of the two, registers 0, 1 / 2, 3
// Steps 1 and 2 are `v_add_f32_dpp` with a bank mask (banks = the row's four quads: exactly bits 3 / 2 of the lane), in place; the
// compiler does not form masked DPP adds, hence the assembly (s_nop 1: a DPP source written by the previous instruction needs two
// wait states, and the hazard recogniser does not look inside an asm statement).
// Every (wave, tile-of-the-wave) has a slot of its own: slots are WRITTEN (no zeroing pass in front, no read-modify-write); a slot that
// no fold reaches must have been zeroed by the caller.
__device__ __forceinline__ void b1_ln_fold(float* part, const f32x4 (&ag)[4], const f32x4 (&ab)[4]) {
    const int lane = lane_now(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lg = lane >> 4;
    float u[16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = ag[ct][r];
            asm volatile("s_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0x3\n\t"
                         "v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc"
                         : "+v"(x) : "v"(ab[ct][r]));
            u[4 * ct + r] = x;
        }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa"
                     : "+v"(u[j]) : "v"(u[8 + j]));
    const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    float w[4], z[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = b1 ? u[4 + j] : u[j], send = b1 ? u[j] : u[4 + j];
        w[j] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x1B, 0xF, 0xF, false));     // quad_perm [3,2,1,0]
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = b0 ? w[2 + j] : w[j], send = b0 ? w[j] : w[2 + j];
        z[j] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
    }
    // the lane's two totals: array = bit 3, column 16 (2 bit2 + bit1) + 4 lg + 2 bit0 (+ 1)
    const int col = 16 * (((lane >> 2) & 1) * 2 + ((lane >> 1) & 1)) + 4 * lg + 2 * (lane & 1);
    float* dst = part + ((lane & 8) ? SB_WAVES * 64 : 0) + wave * 64 + col;
    *reinterpret_cast<float2*>(dst) = make_float2(z[0], z[1]);
}
// NSET slot sets ([2][SB_WAVES][64] floats each, one behind the other)
template <int NSET = 1>
__device__ __forceinline__ void b1_ln_flush(const float* part, float* dg, float* db, int D, bool add) {
    for (int c = threadIdx.x; c < D; c += SB_NT) {
        float g = 0.0f, b = 0.0f;
#pragma unroll
        for (int s = 0; s < NSET; ++s)
#pragma unroll
            for (int w = 0; w < SB_WAVES; ++w) { g += part[(2 * s * SB_WAVES + w) * 64 + c]; b += part[((2 * s + 1) * SB_WAVES + w) * 64 + c]; }
        dg[c] = add ? dg[c] + g : g;
        db[c] = add ? db[c] + b : b;
    }
}

// =====================================================================================================
// phase 1: LN2 + feed-forward backward of every tile of sequence n (both sides)
// =====================================================================================================
// the step counter behind each dropout site's key (wave-uniform; requested at the kernel's start, see drop_ctx)
struct B1Steps { uint32_t ffn2, attn, emb; };
struct B1Acc { f32x4 aw1[2], aw2[2], aw1b[2], aw2b[2], ag[4], ab[4], agF[4], abF[4]; };   // Q side: what phase 1 leaves in registers (aw?b: bias sums at D = 64)

// before_last_products: called by the query side in its last round between the chain and the barrier in front of the round's weight
// gradients (the phase-2 staging's loads go out there: their latency passes under the products)
template <bool SPLIT, int DS, bool QSIDE, int HD, class F>
__device__ __forceinline__ void b1_phase1(const B1Args& a, unsigned char* smem, int n, B1Acc& A, B1Steps& steps, F&& before_last_products) {
    constexpr bool BIAS = DS_EXACT(DS) == 64;
    constexpr bool LATE_Q = !QSIDE && (HD == 2 || DS_EXACT(DS) > 50 || DS <= 0);       // key side: the tile's Q rows are requested late (registers)
    typedef B1Lds<SPLIT> L;
    const cr_block_bwd_desc& bd = a.bd;
    const cr_block_desc& d = bd.f;
    __bf16* Im = reinterpret_cast<__bf16*>(smem);
    __bf16* Wi = Im + L::W_OFF;                           // slot 0: W1 (permuted), slot 1: W2
    float* fl = reinterpret_cast<float*>(smem + L::F_OFF_BYTES);
    float* gam = fl + L::GAM; float* gamF = fl + L::GAMF; float* part = fl + L::PART; float* partF = fl + L::PARTF; float* sdel = fl + L::SDEL;
    constexpr int WST = L::WST, IST = L::IST;
    const int D = DS_D(DS, d.D), T = B1_T(DS, a);
    const DCtx dcx = d_ctx<D_NF(DS)>(D);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the step counters: requested first of all, used behind the barrier that opens the phase
    uint32_t sv_ffn2 = cr_step_request(d.drop_ffn2), sv_attn = cr_step_request(a.ad.drop), sv_emb = a.scatter ? cr_step_request(a.sc.f.drop) : 0u;
    const float scale1 = (d.drop_ffn1.rate > 0.0f) ? 1.0f / (1.0f - d.drop_ffn1.rate) : 1.0f;
    B1_TS(13);
    f32x4 (&aw1)[2] = A.aw1; f32x4 (&aw2)[2] = A.aw2; f32x4 (&ag)[4] = A.ag; f32x4 (&ab)[4] = A.ab; f32x4 (&agF)[4] = A.agF; f32x4 (&abF)[4] = A.abF;
    f32x4 (&aw1b)[2] = A.aw1b; f32x4 (&aw2b)[2] = A.aw2b;
#pragma unroll
    for (int j = 0; j < 2; ++j) { aw1[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; aw2[j] = aw1[j]; aw1b[j] = aw1[j]; aw2b[j] = aw1[j]; }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { ag[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; ab[ct] = ag[ct]; agF[ct] = ag[ct]; abF[ct] = ag[ct]; }
    const int it = wave >> 1, jt0 = 2 * (wave & 1);
    RRaw rdy, rdy2, rhid, rfin, ro, rq, ry;               // rq: q_in (delta); rfin: f_in (Q side) / the tile's Q rows (K side)
    const size_t MD = (size_t)d.M * D;
    auto tile_rows = [&](int rd, int& m, bool& rok) {
        const int q = 16 * (rd * SB_TPR + wave) + (lane_now() & 15);
        rok = q < T;
        m = n * T + min(q, T - 1);
    };
    // the rows of a tile in two halves: what the chain opens with (dy, dy2, y, hid), and what its end needs (f_in / Q, o, q_in)
    auto issue_a = [&](int rd) {
        if (rd * SB_TPR < B1_NKT(DS, a) && wave < min(SB_TPR, B1_NKT(DS, a) - rd * SB_TPR)) {
            int m; bool rok;
            tile_rows(rd, m, rok);
            const u32 mo = (u32)m * (u32)(4 * D);
            if (a.has_ln) {                                          // rows beyond T: zero gradient (they must not reach the LayerNorm sums)
                r_issue(rdy, a.ln.dy, (u32)m * (u32)(4 * a.ln.lddy), dcx, rok);
                if (a.ln_dy2) r_issue(rdy2, a.ln_dy2, (u32)m * (u32)(4 * a.ln.lddy), dcx, rok);
                r_issue(ry, d.y, mo, dcx);
            } else {
                r_issue(rdy, bd.dy, mo, dcx);
                if (a.dy2) r_issue(rdy2, a.dy2, mo, dcx);
            }
            r_issue(rhid, d.hid, mo, dcx);
        }
    };
    auto issue_b = [&](int rd) {
        if (rd * SB_TPR < B1_NKT(DS, a) && wave < min(SB_TPR, B1_NKT(DS, a) - rd * SB_TPR)) {
            int m; bool rok;
            tile_rows(rd, m, rok);
            const u32 mo = (u32)m * (u32)(4 * D);
            if (!LATE_Q) r_issue(rfin, QSIDE ? d.f_in : d.qkv, mo, dcx);     // (LATE_Q: requested inside the chain, below)
            r_issue(ro, d.o, mo, dcx);
            r_issue(rq, d.q_in, mo, dcx);
        }
    };
    auto issue = [&](int rd) { issue_a(rd); issue_b(rd); };
    const int R = B1_NKT(DS, a) > SB_TPR ? 2 : 1;
    {
        // (memory returns in order: the weights are requested first, or their staging would wait for the tile as well)
        WRegs<2, SB_NT> w;
        w_issue<2, SB_NT>(w, D, d.w1, D, 0, d.w2, D, 0, d.w2, D, 0);
        B1_TS(14);
        const int t = threadIdx.x;
        const float gv = (t < D) ? d.ln2_g[t] : 0.0f;
        const float gf = (a.has_ln && t < D) ? a.ln.gamma[t] : 0.0f;
        // (only the first half of the tile's rows in front of the weights' staging: the CU's address pipeline takes ~1.6 us to
        //  issue 7 waves x 24 row loads, and the waves that get theirs out last hold the barrier below: tools/b1_ts.py)
        issue_a(0);
        B1_TS(11);
        w_put_perm<2, SB_NT, SPLIT>(Wi, w, D, d.w1, D, 0, d.w2, D, 0, d.w2, D, 0);
        B1_TS(12);
        if (t < 64) { gam[t] = gv; gamF[t] = gf; }
        if (!QSIDE && (B1_NKT(DS, a) & 1)) {
            // an odd tile count: the key-owner pass reads the absent second tile of the last pair (with zero coefficients):
            // its rows must hold finite values (16 rows x 128 bytes per image half, 16 bytes per thread)
            const int im = t >> 7, o16 = t & 127;
            if (im < L::NIMG) *reinterpret_cast<float4*>(smem + (size_t)im * L::LOB + (size_t)(16 * B1_NKT(DS, a)) * 128 + 16 * o16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    B1_TS(1);
    issue_b(0);                                                    // ... the second half behind it, under the chain's first product
    steps.ffn2 = __builtin_amdgcn_readfirstlane(sv_ffn2); steps.attn = __builtin_amdgcn_readfirstlane(sv_attn); steps.emb = __builtin_amdgcn_readfirstlane(sv_emb);
    const DropCtx d2 = drop_ctx(d.drop_ffn2, steps.ffn2);
#pragma unroll 1
    for (int rd = 0; rd < R; ++rd) {
        const int ntr = min(SB_TPR, B1_NKT(DS, a) - rd * SB_TPR);          // tiles of this round (wave-uniform)
        if (wave < ntr) {
            const int lg = lane_now() >> 4;
            int m; bool rok;
            tile_rows(rd, m, rok);
            const u32 mo = (u32)m * (u32)(4 * D);
            const int trow = 16 * (rd * SB_TPR + wave);              // the tile's first row in the sequence
            const float msk = (rok && d.mask_ids[m] != 0) ? 1.0f : 0.0f;    // rows beyond T contribute nothing
            f32x4 dy[4], g2[4], hid[4];
            r_finish(dy, rdy, dcx);
            if (a.has_ln ? (a.ln_dy2 != nullptr) : (a.dy2 != nullptr)) {
                f32x4 t2[4];
                r_finish(t2, rdy2, dcx);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) dy[ct] += t2[ct];
            }
            if (a.has_ln) {
                // dy = backward of the stack's final LayerNorm (sasrec.py:85) on the gradient rows, x = this block's y
                f32x4 yv[4], dyo[4];
                r_finish(yv, ry, dcx);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) dyo[ct] = dy[ct];
                r_ln_bwd(dy, yv, dyo, gamF, agF, abF, dcx);
            }
            // g2 = dy * mask * keep2 / (1 - rate) (sasrec.py:83, modules.py:309-310)
            const uint32_t e2 = ((d.drop_ffn2.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)(4 * lg)) * CR_PHI + d2.key;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dy[ct][r] *= msk;
                    float v = dy[ct][r];
                    v *= drop_factor_x(d2, e2 + (uint32_t)(16 * ct + r) * CR_PHI);      // (rate 0: threshold 0, factor 1.0 -- no branch per element)
                    g2[ct][r] = v;
                }
            r_finish(hid, rhid, dcx);
            if (QSIDE) {
                img_put<SPLIT>(Im + IST, Im + IST + SB_IMG, 16 * wave, g2);
                f32x4 h1[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) h1[ct] = hid[ct];
                plant_one(h1, D);
                img_put<SPLIT>(Im, Im + SB_IMG, 16 * wave, h1);
            }
            // dhid = g2 W2^T, gated by the stored post-dropout ReLU output -> g1 (modules.py:300-304)
            bf8 gh[2], gl[2];
            f32x4 g1[4];
            // (the tile's Q rows, key side at the widest shapes: held from the round's start they did not fit -- two of their four pieces
            //  went to scratch with a full wait each; requested here they fly under the first product)
            if (LATE_Q) r_issue(rfin, d.qkv, mo, dcx);
            r_split<SPLIT>(g2, gh, gl);
            r_gemm_t<SPLIT, false>(g1, Wi + WST, Wi + WST + ST_WIMG, gh, gl);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) g1[ct][r] = (hid[ct][r] > 0.0f) ? g1[ct][r] * scale1 : 0.0f;
            if (QSIDE) {
                img_put<SPLIT>(Im + 3 * IST, Im + 3 * IST + SB_IMG, 16 * wave, g1);
                f32x4 fin[4];
                r_finish(fin, rfin, dcx);
                plant_one(fin, D);
                img_put<SPLIT>(Im + 2 * IST, Im + 2 * IST + SB_IMG, 16 * wave, fin);
            } else {
                // the tile's Q rows -> the Q image (natural column order; rows beyond T are zero)
                f32x4 qv[4];
                r_finish(qv, rfin, dcx);
                if (!rok) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) qv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                img_put<SPLIT, true>(Im, Im + B1_FSTR, trow, qv);
            }
            // df = g1 W1^T + dy * mask (residual of modules.py:313)
            f32x4 df[4];
            r_split<SPLIT>(g1, gh, gl);
            r_gemm_t<SPLIT, false>(df, Wi, Wi + ST_WIMG, gh, gl);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) df[ct] += dy[ct];
            // LN2 backward: x = o, dy = df -> d_o
            f32x4 o[4], dout[4];
            r_finish(o, ro, dcx);
            r_ln_bwd(dout, o, df, gam, ag, ab, dcx);
            if (QSIDE) {
                r_store(bd.d_o, mo, dout, rok, dcx);                       // (a plain store: other waves read it back within microseconds; streaming: +0.5 us)
            } else {
                if (!rok) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) dout[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                img_put<SPLIT, true>(Im + (L::MATB >> 1), Im + (L::MATB >> 1) + B1_FSTR, trow, dout);
            }
            {
                // delta[row] = sum_c d_o[c] * (o[c] - q_in[c]) (the attention core's output is o - q_in, modules.py:262-269)
                f32x4 qin[4];
                r_finish(qin, rq, dcx);
                float acc[HD];
#pragma unroll
                for (int h = 0; h < HD; ++h) acc[h] = 0.0f;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[HD == 2 ? ct >> 1 : 0] = fmaf(dout[ct][r], o[ct][r] - qin[ct][r], acc[HD == 2 ? ct >> 1 : 0]);   // (a head: 32 columns)
#pragma unroll
                for (int h = 0; h < HD; ++h) {
                    acc[h] = grp_sum(acc[h]);
                    if (lg == 0) sdel[h * B1_ROWS + trow + (lane_now() & 15)] = rok ? acc[h] : 0.0f;
                }
            }
        }
        if (rd + 1 < R) {
            B1_TS(24);
            // (the next round's rows are requested in FRONT of the barrier: a CU's 7 x 35 requests are 1.6 us of its address pipeline and
            //  block the issuing wave wherever they stand; here the waves that end their chains first issue while the others compute.
            //  Behind the barrier -- all waves at once -- the round came out 0.6 us longer: tools/b1_ts.py.)
            issue(rd + 1);
            if (QSIDE) {
                __syncthreads();
                B1_TS(25);
                wgrad_accum<SPLIT, BIAS>(aw2, aw2b, Im, Im + SB_IMG, Im + IST, Im + IST + SB_IMG, ntr, it, jt0);                          // dW2 (+ db2) += hid^T g2
                wgrad_accum<SPLIT, BIAS>(aw1, aw1b, Im + 2 * IST, Im + 2 * IST + SB_IMG, Im + 3 * IST, Im + 3 * IST + SB_IMG, ntr, it, jt0);  // dW1 (+ db1) += f_in^T g1
                B1_TS(26);
                __syncthreads();
                B1_TS(27);
            }
        }
    }
    // the last round's products stand behind the loop: what the caller requests in front of them (the query side: phase 2's K / V
    // rows and Wq) has its addresses formed HERE, not hoisted in front of the round loop and carried through the chains
    B1_TS(28);
    before_last_products();
    if (QSIDE) {
        const int ntr = min(SB_TPR, B1_NKT(DS, a) - (R - 1) * SB_TPR);
        __syncthreads();
        B1_TS(10);
        wgrad_accum<SPLIT, BIAS>(aw2, aw2b, Im, Im + SB_IMG, Im + IST, Im + IST + SB_IMG, ntr, it, jt0);
        wgrad_accum<SPLIT, BIAS>(aw1, aw1b, Im + 2 * IST, Im + 2 * IST + SB_IMG, Im + 3 * IST, Im + 3 * IST + SB_IMG, ntr, it, jt0);
        __syncthreads();
    }
    B1_TS(2);
    if (QSIDE) {
        // d_o rows are read back in phase 2 by OTHER waves (the tiles are dealt differently there): this wave's stores are drained
        // here and a barrier follows (the Q side's staging block ends with one) before any of those reads
        B1_DRAIN();
    }
    (void)MD; (void)part; (void)partF;
}

// ---- the embedding backward of a partial of dx, phase-2 half: g = partial * mask * keep / (1 - rate) in the registers ----
// (cr_embed_bwd's recipe; the rows then wait in `buf` -- d_addend where the graph has one, else the dx buffer -- for phase 3)
__device__ __forceinline__ void b1_scatter_prep(const B1Args& a, f32x4 (&dxl)[4], int m, bool rok, int D, uint32_t step_emb) {
    const cr_embed_desc& e = a.sc.f;
    const DropCtx dce = drop_ctx(e.drop, step_emb);
    const int lg = lane_now() >> 4;
    const int mk = e.mask_ids ? e.mask_ids[m] : 1;
    const float kf = (rok && mk != 0) ? 1.0f : 0.0f;
    const uint32_t eb = ((e.drop.row_offset + (uint32_t)m) * (uint32_t)D + (uint32_t)(4 * lg)) * CR_PHI + dce.key;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = dxl[ct][r] * kf;
            v *= drop_factor_x(dce, eb + (uint32_t)(16 * ct + r) * CR_PHI);
            dxl[ct][r] = v;
        }
}
// last thing a side does: the waiting rows of sequence n, lane = column: one contiguous float-atomic burst per table row.
// A wave takes 16 rows at a time: their ids and gradient rows are requested together, then the atomics go out back to back
// and the wave ends with them in flight (nothing in the kernel waits behind them: placed in front of phase 3 they held its
// loads back for the ~3000 clocks an atomic stays in the memory queue: +17 us on the block that scatters).
// Round 4, measured on the launch that carries it (49.6 us; 43.7 with this function returning at once): the 6 us are the atomics
// themselves -- 1.28 M per side at about one dword per L2 channel and clock -- not this function's two dependent round trips (ids and
// rows requested in front of phase 3's last products: 50.9 us) and not their place (sent in front of those products: no change).
__device__ __forceinline__ void b1_scatter_rows(const B1Args& a, const float* buf, int n, int D) {
    if (!a.sc.table_grad && !a.sc.pos_grad) return;
    const cr_embed_desc& e = a.sc.f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col = threadIdx.x & 63, T = a.T;
    for (int t0 = 16 * wave; t0 < T; t0 += 16 * SB_WAVES) {
        const int nr = min(16, T - t0);
        const int m0 = n * T + t0;
        const int my_id = e.ids[m0 + min(col & 15, nr - 1)];             // lanes 0..15: the id of row t0 + lane
        float g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = (r < nr && col < D) ? buf[(size_t)(m0 + r) * D + col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int id = __shfl(my_id, r, 64);
            if (r < nr && col < D) {
                if (a.sc.table_grad && !(e.zero_pad && id == 0)) atomicAdd(a.sc.table_grad + (size_t)id * D + col, g[r] * e.scale);
                if (a.sc.pos_grad) atomicAdd(a.sc.pos_grad + (size_t)((m0 + r) % e.T) * D + col, g[r]);
            }
        }
    }
}

// Small-table form of the scatter (context tables of 8 .. 256 rows: cr_embed_bwd's small-table mode, its contract): thousands of
// rows land in a handful of table rows, so float atomics would serialise on hot rows.  Each side forms its partial's table
// gradient as ONE matrix product, tab = OneHot^T G: G = the waiting rows of the sequence ([T][D], bf16 hi + lo images in layout W,
// read transposed as the B operand), OneHot[row][id] = (ids[row] == id) built in registers from the ids in LDS as the A operand
// (exact in bf16), k = the sequence's rows.  Wave w owns table rows 16 w .. and 16 (w + 8) ..; the accumulators go straight to the
// side's slab: slab blockIdx.x (Q side) or gridDim.x + blockIdx.x (K side).  Runs last: the image area is free.
// (The first form -- every table row owned by one wave, which added its rows to an LDS image of the table by read-add-write, then
//  the image copied to the slab -- cost 9 us of the launch: 4.4 of them the serial chain of LDS updates; ds_add_f32 was slower
//  still, ~1100 clocks per atomic.  G in two bf16 terms: 2^-17 relative per element, the arithmetic class of the weight gradients.)
// (The rows and ids are requested by b1_small_issue in front of phase 3's products: they are the side's own partial, complete and
//  visible behind the barrier that ends the attention loops.)
struct B1Small { RRaw r0, r1; int my_id; };
template <int NF>
__device__ __forceinline__ void b1_small_issue(B1Small& q, const B1Args& a, const float* buf, int n, int D) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), T = a.T, li = lane_now() & 15;
    const DCtx dcx = d_ctx<NF>(D);
    const int tt0 = wave, tt1 = wave + SB_WAVES;          // the wave's rows: tiles w and w + 8
    if (tt0 < a.nkt) r_issue(q.r0, buf, (u32)(n * T + min(16 * tt0 + li, T - 1)) * (u32)(4 * D), dcx, 16 * tt0 + li < T);
    if (tt1 < a.nkt) r_issue(q.r1, buf, (u32)(n * T + min(16 * tt1 + li, T - 1)) * (u32)(4 * D), dcx, 16 * tt1 + li < T);
    q.my_id = ((int)threadIdx.x < T) ? a.sc.f.ids[n * T + threadIdx.x] : -1;
}
template <bool SPLIT, int NF>
__device__ __forceinline__ void b1_small_table(const B1Args& a, unsigned char* smem, int* ids_lds, B1Small& q, int n, int D, bool add) {
    const cr_embed_desc& e = a.sc.f;
    __bf16* Gh = reinterpret_cast<__bf16*>(smem);
    __bf16* Gl = Gh + B1_FSTR;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nkt = a.nkt;
    const DCtx dcx = d_ctx<NF>(D);
    RRaw& r0 = q.r0; RRaw& r1 = q.r1;
    const int tt0 = wave, tt1 = wave + SB_WAVES;
    const int t = threadIdx.x;
    const int my_id = q.my_id;
    __syncthreads();                                      // the weight-gradient images are dead
    B1_TS(20);
    if (t < B1_ROWS) ids_lds[t] = my_id;                  // rows beyond T: no table row
    {
        f32x4 g[4];
        // (tiles beyond the sequence's last one, up to the 14 the k loop may touch, are written as zeros: 0 x NaN is NaN in the matrix pipe)
        if (tt0 < 2 * SB_TPR) {
            if (tt0 < nkt) r_finish(g, r0, dcx);
            else { for (int ct = 0; ct < 4; ++ct) g[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            img_put<SPLIT>(Gh, Gl, 16 * tt0, g);
        }
        if (tt1 < 2 * SB_TPR) {
            if (tt1 < nkt) r_finish(g, r1, dcx);
            else { for (int ct = 0; ct < 4; ++ct) g[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            img_put<SPLIT>(Gh, Gl, 16 * tt1, g);
        }
    }
    __syncthreads();
    B1_TS(21);
    const int lane = lane_now(), li = lane & 15, lg = lane >> 4;
    const int nit = (e.V + 15) >> 4;                      // table-row tiles (<= 16)
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the lane's table row per tile; the padding row takes nothing (cr_embed_bwd: zero_pad)
    int target[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        target[i] = 16 * (wave + SB_WAVES * i) + li;
        if (e.zero_pad && target[i] == 0) target[i] = -2;
    }
    if (wave < nit) {
        const int nks = (nkt + 1) >> 1;                   // k-steps of 32 rows
#pragma unroll
        for (int ks = 0; ks < SB_TPR; ++ks) {             // (unrolled under a wave-uniform guard: several steps' reads in flight)
            if (ks >= nks) break;
            // k slot j of lane group lg: row 32 ks + 4 lg + j (j < 4), row 32 ks + 16 + 4 lg + (j - 4) -- the order of the transposed reads
            const int4 ia = *reinterpret_cast<const int4*>(ids_lds + 32 * ks + 4 * lg);
            const int4 ib = *reinterpret_cast<const int4*>(ids_lds + 32 * ks + 16 + 4 * lg);
            const int idv[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
            bf8 bh[4], bl[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const bf4 h0 = tr4(Gh, 32 * ks, ct, lane), h1 = tr4(Gh, 32 * ks + 16, ct, lane);
                bh[ct] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                if (SPLIT) {
                    const bf4 l0 = tr4(Gl, 32 * ks, ct, lane), l1 = tr4(Gl, 32 * ks + 16, ct, lane);
                    bl[ct] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (wave + SB_WAVES * i < nit) {          // (wave-uniform)
                    bf8 oh;
                    bool hit = false;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const bool e1 = idv[j] == target[i]; hit |= e1; oh[j] = e1 ? (__bf16)1.0f : (__bf16)0.0f; }
                    // (32 rows hold at most 32 ids: most of the 13 table-row tiles meet none of them -- an all-zero operand adds exactly nothing)
                    if (__any(hit ? 1 : 0) == 0) continue;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {              // tab^T tile: D[column 16 ct + 4 lg + r][table row 16 it + li] = layout R of the table's rows
                        if (SPLIT) acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[ct], oh, acc[i][ct], 0, 0, 0);
                        acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ct], oh, acc[i][ct], 0, 0, 0);
                    }
                }
            }
        }
    }
    B1_TS(22);
    B1_TS(23);
    // the accumulators are the table's rows in layout R (lane = table row 16 it + li): the slab ([V][D], every entry written: rows
    // without an occurrence get 0) is stored like a tile of rows
    float* slab = a.sc.table_grad + (size_t)((blockIdx.y == 0 ? gridDim.x : 0) + blockIdx.x) * a.sc.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int it = wave + SB_WAVES * i;
        if (it < nit) {
            const int row = 16 * it + li;
            const bool rok = row < e.V;
            const u32 ro = (u32)min(row, e.V - 1) * (u32)(4 * D);
            if (add) {
                RRaw ro_;
                f32x4 old[4];
                r_issue(ro_, slab, ro, dcx, rok);
                r_finish(old, ro_, dcx);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[i][ct] = old[ct] + acc[i][ct] * e.scale;
            } else {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[i][ct] *= e.scale;
            }
            r_store(slab, ro, acc[i], rok, dcx);               // (NOT a streaming store: Adam reads this slab right behind the launch)
        }
    }
    (void)lg;
}

// =====================================================================================================
// Q side, phases 2 and 3: query-owner pass (dQ), LN1 + Q projection backward, dWq
// =====================================================================================================
// K rows natural, V rows in the k order of layout R (w_put_perm's column map), hi (+ lo) images; the additive key bias.
// Issue and put are apart: the loads fly under the stores and the LayerNorm fold that end phase 1.
struct B1Stage { float va[4][8], vb[4][8]; float kv0; };
__device__ __forceinline__ void b1_stage_kv_issue(B1Stage& r, const B1Args& a, int base_row, int T16, int D, int M) {
    const cr_attn_desc& d = a.ad;
    const int T = a.T, total = T16 * 8;
    r.kv0 = d.k_valid[base_row + min((int)threadIdx.x, T - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int item = min(tid_now() + u * SB_NT, total - 1);
        const int row = item >> 3, ch = item & 7;
        const bool rok = row < T;
        const int grow = base_row + (rok ? row : 0);
        const bool fix = item_fix(rok, grow == M - 1, 8 * ch, D);
        item_issue(r.va[u], d.K + (size_t)grow * d.ld, 8 * ch, D, fix);
        item_issue(r.vb[u], d.V + (size_t)grow * d.ld, 8 * ch, D, fix);
    }
}
template <bool SPLIT>
__device__ __forceinline__ void b1_stage_kv_put(B1Stage& r, unsigned char* smem, float* kb, const B1Args& a, int base_row, int T16, int D, int M) {
    typedef B1Lds<SPLIT> L;
    const cr_attn_desc& d = a.ad;
    __bf16* Kh = reinterpret_cast<__bf16*>(smem);
    __bf16* Vh = reinterpret_cast<__bf16*>(smem + L::MATB);
    const int T = a.T, total = T16 * 8;                  // T16 <= 224 rows: 1792 items, four per thread
    const bool wg_has_last = base_row + T == M;          // only the last sequence can meet the matrix's last row
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int item = tid_now() + u * SB_NT;
        if (item < total) {
            const int row = item >> 3, ch = item & 7;
            const bool rok = row < T;
            const bool fix = wg_has_last && item_fix(rok, row == T - 1, 8 * ch, D);
            item_mask(r.va[u], 8 * ch, D, rok, fix);
            item_mask(r.vb[u], 8 * ch, D, rok, fix);
            if (__builtin_expect(wg_has_last && fix, 0)) {           // one thread of the grid
                item_refill(r.va[u], d.K + (size_t)(M - 1) * d.ld, 8 * ch, D);
                item_refill(r.vb[u], d.V + (size_t)(M - 1) * d.ld, 8 * ch, D);
            }
            bf8 h, l;
            split8<SPLIT>(r.va[u], h, l);
            const int o = img_off<2>(row, ch);
            *reinterpret_cast<bf8*>(Kh + o) = h;
            if (SPLIT) *reinterpret_cast<bf8*>(Kh + B1_FSTR + o) = l;
            split8<SPLIT>(r.vb[u], h, l);
            const int ks = ch >> 2, c4 = ch & 3, hh = c4 >> 1, lga = 2 * (c4 & 1);
            const int oa = img_off<2>(row, 4 * ks + lga) + 4 * hh, ob = img_off<2>(row, 4 * ks + lga + 1) + 4 * hh;
            *reinterpret_cast<bf4*>(Vh + oa) = __builtin_shufflevector(h, h, 0, 1, 2, 3);
            *reinterpret_cast<bf4*>(Vh + ob) = __builtin_shufflevector(h, h, 4, 5, 6, 7);
            if (SPLIT) {
                *reinterpret_cast<bf4*>(Vh + B1_FSTR + oa) = __builtin_shufflevector(l, l, 0, 1, 2, 3);
                *reinterpret_cast<bf4*>(Vh + B1_FSTR + ob) = __builtin_shufflevector(l, l, 4, 5, 6, 7);
            }
        }
    }
    const int t0 = threadIdx.x;
    if (t0 < T16) kb[t0] = (t0 < T && r.kv0 != 0.0f) ? 0.0f : -INFINITY;
    if (a.nkt & 1) {                                     // the absent second tile of the last pair: finite (zero) rows, masked keys
        const int im = t0 >> 7, o16 = t0 & 127;
        if (im < L::NIMG) *reinterpret_cast<float4*>(smem + (size_t)im * L::LOB + (size_t)T16 * 128 + 16 * o16) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t0 < 16) kb[T16 + t0] = -INFINITY;
    }
}

template <int DS>
__device__ __forceinline__ void b1_q_stage_issue(const B1Args& a, int n, B1Stage& st, WRegs<1, SB_NT>& w) {
    const cr_block_desc& bk = a.bd.f;
    const int D = DS_D(DS, bk.D), T = B1_T(DS, a);
    w_issue<1, SB_NT>(w, D, bk.wqkv, 3 * D, 0, bk.wqkv, 3 * D, 0, bk.wqkv, 3 * D, 0);
    b1_stage_kv_issue(st, a, n * T, 16 * B1_NKT(DS, a), D, bk.M);
}

template <bool SPLIT, int DS, int HD>
__device__ __forceinline__ void b1_q_side(const B1Args& a, unsigned char* smem, int n, bool add, B1Acc& A, const B1Steps& steps, B1Stage& stg, WRegs<1, SB_NT>& w) {
    constexpr bool BIAS = DS_EXACT(DS) == 64;
    typedef B1Lds<SPLIT> L;
    const cr_block_bwd_desc& bd = a.bd;
    const cr_block_desc& bk = bd.f;
    const cr_attn_desc& d = a.ad;
    __bf16* Im = reinterpret_cast<__bf16*>(smem);
    __bf16* Wi = Im + L::W_OFF;                           // slot 0: Wq (permuted)
    float* fl = reinterpret_cast<float*>(smem + L::F_OFF_BYTES);
    float* gam = fl + L::GAM; float* part = fl + L::PART; float* sdel = fl + L::SDEL; float* kb = fl + L::KB;
    constexpr int IST = L::IST;
    const int D = DS_D(DS, bk.D), T = B1_T(DS, a), T16 = 16 * B1_NKT(DS, a);
    const DCtx dcx = d_ctx<D_NF(DS)>(D);
    const int base_row = n * T, M = bk.M;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, lg = lane >> 4;
    const size_t MD = (size_t)M * D;
    float* dQg = const_cast<float*>(bd.dqkv);
    B1_TS(9);
    // ---- end of phase 1 + phase 2 staging: the image slots of phase 1 are dead behind its last barrier, so are W1 / W2.
    // K, V and Wq are requested first; the slab stores and the LayerNorm fold of phase 1 run under those loads.
    {
        float* partF = fl + L::PARTF; float* gamF = fl + L::GAMF;
        (void)gamF;
        // (Wq and the K / V rows were requested in front of phase 1's last weight-gradient products: b1_q_stage_issue)
        const float gv = (threadIdx.x < D) ? bk.ln1_g[threadIdx.x] : 0.0f;
        const size_t so = (size_t)blockIdx.x * bd.slab_stride;
        const int it = wave >> 1, jt0 = 2 * (wave & 1);
        b1_wstore<BIAS>(bd.g_w1 + so, D, bd.g_b1 + so, A.aw1, A.aw1b, D, it, jt0, add);
        b1_wstore<BIAS>(bd.g_w2 + so, D, bd.g_b2 + so, A.aw2, A.aw2b, D, it, jt0, add);
        // (dgamma2 dbeta2 and the final LayerNorm's sums leave on the K side: it runs the same chain and is the shorter side)
        b1_stage_kv_put<SPLIT>(stg, smem, kb, a, base_row, T16, D, M);
        w_put_perm<1, SB_NT, SPLIT>(Wi, w, D, bk.wqkv, 3 * D, 0, bk.wqkv, 3 * D, 0, bk.wqkv, 3 * D, 0);
        if (threadIdx.x < 64) gam[threadIdx.x] = gv;      // (gamma2 is dead behind phase 1's last barrier)
        for (int i = threadIdx.x; i < 4 * SB_WAVES * 64; i += SB_NT) part[i] = 0.0f;      // (PART and PARTF: the folds of a wave's first / second tile)
        (void)partF;
    }
    __syncthreads();
    B1_TS(3);
    const int kt_first = __builtin_amdgcn_readfirstlane(first_valid_key_lds(kb, T16, T) >> 4);     // (a scalar: the pair loop's counter and its branches are)
    const DropCtx dc = drop_ctx(d.drop, steps.attn);
    // per-lane byte offsets of the operand reads at tile 0 of an image (img_off: the swizzle term (row & 6) does not depend on
    // the tile, so a tile adds 2048 bytes)
    const int frk0 = 2 * img_off<2>(li, lg), frk1 = 2 * img_off<2>(li, lg + 4);
    int ftr[4];
    {
        const int q_ = li >> 2, p_ = li & 3;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) ftr[jt] = 2 * (img_off<2>(4 * lg + q_, 2 * jt + (p_ >> 1)) + 4 * (p_ & 1));
    }
    // byte offset of this lane group's four keys in the additive key bias (a key tile adds 64): an opaque value, so that the
    // loop's reads are (one base + pair offset) + immediate
    int vkb = L::F_OFF_BYTES + 4 * (L::KB + 4 * lg);
    asm volatile("" : "+v"(vkb));
    const unsigned tpk = a.qpk[wave];
    // the second-dispatched half of the waves loses every arbitration to its SIMD partner (MI355X_MICROARCH.md, two waves per SIMD: a
    // wave of that half ran its two tiles in 15.9 us where a first-half wave took 12.3): raised for the attention loops
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    typedef float f4s __attribute__((ext_vector_type(4), aligned(4)));
    // a tile's inputs (Q fragment in operand layout, d_o in layout R, the forward's row statistics); the NEXT tile's are
    // requested behind the current tile's loop, under its row chain
    GFrag<2> qn;
    RRaw rdo;
    f4s st[HD];
#pragma unroll
    for (int h = 0; h < HD; ++h) st[h] = (f4s){0.f, 0.f, 2.0f, 0.f};
    float qv_n = 0.0f;
    auto issue_tile = [&](int qt_) {
        const int q0_ = 16 * qt_, q_ = q0_ + (lane_now() & 15);
        const int m_ = base_row + min(q_, T - 1);
        gfrag_issue<2>(qn, d.Q, d.ld, base_row + q0_, 0, T - q0_, D, M);
        r_issue(rdo, bd.d_o, (u32)m_ * (u32)(4 * D), dcx, q_ < T);
#pragma unroll
        for (int h = 0; h < HD; ++h) st[h] = *reinterpret_cast<const f4s*>(d.row_stats + ((size_t)h * d.B * T + m_) * 4);   // [head][sequence][row]
        qv_n = d.q_valid[m_];
    };
    if ((int)(tpk & 31u) < B1_NKT(DS, a)) issue_tile((int)(tpk & 31u));
#pragma unroll 1
    for (int ti = 0; ti < 2; ++ti) {
        const int qt = (int)((tpk >> (5 * ti)) & 31u);
        if (qt >= B1_NKT(DS, a)) break;                           // 31 = none (wave-uniform)
        if (ti == 1) B1_TS(8);
        const int q0 = 16 * qt, q = q0 + li;
        const bool rok = q < T;
        const int m = base_row + min(q, T - 1);
        const u32 mo = (u32)m * (u32)(4 * D);
        const float qvs = rok ? qv_n * dc.scale : 0.0f;               // the kept scores' factor: query validity x 1 / (1 - rate)
        const bool normal = rok && st[0].z == 0.0f;                  // (normal / uniform / dead is the same in every head: the masks are)
        // P[q][k] = exp2(s c - m) / sum = exp2(s c - (m - log2(1 / sum))): the row's 1 / sum goes into the exponent (one multiply per
        // score less); rows without score gradient (uniform, dead, beyond T) get m' = 1e30: P = 0 exactly
        float delta[HD], mrow[HD];
#pragma unroll
        for (int h = 0; h < HD; ++h) {
            delta[h] = sdel[h * B1_ROWS + min(q, T16 - 1)];
            mrow[h] = normal ? st[h].x - __log2f(st[h].y) : 1e30f;
        }
        bf8 qh[2], ql[2], oh[2], ol[2];
        gfrag_finish<SPLIT, 2>(qn, d.Q, d.ld, base_row + q0, 0, T - q0, D, M, qh, ql);
        f32x4 dO[4];                                      // kept for the residual branch behind the loop (dq_in = dQ Wq^T + d_o)
        r_finish(dO, rdo, dcx);
        r_split<SPLIT>(dO, oh, ol);
        RRaw rx;
        r_issue(rx, bk.x, mo, dcx);                       // the block input's rows of this tile (LayerNorm-1 backward), under the loop
        const bool tile_live = __any(normal ? 1 : 0) != 0;              // uniform and dead rows carry no score gradient
        uint32_t xrow[HD];
#pragma unroll
        for (int h = 0; h < HD; ++h) xrow[h] = (attn_row_idx(d, h, n, q) + (uint32_t)(4 * lg)) * CR_PHI + dc.key;
        f32x4 dq[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) dq[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tile_live) {
            const int lo = kt_first, hi = qt;
            for (int kp = lo >> 1; 2 * kp <= hi; ++kp) {                 // pairs of key tiles 2 kp, 2 kp + 1
                const int k0 = 2 * kp, k1 = 2 * kp + 1;
                const int po = 4096 * kp;
                auto rfF = [&](int base, int ks, int second, int lo_) {
                    return *reinterpret_cast<const bf8*>(smem + (base + po + (ks ? frk1 : frk0)) + 2048 * second + L::LOB * lo_);
                };
                auto trF = [&](int base, int jt, int lo_) {
                    const unsigned char* pa = smem + (base + po + ftr[jt]) + L::LOB * lo_;
                    const bf4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(pa));
                    const bf4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(pa + 2048));
                    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                // the additive key bias of the pair's two tiles (the same for every head)
                const float4 b40 = *reinterpret_cast<const float4*>(smem + (vkb + 128 * kp));
                const float4 b41 = *reinterpret_cast<const float4*>(smem + (vkb + 128 * kp) + 64);
                // a head: its 32 columns are one k-step of the score products and two feature tiles of dQ (HD = 1: both k-steps, four tiles)
#pragma unroll
                for (int h = 0; h < HD; ++h) {
                    const int ks_lo = HD == 2 ? h : 0, ks_hi = HD == 2 ? h + 1 : 2;
                    constexpr int NKS = HD == 2 ? 1 : 2;
                    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
                    {
                        bf8 a0h[2], a0l[2], a1h[2], a1l[2];
#pragma unroll
                        for (int ks = ks_lo; ks < ks_hi; ++ks) {
                            a0h[ks] = rfF(0, ks, 0, 0); a1h[ks] = rfF(0, ks, 1, 0);
                            a0l[ks] = SPLIT ? rfF(0, ks, 0, 1) : a0h[ks]; a1l[ks] = SPLIT ? rfF(0, ks, 1, 1) : a1h[ks];
                        }
#pragma unroll
                        for (int ks = ks_lo; ks < ks_hi; ++ks) {
                            s0 = mma<SPLIT>(a0h[ks], a0l[ks], qh[ks], ql[ks], s0);      // S^T[key][q] = K Q^T
                            s1 = mma<SPLIT>(a1h[ks], a1l[ks], qh[ks], ql[ks], s1);
                        }
                        BF_SGB(0x100, (SPLIT ? 4 : 2) * NKS, 0);
                        BF_SGB(0x008, (SPLIT ? 6 : 2) * NKS, 0);
                    }
                    {
                        bf8 v0h[2], v0l[2], v1h[2], v1l[2];
#pragma unroll
                        for (int ks = ks_lo; ks < ks_hi; ++ks) {
                            v0h[ks] = rfF(L::MATB, ks, 0, 0); v1h[ks] = rfF(L::MATB, ks, 1, 0);
                            v0l[ks] = SPLIT ? rfF(L::MATB, ks, 0, 1) : v0h[ks]; v1l[ks] = SPLIT ? rfF(L::MATB, ks, 1, 1) : v1h[ks];
                        }
#pragma unroll
                        for (int ks = ks_lo; ks < ks_hi; ++ks) {
                            p0 = mma<SPLIT>(v0h[ks], v0l[ks], oh[ks], ol[ks], p0);      // dA^T[key][q] = V dO^T (V permuted, dO in layout R)
                            p1 = mma<SPLIT>(v1h[ks], v1l[ks], oh[ks], ol[ks], p1);
                        }
                        BF_SGB(0x100, (SPLIT ? 4 : 2) * NKS, 0);
                        BF_SGB(0x008, (SPLIT ? 6 : 2) * NKS, 0);
                    }
                    // the dQ product's K operand (transposed reads): first batch requested before the element-wise phase that hides it
                    constexpr int JB = (SPLIT || HD == 2) ? 2 : 4;
                    const int jbase = HD == 2 ? 2 * h : 0;
                    bf8 bh[JB], bl[JB];
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) {
                        bh[jt] = trF(0, jbase + jt, 0);
                        bl[jt] = SPLIT ? trF(0, jbase + jt, 1) : bh[jt];
                    }
                    float x[8];
                    // per score: the key mask is an ADDITIVE bias in the exponent (0 / -inf: kb), 1 / sum sits in m', the query's validity
                    // and the dropout scale are one factor, 1 / sqrt(d) is applied to dQ behind the loop; the causal compare exists only
                    // in the code of the pair that holds the diagonal tile (the last one: a wave-uniform branch picks the body):
                    // fma, add, exp2, the keep test (add, xor-shift, multiply, compare, select), fma, mul
                    const float mrow_h = mrow[h], delta_h = delta[h];
                    const uint32_t xrow_h = xrow[h];
                    auto finish = [&](auto diag_c, int kt, const float4& b4, const f32x4& s, const f32x4& p, int xo) {
                        constexpr bool DIAG = decltype(diag_c)::value;
                        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float pn = __builtin_amdgcn_exp2f(fmaf(s[r], a.isd_log2e, -mrow_h) + bb[r]);
                            if (DIAG) pn = (16 * kt + 4 * lg + r <= q) ? pn : 0.0f;                 // causal (kt == qt)
                            const float w = (cr_mix(xrow_h + (uint32_t)(16 * kt + r) * CR_PHI) >= dc.thresh) ? qvs : 0.0f;
                            x[xo + r] = pn * (p[r] * w - delta_h);                            // dS (1 / sqrt(d): behind the loop)
                        }
                    };
                    if (k1 > hi) {                                                            // the diagonal pair, second tile beyond it (or absent)
                        finish(std::true_type{}, k0, b40, s0, p0, 0);
                        x[4] = 0.0f; x[5] = 0.0f; x[6] = 0.0f; x[7] = 0.0f;
                    } else if (k1 == hi) {                                                    // the diagonal pair
                        finish(std::false_type{}, k0, b40, s0, p0, 0);
                        finish(std::true_type{}, k1, b41, s1, p1, 4);
                    } else {                                                                  // (a tile below `lo` holds masked keys only: bias -inf)
                        finish(std::false_type{}, k0, b40, s0, p0, 0);
                        finish(std::false_type{}, k1, b41, s1, p1, 4);
                    }
                    bf8 ah, al;
                    split8<SPLIT>(x, ah, al);
                    // dQ^T += K^T dS^T: the transposed-read fragment as A, dS as B -> D[feature 16 jt + 4 lg + r][query li] = layout R
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) dq[jbase + jt] = mma<SPLIT>(bh[jt], bl[jt], ah, al, dq[jbase + jt]);
                    if (HD == 1) {
#pragma unroll
                        for (int j0 = JB; j0 < 4; j0 += JB) {
#pragma unroll
                            for (int jt = 0; jt < JB; ++jt) {
                                bh[jt] = trF(0, j0 + jt, 0);
                                bl[jt] = SPLIT ? trF(0, j0 + jt, 1) : bh[jt];
                            }
#pragma unroll
                            for (int jt = 0; jt < JB; ++jt) dq[j0 + jt] = mma<SPLIT>(bh[jt], bl[jt], ah, al, dq[j0 + jt]);
                            BF_SGB(0x100, (SPLIT ? 4 : 2) * JB, 0);
                            BF_SGB(0x008, (SPLIT ? 3 : 1) * JB, 0);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) dq[jt] *= a.isd;                                   // dS / sqrt(d), once per output element
        if (ti == 0) B1_TS(6);
        // ---- the tile goes on through registers: dq_in = dQ Wq^T + d_o, LN1 backward -> this side's partial of dx ----
        const B1Args& ar = b1_args_again();                               // (this chain's pointers: not carried across the loop)
        RRaw rdx;
        if (ar.bd.dx_accumulate) r_issue(rdx, ar.bd.dx, mo, dcx, rok);
        r_store<B1_NT_WS>(const_cast<float*>(ar.bd.dqkv), mo, dq, rok, dcx);        // dQ, for the weight-gradient images of phase 3
        f32x4 dqin[4];
        {
            bf8 gh[2], gl[2];
            r_split<SPLIT>(dq, gh, gl);
            r_gemm_t<SPLIT, false>(dqin, Wi, Wi + ST_WIMG, gh, gl);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) dqin[ct] += dO[ct];
        }
        f32x4 x[4], dxl[4], ag[4], ab[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { ag[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; ab[ct] = ag[ct]; }
        r_finish(x, rx, dcx);
        // the NEXT tile's inputs, under the LayerNorm backward and the stores.  Behind the chain's own loads, and only where a next
        // tile exists: an unconditional request in front of them (static load counts, no full drain at the first use) made single
        // columns of dx differ between runs of the same step (tools/diag_repro.py) -- not understood, not kept
        if (ti == 0 && (int)((tpk >> 5) & 31u) < B1_NKT(DS, a)) issue_tile((int)((tpk >> 5) & 31u));
        r_ln_bwd(dxl, x, dqin, gam, ag, ab, dcx);
        b1_ln_fold(part + ti * (2 * SB_WAVES * 64), ag, ab);       // (the wave's first tile: PART, its second: PARTF)
        if (ar.bd.dx_accumulate) {
            f32x4 old[4];
            r_finish(old, rdx, dcx);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) dxl[ct] += old[ct];
        }
        if (ar.scatter) {
            b1_scatter_prep(ar, dxl, m, rok, D, steps.emb);
            r_store(ar.sbuf, mo, dxl, rok, dcx);
        } else {
            r_store(ar.bd.dx, mo, dxl, rok, dcx);
        }
    }
    B1_TS(4);
    __builtin_amdgcn_s_setprio(0);
    // phase 3's rows: wave w images tiles w and w + 8.  Their q_in rows (the forward's: nothing here writes them) are requested in
    // front of the drain, their dQ rows right behind the barrier, all before the first image is built: one exposed round trip
    RRaw rq0, rq1, rg0, rg1;
    const int tt0 = wave, tt1 = wave + SB_WAVES;
    const u32 mo0 = (u32)(base_row + min(16 * tt0 + li, T - 1)) * (u32)(4 * D), mo1 = (u32)(base_row + min(16 * tt1 + li, T - 1)) * (u32)(4 * D);
    if (tt0 < B1_NKT(DS, a)) r_issue(rq0, bk.q_in, mo0, dcx);
    if (tt1 < B1_NKT(DS, a)) r_issue(rq1, bk.q_in, mo1, dcx);
    B1_DRAIN();
    __syncthreads();                                      // every pass is done: the K / V images are dead, dQ rows are visible
    B1_TS(7);
    if (tt0 < B1_NKT(DS, a)) r_issue(rg0, dQg, mo0, dcx, 16 * tt0 + li < T);
    if (tt1 < B1_NKT(DS, a)) r_issue(rg1, dQg, mo1, dcx, 16 * tt1 + li < T);
    // ---- phase 3: dWq dbq from images of q_in and dQ, dgamma1 dbeta1; then the scatter of this side's partial ----
    f32x4 awq[2], nob[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { awq[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; nob[j] = awq[j]; }
    const int it = wave >> 1, jt0 = 2 * (wave & 1);
    // ONE round: the images of q_in and dQ over all tiles (2 x 14 tiles x hi, lo) are exactly the image area
    __bf16* Gm = reinterpret_cast<__bf16*>(smem + L::MATB);
    if (tt0 < B1_NKT(DS, a)) {
        f32x4 qin[4], dQ[4];
        r_finish(qin, rq0, dcx);
        plant_one(qin, D);
        img_put<SPLIT>(Im, Im + B1_FSTR, 16 * tt0, qin);
        r_finish(dQ, rg0, dcx);
        img_put<SPLIT>(Gm, Gm + B1_FSTR, 16 * tt0, dQ);
    }
    if (tt1 < B1_NKT(DS, a)) {
        f32x4 qin[4], dQ[4];
        r_finish(qin, rq1, dcx);
        plant_one(qin, D);
        img_put<SPLIT>(Im, Im + B1_FSTR, 16 * tt1, qin);
        r_finish(dQ, rg1, dcx);
        img_put<SPLIT>(Gm, Gm + B1_FSTR, 16 * tt1, dQ);
    }
    B1Small small;
    if (a.scatter && a.small) b1_small_issue<D_NF(DS)>(small, a, a.sbuf, n, D);
    __syncthreads();
    B1_TS(16);
    wgrad_accum<SPLIT, BIAS, 2 * SB_TPR>(awq, nob, Im, Im + B1_FSTR, Gm, Gm + B1_FSTR, B1_NKT(DS, a), it, jt0);       // dWq (+ dbq) += q_in^T dQ (nob: the bias sums at D = 64)
    B1_TS(17);
    const size_t so = (size_t)blockIdx.x * bd.slab_stride;
    b1_wstore<BIAS>(bd.g_wqkv + so, 3 * D, bd.g_bqkv + so, awq, nob, D, it, jt0, add);
    b1_ln_flush<2>(part, bd.g_ln1_g + so, bd.g_ln1_b + so, D, add);         // (the folds of phase 2 lie behind two barriers)
    B1_TS(18);
    if (a.scatter && a.small) b1_small_table<SPLIT, D_NF(DS)>(a, smem, reinterpret_cast<int*>(part), small, n, D, add);   // (part: the sequence's ids)
    else if (a.scatter) b1_scatter_rows(a, a.sbuf, n, D);
    B1_TS(5);
    (void)MD;
}

// =====================================================================================================
// K side, phases 2 and 3: key-owner pass (dK, dV), K / V projections backward, dWk dWv
// =====================================================================================================
template <bool SPLIT, int DS, int HD>
__device__ __forceinline__ void b1_k_side(const B1Args& a, unsigned char* smem, int n, bool add, B1Acc& A, const B1Steps& steps) {
    constexpr bool BIAS = DS_EXACT(DS) == 64;
    typedef B1Lds<SPLIT> L;
    const cr_block_bwd_desc& bd = a.bd;
    const cr_block_desc& bk = bd.f;
    const cr_attn_desc& d = a.ad;
    __bf16* Im = reinterpret_cast<__bf16*>(smem);
    __bf16* Wi = Im + L::W_OFF;                           // slot 0: Wk, slot 1: Wv (permuted)
    float* fl = reinterpret_cast<float*>(smem + L::F_OFF_BYTES);
    float* sdel = fl + L::SDEL; float* smx = fl + L::SMX; float* suni = fl + L::SUNI; float* sqv = fl + L::SQV;
    float* tile_flag = fl + L::TFLAG;
    constexpr int WST = L::WST, IST = L::IST;
    const int D = DS_D(DS, bk.D), T = B1_T(DS, a), T16 = 16 * B1_NKT(DS, a);
    const DCtx dcx = d_ctx<D_NF(DS)>(D);
    const int base_row = n * T, M = bk.M;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, lg = lane >> 4;
    const size_t MD = (size_t)M * D;
    float* dKg = const_cast<float*>(bd.dqkv) + MD;
    float* dVg = const_cast<float*>(bd.dqkv) + 2 * MD;
    typedef float f4s __attribute__((ext_vector_type(4), aligned(4)));
    // ---- phase 2 staging: row statistics of every query row; Wk, Wv over W1, W2 ----
    const DropCtx dc = drop_ctx(d.drop, steps.attn);
    {
        WRegs<2, SB_NT> w;
        w_issue<2, SB_NT>(w, D, bk.wqkv, 3 * D, D, bk.wqkv, 3 * D, 2 * D, bk.wqkv, 3 * D, 2 * D);
        const int t = threadIdx.x, tc = min(t, T - 1);
        f4s st[HD];
#pragma unroll
        for (int h = 0; h < HD; ++h) st[h] = *reinterpret_cast<const f4s*>(d.row_stats + ((size_t)h * d.B * T + base_row + tc) * 4);   // [head][sequence][row]
        const float qv = d.q_valid[base_row + tc];
        {
            // dgamma2 dbeta2 (+ the final LayerNorm's sums) of this side's phase 1: every wave WRITES its slots (its accumulators cover all
            // its tiles; a wave without a tile writes zeros), one barrier, flush
            float* part = fl + L::PART; float* partF = fl + L::PARTF;
            b1_ln_fold(part, A.ag, A.ab);
            if (a.has_ln) b1_ln_fold(partF, A.agF, A.abF);
            __syncthreads();                              // phase 1 is over in every wave: W1 / W2 are dead, sdel is complete, the folds are visible
            const size_t so = (size_t)blockIdx.x * bd.slab_stride;
            b1_ln_flush(part, bd.g_ln2_g + so, bd.g_ln2_b + so, D, add);
            if (a.has_ln) b1_ln_flush(partF, a.ln.dgamma + so, a.ln.dbeta + so, D, add);
        }
        w_put_perm<2, SB_NT, SPLIT>(Wi, w, D, bk.wqkv, 3 * D, D, bk.wqkv, 3 * D, 2 * D, bk.wqkv, 3 * D, 2 * D);
        // stored so that the inner loop is branch-free: A[q][key] = valid * exp2(s c - smx) + (key < T ? suni : 0);
        // normal row: suni = 0; uniform row: suni = 1/T; dead row: neither (smx = 1e30 wherever the row is not normal)
        const float flag = t < T ? st[0].z : 2.0f;          // (normal / uniform / dead is the same in every head: the masks are)
        const bool normal = flag == 0.0f;
        if (t < T16 + ((B1_NKT(DS, a) & 1) ? 16 : 0)) {             // (an odd tile count: the absent tile of the last pair reads as dead rows)
#pragma unroll
            for (int h = 0; h < HD; ++h) {
                smx[h * B1_ROWS + t] = normal ? st[h].x - __log2f(st[h].y) : 1e30f;      // 1 / sum inside the exponent: P = exp2(s c - m')
                if (!normal || t >= T16) sdel[h * B1_ROWS + t] = 0.0f;
            }
            suni[t] = (flag == 1.0f) ? a.invT : 0.0f;
            sqv[t] = t < T ? qv * dc.scale : 0.0f;               // query validity x 1 / (1 - rate): the kept scores' factor
        }
        // per query tile: 0 nothing flows, 1 normal rows only, 2 has a uniform row -- from the row statistics this thread holds (one
        // lane per row, folded by ballots: no second pass over the vectors, no barrier of its own)
        if ((t & ~63) < T16) {
            const bool live = t < T16;
            const unsigned long long bn = __ballot(live && normal);
            const unsigned long long bu = __ballot(live && flag == 1.0f);
            if ((t & 15) == 0 && live) {
                const int sh = t & 48;
                const bool anyu = ((bu >> sh) & 0xFFFFull) != 0, anyn = ((bn >> sh) & 0xFFFFull) != 0;
                tile_flag[t >> 4] = anyu ? 2.0f : (anyn ? 1.0f : 0.0f);
            }
        }
        if (t == 0 && (B1_NKT(DS, a) & 1)) tile_flag[B1_NKT(DS, a)] = 0.0f;
    }
    __syncthreads();
    B1_TS(3);
    // the tile flags as wave-uniform bit masks: the loops test scalars (a flag read from LDS inside the loop is a read, a full
    // wait and a branch per tile)
    unsigned live_m, uni_m;
    {
        const float tf = tile_flag[lane & 15];
        const unsigned nm = (1u << B1_NKT(DS, a)) - 1u;
        live_m = (unsigned)__ballot(lane < 16 && tf != 0.0f) & nm;
        uni_m = (unsigned)__ballot(lane < 16 && tf == 2.0f) & nm;
    }
    const int frk0 = 2 * img_off<2>(li, lg), frk1 = 2 * img_off<2>(li, lg + 4);
    int ftr[4];
    {
        const int q_ = li >> 2, p_ = li & 3;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) ftr[jt] = 2 * (img_off<2>(4 * lg + q_, 2 * jt + (p_ >> 1)) + 4 * (p_ & 1));
    }
    // byte offset of this lane group's four rows in the row vectors (from sdel; a query tile adds 64): an opaque value, so that
    // the vectors' reads are (one base + tile offset) + immediates instead of one 32-bit add each
    int vrow = L::F_OFF_BYTES + 4 * (L::SDEL + 4 * lg);
    asm volatile("" : "+v"(vrow));
    const unsigned tpk = a.kpk[wave];
    // the second-dispatched half of the waves loses every arbitration to its SIMD partner (MI355X_MICROARCH.md, two waves per SIMD: a
    // wave of that half ran its two tiles in 15.9 us where a first-half wave took 12.3): raised for the attention loops
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    // a tile's own K / V rows (operand layout, from memory); the NEXT tile's are requested behind the current tile's loop
    GFrag<2> kn, vn;
    float kvn = 0.0f;
    auto issue_tile = [&](int kt_) {
        gfrag_issue<2>(kn, d.K, d.ld, base_row + 16 * kt_, 0, T - 16 * kt_, D, M);
        gfrag_issue<2>(vn, d.V, d.ld, base_row + 16 * kt_, 0, T - 16 * kt_, D, M);
        kvn = d.k_valid[base_row + min(16 * kt_ + (lane_now() & 15), T - 1)];
    };
    if ((int)(tpk & 31u) < B1_NKT(DS, a)) issue_tile((int)(tpk & 31u));
#pragma unroll 1
    for (int ti = 0; ti < 2; ++ti) {
        const int kt = (int)((tpk >> (5 * ti)) & 31u);
        if (kt >= B1_NKT(DS, a)) break;
        if (ti == 1) B1_TS(8);
        const int key0 = 16 * kt, key = key0 + li;
        const bool rok = key < T;
        const int m = base_row + min(key, T - 1);
        const u32 mo = (u32)m * (u32)(4 * D);
        const float key_in_T = rok ? 1.0f : 0.0f;
        const uint32_t drop_base = attn_row_idx(d, 0, n, 0) + (uint32_t)key;
        bf8 kh[2], kl[2], vh[2], vl[2];
        gfrag_finish<SPLIT, 2>(kn, d.K, d.ld, base_row + key0, 0, T - key0, D, M, kh, kl);
        gfrag_finish<SPLIT, 2>(vn, d.V, d.ld, base_row + key0, 0, T - key0, D, M, vh, vl);
        const bool kvk = rok && kvn != 0.0f;
        const float kbias = kvk ? 0.0f : -INFINITY;                     // this lane's key: valid or masked (padding, beyond T)
        const bool tile_has_key = __any(kvk ? 1 : 0) != 0;              // all-padding key tile: only uniform rows reach it
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            dk[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int ntile = B1_NKT(DS, a);
        // x = idx * PHI + key of attention_weights[n, 4 lg, key]: a query tile adds 16 T PHI, a row T PHI (scalars)
        const uint32_t xbase = (drop_base + (uint32_t)(4 * lg) * (uint32_t)T) * CR_PHI + dc.key;
        const uint32_t xT = (uint32_t)T * CR_PHI;
        const uint32_t xhead = (uint32_t)d.batch_global * (uint32_t)T * (uint32_t)T * CR_PHI;      // a head further in attention_weights (attn_row_idx)
        for (int qp = 0; 2 * qp < ntile; ++qp) {                         // pairs of query tiles 2 qp, 2 qp + 1
            const int l0 = 2 * qp, l1 = 2 * qp + 1;
            // nothing flows through dead query tiles; uniform rows see every key; else the causal / padding skip (scalar bit tests)
            const unsigned pairbits = (live_m >> l0) & 3u, unibits = (uni_m >> l0) & 3u;
            const bool w0 = (pairbits & 1u) && ((unibits & 1u) || (l0 >= kt && tile_has_key));
            const bool w1 = (pairbits & 2u) && ((unibits & 2u) || (l1 >= kt && tile_has_key));
            if (!w0 && !w1) continue;
            const int po = 4096 * qp;
            auto rfF = [&](int base, int ks, int second, int lo) {
                return *reinterpret_cast<const bf8*>(smem + (base + po + (ks ? frk1 : frk0)) + 2048 * second + L::LOB * lo);
            };
            auto trF = [&](int base, int jt, int lo) {
                const unsigned char* pa = smem + (base + po + ftr[jt]) + L::LOB * lo;
                const bf4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(pa));
                const bf4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(pa + 2048));
                return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            const unsigned char* vq = smem + (vrow + 128 * qp);
            // a head: its 32 columns are one k-step of the score products and two feature tiles of dK / dV (HD = 1: both k-steps, four tiles)
#pragma unroll
            for (int h = 0; h < HD; ++h) {
                const int ks_lo = HD == 2 ? h : 0, ks_hi = HD == 2 ? h + 1 : 2;
                constexpr int NKS = HD == 2 ? 1 : 2;
                f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
                {
                    bf8 a0h[2], a0l[2], a1h[2], a1l[2];
#pragma unroll
                    for (int ks = ks_lo; ks < ks_hi; ++ks) {
                        a0h[ks] = rfF(0, ks, 0, 0); a1h[ks] = rfF(0, ks, 1, 0);
                        a0l[ks] = SPLIT ? rfF(0, ks, 0, 1) : a0h[ks]; a1l[ks] = SPLIT ? rfF(0, ks, 1, 1) : a1h[ks];
                    }
#pragma unroll
                    for (int ks = ks_lo; ks < ks_hi; ++ks) {
                        s0 = mma<SPLIT>(a0h[ks], a0l[ks], kh[ks], kl[ks], s0);      // S[q][key]
                        s1 = mma<SPLIT>(a1h[ks], a1l[ks], kh[ks], kl[ks], s1);
                    }
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * NKS, 0);
                    BF_SGB(0x008, (SPLIT ? 6 : 2) * NKS, 0);
                }
                {
                    bf8 o0h[2], o0l[2], o1h[2], o1l[2];
#pragma unroll
                    for (int ks = ks_lo; ks < ks_hi; ++ks) {
                        o0h[ks] = rfF(L::MATB, ks, 0, 0); o1h[ks] = rfF(L::MATB, ks, 1, 0);
                        o0l[ks] = SPLIT ? rfF(L::MATB, ks, 0, 1) : o0h[ks]; o1l[ks] = SPLIT ? rfF(L::MATB, ks, 1, 1) : o1h[ks];
                    }
#pragma unroll
                    for (int ks = ks_lo; ks < ks_hi; ++ks) {
                        p0 = mma<SPLIT>(o0h[ks], o0l[ks], vh[ks], vl[ks], p0);      // dA[q][key] = dO V^T
                        p1 = mma<SPLIT>(o1h[ks], o1l[ks], vh[ks], vl[ks], p1);
                    }
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * NKS, 0);
                    BF_SGB(0x008, (SPLIT ? 6 : 2) * NKS, 0);
                }
                // the rows' vectors of both query tiles (this head's m' and delta, validity x dropout scale), in front of the transposed
                // reads (LDS returns in order: they are there when the element-wise phase opens)
                float4 m4[2], d4[2], w4[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    m4[t2] = *reinterpret_cast<const float4*>(vq + 64 * t2 + 4 * (L::SMX - L::SDEL) + 4 * B1_ROWS * h);
                    d4[t2] = *reinterpret_cast<const float4*>(vq + 64 * t2 + 4 * B1_ROWS * h);
                    w4[t2] = *reinterpret_cast<const float4*>(vq + 64 * t2 + 4 * (L::SQV - L::SDEL));
                }
                // dOut columns for the dV product: first batch requested before the element-wise phase that hides it
                constexpr int JB = (SPLIT || HD == 2) ? 2 : 4;
                const int jbase = HD == 2 ? 2 * h : 0;
                bf8 oth[JB], otl[JB];
#pragma unroll
                for (int jt = 0; jt < JB; ++jt) {
                    oth[jt] = trF(L::MATB, jbase + jt, 0);
                    otl[jt] = SPLIT ? trF(L::MATB, jbase + jt, 1) : oth[jt];
                }
                float xa[8], xd[8];
                // per score: the key's validity (this lane's key: loop-invariant) and "query tile above the key tile" are an ADDITIVE bias
                // in the exponent (0 / -inf), 1 / sum sits in m' (smx), the query's validity and the dropout scale in one factor (sqv),
                // 1 / sqrt(d) is applied to dK behind the loop; the causal compare exists only in the code of the one pair that holds
                // the diagonal tile (DIAG: a wave-uniform branch picks the body), the uniform-row term only in tiles that hold such a
                // row: 11 vector instructions per score.  (The absent second tile of an odd count reads as dead rows: m' = 1e30.)
                const uint32_t xb_h = xbase + (uint32_t)h * xhead;
                auto finish = [&](auto diag_c, int t2, int xo, const f32x4& s, const f32x4& p) {
                    constexpr bool DIAG = decltype(diag_c)::value;
                    const int lt = 2 * qp + t2;
                    const float mm[4] = {m4[t2].x, m4[t2].y, m4[t2].z, m4[t2].w}, dd[4] = {d4[t2].x, d4[t2].y, d4[t2].z, d4[t2].w}, ww[4] = {w4[t2].x, w4[t2].y, w4[t2].z, w4[t2].w};
                    const uint32_t x0 = xb_h + (uint32_t)lt * (16u * xT);        // counter of attention_weights[head, n, 16 lt + 4 lg, key]
                    const float bias = (lt < kt) ? -INFINITY : kbias;            // query tile above the key tile: causally masked as a whole
                    float pn[4], w[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pn[r] = __builtin_amdgcn_exp2f(fmaf(s[r], a.isd_log2e, -mm[r]) + bias);
                        if (DIAG) pn[r] = (lt != kt || key <= 16 * lt + 4 * lg + r) ? pn[r] : 0.0f;        // causal
                        w[r] = (cr_mix(x0 + (uint32_t)r * xT) >= dc.thresh) ? ww[r] : 0.0f;                // validity x keep / (1 - rate)
                        xd[xo + r] = pn[r] * (p[r] * w[r] - dd[r]);              // dS (1 / sqrt(d): behind the loop)
                    }
                    if ((uni_m >> lt) & 1u) {                                    // rows without a valid key: 1 / T on every key < T
                        const float4 u4 = *reinterpret_cast<const float4*>(vq + 64 * t2 + 4 * (L::SUNI - L::SDEL));
                        const float uu[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) xa[xo + r] = (pn[r] + key_in_T * uu[r]) * w[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) xa[xo + r] = pn[r] * w[r];   // A after mask + dropout
                    }
                };
                if (qp == (kt >> 1)) {
                    finish(std::true_type{}, 0, 0, s0, p0);
                    finish(std::true_type{}, 1, 4, s1, p1);
                } else {
                    finish(std::false_type{}, 0, 0, s0, p0);
                    finish(std::false_type{}, 1, 4, s1, p1);
                }
                bf8 ah, al, dh, dl;
                split8<SPLIT>(xa, ah, al);
                split8<SPLIT>(xd, dh, dl);
                // dV^T += dO^T A, dK^T += Q^T dS: the transposed-read fragment as A, the coefficients as B -> layout R
#pragma unroll
                for (int jt = 0; jt < JB; ++jt) dv[jbase + jt] = mma<SPLIT>(oth[jt], otl[jt], ah, al, dv[jbase + jt]);
                if (HD == 1) {
#pragma unroll
                    for (int j0 = JB; j0 < 4; j0 += JB) {
#pragma unroll
                        for (int jt = 0; jt < JB; ++jt) {
                            oth[jt] = trF(L::MATB, j0 + jt, 0);
                            otl[jt] = SPLIT ? trF(L::MATB, j0 + jt, 1) : oth[jt];
                        }
#pragma unroll
                        for (int jt = 0; jt < JB; ++jt) dv[j0 + jt] = mma<SPLIT>(oth[jt], otl[jt], ah, al, dv[j0 + jt]);
                        BF_SGB(0x100, (SPLIT ? 4 : 2) * JB, 0);
                        BF_SGB(0x008, (SPLIT ? 3 : 1) * JB, 0);
                    }
                }
#pragma unroll
                for (int j0 = jbase; j0 < (HD == 2 ? jbase + 2 : 4); j0 += JB) {
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) {
                        oth[jt] = trF(0, j0 + jt, 0);
                        otl[jt] = SPLIT ? trF(0, j0 + jt, 1) : oth[jt];
                    }
#pragma unroll
                    for (int jt = 0; jt < JB; ++jt) dk[j0 + jt] = mma<SPLIT>(oth[jt], otl[jt], dh, dl, dk[j0 + jt]);
                    BF_SGB(0x100, (SPLIT ? 4 : 2) * JB, 0);
                    BF_SGB(0x008, (SPLIT ? 3 : 1) * JB, 0);
                }
            }
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) dk[jt] *= a.isd;                                   // dS / sqrt(d), once per output element
        if (ti == 0) B1_TS(6);
        if (ti == 0 && (int)((tpk >> 5) & 31u) < B1_NKT(DS, a)) issue_tile((int)((tpk >> 5) & 31u));   // the next tile's K / V rows, under this tile's row chain
        // ---- the tile goes on through registers: this side's partial of dx = dK Wk^T + dV Wv^T ----
        const B1Args& ar = b1_args_again();                               // (this chain's pointers: not carried across the loop)
        r_store<B1_NT_WS>(const_cast<float*>(ar.bd.dqkv) + MD, mo, dk, rok, dcx);   // dK, dV: for the weight-gradient images of phase 3
        r_store<B1_NT_WS>(const_cast<float*>(ar.bd.dqkv) + 2 * MD, mo, dv, rok, dcx);
        f32x4 dxp[4];
        {
            bf8 gh[2], gl[2];
            r_split<SPLIT>(dk, gh, gl);
            r_gemm_t<SPLIT, false>(dxp, Wi, Wi + ST_WIMG, gh, gl);
            r_split<SPLIT>(dv, gh, gl);
            r_gemm_t<SPLIT, true>(dxp, Wi + WST, Wi + WST + ST_WIMG, gh, gl);
        }
        if (ar.scatter) {
            b1_scatter_prep(ar, dxp, m, rok, D, steps.emb);
            r_store(ar.sbuf2, mo, dxp, rok, dcx);
        } else {
            r_store(ar.dx2, mo, dxp, rok, dcx);
        }
    }
    B1_TS(4);
    __builtin_amdgcn_s_setprio(0);
    B1_DRAIN();
    __syncthreads();                                      // every pass is done: the Q / dOut images are dead, dK / dV rows are visible
    B1_TS(7);
    // ---- phase 3: dWk dbk dWv dbv from images of x, dK, dV; then the scatter of this side's partial ----
    f32x4 awk[2], awv[2], awkb[2], awvb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { awk[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; awv[j] = awk[j]; awkb[j] = awk[j]; awvb[j] = awk[j]; }
    const int it = wave >> 1, jt0 = 2 * (wave & 1);
    const int R = B1_NKT(DS, a) > SB_TPR ? 2 : 1;
    // a round's rows (x, dK, dV of the wave's tile) are requested one round ahead: the second round's fly under the first round's products
    RRaw r1, r2, r3;
    auto issue3 = [&](int rd) {
        if (rd < R && wave < min(SB_TPR, B1_NKT(DS, a) - rd * SB_TPR)) {
            const int qq = 16 * (rd * SB_TPR + wave) + (lane_now() & 15);
            const u32 mo = (u32)(base_row + min(qq, T - 1)) * (u32)(4 * D);
            r_issue(r1, bk.x, mo, dcx);
            r_issue(r2, dKg, mo, dcx, qq < T);
            r_issue(r3, dVg, mo, dcx, qq < T);
        }
    };
    issue3(0);
    B1Small small;
#pragma unroll 1
    for (int rd = 0; rd < R; ++rd) {
        const int ntr = min(SB_TPR, B1_NKT(DS, a) - rd * SB_TPR);
        if (wave < ntr) {
            f32x4 x[4], g[4];
            r_finish(x, r1, dcx);
            plant_one(x, D);
            img_put<SPLIT>(Im, Im + SB_IMG, 16 * wave, x);
            r_finish(g, r2, dcx);
            img_put<SPLIT>(Im + IST, Im + IST + SB_IMG, 16 * wave, g);
            r_finish(g, r3, dcx);
            img_put<SPLIT>(Im + 2 * IST, Im + 2 * IST + SB_IMG, 16 * wave, g);
        }
        issue3(rd + 1);
        if (rd == R - 1 && a.scatter && a.small) b1_small_issue<D_NF(DS)>(small, a, a.sbuf2, n, D);     // (the small table's rows: under the last products)
        __syncthreads();
        if (rd == 0) B1_TS(16);
        wgrad_accum<SPLIT, BIAS>(awk, awkb, Im, Im + SB_IMG, Im + IST, Im + IST + SB_IMG, ntr, it, jt0);                // dWk (+ dbk) += x^T dK
        wgrad_accum<SPLIT, BIAS>(awv, awvb, Im, Im + SB_IMG, Im + 2 * IST, Im + 2 * IST + SB_IMG, ntr, it, jt0);        // dWv (+ dbv) += x^T dV
        __syncthreads();
        if (rd == 0) B1_TS(19);
    }
    B1_TS(17);
    const size_t so = (size_t)blockIdx.x * bd.slab_stride;
    b1_wstore<BIAS>(bd.g_wqkv + so + D, 3 * D, bd.g_bqkv + so + D, awk, awkb, D, it, jt0, add);
    b1_wstore<BIAS>(bd.g_wqkv + so + 2 * D, 3 * D, bd.g_bqkv + so + 2 * D, awv, awvb, D, it, jt0, add);
    B1_TS(18);
    if (a.scatter && a.small) b1_small_table<SPLIT, D_NF(DS)>(a, smem, reinterpret_cast<int*>(fl + L::PART), small, n, D, add);
    else if (a.scatter) b1_scatter_rows(a, a.sbuf2, n, D);
    B1_TS(5);
}

template <bool SPLIT, int DS, int HD>
__global__ __launch_bounds__(SB_NT) void k_stack_block_bwd(B1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    B1_TS(0);
    cr_kernarg_touch<sizeof(B1Args)>();
    B1_TS(15);
    // ONE sequence per workgroup pair and launch: sequence a.n0 + blockIdx.x, slab blockIdx.x.  (A loop over sequences in here
    // made every per-lane address and mask of both sides loop-invariant: hoisted to the top of the kernel and spilled, 269
    // registers.  More sequences than slabs are further launches that ADD to the slabs -- the host's loop.)
    const int n = a.n0 + (int)blockIdx.x;
    if (blockIdx.y == 0) {
        B1Acc acc;
        B1Steps steps;
        b1_phase1<SPLIT, DS, false, HD>(a, smem_raw, n, acc, steps, [] {});
        b1_k_side<SPLIT, DS, HD>(a, smem_raw, n, a.add != 0, acc, steps);
    } else {
        B1Acc acc;
        B1Stage st;
        WRegs<1, SB_NT> w;
        B1Steps steps;
        b1_phase1<SPLIT, DS, true, HD>(a, smem_raw, n, acc, steps, [&] { b1_q_stage_issue<DS>(a, n, st, w); });
        b1_q_side<SPLIT, DS, HD>(a, smem_raw, n, a.add != 0, acc, steps, st, w);
    }
    B1_TS(31);
}

// =====================================================================================================
// host side
// =====================================================================================================
// Tiles of the attention passes' waves: at most two per wave (what a wave carries through registers).  A query tile qt meets key tiles
// 0..qt, a key tile kt query tiles kt..nkt-1, two per loop iteration; a tile costs its pair iterations + 3 (fragments from memory, the
// row chain behind the loop, stores: tools/b1_ts.py).  Waves w and w + 4 share a SIMD, and what the timeline shows is that the SIMD's SUM
// sets the pace (the wave whose SIMD carried 20 units finished 3 us behind the ones at 18-19): so the deal minimises the largest SIMD sum
// first (a small branch-and-bound over the 4^nkt assignments, heaviest tile first, equal partial SIMDs tried once), then the largest wave;
// inside a SIMD the heavier wave takes the upper half (w + 4: raised priority in the loops, b1_q_side / b1_k_side), inside a wave the
// heavier tile goes first.  Round 5: the greedy deal before this one (heaviest tile to the lightest SIMD) left SIMD sums of 19 / 18 / 18 / 20
// at 13 tiles where 19 / 19 / 19 / 18 exists: 0.3163 -> 0.3125 ms per step (tools/probes/deal_ab.sh, three interleaved rounds).
struct B1Deal { unsigned pk[8]; bool done; };
static void b1_deal_search(int nkt, bool query_pass, unsigned (&pk)[8]) {
    int wave_of[16], order[16];
    cr_deal_search(nkt, query_pass, wave_of, order);
    for (int w = 0; w < 8; ++w) pk[w] = 0x3FFu;                           // two "none" entries
    int wc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < nkt; ++k) {                                        // heaviest first: a wave's heavier tile is its first
        const int t = order[k], w = wave_of[t];
        pk[w] = (pk[w] & ~(31u << (5 * wc[w]))) | ((unsigned)t << (5 * wc[w]));
        ++wc[w];
    }
}
static void b1_deal_tiles(int nkt, bool query_pass, unsigned (&pk)[8]) {
    static B1Deal cache[2][16];
    static std::mutex mu;
    {
        std::lock_guard<std::mutex> lock(mu);
        B1Deal& c = cache[query_pass ? 1 : 0][nkt & 15];
        if (!c.done) { b1_deal_search(nkt, query_pass, c.pk); c.done = true; }
        for (int w = 0; w < 8; ++w) pk[w] = c.pk[w];
    }
    // measurement override: CASTREC_B1_QPK / CASTREC_B1_KPK = eight entries "a" or "a:b" (tile numbers; "-" = none), comma separated, wave 0 first
    if (const char* ov = getenv(query_pass ? "CASTREC_B1_QPK" : "CASTREC_B1_KPK")) {
        unsigned t[8];
        int w = 0;
        const char* c = ov;
        while (w < 8 && *c) {
            unsigned a0 = 31, a1 = 31;
            if (*c != '-') { a0 = (unsigned)strtol(c, const_cast<char**>(&c), 10); if (*c == ':') { ++c; a1 = (unsigned)strtol(c, const_cast<char**>(&c), 10); } }
            else ++c;
            t[w++] = (a0 & 31u) | ((a1 & 31u) << 5);
            if (*c == ',') ++c;
        }
        if (w == 8) for (int i = 0; i < 8; ++i) pk[i] = t[i];
    }
}

// (for tests: the deal of one pass -- pk[w] = two 5-bit tile numbers of wave w, 31 = none)
extern "C" int cr_stack_block_bwd_deal(int nkt, int query_pass, uint32_t* pk) {
    CR_REQUIRE(pk && nkt >= 1 && nkt <= 2 * SB_TPR, "cr_stack_block_bwd_deal: 1 <= tiles <= 14");
    unsigned t[8];
    b1_deal_tiles(nkt, query_pass != 0, t);
    for (int w = 0; w < 8; ++w) pk[w] = t[w];
    return CR_OK;
}

static const char* b1_unsupported(const cr_block_bwd_desc* bd, const cr_attn_desc* ad, int B, int T, int precision) {
    if (!bd || !ad) return "NULL description";
    const cr_block_desc& d = bd->f;
    const bool two_heads = d.D == 64 && ad->H == 2 && ad->d == 32;   // (C3's shape: the bias gradients come from an all-ones product there)
    if (!two_heads && (d.D < 8 || d.D >= 64)) return "hidden size 8..63 (one head), or 64 with two heads of 32";
    if (precision != CR_PREC_BF16X3 && precision != CR_PREC_BF16) return "bf16 arithmetic (precision) only";
    if (B < 1 || T < 1 || d.M != B * T) return "M = B T";
    if ((T + 15) / 16 > 2 * SB_TPR) return "T <= 224 (two rounds of 7 row tiles)";
    if ((size_t)d.M * d.D * 4 >= ((size_t)1 << 32)) return "activations of 4 GiB or more (32-bit row offsets)";
    if (bd->n_slabs < 1) return "n_slabs";
    if (!two_heads && (ad->H != 1 || ad->d != d.D)) return "one head of d = D (or two heads of 32 at D = 64)";
    if (ad->B != B || ad->T != T) return "the block's B and T";
    if (ad->ld != d.D || ad->Q != d.qkv || ad->K != d.qkv + (size_t)d.M * d.D || ad->V != d.qkv + 2 * (size_t)d.M * d.D) return "Q / K / V = the parts of the block's qkv";
    if (ad->residual != d.q_in || ad->out != d.o || ad->k_valid != d.k_valid || ad->q_valid != d.q_valid) return "the attention call of this block (residual = q_in, out = o, masks)";
    if (!ad->row_stats) return "row_stats (saved by the forward)";
    if (ad->attn_weights) return "attention weights are not taken";
    return nullptr;
}
extern "C" int cr_stack_block_bwd_supported(const cr_block_bwd_desc* bd, const cr_attn_desc* ad, int B, int T, int precision) {
    return b1_unsupported(bd, ad, B, T, precision) == nullptr;
}

template <bool SPLIT, int DS, int HD = 1>
static int launch_b1(B1Args& a, int nwg, hipStream_t s) {
    static cr_devmask attr = 0;
    int rc = cr_raise_lds_limit(reinterpret_cast<const void*>(&k_stack_block_bwd<SPLIT, DS, HD>), &attr);
    if (rc) return rc;
    for (int n0 = 0; n0 < a.B; n0 += nwg) {               // more sequences than slabs: further launches add to the slabs
        a.n0 = n0;
        a.add = n0 > 0;
        hipLaunchKernelGGL((k_stack_block_bwd<SPLIT, DS, HD>), dim3(a.B - n0 < nwg ? a.B - n0 : nwg, 2), dim3(SB_NT), B1Lds<SPLIT>::BYTES, s, a);
    }
    return cr_check_launch("cr_stack_block_bwd");
}

extern "C" int cr_stack_block_bwd(const cr_block_bwd_desc* bd, const cr_attn_desc* ad, const cr_block_bwd1_ext* x, const cr_ln_bwd_desc* lnf,
                                  const cr_embed_bwd_desc* sc, int B, int T, int precision, void* stream) {
    const char* why = b1_unsupported(bd, ad, B, T, precision);
    CR_REQUIRE(why == nullptr, "cr_stack_block_bwd: unsupported (%s)", why ? why : "");
    CR_REQUIRE(x != nullptr, "cr_stack_block_bwd: NULL extension");
    B1Args a;
    memset(static_cast<void*>(&a), 0, sizeof(a));
    a.bd = *bd; a.ad = *ad;
    a.B = B; a.T = T; a.nkt = (T + 15) / 16;
    const cr_block_desc* d = &bd->f;
    CR_REQUIRE(bd->d_o && bd->dqkv && d->hid && d->f_in && d->o && d->q_in && d->x && d->mask_ids && d->w1 && d->w2 && d->wqkv && d->ln1_g && d->ln2_g,
               "cr_stack_block_bwd: NULL pointer");
    CR_REQUIRE(bd->g_w1 && bd->g_b1 && bd->g_w2 && bd->g_b2 && bd->g_ln2_g && bd->g_ln2_b && bd->g_wqkv && bd->g_bqkv && bd->g_ln1_g && bd->g_ln1_b,
               "cr_stack_block_bwd: NULL gradient pointer");
    CR_REQUIRE(bd->dq_part == nullptr, "cr_stack_block_bwd: dq_part is not taken");
    a.dy2 = x->dy2; a.dx2 = x->dx2; a.ln_dy2 = x->lnf_dy2; a.d_addend2 = x->d_addend2;
    if (lnf) {
        CR_REQUIRE(lnf->x == d->y && lnf->ldx == d->D && lnf->M == d->M && lnf->D == d->D, "cr_stack_block_bwd: the LayerNorm's input must be this block's y");
        CR_REQUIRE(lnf->gamma && lnf->dy && lnf->dgamma && lnf->dbeta && lnf->accumulate == 0, "cr_stack_block_bwd: LayerNorm backward arguments");
        CR_REQUIRE(lnf->slab_stride == bd->slab_stride && lnf->n_slabs == bd->n_slabs, "cr_stack_block_bwd: the LayerNorm's slabs must be the block's");
        CR_REQUIRE((size_t)lnf->M * lnf->lddy * 4 < ((size_t)1 << 32), "cr_stack_block_bwd: dy of 4 GiB or more");
        a.ln = *lnf;
        a.has_ln = 1;
    } else {
        CR_REQUIRE(bd->dy, "cr_stack_block_bwd: dy is NULL");
        CR_REQUIRE(x->lnf_dy2 == nullptr, "cr_stack_block_bwd: lnf_dy2 without a LayerNorm");
    }
    if (sc) {
        const cr_embed_desc* e = &sc->f;
        CR_REQUIRE(e->ids && e->M == d->M && e->D == d->D && e->ld_out == d->D && e->col_off == 0 && e->T > 0 && e->V > 0,
                   "cr_stack_block_bwd: the embedding recipe must describe the block's dense input x");
        CR_REQUIRE(!bd->dx_accumulate, "cr_stack_block_bwd: dx_accumulate with a scatter (this kernel must be the only producer of dx)");
        // (table_grad, pos_grad and d_addend all NULL is the occurrence-index form: the masked, dropped-out partial rows are LEFT in
        //  dx / dx2 -- d_addend / d_addend2 with an addend -- for cr_table_grad / cr_adam_desc.tg to gather; no atomics here)
        if (sc->n_slabs > 0) {
            // small-table mode (cr_embed_bwd's contract: every slab in use written, their sum is the gradient): two slabs per pair
            const int nw = B < bd->n_slabs ? B : bd->n_slabs;
            CR_REQUIRE(sc->table_grad && !sc->pos_grad && e->V <= 256 && sc->n_slabs >= 2 * nw,
                       "cr_stack_block_bwd: small-table scatter needs table_grad, no pos_grad, V <= 256 and n_slabs >= 2 * min(B, n_slabs of the block)");
            a.small = 1;
        }
        CR_REQUIRE(sc->d_addend == nullptr || (e->ld_add == d->D && x->d_addend2 != nullptr), "cr_stack_block_bwd: d_addend must be dense [M, D] and come with d_addend2");
        a.sc = *sc;
        a.scatter = 1;
        // the masked partial rows wait for phase 3 in d_addend / d_addend2 where the graph has an addend, else in dx / dx2
        a.sbuf = sc->d_addend ? sc->d_addend : bd->dx;
        a.sbuf2 = sc->d_addend ? x->d_addend2 : x->dx2;
        CR_REQUIRE(a.sbuf && a.sbuf2 && a.sbuf != a.sbuf2, "cr_stack_block_bwd: the scatter needs two [M, D] row buffers (d_addend + d_addend2, or dx + dx2)");
    } else {
        CR_REQUIRE(bd->dx && x->dx2 && bd->dx != x->dx2, "cr_stack_block_bwd: dx / dx2");
    }
    a.isd = 1.0f / sqrtf((float)ad->d);
    a.isd_log2e = a.isd * 1.4426950408889634f;
    a.invT = 1.0f / (float)T;
    b1_deal_tiles(a.nkt, true, a.qpk);
    b1_deal_tiles(a.nkt, false, a.kpk);
    a.ts = g_attn_ts_which == 9 ? g_attn_ts : nullptr;
    if (a.ts) {
        // CASTREC_TS_LAUNCH=k: stamps of the k-th launch after the hook was set (default: every launch, the last one stays)
        static const char* only = getenv("CASTREC_TS_LAUNCH");
        static unsigned long long* seen_for = nullptr;
        static int count = 0;
        if (seen_for != a.ts) { seen_for = a.ts; count = 0; }
        if (only && atoi(only) != count) a.ts = nullptr;
        ++count;
    }
    const int nwg = B < bd->n_slabs ? B : bd->n_slabs;
    const bool split = precision == CR_PREC_BF16X3;
    hipStream_t s = cr_stream(stream);
    // the hidden size as a compile-time constant where it is a common one (column-tile predicates fold: 240 .. 255 registers, no
    // scratch); any other size runs the generic instantiation (live predicates: it spills -- see DESIGN.md section 4)
    switch (d->D) {
    case 64: return split ? launch_b1<true, 64, 2>(a, nwg, s) : launch_b1<false, 64, 2>(a, nwg, s);      // two heads of 32 columns
    case 50:
        // (the headline's length -- 200 positions, 13 tiles -- as constants too: DS = D | nkt << 8 | T << 16)
        static const bool no_nk = getenv("CASTREC_B1_NO_NKT") != nullptr;       // (measurement switch)
        if (a.T == 200 && !no_nk) return split ? launch_b1<true, 50 | (13 << 8) | (200 << 16)>(a, nwg, s) : launch_b1<false, 50 | (13 << 8) | (200 << 16)>(a, nwg, s);
        return split ? launch_b1<true, 50>(a, nwg, s) : launch_b1<false, 50>(a, nwg, s);
    case 32: return split ? launch_b1<true, 32>(a, nwg, s) : launch_b1<false, 32>(a, nwg, s);
    case 40: return split ? launch_b1<true, 40>(a, nwg, s) : launch_b1<false, 40>(a, nwg, s);
    case 48: return split ? launch_b1<true, 48>(a, nwg, s) : launch_b1<false, 48>(a, nwg, s);
    case 56: return split ? launch_b1<true, 56>(a, nwg, s) : launch_b1<false, 56>(a, nwg, s);
    default: break;
    }
    // any other size: the instantiation of its FAMILY -- the number of whole 16-column tiles as a constant, the hidden size itself at run
    // time (cr_rlayout.hpp d_ctx<NF>).  Hidden sizes below 48: no scratch (238 / 252 registers); 49 .. 63 other than 50 / 56: 48 / 132 bytes
    // per lane, where round 4's fully generic instantiation (every size outside the constants) spilled 108 / 228
    // (-Rpass-analysis=kernel-resource-usage, tools/res_usage.sh cr_stack_bwd1.hip).
    switch (d->D / 16) {
    case 0: return split ? launch_b1<true, -1>(a, nwg, s) : launch_b1<false, -1>(a, nwg, s);
    case 1: return split ? launch_b1<true, -2>(a, nwg, s) : launch_b1<false, -2>(a, nwg, s);
    case 2: return split ? launch_b1<true, -3>(a, nwg, s) : launch_b1<false, -3>(a, nwg, s);
    default: return split ? launch_b1<true, -4>(a, nwg, s) : launch_b1<false, -4>(a, nwg, s);
    }
}
