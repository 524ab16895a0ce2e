// Prediction head (sasrec.py:87-115): pos/neg item gathers, row dot products, masked BCE + AUC sums,
// and -- fused in the same pass -- the gradients wrt the sequence embedding and the item table.
// Gradients are UN-normalised (scaled by n_target); cr_adam_step divides by n_target, which is only
// known after the whole batch (all ranks) has been reduced (sasrec.py:104-108).
#include <math.h>

#include <stdlib.h>
#include "cr_common.hpp"

// Row mapping as in cr_layernorm.hip: hidden sizes <= 64 give a row 16 lanes (4 rows per wave in flight,
// DPP row sums); larger ones a whole wave.
template <int LPR>
__device__ __forceinline__ float head_row_sum(float v) {
    if (LPR == 16) return cr_row16_sum(v);
    if (LPR == 32) {                                     // two DPP rows of 16 lanes: row sums, then the neighbouring row's
        v = cr_row16_sum(v);
        return v + __shfl_xor(v, 16, 64);
    }
    return wave_sum(v);
}

// (head_snapshot: cr_common.hpp)
template <int LPR, int MAXC>
__global__ __launch_bounds__(256) void k_head(cr_head_desc d) {
    constexpr int RPW = 64 / LPR;
    __shared__ float red[3][4 * RPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    float loss_acc = 0.0f, auc_acc = 0.0f, n_acc = 0.0f;
    for (int mb = (blockIdx.x * 4 + wave) * RPW; mb < d.M; mb += gridDim.x * 4 * RPW) {
        const int m = mb + sub;
        const bool act = m < d.M;
        const int mm = act ? m : d.M - 1;
        const int p = act ? d.pos[mm] : 0, ng = act ? d.neg[mm] : 0;
        float s[MAXC], ep[MAXC], en[MAXC];
        float pl = 0.0f, nl = 0.0f, gp = 0.0f, gn = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = l + LPR * i;
            const bool ok = c < d.D;
            s[i] = ok ? d.seq_emb[(size_t)mm * d.ld + c] : 0.0f;
            ep[i] = (ok && p != 0) ? d.table[(size_t)p * d.D + c] : 0.0f;      // row 0 == zeros (modules.py:154-156)
            en[i] = (ok && ng != 0) ? d.table[(size_t)ng * d.D + c] : 0.0f;
            pl += ep[i] * s[i];
            nl += en[i] * s[i];
        }
        pl = head_row_sum<LPR>(pl);                                            // sasrec.py:100
        nl = head_row_sum<LPR>(nl);                                            // sasrec.py:101
        const float ist = (p != 0) ? 1.0f : 0.0f;                              // sasrec.py:104
        const float sp = 1.0f / (1.0f + expf(-pl)), sn = 1.0f / (1.0f + expf(-nl));
        if (l == 0 && act) {
            loss_acc += ist * (-logf(sp + 1e-24f) - logf(1.0f - sn + 1e-24f)); // sasrec.py:105-108
            const float dlt = pl - nl;
            const float sg = (dlt > 0.0f) ? 1.0f : ((dlt < 0.0f) ? -1.0f : 0.0f);
            auc_acc += ist * (sg + 1.0f) * 0.5f;                               // sasrec.py:113-115
            n_acc += ist;
            if (d.pos_logits) d.pos_logits[m] = pl;
            if (d.neg_logits) d.neg_logits[m] = nl;
        }
        if (act && (d.d_seq_emb || d.table_grad || d.coef_out)) {
            // d/dpl [-log(sig(pl)+e)] = -sig(1-sig)/(sig+e);  d/dnl [-log(1-sig(nl)+e)] = sig(1-sig)/(1-sig+e)
            const float dpl = -ist * sp * (1.0f - sp) / (sp + 1e-24f);
            const float dnl = ist * sn * (1.0f - sn) / (1.0f - sn + 1e-24f);
            if (d.coef_out && l == 0) {                  // the item table's gradient is gathered from the batch's occurrence index
                d.coef_out[m] = dpl;
                d.coef_out[(size_t)d.M + m] = dnl;
            }
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = l + LPR * i;
                if (c < d.D) {
                    if (d.d_seq_emb) d.d_seq_emb[(size_t)m * d.ldd + c] = dpl * ep[i] + dnl * en[i];
                    if (LPR != 16 && d.table_grad && ist != 0.0f) {
                        if (p != 0) atomicAdd(d.table_grad + (size_t)p * d.D + c, dpl * s[i]);
                        if (ng != 0) atomicAdd(d.table_grad + (size_t)ng * d.D + c, dnl * s[i]);
                    }
                }
            }
            gp = dpl;
            gn = dnl;
        }
        if (LPR == 16 && d.table_grad) {
            // Table rows get one CONTIGUOUS 4*D-byte float-atomic burst each (lane = column), the shape the atomic
            // units take at full rate; with the 16-lanes-per-row mapping above one instruction carried four 64-byte
            // pieces of four different rows.  The row's scalars come from its lane group, s[] is re-read (L1 hit).
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int mr = mb + rr;
                const float gpr = __shfl(gp, rr * LPR, 64), gnr = __shfl(gn, rr * LPR, 64);
                const int pr = __shfl(p, rr * LPR, 64), nr = __shfl(ng, rr * LPR, 64);
                if (mr < d.M && pr != 0 && lane < d.D) {                       // ist == (pos id != 0)
                    const float sv = d.seq_emb[(size_t)mr * d.ld + lane];
                    atomicAdd(d.table_grad + (size_t)pr * d.D + lane, gpr * sv);
                    if (nr != 0) atomicAdd(d.table_grad + (size_t)nr * d.D + lane, gnr * sv);
                }
            }
        }
    }
    if (l == 0) { red[0][wave * RPW + sub] = loss_acc; red[1][wave * RPW + sub] = auc_acc; red[2][wave * RPW + sub] = n_acc; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float v = 0.0f;
        for (int i = 0; i < 4 * RPW; ++i) v += red[threadIdx.x][i];
        if (v != 0.0f) atomicAdd(d.state + threadIdx.x, v);
    }
    head_snapshot(d.state, gridDim.x);
}

extern "C" int cr_head_fwd_bwd(const cr_head_desc* d, void* stream) {
    CR_REQUIRE(d && d->seq_emb && d->table && d->pos && d->neg && d->state, "cr_head_fwd_bwd: NULL pointer");
    CR_REQUIRE(d->M > 0 && d->D > 0 && d->V > 0 && d->ld >= d->D, "cr_head_fwd_bwd: bad shape");
    if (d->D > 512) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_head_fwd_bwd: D=%d > 512", d->D);
    if (d->D <= 64) {
        int grid = cr_ceil_div(d->M, 16);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL((k_head<16, 4>), dim3(grid), dim3(256), 0, cr_stream(stream), *d);
    } else if (d->D <= 128) {
        int grid = cr_ceil_div(d->M, 8);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL((k_head<32, 4>), dim3(grid), dim3(256), 0, cr_stream(stream), *d);
    } else {
        int grid = cr_ceil_div(d->M, 4);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL((k_head<64, 8>), dim3(grid), dim3(256), 0, cr_stream(stream), *d);
    }
    return cr_check_launch("cr_head_fwd_bwd");
}

// Head + backward of the LayerNorm that produced seq_emb (sasrec.py:85), in one pass: the gradient row
// dy = dpl * E[pos] + dnl * E[neg] never leaves the registers it is computed in -- it goes straight through the
// LayerNorm backward of cr_layernorm.hip (same row mapping, same arithmetic, same slab reduction: n_slabs workgroups
// of 16 waves, workgroup s owns rows [s*rps, (s+1)*rps) and writes slab s of dgamma / dbeta).
template <int LPR, int MAXC>
__global__ __launch_bounds__(1024) void k_head_ln(cr_head_desc d, cr_ln_bwd_desc n) {
    constexpr int RPW = 64 / LPR;
    __shared__ float red[3][16 * RPW];
    cr_kernarg_touch<sizeof(cr_head_desc) + sizeof(cr_ln_bwd_desc)>();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    const int rps = (d.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * rps, m1 = min(d.M, m0 + rps);
    float gam[MAXC], ag[MAXC], ab[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = l + LPR * i;
        gam[i] = (c < d.D) ? n.gamma[c] : 0.0f;
        ag[i] = 0.0f; ab[i] = 0.0f;
    }
    const float invD = 1.0f / (float)d.D;
    float loss_acc = 0.0f, auc_acc = 0.0f, n_acc = 0.0f;
    // the ids of a row group are requested one iteration ahead: id -> table row is a dependent pair of memory round trips, and a
    // wave runs only two iterations (100 rows per workgroup, 16 waves x 4 rows)
    int p_n = 0, ng_n = 0;
    {
        const int m = m0 + wave * RPW + sub;
        if (m < m1) { p_n = d.pos[m]; ng_n = d.neg[m]; }
    }
    for (int mb = m0 + wave * RPW; mb < m1; mb += 16 * RPW) {
        const int m = mb + sub;
        const bool act = m < m1;
        const int mm = act ? m : m0;
        const int p = act ? p_n : 0, ng = act ? ng_n : 0;
        {
            const int mn = m + 16 * RPW;
            p_n = 0; ng_n = 0;
            if (mn < m1) { p_n = d.pos[mn]; ng_n = d.neg[mn]; }
        }
        float s[MAXC], ep[MAXC], en[MAXC], x[MAXC];
        float pl = 0.0f, nl = 0.0f, xs = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = l + LPR * i;
            const bool ok = c < d.D;
            s[i] = ok ? d.seq_emb[(size_t)mm * d.ld + c] : 0.0f;
            x[i] = ok ? n.x[(size_t)mm * n.ldx + c] : 0.0f;
            ep[i] = (ok && p != 0) ? d.table[(size_t)p * d.D + c] : 0.0f;      // row 0 == zeros (modules.py:154-156)
            en[i] = (ok && ng != 0) ? d.table[(size_t)ng * d.D + c] : 0.0f;
            pl += ep[i] * s[i];
            nl += en[i] * s[i];
            xs += x[i];
        }
        pl = head_row_sum<LPR>(pl);                                            // sasrec.py:100
        nl = head_row_sum<LPR>(nl);                                            // sasrec.py:101
        const float ist = (p != 0) ? 1.0f : 0.0f;                              // sasrec.py:104
        const float sp = 1.0f / (1.0f + expf(-pl)), sn = 1.0f / (1.0f + expf(-nl));
        if (l == 0 && act) {
            loss_acc += ist * (-logf(sp + 1e-24f) - logf(1.0f - sn + 1e-24f)); // sasrec.py:105-108
            const float dlt = pl - nl;
            const float sgn = (dlt > 0.0f) ? 1.0f : ((dlt < 0.0f) ? -1.0f : 0.0f);
            auc_acc += ist * (sgn + 1.0f) * 0.5f;                              // sasrec.py:113-115
            n_acc += ist;
            if (d.pos_logits) d.pos_logits[m] = pl;
            if (d.neg_logits) d.neg_logits[m] = nl;
        }
        const float dpl = act ? -ist * sp * (1.0f - sp) / (sp + 1e-24f) : 0.0f;
        const float dnl = act ? ist * sn * (1.0f - sn) / (1.0f - sn + 1e-24f) : 0.0f;
        if (d.coef_out && l == 0 && act) {               // the item table's gradient is gathered from the batch's occurrence index
            d.coef_out[m] = dpl;
            d.coef_out[(size_t)d.M + m] = dnl;
        }
        // ---- LayerNorm backward of this row (modules.py:74-78), dy in registers
        const float mean = head_row_sum<LPR>(xs) * invD;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const float dxm = (l + LPR * i < d.D) ? (x[i] - mean) : 0.0f;
            v += dxm * dxm;
        }
        const float rstd = 1.0f / sqrtf(head_row_sum<LPR>(v) * invD + n.eps);
        float dy[MAXC], c1 = 0.0f, c2 = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = l + LPR * i;
            dy[i] = dpl * ep[i] + dnl * en[i];
            if (act && c < d.D && d.d_seq_emb) d.d_seq_emb[(size_t)m * d.ldd + c] = dy[i];
            const float xh = (c < d.D) ? (x[i] - mean) * rstd : 0.0f;
            x[i] = xh;
            const float dg = dy[i] * gam[i];
            c1 += dg;
            c2 += dg * xh;
            ag[i] += dy[i] * xh;
            ab[i] += dy[i];
        }
        c1 = head_row_sum<LPR>(c1) * invD;
        c2 = head_row_sum<LPR>(c2) * invD;
        if (act) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = l + LPR * i;
                if (c < d.D) n.dx[(size_t)m * n.lddx + c] = cr_ln_bwd_tail(dy[i] * gam[i], c1, x[i], c2, rstd);
            }
        }
        if (d.table_grad) {
            if (LPR == 16) {
                // contiguous 4*D-byte float-atomic burst per table row (lane = column), as in k_head
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const int mr = mb + rr;
                    const float gpr = __shfl(dpl, rr * LPR, 64), gnr = __shfl(dnl, rr * LPR, 64);
                    const int pr = __shfl(p, rr * LPR, 64), nr = __shfl(ng, rr * LPR, 64);
                    if (mr < m1 && pr != 0 && lane < d.D) {
                        const float sv = d.seq_emb[(size_t)mr * d.ld + lane];
                        atomicAdd(d.table_grad + (size_t)pr * d.D + lane, gpr * sv);
                        if (nr != 0) atomicAdd(d.table_grad + (size_t)nr * d.D + lane, gnr * sv);
                    }
                }
            } else if (act && ist != 0.0f) {
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = l + LPR * i;
                    if (c < d.D) {
                        if (p != 0) atomicAdd(d.table_grad + (size_t)p * d.D + c, dpl * s[i]);
                        if (ng != 0) atomicAdd(d.table_grad + (size_t)ng * d.D + c, dnl * s[i]);
                    }
                }
            }
        }
    }
    if (l == 0) { red[0][wave * RPW + sub] = loss_acc; red[1][wave * RPW + sub] = auc_acc; red[2][wave * RPW + sub] = n_acc; }
    // dgamma / dbeta: row groups of the wave by shuffles, then the 16 waves in a fixed order (cr_layernorm.hip)
    __syncthreads();
    if (threadIdx.x < 3) {
        float v = 0.0f;
        for (int i = 0; i < 16 * RPW; ++i) v += red[threadIdx.x][i];
        if (v != 0.0f) atomicAdd(d.state + threadIdx.x, v);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
            ag[i] += __shfl_xor(ag[i], o, 64);
            ab[i] += __shfl_xor(ab[i], o, 64);
        }
    }
    if (LPR == 16) {                                     // per-wave slots + one barrier, as in cr_layernorm.hip
        __shared__ float wg[16][64], wb[16][64];
        if (sub == 0) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = l + LPR * i;
                if (c < 64) { wg[wave][c] = ag[i]; wb[wave][c] = ab[i]; }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < d.D) {
            float g = 0.0f, b = 0.0f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { g += wg[w][threadIdx.x]; b += wb[w][threadIdx.x]; }
            n.dgamma[(size_t)blockIdx.x * n.slab_stride + threadIdx.x] = g;
            n.dbeta[(size_t)blockIdx.x * n.slab_stride + threadIdx.x] = b;
        }
    } else {
        __shared__ float wide[16][512];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (sub == 0) {
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = l + LPR * i;
                    if (c < 512) wide[wave][c] = pass == 0 ? ag[i] : ab[i];
                }
            }
            __syncthreads();
            float* dst = pass == 0 ? n.dgamma : n.dbeta;
            for (int c = threadIdx.x; c < d.D; c += 1024) {
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < 16; ++w) v += wide[w][c];
                dst[(size_t)blockIdx.x * n.slab_stride + c] = v;
            }
            __syncthreads();
        }
    }
    head_snapshot(d.state, gridDim.x);
}

// The same kernel with VECTOR row accesses (round 5): a lane owns VEC consecutive columns (16- or 8-byte loads: D a multiple of 4,
// or even), LPR lanes a row (LPR * VEC >= D), and the loop is software-pipelined: the four rows an iteration reads (sequence
// embedding, LayerNorm input, pos row, neg row) are requested one iteration ahead, the ids two.  k_head_ln above reads a row as
// LPR-strided dwords (four load instructions per array and row group) and starts each iteration's loads behind the previous one's
// stores: 14.3 us at the headline shape (two dependent iterations per wave), 81 us for one C5 step's 131 072 table rows (0.52 of the
// HBM roof over all its bytes).  This kernel is bound by load ISSUE and dependent round trips, so both changes go straight to time.
typedef float hd_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float hd_f2 __attribute__((ext_vector_type(2), aligned(4)));
template <int VEC> struct HdVec { float v[VEC]; };
template <int VEC>
__device__ __forceinline__ HdVec<VEC> hd_load(const float* p, bool ok) {
    HdVec<VEC> r;
    if constexpr (VEC == 4) {
        hd_f4 t = (hd_f4){0.f, 0.f, 0.f, 0.f};
        if (ok) t = *reinterpret_cast<const hd_f4*>(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
        hd_f2 t = (hd_f2){0.f, 0.f};
        if (ok) t = *reinterpret_cast<const hd_f2*>(p);
        r.v[0] = t.x; r.v[1] = t.y;
    }
    return r;
}
template <int VEC>
__device__ __forceinline__ void hd_store(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) *reinterpret_cast<hd_f4*>(p) = (hd_f4){v[0], v[1], v[2], v[3]};
    else *reinterpret_cast<hd_f2*>(p) = (hd_f2){v[0], v[1]};
}

template <int LPR, int VEC>
__global__ __launch_bounds__(1024) void k_head_ln_v(cr_head_desc d, cr_ln_bwd_desc n) {
    constexpr int RPW = 64 / LPR;
    __shared__ float red[3][16 * RPW];
    __shared__ float wg[16][LPR * VEC], wb[16][LPR * VEC];
    cr_kernarg_touch<sizeof(cr_head_desc) + sizeof(cr_ln_bwd_desc)>();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    const int c0 = VEC * l;
    const bool cok = c0 < d.D;
    const int rps = (d.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * rps, m1 = min(d.M, m0 + rps);
    constexpr int STEP = 16 * RPW;                        // rows per iteration of the workgroup
    float gam[VEC], ag[VEC], ab[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) { gam[u] = (c0 + u < d.D) ? n.gamma[c0 + u] : 0.0f; ag[u] = 0.0f; ab[u] = 0.0f; }
    const float invD = 1.0f / (float)d.D;
    float loss_acc = 0.0f, auc_acc = 0.0f, n_acc = 0.0f;
    // pipeline: ids of iterations 0 and 1, rows of iteration 0
    const int mfirst = m0 + wave * RPW + sub;
    int p_a = 0, ng_a = 0, p_b = 0, ng_b = 0;             // ids of this iteration / the next one
    if (mfirst < m1) { p_a = d.pos[mfirst]; ng_a = d.neg[mfirst]; }
    if (mfirst + STEP < m1) { p_b = d.pos[mfirst + STEP]; ng_b = d.neg[mfirst + STEP]; }
    HdVec<VEC> s_n, x_n, ep_n, en_n;
    {
        const bool act = mfirst < m1;
        const int mm = act ? mfirst : m0;
        s_n = hd_load<VEC>(d.seq_emb + (size_t)mm * d.ld + c0, cok);
        x_n = hd_load<VEC>(n.x + (size_t)mm * n.ldx + c0, cok);
        ep_n = hd_load<VEC>(d.table + (size_t)p_a * d.D + c0, cok && p_a != 0);      // row 0 == zeros (modules.py:154-156)
        en_n = hd_load<VEC>(d.table + (size_t)ng_a * d.D + c0, cok && ng_a != 0);
    }
    for (int mb = m0 + wave * RPW; mb < m1; mb += STEP) {
        const int m = mb + sub;
        const bool act = m < m1;
        const int p = act ? p_a : 0;
        const HdVec<VEC> s = s_n, xv = x_n, ep = ep_n, en = en_n;
        // the next iteration's rows (its ids arrived an iteration ago), the ids of the one after
        {
            const int mn = m + STEP;
            const bool actn = mn < m1;
            const int mmn = actn ? mn : m0;
            const int pn = actn ? p_b : 0, nn = actn ? ng_b : 0;
            s_n = hd_load<VEC>(d.seq_emb + (size_t)mmn * d.ld + c0, cok);
            x_n = hd_load<VEC>(n.x + (size_t)mmn * n.ldx + c0, cok);
            ep_n = hd_load<VEC>(d.table + (size_t)pn * d.D + c0, cok && pn != 0);
            en_n = hd_load<VEC>(d.table + (size_t)nn * d.D + c0, cok && nn != 0);
            p_a = p_b; ng_a = ng_b;
            p_b = 0; ng_b = 0;
            if (mn + STEP < m1) { p_b = d.pos[mn + STEP]; ng_b = d.neg[mn + STEP]; }
        }
        float pl = 0.0f, nl = 0.0f, xs = 0.0f;
#pragma unroll
        for (int u = 0; u < VEC; ++u) { pl += ep.v[u] * s.v[u]; nl += en.v[u] * s.v[u]; xs += xv.v[u]; }
        pl = head_row_sum<LPR>(pl);                                            // sasrec.py:100
        nl = head_row_sum<LPR>(nl);                                            // sasrec.py:101
        const float ist = (p != 0) ? 1.0f : 0.0f;                              // sasrec.py:104
        const float sp = 1.0f / (1.0f + expf(-pl)), sn = 1.0f / (1.0f + expf(-nl));
        if (l == 0 && act) {
            loss_acc += ist * (-logf(sp + 1e-24f) - logf(1.0f - sn + 1e-24f)); // sasrec.py:105-108
            const float dlt = pl - nl;
            const float sgn = (dlt > 0.0f) ? 1.0f : ((dlt < 0.0f) ? -1.0f : 0.0f);
            auc_acc += ist * (sgn + 1.0f) * 0.5f;                              // sasrec.py:113-115
            n_acc += ist;
            if (d.pos_logits) d.pos_logits[m] = pl;
            if (d.neg_logits) d.neg_logits[m] = nl;
        }
        const float dpl = act ? -ist * sp * (1.0f - sp) / (sp + 1e-24f) : 0.0f;
        const float dnl = act ? ist * sn * (1.0f - sn) / (1.0f - sn + 1e-24f) : 0.0f;
        if (d.coef_out && l == 0 && act) {
            d.coef_out[m] = dpl;
            d.coef_out[(size_t)d.M + m] = dnl;
        }
        // ---- LayerNorm backward of this row (modules.py:74-78), dy in registers
        const float mean = head_row_sum<LPR>(xs) * invD;
        float v = 0.0f, xh[VEC];
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            const float dxm = (c0 + u < d.D) ? (xv.v[u] - mean) : 0.0f;
            xh[u] = dxm;
            v += dxm * dxm;
        }
        const float rstd = 1.0f / sqrtf(head_row_sum<LPR>(v) * invD + n.eps);
        float dy[VEC], c1 = 0.0f, c2 = 0.0f;
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            dy[u] = dpl * ep.v[u] + dnl * en.v[u];
            xh[u] *= rstd;
            const float dg = dy[u] * gam[u];
            c1 += dg;
            c2 += dg * xh[u];
            ag[u] += dy[u] * xh[u];
            ab[u] += dy[u];
        }
        if (act && cok && d.d_seq_emb) hd_store<VEC>(d.d_seq_emb + (size_t)m * d.ldd + c0, dy);
        c1 = head_row_sum<LPR>(c1) * invD;
        c2 = head_row_sum<LPR>(c2) * invD;
        if (act && cok) {
            float o[VEC];
#pragma unroll
            for (int u = 0; u < VEC; ++u) o[u] = cr_ln_bwd_tail(dy[u] * gam[u], c1, xh[u], c2, rstd);
            hd_store<VEC>(n.dx + (size_t)m * n.lddx + c0, o);
        }
    }
    if (l == 0) { red[0][wave * RPW + sub] = loss_acc; red[1][wave * RPW + sub] = auc_acc; red[2][wave * RPW + sub] = n_acc; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float v = 0.0f;
        for (int i = 0; i < 16 * RPW; ++i) v += red[threadIdx.x][i];
        if (v != 0.0f) atomicAdd(d.state + threadIdx.x, v);
    }
    // dgamma / dbeta: the wave's row groups by shuffles, then the 16 waves in a fixed order
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
            ag[u] += __shfl_xor(ag[u], o, 64);
            ab[u] += __shfl_xor(ab[u], o, 64);
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int u = 0; u < VEC; ++u) { wg[wave][c0 + u] = ag[u]; wb[wave][c0 + u] = ab[u]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d.D; c += 1024) {
        float g = 0.0f, b = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { g += wg[w][c]; b += wb[w][c]; }
        n.dgamma[(size_t)blockIdx.x * n.slab_stride + c] = g;
        n.dbeta[(size_t)blockIdx.x * n.slab_stride + c] = b;
    }
    head_snapshot(d.state, gridDim.x);
}

extern "C" int cr_head_fwd_bwd_ln(const cr_head_desc* d, const cr_ln_bwd_desc* n, void* stream) {
    CR_REQUIRE(d && d->seq_emb && d->table && d->pos && d->neg && d->state, "cr_head_fwd_bwd_ln: NULL pointer");
    CR_REQUIRE(d->M > 0 && d->D > 0 && d->V > 0 && d->ld >= d->D, "cr_head_fwd_bwd_ln: bad shape");
    CR_REQUIRE(n && n->x && n->gamma && n->dx && n->dgamma && n->dbeta, "cr_head_fwd_bwd_ln: NULL LayerNorm pointer");
    CR_REQUIRE(n->M == d->M && n->D == d->D && n->n_slabs > 0 && n->ldx >= d->D && n->lddx >= d->D,
               "cr_head_fwd_bwd_ln: the LayerNorm description must match the head's rows");
    CR_REQUIRE(!n->accumulate, "cr_head_fwd_bwd_ln: accumulate is not supported");
    if (d->D > 512) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_head_fwd_bwd_ln: D=%d > 512", d->D);
    // lanes per row: 16 (D <= 64), 32 (D <= 128: two rows per wave and DPP row sums -- with a whole wave per row the eight rows
    // of a wave at config C4 were eight serial gather -> reduce -> atomics passes: 36 us), else 64
    // vector row accesses where the shape allows them, and nothing scatters from this kernel (the occurrence-index form)
    static const bool no_vec = getenv("CASTREC_HEAD_NO_VEC") != nullptr;
    const bool al = ((d->ld | n->ldx | n->lddx | (d->d_seq_emb ? d->ldd : 0)) % 2) == 0;
    // (D = 50: 32 lanes x 8 bytes halve the rows a wave holds per iteration -- four dependent iterations instead of two at the headline
    //  shape: 17.6 against 16.5 us; 16-byte lanes keep four rows per wave up to 64 columns, and above 64 the scalar form's strided dwords
    //  lose outright: 82.2 -> 65.1 us for one C5 step's rows, 0.51 -> 0.65 of the HBM roof, bench.py gather.head_ln)
    if (!no_vec && !d->table_grad && al && d->D % 2 == 0 && d->D <= 256 && (d->D % 4 == 0 || d->D > 64)) {
        const bool v4 = d->D % 4 == 0 && ((d->ld | n->ldx | n->lddx | (d->d_seq_emb ? d->ldd : 0)) % 4) == 0;
#define HD_LAUNCH(L, V) hipLaunchKernelGGL((k_head_ln_v<L, V>), dim3(n->n_slabs), dim3(1024), 0, cr_stream(stream), *d, *n)
        if (v4 && d->D <= 64) HD_LAUNCH(16, 4);
        else if (v4 && d->D <= 128) HD_LAUNCH(32, 4);
        else if (v4) HD_LAUNCH(64, 4);
        else if (d->D <= 32) HD_LAUNCH(16, 2);
        else if (d->D <= 64) HD_LAUNCH(32, 2);
        else if (d->D <= 128) HD_LAUNCH(64, 2);
        else goto scalar_form;
#undef HD_LAUNCH
        return cr_check_launch("cr_head_fwd_bwd_ln");
    }
scalar_form:
    if (d->D <= 64)
        hipLaunchKernelGGL((k_head_ln<16, 4>), dim3(n->n_slabs), dim3(1024), 0, cr_stream(stream), *d, *n);
    else if (d->D <= 128)
        hipLaunchKernelGGL((k_head_ln<32, 4>), dim3(n->n_slabs), dim3(1024), 0, cr_stream(stream), *d, *n);
    else
        hipLaunchKernelGGL((k_head_ln<64, 8>), dim3(n->n_slabs), dim3(1024), 0, cr_stream(stream), *d, *n);
    return cr_check_launch("cr_head_fwd_bwd_ln");
}

// test_logits (sasrec.py:93-97): last position of every sequence against its candidate items.
__global__ __launch_bounds__(256) void k_test_logits(const float* seq_emb, int ld, const float* table, const int32_t* cand,
                                                     int B, int T, int D, int n_cand, float* logits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const float* s = seq_emb + (size_t)(b * T + T - 1) * ld;
    for (int j = wave; j < n_cand; j += 4) {
        const int id = cand[(size_t)b * n_cand + j];
        float acc = 0.0f;
        if (id != 0)
            for (int c = lane; c < D; c += 64) acc += s[c] * table[(size_t)id * D + c];
        acc = wave_sum(acc);
        if (lane == 0) logits[(size_t)b * n_cand + j] = acc;
    }
}

// The same for hidden sizes that are multiples of 4 (16-byte rows): a read-only row gather, the form the HBM-read
// roofline of the item table is measured on (bench.py "gather" block).  16 lanes own a candidate row (float4 per lane
// per 64 columns), a wave keeps 4 rows x NR rounds in flight, the row sum is four DPP adds.  NV = D / 64 rounded up.
template <int NV, int NR, bool STREAM>
__global__ __launch_bounds__(256) void k_test_logits_v4(const float* seq_emb, int ld, const float* table, const int32_t* cand,
                                                        int B, int T, int D, int n_cand, float* logits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, grp = (threadIdx.x >> 4);   // 16 groups per block
    const int b = blockIdx.x;
    const float* s = seq_emb + (size_t)(b * T + T - 1) * ld;
    typedef float f4a __attribute__((ext_vector_type(4), aligned(4)));
    f4a sv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * li + 64 * i;
        sv[i] = c < D ? *reinterpret_cast<const f4a*>(s + c) : (f4a){0.f, 0.f, 0.f, 0.f};
    }
    // NR candidates per group and iteration
    for (int j0 = grp * NR; j0 < n_cand; j0 += 16 * NR) {
        int id[NR];
        float4 rv[NR][NV];
#pragma unroll
        for (int r = 0; r < NR; ++r) id[r] = (j0 + r < n_cand) ? cand[(size_t)b * n_cand + j0 + r] : 0;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = 4 * li + 64 * i;
                // STREAM (a table larger than the Infinity Cache: a row is not read again before it has left every cache): streaming loads --
                // 76.7 -> 69.2 us for 425 MB of fresh rows at config C5's table, 69 -> 77 % of the 8 TB/s peak (bench.py gather block).  NOT in
                // cr_embed_fwd, the training-path gather, which writes as many bytes as it reads: there streaming loads cost 5 %, and with
                // both kernels streaming this one's gain was gone as well (tools/probes/run_gather.sh: CASTREC_GATHER_STREAM)
                typedef float f4n __attribute__((ext_vector_type(4)));
                const f4n* src = reinterpret_cast<const f4n*>(table + (size_t)id[r] * D + (c < D ? c : 0));
                const f4n t4 = STREAM ? __builtin_nontemporal_load(src) : *src;
                rv[r][i] = make_float4(t4.x, t4.y, t4.z, t4.w);
            }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool ok = 4 * li + 64 * i < D;
                acc += ok ? (sv[i].x * rv[r][i].x + sv[i].y * rv[r][i].y + sv[i].z * rv[r][i].z + sv[i].w * rv[r][i].w) : 0.0f;
            }
            acc = cr_row16_sum(acc);
            if (li == 0 && j0 + r < n_cand) logits[(size_t)b * n_cand + j0 + r] = id[r] != 0 ? acc : 0.0f;   // row 0 reads as zeros
        }
    }
    (void)wave;
}

extern "C" int cr_test_logits(const float* seq_emb, int ld, const float* table, const int32_t* cand, int B, int T, int D,
                              int V, int n_cand, float* logits, void* stream) {
    CR_REQUIRE(seq_emb && table && cand && logits, "cr_test_logits: NULL pointer");
    CR_REQUIRE(B > 0 && T > 0 && D > 0 && V > 0 && n_cand > 0 && ld >= D, "cr_test_logits: bad shape");
    hipStream_t st = cr_stream(stream);
    if (D % 4 == 0 && D <= 256 && (reinterpret_cast<uintptr_t>(table) & 15) == 0) {
        // rows in flight per 16-lane group at D > 128 (1 KB rows): 7 puts the evaluator's 101 candidates of a query into ONE batch
        // (16 groups x 7): 68.2 % / 68.7 % / 70.0 % / 71.1 % of the 8 TB/s HBM peak for 1 / 2 / 4 / 7 at config C5's table
        // (tools/gather_sweep.py, a fresh row set per launch)
        static const int nr = getenv("CASTREC_TL_NR") ? atoi(getenv("CASTREC_TL_NR")) : 7;
        static const char* gs = getenv("CASTREC_GATHER_STREAM");            // (measurement override: 0 / 1)
        const bool big = gs ? atoi(gs) != 0 : (size_t)V * D * 4 >= ((size_t)256 << 20);       // streaming row loads: see k_test_logits_v4
#define TL_LAUNCH(NV, NRR)                                                                                                                    \
    do {                                                                                                                                      \
        if (big) hipLaunchKernelGGL((k_test_logits_v4<NV, NRR, true>), dim3(B), dim3(256), 0, st, seq_emb, ld, table, cand, B, T, D, n_cand, logits);   \
        else hipLaunchKernelGGL((k_test_logits_v4<NV, NRR, false>), dim3(B), dim3(256), 0, st, seq_emb, ld, table, cand, B, T, D, n_cand, logits);     \
    } while (0)
        if (D <= 64) TL_LAUNCH(1, 2);
        else if (D <= 128) TL_LAUNCH(2, 2);
        else if (nr == 4) TL_LAUNCH(4, 4);
        else if (nr == 1) TL_LAUNCH(4, 1);
        else if (nr == 7) TL_LAUNCH(4, 7);
        else TL_LAUNCH(4, 2);
#undef TL_LAUNCH
        return cr_check_launch("cr_test_logits");
    }
    hipLaunchKernelGGL(k_test_logits, dim3(B), dim3(256), 0, st, seq_emb, ld, table, cand, B, T, D, n_cand, logits);
    return cr_check_launch("cr_test_logits");
}
