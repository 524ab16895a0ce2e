// Embedding gather (+ scale, positional row, addend, dropout, row mask) and its backward.
// Reference: modules.py:83-164 `embedding`; input composition sasrec.py:27-62, cast_1.py:86-91.
// HBM-bound op: one wavefront per row, lanes sweep the row's columns (coalesced 256-B segments).
#include "cr_common.hpp"

__global__ __launch_bounds__(256) void k_embed_fwd(cr_embed_desc d) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= d.M) return;
    const int id = d.ids[m];
    const bool zero = (d.zero_pad && id == 0);
    const float keep_row = (d.mask_ids && d.mask_ids[m] == 0) ? 0.0f : 1.0f;
    const DropCtx dc = drop_ctx(d.drop);
    const int t = m % d.T;
    const float* row = d.table + (size_t)id * d.D;
    for (int c = lane; c < d.D; c += 64) {
        float v = zero ? 0.0f : row[c] * d.scale;
        if (d.pos_table) v += d.pos_table[(size_t)t * d.D + c];
        if (d.addend) v += d.addend[(size_t)m * d.ld_add + c];
        v = drop_apply(dc, (d.drop.row_offset + (uint32_t)m) * (uint32_t)d.D + (uint32_t)c, v);
        d.out[(size_t)m * d.ld_out + d.col_off + c] = v * keep_row;
    }
}

// Vectorised gather: a row is ceil(D/4) 16-byte chunks (dword aligned is all gfx950 global memory needs),
// LPR lanes per row (the next power of two, <= 64), 64/LPR rows per wave-instruction, and each wave keeps
// R = 4 row groups in flight (independent loads issued before any store) -- the shape the HBM-bound C5 gather
// (1 KiB rows out of a 10 GB table) needs, and 4x fewer vector-memory instructions than one dword per lane at
// the small hidden sizes.  The chunk that crosses column D is read shifted back to [D-4, D) and rotated, so
// nothing outside a row is touched (D >= 4).
typedef float f4e __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void rot4(f4e v, int shift, float (&e)[4]) {
    e[0] = shift == 0 ? v.x : (shift == 1 ? v.y : (shift == 2 ? v.z : v.w));
    e[1] = shift == 0 ? v.y : (shift == 1 ? v.z : (shift == 2 ? v.w : 0.0f));
    e[2] = shift == 0 ? v.z : (shift == 1 ? v.w : 0.0f);
    e[3] = shift == 0 ? v.w : 0.0f;
}
template <int LPR>
__global__ __launch_bounds__(256) void k_embed_fwd_vec(cr_embed_desc d) {
    // R row groups in flight per wave.  Two batches of independent loads: first every group's id (+ row mask), then every
    // group's table row and positional row, unconditionally (the zero-pad row 0 is read like any other and zeroed by a select:
    // a branch per group put each group's id -> row chain behind the previous one); then the arithmetic and the stores.
    constexpr int RPW = 64 / LPR, R = LPR >= 32 ? 8 : 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;          // row within the group, 4-column chunk within the row
    const int nchunk = (d.D + 3) >> 2;
    const int c = 4 * l, col0 = min(c, d.D - 4), shift = c - col0;
    const DropCtx dc = drop_ctx(d.drop);
    const int rows_per_iter = RPW * R;
    for (int mb = (blockIdx.x * 4 + wave) * rows_per_iter; mb < d.M; mb += gridDim.x * 4 * rows_per_iter) {
        f4e v[R], pv[R];
        int mrow[R], id[R], mk[R];
        bool act[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int m = mb + u * RPW + sub;
            mrow[u] = m;
            act[u] = (m < d.M) && (l < nchunk);
            const int mc = min(m, d.M - 1);
            id[u] = d.ids[mc];
            mk[u] = d.mask_ids ? d.mask_ids[mc] : 1;
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            v[u] = *reinterpret_cast<const f4e*>(d.table + (size_t)id[u] * d.D + col0);
            pv[u] = (f4e){0.f, 0.f, 0.f, 0.f};
            if (d.pos_table) pv[u] = *reinterpret_cast<const f4e*>(d.pos_table + (size_t)(min(mrow[u], d.M - 1) % d.T) * d.D + col0);   // (wave-uniform branch)
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (!act[u]) continue;
            const int m = mrow[u];
            float x[4], p[4];
            if (d.zero_pad && id[u] == 0) v[u] = (f4e){0.f, 0.f, 0.f, 0.f};
            rot4(v[u], shift, x);
            rot4(pv[u], shift, p);
            const bool dead = mk[u] == 0;
            const uint32_t base = (d.drop.row_offset + (uint32_t)m) * (uint32_t)d.D + (uint32_t)c;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float y = x[t] * d.scale + p[t];
                if (d.addend && c + t < d.D) y += d.addend[(size_t)m * d.ld_add + c + t];
                if (dc.on) y = drop_apply(dc, base + (uint32_t)t, y);
                x[t] = dead ? 0.0f : y;
            }
            float* o = d.out + (size_t)m * d.ld_out + d.col_off + c;
            if (c + 3 < d.D) {
                *reinterpret_cast<f4e*>(o) = (f4e){x[0], x[1], x[2], x[3]};
            } else {
                o[0] = x[0];
                if (c + 1 < d.D) o[1] = x[1];
                if (c + 2 < d.D) o[2] = x[2];
            }
        }
    }
}

static int embed_lpr(int D) {                              // lanes per row: next power of two >= ceil(D/4)
    const int nchunk = (D + 3) / 4;
    int lpr = 1;
    while (lpr < nchunk) lpr <<= 1;
    return lpr;
}
static bool embed_vec_ok(const cr_embed_desc* d) { return d->D >= 4 && (d->D + 3) / 4 <= 64; }

extern "C" int cr_embed_fwd(const cr_embed_desc* d, void* stream) {
    CR_REQUIRE(d && d->ids && d->table && d->out, "cr_embed_fwd: NULL pointer");
    CR_REQUIRE(d->M > 0 && d->T > 0 && d->D > 0 && d->V > 0 && d->M % d->T == 0, "cr_embed_fwd: bad shape M=%d T=%d D=%d", d->M, d->T, d->D);
    CR_REQUIRE(d->ld_out >= d->col_off + d->D, "cr_embed_fwd: ld_out too small");
    if (embed_vec_ok(d)) {
        const int lpr = embed_lpr(d->D);
        const int rows_per_block = 4 * (64 / lpr) * (lpr >= 32 ? 8 : 4);     // 4 waves x rows per group x groups in flight (k_embed_fwd_vec's R)
        int grid = cr_ceil_div(d->M, rows_per_block);
        if (grid > 8192) grid = 8192;
        hipStream_t s = cr_stream(stream);
        switch (lpr) {
            case 64: hipLaunchKernelGGL((k_embed_fwd_vec<64>), dim3(grid), dim3(256), 0, s, *d); break;
            case 32: hipLaunchKernelGGL((k_embed_fwd_vec<32>), dim3(grid), dim3(256), 0, s, *d); break;
            case 16: hipLaunchKernelGGL((k_embed_fwd_vec<16>), dim3(grid), dim3(256), 0, s, *d); break;
            case 8: hipLaunchKernelGGL((k_embed_fwd_vec<8>), dim3(grid), dim3(256), 0, s, *d); break;
            case 4: hipLaunchKernelGGL((k_embed_fwd_vec<4>), dim3(grid), dim3(256), 0, s, *d); break;
            case 2: hipLaunchKernelGGL((k_embed_fwd_vec<2>), dim3(grid), dim3(256), 0, s, *d); break;
            default: hipLaunchKernelGGL((k_embed_fwd_vec<1>), dim3(grid), dim3(256), 0, s, *d); break;
        }
        return cr_check_launch("cr_embed_fwd(vec)");
    }
    hipLaunchKernelGGL(k_embed_fwd, dim3(cr_ceil_div(d->M, 4)), dim3(256), 0, cr_stream(stream), *d);
    return cr_check_launch("cr_embed_fwd");
}

// Backward: one workgroup per position t sweeps the batch; the positional-table gradient
// (sum over the batch) is reduced in registers + LDS (no atomics), the item rows are
// scatter-added with one 4*D-byte contiguous float-atomic burst per row.
#define EMB_MAXC 8   // columns per lane: D <= 512
#define EMB_BW 16    // waves per workgroup: a wave's rows are a serial chain of load -> atomics passes (4 waves x 8 passes at B = 128
                     // made the 200 workgroups of config C4 take 29 us; 16 waves x 2 passes: the chip has the wave slots)

// gradient of the forward's output at element idx: f.out (+ out2: a gradient that arrives as two partials).  HAS2 is a
// template constant of the kernels: a run-time `out2 ? ... : ...` per element put a branch between the two loads of every
// element (k_embed_bwd_small: 10.9 -> 22.9 us), with the constant both are plain loads of one batch
template <bool HAS2>
__device__ __forceinline__ float dout_at(const cr_embed_bwd_desc& bd, size_t idx) {
    const float v = bd.f.out[idx];
    return HAS2 ? v + bd.out2[idx] : v;
}

template <bool HAS2>
__global__ __launch_bounds__(64 * EMB_BW) void k_embed_bwd(cr_embed_bwd_desc bd) {
    const cr_embed_desc& d = bd.f;
    __shared__ float red[EMB_BW][64 * EMB_MAXC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x;
    const int B = d.M / d.T;
    const DropCtx dc = drop_ctx(d.drop);
    float acc[EMB_MAXC];
#pragma unroll
    for (int i = 0; i < EMB_MAXC; ++i) acc[i] = 0.0f;
    // U rows per pass: their ids, masks and gradient rows are requested together (one latency for U rows),
    // then scattered; the per-column accumulation order over b stays fixed
    constexpr int U = 4;
    for (int b0 = wave; b0 < B; b0 += EMB_BW * U) {
        int mm[U], id[U];
        float keep[U], g[U][EMB_MAXC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + EMB_BW * u;
            mm[u] = (b < B ? b : b0) * d.T + t;
            id[u] = d.ids[mm[u]];
            keep[u] = (b < B && !(d.mask_ids && d.mask_ids[mm[u]] == 0)) ? 1.0f : 0.0f;
#pragma unroll
            for (int i = 0; i < EMB_MAXC; ++i) {
                const int c = lane + 64 * i;
                g[u][i] = (c < d.D) ? dout_at<HAS2>(bd, (size_t)mm[u] * d.ld_out + d.col_off + c) : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (b0 + EMB_BW * u >= B) break;
            const int m = mm[u];
            const bool skip_table = (d.zero_pad && id[u] == 0) || bd.table_grad == nullptr;
#pragma unroll
            for (int i = 0; i < EMB_MAXC; ++i) {
                const int c = lane + 64 * i;
                if (c < d.D) {
                    float gv = g[u][i] * keep[u];
                    gv = drop_apply(dc, (d.drop.row_offset + (uint32_t)m) * (uint32_t)d.D + (uint32_t)c, gv);
                    acc[i] += gv;
                    if (bd.d_addend) bd.d_addend[(size_t)m * d.ld_add + c] = gv;
                    if (!skip_table) atomicAdd(bd.table_grad + (size_t)id[u] * d.D + c, gv * d.scale);
                }
            }
        }
    }
    if (bd.pos_grad) {
#pragma unroll
        for (int i = 0; i < EMB_MAXC; ++i) red[wave][lane + 64 * i] = acc[i];
        __syncthreads();
        for (int c = threadIdx.x; c < d.D; c += 64 * EMB_BW) {
            float sum = 0.0f;                                 // wave order: fixed
#pragma unroll
            for (int w = 0; w < EMB_BW; ++w) sum += red[w][c];
            bd.pos_grad[(size_t)t * d.D + c] = sum;
        }
    }
}

// Small-table mode (context tables: 8 / 25 / max_bins+1 rows): thousands of rows scatter into a
// handful of table rows, so global float atomics would serialise on hot rows.  Each workgroup
// reduces its share of the rows into an LDS image of the table and writes it out as one slab.
#define EMB_SMALL_MAX 12288   // floats of LDS (48 KiB)
template <bool HAS2>
__global__ __launch_bounds__(256) void k_embed_bwd_small(cr_embed_bwd_desc bd) {
    const cr_embed_desc& d = bd.f;
    __shared__ __attribute__((aligned(16))) float tab[EMB_SMALL_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = d.V * d.D;
    for (int i = threadIdx.x; i < (n + 3) >> 2; i += 256) *reinterpret_cast<f4e*>(tab + 4 * i) = (f4e){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const DropCtx dc = drop_ctx(d.drop);
    const int rps = (d.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * rps, m1 = min(d.M, m0 + rps);
    if (d.D <= 64) {
        // LDS float atomics (ds_add_f32) retire only a few lanes per clock: the 6 400 of them a workgroup needed for
        // its 128 rows were 10 of this kernel's 14 us.  Instead every table row has ONE owner wave (id mod 4): a wave
        // scans the ids of 64 gradient rows at a time, requests up to 16 of the rows it owns together (lane = column),
        // and adds them into the LDS image with plain read-add-write -- no other wave touches those table rows, and
        // a wave's own LDS operations stay in order.  d_addend (every row, owned or not) is a separate streaming pass.
        if (bd.d_addend) {
            const int sub = lane >> 4, l = lane & 15;
            for (int mb = m0 + wave * 4; mb < m1; mb += 16) {
                const int m = mb + sub;
                if (m >= m1) continue;
                const float keep = (d.mask_ids && d.mask_ids[m] == 0) ? 0.0f : 1.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = l + 16 * i;
                    if (c < d.D) {
                        const float gv = dout_at<HAS2>(bd, (size_t)m * d.ld_out + d.col_off + c) * keep;
                        bd.d_addend[(size_t)m * d.ld_add + c] =
                            drop_apply(dc, (d.drop.row_offset + (uint32_t)m) * (uint32_t)d.D + (uint32_t)c, gv);
                    }
                }
            }
        }
        constexpr int U = 16;
        const bool col = lane < d.D;
        for (int h0 = m0; h0 < m1; h0 += 64) {
            const int m = h0 + lane;
            const bool act = m < m1;
            const int id = act ? d.ids[m] : 0;
            const int keep = (act && !(d.mask_ids && d.mask_ids[m] == 0)) ? 1 : 0;
            const bool mine = act && !(d.zero_pad && id == 0) && ((id & 3) == wave);
            unsigned long long todo = __ballot(mine ? 1 : 0);
            while (todo) {
                float g[U];
                int rr[U];
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (todo) {
                        const int r = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        rr[k] = r;
                        g[k] = col ? dout_at<HAS2>(bd, (size_t)(h0 + r) * d.ld_out + d.col_off + lane) : 0.0f;
                        cnt = k + 1;
                    } else {
                        rr[k] = 0;
                        g[k] = 0.0f;
                    }
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (k < cnt) {
                        const int idk = __builtin_amdgcn_readlane(id, rr[k]);
                        const int kk = __builtin_amdgcn_readlane(keep, rr[k]);
                        float gv = kk ? g[k] : 0.0f;
                        gv = drop_apply(dc, (d.drop.row_offset + (uint32_t)(h0 + rr[k])) * (uint32_t)d.D + (uint32_t)lane, gv);
                        if (col) tab[idk * d.D + lane] += gv * d.scale;
                    }
                }
            }
        }
    } else
    for (int m = m0 + wave; m < m1; m += 4) {
        const int id = d.ids[m];
        const float keep_row = (d.mask_ids && d.mask_ids[m] == 0) ? 0.0f : 1.0f;
        const bool skip_table = (d.zero_pad && id == 0);
        for (int c = lane; c < d.D; c += 64) {
            float g = dout_at<HAS2>(bd, (size_t)m * d.ld_out + d.col_off + c) * keep_row;
            g = drop_apply(dc, (d.drop.row_offset + (uint32_t)m) * (uint32_t)d.D + (uint32_t)c, g);
            if (bd.d_addend) bd.d_addend[(size_t)m * d.ld_add + c] = g;
            if (!skip_table) atomicAdd(&tab[id * d.D + c], g * d.scale);
        }
    }
    __syncthreads();
    // LDS image -> this workgroup's slab: 16-byte pieces, four of them in flight per thread (the one-dword loop was a
    // chain of 40 dependent LDS-read -> store pairs per thread)
    float* slab = bd.table_grad + (size_t)blockIdx.x * bd.slab_stride;
    const int n4 = n >> 2;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * 256) {
        f4e v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            if (i < n4) v[u] = *reinterpret_cast<const f4e*>(tab + 4 * i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            if (i < n4) *reinterpret_cast<f4e*>(slab + 4 * i) = v[u];
        }
    }
    for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) slab[i] = tab[i];
}

// Large table without a positional-table gradient (static sinusoid graphs, context-free inputs): nothing
// couples the rows, so the scatter-add runs row-parallel with 16 lanes per row (D <= 64) instead of one
// workgroup per position.
template <bool HAS2>
__global__ __launch_bounds__(256) void k_embed_bwd_rows16(cr_embed_bwd_desc bd) {
    const cr_embed_desc& d = bd.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 4, l = lane & 15;
    const DropCtx dc = drop_ctx(d.drop);
    for (int mb = (blockIdx.x * 4 + wave) * 4; mb < d.M; mb += gridDim.x * 16) {
        const int m = mb + sub;
        if (m >= d.M) continue;
        const int id = d.ids[m];
        const float keep_row = (d.mask_ids && d.mask_ids[m] == 0) ? 0.0f : 1.0f;
        const bool skip_table = (d.zero_pad && id == 0) || bd.table_grad == nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = l + 16 * i;
            if (c < d.D) {
                float g = dout_at<HAS2>(bd, (size_t)m * d.ld_out + d.col_off + c) * keep_row;
                g = drop_apply(dc, (d.drop.row_offset + (uint32_t)m) * (uint32_t)d.D + (uint32_t)c, g);
                if (bd.d_addend) bd.d_addend[(size_t)m * d.ld_add + c] = g;
                if (!skip_table && g != 0.0f) atomicAdd(bd.table_grad + (size_t)id * d.D + c, g * d.scale);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_embed_zero_slabs(float* slab0, int slab_stride, int n_slabs, long long n) {
    const long long total = n * n_slabs;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        slab0[(size_t)(i / n) * slab_stride + (size_t)(i % n)] = 0.0f;
}

extern "C" int cr_embed_bwd(const cr_embed_bwd_desc* bd, void* stream) {
    CR_REQUIRE(bd && bd->f.ids && bd->f.out, "cr_embed_bwd: NULL pointer");
    const cr_embed_desc* d = &bd->f;
    CR_REQUIRE(d->M > 0 && d->T > 0 && d->D > 0 && d->M % d->T == 0, "cr_embed_bwd: bad shape");
    if (bd->n_slabs > 0) {
        CR_REQUIRE(bd->table_grad != nullptr && bd->pos_grad == nullptr, "cr_embed_bwd: small-table mode needs table_grad and no pos_grad");
        if ((long long)d->V * d->D > EMB_SMALL_MAX) {
            // The table image does not fit one workgroup's LDS (e.g. time_emb at max_bins 200 and hidden_units >= 62):
            // same contract -- every slab written, their sum is the gradient -- through the large-table kernels:
            // all slabs zeroed, float atomics into slab 0.
            if (d->D > 64 * EMB_MAXC) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_embed_bwd: D=%d > %d", d->D, 64 * EMB_MAXC);
            const long long n = (long long)d->V * d->D;
            int zgrid = (int)((n * bd->n_slabs + 255) / 256);
            if (zgrid > 2048) zgrid = 2048;
            hipLaunchKernelGGL(k_embed_zero_slabs, dim3(zgrid), dim3(256), 0, cr_stream(stream), bd->table_grad, bd->slab_stride, bd->n_slabs, n);
            cr_embed_bwd_desc big = *bd;
            big.n_slabs = 0; big.slab_stride = 0;
            if (d->D <= 64) {
                int grid = cr_ceil_div(d->M, 16);
                if (grid > 2048) grid = 2048;
                if (bd->out2) hipLaunchKernelGGL((k_embed_bwd_rows16<true>), dim3(grid), dim3(256), 0, cr_stream(stream), big); else hipLaunchKernelGGL((k_embed_bwd_rows16<false>), dim3(grid), dim3(256), 0, cr_stream(stream), big);
            } else {
                if (bd->out2) hipLaunchKernelGGL((k_embed_bwd<true>), dim3(d->T), dim3(64 * EMB_BW), 0, cr_stream(stream), big); else hipLaunchKernelGGL((k_embed_bwd<false>), dim3(d->T), dim3(64 * EMB_BW), 0, cr_stream(stream), big);
            }
            return cr_check_launch("cr_embed_bwd(small table through the large-table kernels)");
        }
        if (bd->out2) hipLaunchKernelGGL((k_embed_bwd_small<true>), dim3(bd->n_slabs), dim3(256), 0, cr_stream(stream), *bd); else hipLaunchKernelGGL((k_embed_bwd_small<false>), dim3(bd->n_slabs), dim3(256), 0, cr_stream(stream), *bd);
        return cr_check_launch("cr_embed_bwd(small)");
    }
    if (d->D > 64 * EMB_MAXC) return cr_set_error(CR_ERR_UNSUPPORTED, "cr_embed_bwd: D=%d > %d", d->D, 64 * EMB_MAXC);
    if (bd->pos_grad == nullptr && d->D <= 64) {
        int grid = cr_ceil_div(d->M, 16);
        if (grid > 2048) grid = 2048;
        if (bd->out2) hipLaunchKernelGGL((k_embed_bwd_rows16<true>), dim3(grid), dim3(256), 0, cr_stream(stream), *bd); else hipLaunchKernelGGL((k_embed_bwd_rows16<false>), dim3(grid), dim3(256), 0, cr_stream(stream), *bd);
        return cr_check_launch("cr_embed_bwd(rows)");
    }
    if (bd->out2) hipLaunchKernelGGL((k_embed_bwd<true>), dim3(d->T), dim3(64 * EMB_BW), 0, cr_stream(stream), *bd); else hipLaunchKernelGGL((k_embed_bwd<false>), dim3(d->T), dim3(64 * EMB_BW), 0, cr_stream(stream), *bd);
    return cr_check_launch("cr_embed_bwd");
}
