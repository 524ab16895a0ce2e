// Hardware probe (round 4), fifth part: read-after-write between packed-fp32 instructions.  The round-3 flake, pinned by
// tools/diag_repro2.py + an in-kernel dump (profiles/r04_flake_evidence.md): in the LayerNorm backward's last statement
//     v_pk_fma_f32 v[202:203], v[164:165], v[236:237], v[202:203] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]     ; t -= xc * c2
//     v_pk_fma_f32 v[204:205], v[160:161], v[236:237], v[204:205] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]
//     v_pk_mul_f32 v[208:209], v[138:139], v[202:203] op_sel_hi:[0,1]                                            ; rstd * t
// the product came out as rstd * (t BEFORE the first instruction) in the LOW half of lanes 48..63: the third instruction read
// v202 before the first one's write.  Here: that triple and neighbours of it (other fillers, other distances), ~1e9 wave-times
// each, beside partner waves of several kinds, checked bit for bit.
//   build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -o probe_pk_raw probe_pk_raw.hip ; run: ./probe_pk_raw [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NT 8
#define NC 5
struct Res { unsigned long long trials, bad, stale, lo[4], hi[4]; };
__device__ __forceinline__ float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return __uint_as_float(0x3f800000u | (s >> 9)) - 1.5f; }
#define CLOB "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211"
// xc = v[200:201], c = v[202:203] (c1, c2), t = v[204:205], t' = v[206:207] (second chain), r = v[208:209] (rstd in v208), out = v[210:211]
#define LOAD "v_mov_b32 v200, %2\n\tv_mov_b32 v201, %3\n\tv_mov_b32 v202, %4\n\tv_mov_b32 v203, %5\n\tv_mov_b32 v204, %6\n\tv_mov_b32 v205, %7\n\t" \
             "v_mov_b32 v206, %6\n\tv_mov_b32 v207, %7\n\tv_mov_b32 v208, %8\n\tv_mov_b32 v209, %8\n\ts_nop 7\n\t"
#define OUT2 "s_nop 7\n\tv_mov_b32 %0, v210\n\tv_mov_b32 %1, v211\n\t"
#define ARGS : "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1), "v"(c1), "v"(c2), "v"(t0), "v"(t1), "v"(rs) : CLOB
#define FMA_A "v_pk_fma_f32 v[204:205], v[200:201], v[202:203], v[204:205] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
#define FMA_B "v_pk_fma_f32 v[206:207], v[200:201], v[202:203], v[206:207] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
#define MUL_A "v_pk_mul_f32 v[210:211], v[208:209], v[204:205] op_sel_hi:[0,1]\n\t"
__device__ __forceinline__ void run(int t, float x0, float x1, float c1, float c2, float t0, float t1, float rs, float& r0, float& r1) {
    switch (t) {
    case 0: asm volatile(LOAD FMA_A FMA_B MUL_A OUT2 ARGS); break;                                   // the listing's triple
    case 1: asm volatile(LOAD FMA_A MUL_A OUT2 ARGS); break;                                         // nothing in between
    case 2: asm volatile(LOAD FMA_A "s_nop 0\n\t" MUL_A OUT2 ARGS); break;                           // one nop
    case 3: asm volatile(LOAD FMA_A "v_sub_f32 v206, v206, v202\n\t" MUL_A OUT2 ARGS); break;        // one plain VALU
    case 4: asm volatile(LOAD FMA_A FMA_B FMA_B MUL_A OUT2 ARGS); break;                             // two packed fillers
    case 5: asm volatile(LOAD "v_sub_f32 v205, v205, v202\n\tv_sub_f32 v204, v204, v202\n\t" FMA_A FMA_B MUL_A OUT2 ARGS); break;   // with the two subtractions in front, as listed (t = t - c1 first)
    case 6: asm volatile(LOAD FMA_A FMA_B "v_mul_f32 v210, v208, v204\n\tv_mul_f32 v211, v208, v205\n\t" OUT2 ARGS); break;          // scalar consumers
    default: asm volatile(LOAD FMA_A "s_nop 1\n\t" MUL_A OUT2 ARGS); break;                          // two nops
    }
}
__global__ __launch_bounds__(512) void k_probe(Res* res, int iters, int company, float* sink) {
    __shared__ __attribute__((aligned(16))) __bf16 img[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 512) img[i] = (__bf16)(float)(i & 7);
    __syncthreads();
    unsigned seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 271828u;
    if (company > 0 && company < 4 && wave >= 4) {
        float x = frand(seed), y = frand(seed);
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters * 16; ++it) {
            if (company == 1) {
                asm volatile("v_fma_f32 %0, %1, %2, %0\n\tv_pk_mul_f32 v[220:221], v[222:223], v[224:225]\n\tv_fma_f32 %1, %0, %2, %1\n\tv_pk_fma_f32 v[220:221], v[222:223], v[224:225], v[220:221]"
                             : "+v"(x), "+v"(y) : "v"(0.5f) : "v220", "v221", "v222", "v223", "v224", "v225");
            } else if (company == 2) {
                const bf8 a = *reinterpret_cast<const bf8*>(img + ((lane * 8 + it * 64) & 4088));
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c, 0, 0, 0);
            } else {
                asm volatile("v_exp_f32 %0, %1\n\ts_nop 1\n\tv_add_f32_dpp %1, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
            }
        }
        if (x + y + c[0] == 12345.678f) sink[0] = x;
        return;
    }
    if (company == 4 && wave >= 4) return;                  // one wave per SIMD
    unsigned long long bad[NT] = {0}, stale[NT] = {0}, lo[NT][4] = {{0}}, hi[NT][4] = {{0}}, trials = 0;
    for (int it = 0; it < iters; ++it) {
        const float x0 = frand(seed), x1 = frand(seed), c1 = frand(seed), c2 = frand(seed), t0 = frand(seed), t1 = frand(seed), rs = frand(seed);
        const float f0 = __fmaf_rn(-x0, c2, t0), f1 = __fmaf_rn(-x1, c2, t1);
        const float g0 = __fmaf_rn(-x0, c2, __fsub_rn(t0, c1)), g1 = __fmaf_rn(-x1, c2, __fsub_rn(t1, c1));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float r0, r1;
            run(t, x0, x1, c1, c2, t0, t1, rs, r0, r1);
            const float e0 = __fmul_rn(rs, t == 5 ? g0 : f0), e1 = __fmul_rn(rs, t == 5 ? g1 : f1);
            const float s0 = __fmul_rn(rs, t == 5 ? __fsub_rn(t0, c1) : t0);                         // what a stale read of the low half gives
            const unsigned long long m0 = __ballot(__float_as_uint(r0) != __float_as_uint(e0)), m1 = __ballot(__float_as_uint(r1) != __float_as_uint(e1));
            if (m0 | m1) {
                bad[t] += 1;
                if (__ballot(__float_as_uint(r0) != __float_as_uint(e0) && __float_as_uint(r0) == __float_as_uint(s0))) stale[t] += 1;
                for (int g = 0; g < 4; ++g) { lo[t][g] += ((m0 >> (16 * g)) & 0xFFFFull) != 0; hi[t][g] += ((m1 >> (16 * g)) & 0xFFFFull) != 0; }
            }
        }
        trials += 1;
    }
    if (lane == 0)
        for (int t = 0; t < NT; ++t) {
            Res& r = res[company * NT + t];
            atomicAdd(&r.trials, trials); atomicAdd(&r.bad, bad[t]); atomicAdd(&r.stale, stale[t]);
            for (int g = 0; g < 4; ++g) { atomicAdd(&r.lo[g], lo[t][g]); atomicAdd(&r.hi[g], hi[t][g]); }
        }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100000;
    Res* res; float* sink;
    (void)hipMalloc(&res, NC * NT * sizeof(Res)); (void)hipMemset(res, 0, NC * NT * sizeof(Res));
    (void)hipMalloc(&sink, 64);
    const char* comp[NC] = {"eight testers per CU", "partner: VALU + packed", "partner: MFMA + LDS", "partner: trans + DPP + swap", "one wave per SIMD"};
    const char* names[NT] = {"pk_fma A ; pk_fma B ; pk_mul(A)", "pk_fma A ; pk_mul(A)", "pk_fma A ; s_nop 0 ; pk_mul(A)", "pk_fma A ; v_sub ; pk_mul(A)", "pk_fma A ; pk_fma B x2 ; pk_mul(A)",
                             "v_sub x2 ; pk_fma A ; pk_fma B ; pk_mul(A)", "pk_fma A ; pk_fma B ; v_mul x2 (A)", "pk_fma A ; s_nop 1 ; pk_mul(A)"};
    for (int c = 0; c < NC; ++c) {
        hipLaunchKernelGGL(k_probe, dim3(512), dim3(512), 0, 0, res, iters, c, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    }
    std::vector<Res> h(NC * NT);
    (void)hipMemcpy(h.data(), res, h.size() * sizeof(Res), hipMemcpyDeviceToHost);
    int rc = 0;
    for (int c = 0; c < NC; ++c)
        for (int t = 0; t < NT; ++t) {
            const Res& r = h[c * NT + t];
            printf("%-28s | %-44s | wave-trials %llu wrong %llu (stale low half %llu) | low half by lane quarter %llu %llu %llu %llu | high half %llu %llu %llu %llu\n", comp[c], names[t], r.trials, r.bad, r.stale,
                   r.lo[0], r.lo[1], r.lo[2], r.lo[3], r.hi[0], r.hi[1], r.hi[2], r.hi[3]);
            if (r.bad) rc = 1;
        }
    printf(rc ? "MISMATCHES SEEN\n" : "ALL FORMS EXACT\n");
    return rc;
}
