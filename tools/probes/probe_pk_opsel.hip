// Hardware probe (round 4), fourth part: the packed-fp32 forms with operand selects that hipcc's SLP vectoriser emits in the LayerNorm
// backward of cr_stack_bwd1.hip (v_pk_fma_f32 ... op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0] -- the LOW half reads the HIGH register
// of a pair -- and v_pk_mul_f32 ... op_sel_hi:[0,1] / [1,0]) executed ~1e9 times per form beside partner waves that keep the same
// SIMD's vector pipe busy (plain VALU, transcendentals, DPP, permlane swaps, MFMA).  Every result is compared bit for bit with the
// same arithmetic by scalar v_fma_f32 / v_mul_f32.  The round-3 flake's signature: low half, lanes 48..63.
//   build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -o probe_pk_opsel probe_pk_opsel.hip ; run: ./probe_pk_opsel [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NT 6
#define NC 5
struct Res { unsigned long long trials, bad, lo[4], hi[4]; };
__device__ __forceinline__ float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return __uint_as_float(0x3f800000u | (s >> 9)) - 1.5f; }
#define CLOB "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207"
// a = v[200:201], b = v[202:203], c = v[204:205]; d = v[206:207]
#define LOAD6 "v_mov_b32 v200, %2\n\tv_mov_b32 v201, %3\n\tv_mov_b32 v202, %4\n\tv_mov_b32 v203, %5\n\tv_mov_b32 v204, %6\n\tv_mov_b32 v205, %7\n\ts_nop 7\n\t"
#define OUT2 "s_nop 7\n\tv_mov_b32 %0, v206\n\tv_mov_b32 %1, v207\n\t"
#define ARGS : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1) : CLOB
__device__ __forceinline__ void run(int t, float a0, float a1, float b0, float b1, float c0, float c1, float& r0, float& r1) {
    switch (t) {
    case 0: asm volatile(LOAD6 "v_pk_fma_f32 v[206:207], v[200:201], v[202:203], v[204:205] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t" OUT2 ARGS); break;   // {c0 - a0 b1, c1 - a1 b1}
    case 1: asm volatile(LOAD6 "v_pk_fma_f32 v[206:207], v[200:201], v[202:203], v[204:205] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t" OUT2 ARGS); break; // {a0 b0 - c0, a1 b1 - c0}
    case 2: asm volatile(LOAD6 "v_pk_mul_f32 v[206:207], v[200:201], v[202:203] op_sel_hi:[0,1]\n\t" OUT2 ARGS); break;                                           // {a0 b0, a0 b1}
    case 3: asm volatile(LOAD6 "v_pk_mul_f32 v[206:207], v[200:201], v[202:203] op_sel_hi:[1,0]\n\t" OUT2 ARGS); break;                                           // {a0 b0, a1 b0}
    case 4: asm volatile(LOAD6 "v_pk_add_f32 v[206:207], v[200:201], v[202:203] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" OUT2 ARGS); break;                 // {a0 - b0, a1 - b0}
    // the listing's chain: producer pair by v_pk_mul (scalar factor), one state, two consumers reading it with op_sel
    default: asm volatile(LOAD6 "v_pk_mul_f32 v[202:203], v[202:203], v[204:205] op_sel_hi:[1,0]\n\ts_nop 0\n\t"
                                "v_pk_fma_f32 v[206:207], v[200:201], v[202:203], v[200:201] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t" OUT2 ARGS); break;   // b' = {b0 c0, b1 c0}; {a0 - a0 b1', a1 - a1 b1'}
    }
}
__global__ __launch_bounds__(512) void k_probe(Res* res, int iters, int company, float* sink) {
    __shared__ __attribute__((aligned(16))) __bf16 img[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 512) img[i] = (__bf16)(float)(i & 7);
    __syncthreads();
    unsigned seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 31337u;
    if (company > 0 && wave >= 4) {
        // partner waves of the testers' SIMDs
        float x = frand(seed), y = frand(seed), z = 0.f;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters * 12; ++it) {
            if (company == 1) {                       // plain + packed VALU, back to back
                asm volatile("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %0, %2, %1\n\tv_pk_mul_f32 v[210:211], v[212:213], v[214:215]\n\tv_fma_f32 %0, %1, %2, %0\n\tv_pk_add_f32 v[210:211], v[212:213], v[214:215]\n\tv_fma_f32 %1, %0, %2, %1"
                             : "+v"(x), "+v"(y) : "v"(0.5f) : "v210", "v211", "v212", "v213", "v214", "v215");
            } else if (company == 2) {                // transcendentals + DPP + permlane swaps
                asm volatile("v_exp_f32 %0, %1\n\tv_rcp_f32 %1, %0\n\ts_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_rsq_f32 %1, %0"
                             : "+v"(x), "+v"(y));
            } else if (company == 3) {                // MFMA + LDS reads
                const bf8 a = *reinterpret_cast<const bf8*>(img + ((lane * 8 + it * 64) & 4088));
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c, 0, 0, 0);
            } else {                                  // a mix with quarter-rate integer multiplies and selects (the dropout hash)
                asm volatile("v_mul_lo_u32 %0, %0, %2\n\tv_cmp_ge_u32 vcc, %0, %1\n\tv_cndmask_b32 %1, %1, %0, vcc\n\tv_xor_b32 %0, %0, %1" : "+v"(x), "+v"(y) : "v"(0x9E3779B1u) : "vcc");
            }
        }
        if (x + y + z + c[0] == 12345.678f) sink[0] = x;
        return;
    }
    unsigned long long bad[NT] = {0}, lo[NT][4] = {{0}}, hi[NT][4] = {{0}}, trials = 0;
    for (int it = 0; it < iters; ++it) {
        const float a0 = frand(seed), a1 = frand(seed), b0 = frand(seed), b1 = frand(seed), c0 = frand(seed), c1 = frand(seed);
        const float e0[NT] = {__fmaf_rn(-a0, b1, c0), __fmaf_rn(a0, b0, -c0), __fmul_rn(a0, b0), __fmul_rn(a0, b0), __fsub_rn(a0, b0), __fmaf_rn(-a0, __fmul_rn(b1, c0), a0)};
        const float e1[NT] = {__fmaf_rn(-a1, b1, c1), __fmaf_rn(a1, b1, -c0), __fmul_rn(a0, b1), __fmul_rn(a1, b0), __fsub_rn(a1, b0), __fmaf_rn(-a1, __fmul_rn(b1, c0), a1)};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float r0, r1;
            run(t, a0, a1, b0, b1, c0, c1, r0, r1);
            const unsigned long long m0 = __ballot(__float_as_uint(r0) != __float_as_uint(e0[t])), m1 = __ballot(__float_as_uint(r1) != __float_as_uint(e1[t]));
            if (m0 | m1) {
                bad[t] += 1;
                for (int g = 0; g < 4; ++g) { lo[t][g] += ((m0 >> (16 * g)) & 0xFFFFull) != 0; hi[t][g] += ((m1 >> (16 * g)) & 0xFFFFull) != 0; }
            }
        }
        trials += 1;
    }
    if (lane == 0)
        for (int t = 0; t < NT; ++t) {
            Res& r = res[company * NT + t];
            atomicAdd(&r.trials, trials); atomicAdd(&r.bad, bad[t]);
            for (int g = 0; g < 4; ++g) { atomicAdd(&r.lo[g], lo[t][g]); atomicAdd(&r.hi[g], hi[t][g]); }
        }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100000;
    Res* res; float* sink;
    (void)hipMalloc(&res, NC * NT * sizeof(Res)); (void)hipMemset(res, 0, NC * NT * sizeof(Res));
    (void)hipMalloc(&sink, 64);
    const char* comp[NC] = {"eight testers per CU", "partner: VALU + packed", "partner: trans + DPP + swaps", "partner: MFMA + LDS", "partner: mul_lo + cmp + cndmask"};
    const char* names[NT] = {"pk_fma op_sel:[0,1,0] neg a", "pk_fma op_sel_hi:[1,1,0] neg c", "pk_mul op_sel_hi:[0,1]", "pk_mul op_sel_hi:[1,0]", "pk_add op_sel_hi:[1,0] neg b", "pk_mul ; s_nop 0 ; pk_fma op_sel:[0,1,0]"};
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
            printf("%-32s | %-42s | wave-trials %llu wrong %llu | low half by lane quarter %llu %llu %llu %llu | high half %llu %llu %llu %llu\n", comp[c], names[t], r.trials, r.bad,
                   r.lo[0], r.lo[1], r.lo[2], r.lo[3], r.hi[0], r.hi[1], r.hi[2], r.hi[3]);
            if (r.bad) rc = 1;
        }
    printf(rc ? "MISMATCHES SEEN\n" : "ALL FORMS EXACT\n");
    return rc;
}
