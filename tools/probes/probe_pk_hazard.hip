// Hardware probe (round 4): does any of the instruction sequences hipcc's SLP vectoriser put into cr_stack_bwd1.hip misbehave on
// gfx950?  Round 3 blamed "a compiler-made v_pk_mul_f32 whose source pair is overwritten by the next instruction: wrong LOW half
// in lanes 48..63, ~1 in 350 steps".  Each test below is that sequence (or a neighbour of it from the same ISA listing), written
// as ONE asm statement on fixed registers so that nothing is scheduled in between, executed ~1e9 wave-times in one launch under
// three kinds of company on the SIMD (alone, a partner wave issuing MFMA + LDS reads, loads returning into other registers), and
// checked element by element against the same arithmetic done by plain scalar instructions far away from any hazard window.
//   build: hipcc --offload-arch=gfx950 -O2 -o probe_pk_hazard probe_pk_hazard.hip ; run: ./probe_pk_hazard [iters]
// Output: one line per (test, company): trials, mismatches, mismatching lane mask.  0 mismatches everywhere = the sequence is safe.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NTEST 8
struct Res { unsigned long long trials, bad, lanes; };

__device__ __forceinline__ float frand(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return __uint_as_float(0x3f800000u | (s >> 9)) - 1.5f;          // [-0.5, 0.5), full mantissa
}

// ---- the sequences; registers v200..v215 are the stage ---------------------------------------------------
#define CLOB "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211"
#define LOAD4 "v_mov_b32 v200, %2\n\tv_mov_b32 v201, %3\n\tv_mov_b32 v202, %4\n\tv_mov_b32 v203, %5\n\ts_nop 7\n\t"
#define OUT2(lo, hi) "s_nop 7\n\tv_mov_b32 %0, " lo "\n\tv_mov_b32 %1, " hi "\n\t"

// T0: v_pk_mul_f32 d, a(op_sel_hi 0: a.lo both halves), b ; next instruction overwrites b.lo with a full-rate v_mov
__device__ __forceinline__ void t0(float a0, float a1, float b0, float b1, float j, float& r0, float& r1) {
    asm volatile(LOAD4 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel_hi:[0,1]\n\tv_mov_b32 v202, %6\n\t" OUT2("v204", "v205")
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(j) : CLOB);
}
// T1: ... next instruction is v_pk_fma_f32 writing the whole source pair (the pair found in the SLP listing)
__device__ __forceinline__ void t1(float a0, float a1, float b0, float b1, float j, float& r0, float& r1) {
    asm volatile(LOAD4 "v_mov_b32 v206, %6\n\tv_mov_b32 v207, %6\n\ts_nop 7\n\t"
                 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel_hi:[0,1]\n\tv_pk_fma_f32 v[202:203], v[206:207], v[206:207], v[206:207]\n\t" OUT2("v204", "v205")
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(j) : CLOB);
}
// T2: v_pk_add_f32 d, a, -b ; next: v_and_b32 overwrites b.hi (36 of these in the production build: split8)
__device__ __forceinline__ void t2(float a0, float a1, float b0, float b1, float j, float& r0, float& r1) {
    asm volatile(LOAD4 "v_pk_add_f32 v[204:205], v[200:201], v[202:203] neg_lo:[0,1] neg_hi:[0,1]\n\tv_and_b32 v203, 0xffff0000, %6\n\t" OUT2("v204", "v205")
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(j) : CLOB);
}
// T3: two row reductions finished by back-to-back v_permlane32_swap with the 2 wait states hipcc leaves (v_mov; s_nop 0; swap; swap),
//     consumed at once by v_pk_add_f32 + v_pk_mul_f32 with a scalar pair: the c1 / c2 tail of the LayerNorm backward
__device__ __forceinline__ void t3(float a0, float a1, float b0, float b1, float j, float& r0, float& r1) {
    // v200 = a0, v201 = a1 (per-lane values); result lanes: sum over the lane and its partner lane ^ 32, times b0 (made uniform by the caller)
    asm volatile(LOAD4 "v_mov_b32 v204, v200\n\tv_mov_b32 v205, v201\n\ts_nop 0\n\t"
                 "v_permlane32_swap_b32 v200, v204\n\tv_permlane32_swap_b32 v201, v205\n\t"
                 "v_pk_add_f32 v[200:201], v[200:201], v[204:205]\n\tv_pk_mul_f32 v[200:201], v[200:201], v[202:203] op_sel_hi:[1,0]\n\t" OUT2("v200", "v201")
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(j) : CLOB);
}
// T4: the same through v_permlane16_swap first (the whole grp_sum pair as the SLP build emits it)
__device__ __forceinline__ void t4(float a0, float a1, float b0, float b1, float j, float& r0, float& r1) {
    asm volatile(LOAD4 "v_mov_b32 v204, v200\n\ts_nop 0\n\tv_fmac_f32 v206, v202, v203\n\t"
                 "v_permlane16_swap_b32 v200, v204\n\tv_add_f32 v204, v200, v204\n\tv_mov_b32 v200, v201\n\ts_nop 1\n\t"
                 "v_permlane16_swap_b32 v201, v200\n\tv_add_f32 v205, v201, v200\n\tv_mov_b32 v200, v204\n\tv_mov_b32 v201, v205\n\ts_nop 0\n\t"
                 "v_permlane32_swap_b32 v204, v200\n\tv_permlane32_swap_b32 v205, v201\n\t"
                 "v_pk_add_f32 v[204:205], v[204:205], v[200:201]\n\tv_pk_mul_f32 v[204:205], v[204:205], v[202:203] op_sel_hi:[1,0]\n\t" OUT2("v204", "v205")
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(j) : CLOB);
}
// T5: packed producer -> DPP consumer at hipcc's 2 wait states (a half-rate producer in front of a consumer without interlock)
__device__ __forceinline__ void t5(float a0, float a1, float b0, float b1, float j, float& r0, float& r1) {
    asm volatile(LOAD4 "v_mov_b32 v204, %6\n\tv_mov_b32 v205, %6\n\ts_nop 7\n\tv_pk_mul_f32 v[204:205], v[200:201], v[202:203]\n\ts_nop 1\n\t"
                 "v_add_f32_dpp v206, v204, v204 row_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v207, v205, v205 row_mirror row_mask:0xf bank_mask:0xf\n\t" OUT2("v206", "v207")
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(j) : CLOB);
}
// T6: packed producer -> v_permlane32_swap at 2 wait states
__device__ __forceinline__ void t6(float a0, float a1, float b0, float b1, float j, float& r0, float& r1) {
    asm volatile(LOAD4 "v_mov_b32 v204, %6\n\tv_mov_b32 v205, %6\n\ts_nop 7\n\tv_pk_mul_f32 v[204:205], v[200:201], v[202:203]\n\ts_nop 1\n\t"
                 "v_permlane32_swap_b32 v204, v205\n\t" OUT2("v204", "v205")
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(j) : CLOB);
}
// T7: v_pk_mul_f32 ; next instruction a 16-byte global store whose data registers are the packed result (store data read over
//     several cycles behind a half-rate producer: the dx store of the LayerNorm backward's tail)
__device__ __forceinline__ void t7(float a0, float a1, float b0, float b1, float* p, float& r0, float& r1) {
    asm volatile(LOAD4 "v_mov_b32 v206, 0\n\tv_mov_b32 v207, 0\n\ts_nop 7\n\t"
                 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203]\n\tglobal_store_dwordx4 %6, v[204:207], off\n\t"
                 "s_waitcnt vmcnt(0)\n\tglobal_load_dwordx2 v[208:209], %6, off sc1\n\ts_waitcnt vmcnt(0)\n\t" OUT2("v208", "v209")
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(p) : CLOB, "memory");
}

__global__ __launch_bounds__(512) void k_probe(Res* res, int iters, int company, float* scratch, const float* junk, float* dbg) {
    __shared__ __attribute__((aligned(16))) __bf16 img[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 512) img[i] = (__bf16)(float)(i & 7);
    __syncthreads();
    unsigned seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    if (company == 1 && wave >= 4) {
        // the partner waves of SIMDs 0..3: MFMA + LDS reads + some vector arithmetic, for as long as the testers run (same trip count)
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters * NTEST; ++it) {
            const bf8 a = *reinterpret_cast<const bf8*>(img + ((lane * 8 + it * 64) & 4088));
            const bf8 b = *reinterpret_cast<const bf8*>(img + ((lane * 8 + it * 128 + 512) & 4088));
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
            c[0] = c[0] * 0.5f + 1.0f;
        }
        if (c[0] == 12345.678f) scratch[0] = c[1];
        return;
    }
    float* myp = scratch + 16 + (size_t)(blockIdx.x * 512 + threadIdx.x) * 4;
    unsigned long long bad[NTEST] = {0}, lanes[NTEST] = {0}, trials = 0;
    for (int it = 0; it < iters; ++it) {
        const float a0 = frand(seed), a1 = frand(seed), b0 = frand(seed), b1 = frand(seed), j = frand(seed);
        const float bu = __shfl(b0, 0, 64);                             // wave-uniform factor for T3 / T4
        float ld0 = 0.f, ld1 = 0.f, ld2 = 0.f, ld3 = 0.f;
        if (company == 2) {                                             // loads whose data returns while the sequences run
            const float4 q = *reinterpret_cast<const float4*>(junk + (((size_t)it * 2048 + blockIdx.x * 64 + lane) & 0xFFFFF) * 4);
            ld0 = q.x; ld1 = q.y; ld2 = q.z; ld3 = q.w;
        }
        float r0, r1;
#define CHECK(k, e0, e1)                                                                                    \
        {                                                                                                   \
            const bool w = __float_as_uint(r0) != __float_as_uint(e0) || __float_as_uint(r1) != __float_as_uint(e1); \
            const unsigned long long m = __ballot(w);                                                       \
            if (m) { bad[k] += 1; lanes[k] |= m; }                                                          \
        }
        t0(a0, a1, b0, b1, j, r0, r1); CHECK(0, a0 * b0, a0 * b1)
        t1(a0, a1, b0, b1, j, r0, r1); CHECK(1, a0 * b0, a0 * b1)
        t2(a0, a1, b0, b1, j, r0, r1); CHECK(2, a0 - b0, a1 - b1)
        {
            const float s0 = a0 + __shfl_xor(a0, 32, 64), s1 = a1 + __shfl_xor(a1, 32, 64);
            // (the swap leaves lane l with (own, partner) in an order that depends on the half: the sum is commutative, bits equal)
            t3(a0, a1, bu, b1, j, r0, r1); CHECK(3, s0 * bu, s1 * bu)
            const float q0 = a0 + __shfl_xor(a0, 16, 64), q1 = a1 + __shfl_xor(a1, 16, 64);
            const float u0 = q0 + __shfl_xor(q0, 32, 64), u1 = q1 + __shfl_xor(q1, 32, 64);
            t4(a0, a1, bu, b1, j, r0, r1); CHECK(4, u0 * bu, u1 * bu)
        }
        {
            const float p0 = __fmul_rn(a0, b0), p1 = __fmul_rn(a1, b1);             // (not contracted into the sums below)
            const int mir = (lane & 48) | (15 - (lane & 15));
            const float e0 = __fadd_rn(p0, __shfl(p0, mir, 64)), e1 = __fadd_rn(p1, __shfl(p1, mir, 64));
            t5(a0, a1, b0, b1, j, r0, r1); CHECK(5, e0, e1)
            // v_permlane32_swap v204, v205: lanes 32..63 of v204 <-> lanes 0..31 of v205
            const float p0o = __shfl(p0, (lane + 32) & 63, 64), p1o = __shfl(p1, (lane + 32) & 63, 64);    // (all lanes active: a shuffle inside the select reads 0 from lanes the branch masked)
            const float x0 = lane < 32 ? p0 : p1o, x1 = lane < 32 ? p0o : p1;
            t6(a0, a1, b0, b1, j, r0, r1); CHECK(6, x0, x1)
            if (company == 0 && blockIdx.x == 0 && wave == 0 && it == 0) {       // one wave's registers after the swap, for the record
                float* d = dbg + lane * 6;
                d[0] = r0; d[1] = r1; d[2] = x0; d[3] = x1; d[4] = p0; d[5] = p1;
            }
            t7(a0, a1, b0, b1, myp, r0, r1); CHECK(7, p0, p1)
        }
        trials += 1;
        if (ld0 + ld1 + ld2 + ld3 == 12345.678f) scratch[1] = ld0;
    }
    if (lane == 0)
        for (int k = 0; k < NTEST; ++k) {
            atomicAdd(&res[company * NTEST + k].trials, trials);
            atomicAdd(&res[company * NTEST + k].bad, bad[k]);
            atomicOr(&res[company * NTEST + k].lanes, lanes[k]);
        }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    const int nwg = 512;
    Res* res; float* scratch; float* junk; float* dbg;
    hipMalloc(&dbg, 64 * 6 * 4); hipMemset(dbg, 0, 64 * 6 * 4);
    hipMalloc(&res, 3 * NTEST * sizeof(Res)); hipMemset(res, 0, 3 * NTEST * sizeof(Res));
    hipMalloc(&scratch, (16 + (size_t)nwg * 512 * 4) * 4); hipMemset(scratch, 0, (16 + (size_t)nwg * 512 * 4) * 4);
    hipMalloc(&junk, (size_t)(1 << 20) * 16 + 64); hipMemset(junk, 0, (size_t)(1 << 20) * 16 + 64);
    const char* names[NTEST] = {"pk_mul ; v_mov src.lo (WAR)", "pk_mul ; pk_fma src pair (WAR)", "pk_add ; v_and src.hi (WAR)",
                                "mov ; s_nop 0 ; swap32 ; swap32 ; pk_add ; pk_mul", "swap16 x2 ; swap32 x2 ; pk_add ; pk_mul",
                                "pk_mul ; s_nop 1 ; add_dpp row_mirror", "pk_mul ; s_nop 1 ; swap32", "pk_mul ; store x4 of the result"};
    const char* comp[3] = {"eight testers per CU", "testers + MFMA/LDS partners", "testers + loads in flight"};
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(512), 0, 0, res, iters, c, scratch, junk, dbg);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    }
    std::vector<Res> h(3 * NTEST);
    hipMemcpy(h.data(), res, 3 * NTEST * sizeof(Res), hipMemcpyDeviceToHost);
    int rc = 0;
    for (int c = 0; c < 3; ++c)
        for (int k = 0; k < NTEST; ++k) {
            const Res& r = h[c * NTEST + k];
            printf("%-28s | T%d %-52s | wave-trials %llu mismatching %llu lanes %016llx\n", comp[c], k, names[k], r.trials, r.bad, r.lanes);
            if (r.bad) rc = 1;
        }
    if (h[6].bad) {
        std::vector<float> d(64 * 6);
        hipMemcpy(d.data(), dbg, 64 * 6 * 4, hipMemcpyDeviceToHost);
        printf("T6, first wave, first trial: lane | got v204 v205 | model v204 v205 | product lo hi\n");
        for (int l = 0; l < 64; l += 7) printf("  %2d | %+.6f %+.6f | %+.6f %+.6f | %+.6f %+.6f\n", l, d[l * 6], d[l * 6 + 1], d[l * 6 + 2], d[l * 6 + 3], d[l * 6 + 4], d[l * 6 + 5]);
    }
    printf(rc ? "HAZARD SEEN\n" : "ALL SEQUENCES EXACT\n");
    return rc;
}
