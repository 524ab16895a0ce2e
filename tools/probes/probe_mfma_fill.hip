// Hardware probe (round 4), second part: hipcc's hazard recogniser counts EVERY instruction between an MFMA and the first VALU read of
// its result as one wait state (8 needed for v_mfma_f32_16x16x32_bf16 on gfx950: tools/probes/probe_mfma_raw.hip shows 7 stale, 8 exact).
// Which instruction kinds really last a wait state?  The 8 states are filled with N instructions of one kind + s_nop (7 - N):
// stale reads say that kind retires faster than a state.  Per lane quarter, to see which MFMA pass is the late one.
//   build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -o probe_mfma_fill probe_mfma_fill.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define NKIND 8
struct Res { unsigned long long trials, stale, q[4]; };
#define CLOB "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "s40", "s41", "vcc"
#define SETUP                                                                                                         \
    "v_mov_b32 v204, %2\n\tv_mov_b32 v205, %3\n\tv_mov_b32 v206, %2\n\tv_mov_b32 v207, %3\n\t"                          \
    "v_mov_b32 v208, %3\n\tv_mov_b32 v209, %2\n\tv_mov_b32 v210, %3\n\tv_mov_b32 v211, %2\n\t"                          \
    "v_mov_b32 v200, %4\n\tv_mov_b32 v201, %4\n\tv_mov_b32 v202, %4\n\tv_mov_b32 v203, %4\n\ts_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#define MFMA "v_mfma_f32_16x16x32_bf16 v[200:203], v[204:207], v[208:211], v[200:203]\n\t"
#define TAIL "v_add_f32 %0, 0, v202\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_add_f32 %1, 0, v202\n\t"
#define X7(s) s s s s s s s
#define X8(s) s s s s s s s s
#define X4(s) s s s s
__device__ __forceinline__ void run(int kind, unsigned a, unsigned b, float c, float& e, float& l) {
#define SEQ(body) asm volatile(SETUP MFMA body TAIL : "=&v"(e), "=&v"(l) : "v"(a), "v"(b), "v"(c) : CLOB)
    switch (kind) {
    case 0: SEQ("s_nop 7\n\t"); break;                                              // reference: exact
    case 1: SEQ(X8("s_nop 0\n\t")); break;                                          // eight one-state nops
    case 2: SEQ(X8("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t")); break;                   // eight satisfied waits
    case 3: SEQ(X8("s_mov_b32 s40, 0\n\t")); break;                                 // eight scalar moves
    case 4: SEQ(X8("v_mov_b32 v196, v197\n\t")); break;                             // eight vector moves (unrelated registers)
    case 5: SEQ(X4("s_waitcnt lgkmcnt(0)\n\t") "s_nop 3\n\t"); break;              // four satisfied waits + four states
    case 6: SEQ(X4("v_cmp_eq_u32 vcc, v196, v197\n\ts_and_b64 s[40:41], vcc, exec\n\t")); break;   // compare / scalar pairs
    case 7: SEQ(X7("s_nop 0\n\t")); break;                                          // seven states: must be stale (the probe sees the hazard)
    }
}
__global__ __launch_bounds__(512) void k_probe(Res* res, int iters, int company) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 99u;
    if (company == 1 && wave >= 4) return;                                         // one wave per SIMD: nothing interleaves with the tester
    unsigned long long stale[NKIND] = {0}, q[NKIND][4] = {{0}}, trials = 0;
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const unsigned a = 0x3f803f80u ^ (seed & 0x007f007fu), b = 0x3f803f80u ^ ((seed >> 9) & 0x007f007fu);
        const float c = __uint_as_float(0x3f800000u | (seed >> 9));
#pragma unroll
        for (int k = 0; k < NKIND; ++k) {
            float e, l;
            run(k, a, b, c, e, l);
            const unsigned long long m = __ballot(__float_as_uint(e) != __float_as_uint(l));
            if (m) { stale[k] += 1; for (int g = 0; g < 4; ++g) q[k][g] += ((m >> (16 * g)) & 0xFFFFull) != 0; }
        }
        trials += 1;
    }
    if (lane == 0)
        for (int k = 0; k < NKIND; ++k) {
            atomicAdd(&res[company * NKIND + k].trials, trials);
            atomicAdd(&res[company * NKIND + k].stale, stale[k]);
            for (int g = 0; g < 4; ++g) atomicAdd(&res[company * NKIND + k].q[g], q[k][g]);
        }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50000;
    Res* res;
    (void)hipMalloc(&res, 2 * NKIND * sizeof(Res)); (void)hipMemset(res, 0, 2 * NKIND * sizeof(Res));
    const char* kinds[NKIND] = {"s_nop 7", "8 x s_nop 0", "8 x s_waitcnt (satisfied)", "8 x s_mov_b32", "8 x v_mov_b32", "4 x s_waitcnt + s_nop 3", "4 x (v_cmp + s_and_b64)", "7 x s_nop 0 (one short)"};
    const char* comp[2] = {"two waves per SIMD", "one wave per SIMD"};
    for (int c = 0; c < 2; ++c) {
        hipLaunchKernelGGL(k_probe, dim3(512), dim3(512), 0, 0, res, iters, c);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    }
    std::vector<Res> h(2 * NKIND);
    (void)hipMemcpy(h.data(), res, h.size() * sizeof(Res), hipMemcpyDeviceToHost);
    for (int c = 0; c < 2; ++c)
        for (int k = 0; k < NKIND; ++k) {
            const Res& r = h[c * NKIND + k];
            printf("%-20s | MFMA ; %-28s ; read | wave-reads %llu stale %llu | by lane quarter 0-15 %llu 16-31 %llu 32-47 %llu 48-63 %llu\n", comp[c], kinds[k], r.trials, r.stale, r.q[0], r.q[1], r.q[2], r.q[3]);
        }
    return 0;
}
