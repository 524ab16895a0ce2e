// Hardware probe (round 4), third part: v_mfma_f32_16x16x32_bf16 D, A, B, C with C != D, and a VALU write to one register of C a few
// wait states later (write-after-read on SrcC).  hipcc leaves 3 wait states there (cr_stack_bwd1.hip, round-3 builds: an address
// add for the next LDS read lands in the accumulator an MFMA issued three states earlier still reads as C).  If the matrix pipe
// reads C for its last rows later than that, D comes out as A B + (the overwriting value) in the rows of the last pass: ONE
// register of lanes 48..63 -- the signature of the round-3 flake.
//   build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -o probe_mfma_war probe_mfma_war.hip ; run: ./probe_mfma_war [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NK 8                          // wait states 0 .. 7 between the MFMA and the overwrite of C[2]
struct Res { unsigned long long trials, bad, q[4]; };
#define CLOB "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211"
#define SETUP                                                                                                         \
    "v_mov_b32 v204, %2\n\tv_mov_b32 v205, %3\n\tv_mov_b32 v206, %2\n\tv_mov_b32 v207, %3\n\t"                          \
    "v_mov_b32 v208, %3\n\tv_mov_b32 v209, %2\n\tv_mov_b32 v210, %3\n\tv_mov_b32 v211, %2\n\t"                          \
    "v_mov_b32 v200, %4\n\tv_mov_b32 v201, %4\n\tv_mov_b32 v202, %4\n\tv_mov_b32 v203, %4\n\t"                          \
    "v_mov_b32 v196, 0\n\tv_mov_b32 v197, 0\n\tv_mov_b32 v198, 0\n\tv_mov_b32 v199, 0\n\ts_nop 15\n\ts_nop 15\n\t"
#define MFMA "v_mfma_f32_16x16x32_bf16 v[196:199], v[204:207], v[208:211], v[200:203]\n\t"
#define TAIL "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v198\n\t"

// PRE: 0 = the MFMA alone; 1 = two more MFMAs (other accumulators) in FRONT of it, back to back: the pipe is busy when it issues
template <int K, int PRE>
__device__ __forceinline__ float seq(unsigned a, unsigned b, float c, float junk) {
    float got;
    if (PRE == 0) {
        if (K == 0) asm volatile(SETUP MFMA "v_mov_b32 v202, %5\n\t" TAIL : "=&v"(got), "=&v"(junk) : "v"(a), "v"(b), "v"(c), "v"(junk) : CLOB);
        else asm volatile(SETUP MFMA "s_nop %6\n\tv_mov_b32 v202, %5\n\t" TAIL : "=&v"(got), "=&v"(junk) : "v"(a), "v"(b), "v"(c), "v"(junk), "n"(K > 0 ? K - 1 : 0) : CLOB);
    } else {
#define PREM "v_mfma_f32_16x16x32_bf16 v[192:195], v[204:207], v[208:211], v[192:195]\n\tv_mfma_f32_16x16x32_bf16 v[192:195], v[208:211], v[204:207], v[192:195]\n\t"
        if (K == 0) asm volatile(SETUP PREM MFMA "v_mov_b32 v202, %5\n\t" TAIL : "=&v"(got), "=&v"(junk) : "v"(a), "v"(b), "v"(c), "v"(junk) : CLOB);
        else asm volatile(SETUP PREM MFMA "s_nop %6\n\tv_mov_b32 v202, %5\n\t" TAIL : "=&v"(got), "=&v"(junk) : "v"(a), "v"(b), "v"(c), "v"(junk), "n"(K > 0 ? K - 1 : 0) : CLOB);
    }
    return got;
}
// reference: the same MFMA, C left alone
__device__ __forceinline__ float ref(unsigned a, unsigned b, float c) {
    float got, d;
    asm volatile(SETUP MFMA TAIL : "=&v"(got), "=&v"(d) : "v"(a), "v"(b), "v"(c) : CLOB);
    return got;
}
template <int K, int PRE>
__device__ __forceinline__ void one(unsigned a, unsigned b, float c, float r, unsigned long long* bad, unsigned long long (*q)[4]) {
    const float g = seq<K, PRE>(a, b, c, -777.0f);
    const unsigned long long m = __ballot(__float_as_uint(g) != __float_as_uint(r));
    if (m) { bad[PRE * NK + K] += 1; for (int j = 0; j < 4; ++j) q[PRE * NK + K][j] += ((m >> (16 * j)) & 0xFFFFull) != 0; }
}
__global__ __launch_bounds__(512) void k_probe(Res* res, int iters, int company, float* sink) {
    __shared__ __attribute__((aligned(16))) __bf16 img[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 512) img[i] = (__bf16)(float)(i & 7);
    __syncthreads();
    unsigned seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 4242u;
    if (company == 2 && wave >= 4) return;                                         // one wave per SIMD
    if (company == 1 && wave >= 4) {                                               // partners: back-to-back MFMAs on the shared matrix pipe + LDS reads
        f32x4 c = {0.f, 0.f, 0.f, 0.f}, c2 = c;
        for (int it = 0; it < iters * 30; ++it) {
            const bf8 a = *reinterpret_cast<const bf8*>(img + ((lane * 8 + it * 64) & 4088));
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0);
        }
        if (c[0] + c2[1] == 12345.678f) sink[0] = c[1];
        return;
    }
    unsigned long long bad[2 * NK] = {0}, q[2 * NK][4] = {{0}}, trials = 0;
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const unsigned a = 0x3f803f80u ^ (seed & 0x007f007fu), b = 0x3f803f80u ^ ((seed >> 9) & 0x007f007fu);
        const float c = __uint_as_float(0x3f800000u | (seed >> 9));
        const float r = ref(a, b, c);
        one<0, 0>(a, b, c, r, bad, q); one<1, 0>(a, b, c, r, bad, q); one<2, 0>(a, b, c, r, bad, q); one<3, 0>(a, b, c, r, bad, q);
        one<4, 0>(a, b, c, r, bad, q); one<5, 0>(a, b, c, r, bad, q); one<6, 0>(a, b, c, r, bad, q); one<7, 0>(a, b, c, r, bad, q);
        one<0, 1>(a, b, c, r, bad, q); one<1, 1>(a, b, c, r, bad, q); one<2, 1>(a, b, c, r, bad, q); one<3, 1>(a, b, c, r, bad, q);
        one<4, 1>(a, b, c, r, bad, q); one<5, 1>(a, b, c, r, bad, q); one<6, 1>(a, b, c, r, bad, q); one<7, 1>(a, b, c, r, bad, q);
        trials += 1;
    }
    if (lane == 0)
        for (int k = 0; k < 2 * NK; ++k) {
            atomicAdd(&res[company * 2 * NK + k].trials, trials);
            atomicAdd(&res[company * 2 * NK + k].bad, bad[k]);
            for (int j = 0; j < 4; ++j) atomicAdd(&res[company * 2 * NK + k].q[j], q[k][j]);
        }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 30000;
    Res* res; float* sink;
    (void)hipMalloc(&res, 3 * 2 * NK * sizeof(Res)); (void)hipMemset(res, 0, 3 * 2 * NK * sizeof(Res));
    (void)hipMalloc(&sink, 64);
    const char* comp[3] = {"two testers per SIMD", "tester + MFMA partner per SIMD", "one wave per SIMD"};
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL(k_probe, dim3(512), dim3(512), 0, 0, res, iters, c, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    }
    std::vector<Res> h(3 * 2 * NK);
    (void)hipMemcpy(h.data(), res, h.size() * sizeof(Res), hipMemcpyDeviceToHost);
    for (int c = 0; c < 3; ++c)
        for (int p = 0; p < 2; ++p)
            for (int k = 0; k < NK; ++k) {
                const Res& r = h[c * 2 * NK + p * NK + k];
                printf("%-30s | %s MFMA(C) ; %d wait states ; v_mov C[2] | wave-trials %llu wrong D[2] %llu | lane quarters 0-15 %llu 16-31 %llu 32-47 %llu 48-63 %llu\n",
                       comp[c], p ? "2 MFMAs ;" : "          ", k, r.trials, r.bad, r.q[0], r.q[1], r.q[2], r.q[3]);
            }
    return 0;
}
