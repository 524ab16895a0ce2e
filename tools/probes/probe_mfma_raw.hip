// Hardware probe (round 4): is the number of wait states hipcc leaves between a v_mfma_f32_16x16x32_bf16 and the first VALU read of
// its result enough on gfx950 under every kind of company?  Evidence that led here (tools/diag_repro2.py on the two round-3
// revisions that were not reproducible, profiles/r04_flake_*.json): the deviating element is always ONE accumulator register of
// lanes 48..63 (the rows an MFMA writes in its last pass) of the wave that finishes its attention loop last, and in the ISA of those
// builds that register is the first one a VALU instruction reads after the dq_in product's MFMA burst, at exactly the distance
// hipcc's hazard recogniser leaves (8 wait states).
// Each test: C = known, D = A B + C by MFMA(s), then `DIST` wait states (s_nop), an EARLY read of D[2] by v_add_f32, a long wait, a
// LATE read of the same register.  early != late  <=>  the early read saw the accumulator before the MFMA's write.
//   build: hipcc --offload-arch=gfx950 -O2 -o probe_mfma_raw probe_mfma_raw.hip ; run: ./probe_mfma_raw [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NDIST 9                      // wait states 1 .. 9 between the (last) MFMA and the early read
#define NSHAPE 3                     // 0: one MFMA; 1: the burst of the listing (read of the FIRST of three MFMAs' result, two more behind it); 2: read of the LAST of three back-to-back MFMAs
struct Res { unsigned long long trials, stale, lanes; };

#define CLOB "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", \
             "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215"
// operands: A = v[204:207], B = v[208:211] (bf16 pairs, any finite values), C / D = v[200:203]; second and third accumulators v[192:195], v[196:199]
#define SETUP                                                                                                         \
    "v_mov_b32 v204, %2\n\tv_mov_b32 v205, %3\n\tv_mov_b32 v206, %2\n\tv_mov_b32 v207, %3\n\t"                          \
    "v_mov_b32 v208, %3\n\tv_mov_b32 v209, %2\n\tv_mov_b32 v210, %3\n\tv_mov_b32 v211, %2\n\t"                          \
    "v_mov_b32 v200, %4\n\tv_mov_b32 v201, %4\n\tv_mov_b32 v202, %4\n\tv_mov_b32 v203, %4\n\t"                          \
    "v_mov_b32 v192, %4\n\tv_mov_b32 v193, %4\n\tv_mov_b32 v194, %4\n\tv_mov_b32 v195, %4\n\t"                          \
    "v_mov_b32 v196, %4\n\tv_mov_b32 v197, %4\n\tv_mov_b32 v198, %4\n\tv_mov_b32 v199, %4\n\ts_nop 15\n\ts_nop 15\n\t"
#define TAIL "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_add_f32 %1, 0, v202\n\t"
#define MF(d) "v_mfma_f32_16x16x32_bf16 v[" d "], v[204:207], v[208:211], v[" d "]\n\t"

template <int SHAPE, int DIST>
__device__ __forceinline__ void seq(unsigned a, unsigned b, float c, float& early, float& late) {
    static_assert(DIST >= 1 && DIST <= 16, "s_nop range");
    if (SHAPE == 0) {
        asm volatile(SETUP MF("200:203") "s_nop %5\n\tv_add_f32 %0, 0, v202\n\t" TAIL
                     : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "v"(c), "n"(DIST - 1) : CLOB);
    } else if (SHAPE == 1) {
        // as in the listing: MFMA (read later), two more MFMAs on other accumulators, s_nop, read: DIST counts the two MFMAs as one state each
        asm volatile(SETUP MF("200:203") MF("192:195") MF("196:199") "s_nop %5\n\tv_add_f32 %0, 0, v202\n\t" TAIL
                     : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "v"(c), "n"(DIST >= 3 ? DIST - 3 : 0) : CLOB);
    } else {
        asm volatile(SETUP MF("192:195") MF("196:199") MF("200:203") "s_nop %5\n\tv_add_f32 %0, 0, v202\n\t" TAIL
                     : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "v"(c), "n"(DIST - 1) : CLOB);
    }
}

// company 2: every sequence runs with two loads of its own in flight (returning into other registers)
template <int SHAPE, int DIST>
__device__ __forceinline__ void one(unsigned a, unsigned b, float c, unsigned long long* stale, unsigned long long* lanes, const float4* junk = nullptr, float* acc = nullptr) {
    float e, l;
    float4 q0 = {0, 0, 0, 0}, q1 = q0;
    if (junk) {
        const unsigned h = (a * 2654435761u) ^ (unsigned)(SHAPE * 16 + DIST) * 40503u;
        q0 = junk[((size_t)h * 64 + (threadIdx.x & 63)) & 0x3FFFFF];
        q1 = junk[((size_t)(h >> 7) * 64 + (threadIdx.x & 63)) & 0x3FFFFF];
    }
    seq<SHAPE, DIST>(a, b, c, e, l);
    if (junk) *acc += q0.x + q1.y;
    const unsigned long long m = __ballot(__float_as_uint(e) != __float_as_uint(l));
    if (m) { stale[SHAPE * NDIST + DIST - 1] += 1; lanes[SHAPE * NDIST + DIST - 1] |= m; }
}
template <int SHAPE>
__device__ __forceinline__ void all_dist(unsigned a, unsigned b, float c, unsigned long long* stale, unsigned long long* lanes, const float4* j = nullptr, float* acc = nullptr) {
    if constexpr (SHAPE != 1) { one<SHAPE, 1>(a, b, c, stale, lanes, j, acc); one<SHAPE, 2>(a, b, c, stale, lanes, j, acc); }
    one<SHAPE, 3>(a, b, c, stale, lanes, j, acc); one<SHAPE, 4>(a, b, c, stale, lanes, j, acc); one<SHAPE, 5>(a, b, c, stale, lanes, j, acc);
    one<SHAPE, 6>(a, b, c, stale, lanes, j, acc); one<SHAPE, 7>(a, b, c, stale, lanes, j, acc); one<SHAPE, 8>(a, b, c, stale, lanes, j, acc);
    one<SHAPE, 9>(a, b, c, stale, lanes, j, acc);
}

__global__ __launch_bounds__(512) void k_probe(Res* res, int iters, int company, const float4* junk, float* sink) {
    __shared__ __attribute__((aligned(16))) __bf16 img[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 512) img[i] = (__bf16)(float)(i & 7);
    __syncthreads();
    unsigned seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 777u;
    if ((company == 1 || company == 3) && wave >= 4) {
        // the partner waves of the testers' SIMDs: company 1: back-to-back MFMAs (the matrix pipe is shared per SIMD); 3: LDS reads + VALU
        f32x4 c = {0.f, 0.f, 0.f, 0.f}, c2 = c;
        for (int it = 0; it < iters * 40; ++it) {
            const bf8 a = *reinterpret_cast<const bf8*>(img + ((lane * 8 + it * 64) & 4088));
            if (company == 1) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0);
            } else {
                c[0] += (float)a[0] * 0.5f; c[1] += (float)a[3];
            }
        }
        if (c[0] + c2[1] == 12345.678f) sink[0] = c[1];
        return;
    }
    unsigned long long stale[NSHAPE * NDIST] = {0}, lanes[NSHAPE * NDIST] = {0}, trials = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const unsigned a = 0x3f803f80u ^ (seed & 0x007f007fu), b = 0x3f803f80u ^ ((seed >> 9) & 0x007f007fu);   // bf16 pairs in [1, 2)
        const float c = __uint_as_float(0x3f800000u | (seed >> 9));
        if (company == 2) {
            // loads whose data returns into OTHER registers while the sequences run: spread over 64 MiB (cache hits and misses mixed)
            all_dist<0>(a, b, c, stale, lanes, junk, &acc);
            all_dist<1>(a, b, c, stale, lanes, junk, &acc);
            all_dist<2>(a, b, c, stale, lanes, junk, &acc);
        } else {
            all_dist<0>(a, b, c, stale, lanes);
            all_dist<1>(a, b, c, stale, lanes);
            all_dist<2>(a, b, c, stale, lanes);
        }
        trials += 1;
    }
    if (acc == 12345.678f) sink[1] = acc;
    if (lane == 0)
        for (int k = 0; k < NSHAPE * NDIST; ++k) {
            atomicAdd(&res[company * NSHAPE * NDIST + k].trials, trials);
            atomicAdd(&res[company * NSHAPE * NDIST + k].stale, stale[k]);
            atomicOr(&res[company * NSHAPE * NDIST + k].lanes, lanes[k]);
        }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int nwg = 512, NC = 4;
    Res* res; float4* junk; float* sink;
    (void)hipMalloc(&res, NC * NSHAPE * NDIST * sizeof(Res)); (void)hipMemset(res, 0, NC * NSHAPE * NDIST * sizeof(Res));
    (void)hipMalloc(&junk, (size_t)(1 << 22) * 16 + 64); (void)hipMemset(junk, 0, (size_t)(1 << 22) * 16 + 64);
    (void)hipMalloc(&sink, 64);
    const char* comp[NC] = {"eight testers per CU", "testers + MFMA partners", "testers, loads in flight", "testers + LDS/VALU partners"};
    const char* shp[NSHAPE] = {"one MFMA", "first of three MFMAs", "last of three MFMAs"};
    for (int c = 0; c < NC; ++c) {
        hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(512), 0, 0, res, iters, c, junk, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    }
    std::vector<Res> h(NC * NSHAPE * NDIST);
    (void)hipMemcpy(h.data(), res, h.size() * sizeof(Res), hipMemcpyDeviceToHost);
    for (int c = 0; c < NC; ++c)
        for (int s = 0; s < NSHAPE; ++s) {
            printf("%-28s | %-22s | wait states -> stale wave-reads of %llu (lanes):", comp[c], shp[s], h[(c * NSHAPE + s) * NDIST + 4].trials);
            for (int d = 0; d < NDIST; ++d) {
                const Res& r = h[(c * NSHAPE + s) * NDIST + d];
                if (s == 1 && d < 2) continue;
                printf("  %d: %llu", d + 1, r.stale);
                if (r.stale) printf(" (%016llx)", r.lanes);
            }
            printf("\n");
        }
    return 0;
}
