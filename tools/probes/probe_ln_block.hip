// Hardware probe (round 4), sixth part: the LayerNorm backward's closing block exactly as hipcc's SLP vectoriser scheduled it in the
// round-3 build that lost reproducibility (same registers, modifiers and interleaved scalar instructions: /tmp listing of
// tools/probes/wt_b, FLK=20, .LBB0_385), run ~2e8 wave-times beside several kinds of partner waves and checked element by
// element against scalar arithmetic.  In the kernel one LOW-half output of lanes 48..63 came out as rstd * (dg - c1): the
// "- xhat * c2" of its v_pk_fma_f32 missing (profiles/r04_flake_evidence.md).
//   build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -o probe_ln_block probe_ln_block.hip ; run: ./probe_ln_block [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NC 5
struct Res { unsigned long long trials, bad, noterm, lo[4], hi[4], which[12]; };
__device__ __forceinline__ float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return __uint_as_float(0x3f800000u | (s >> 9)) - 1.5f; }

// inputs: dg[12] -> v60,v61,v62,v63,v56,v57,v58,v59,v54,v55,v124,v125 ; xh[12] -> v136..v147 ; sums (after the swaps) v52,v53 and partners v150,v151 ; rstd v134
// outputs: v60,v61,v62,v63,v56,v57,v58,v59,v52,v53,v54,v55
struct In { float dg[12], xh[12], s0, s1, p0, p1, rstd; };
__device__ __forceinline__ void block(const In& i, float (&o)[12]) {
    asm volatile(
        "v_mov_b32 v60, %12\n\tv_mov_b32 v61, %13\n\tv_mov_b32 v62, %14\n\tv_mov_b32 v63, %15\n\tv_mov_b32 v56, %16\n\tv_mov_b32 v57, %17\n\t"
        "v_mov_b32 v58, %18\n\tv_mov_b32 v59, %19\n\tv_mov_b32 v54, %20\n\tv_mov_b32 v55, %21\n\tv_mov_b32 v124, %22\n\tv_mov_b32 v125, %23\n\t"
        "v_mov_b32 v136, %24\n\tv_mov_b32 v137, %25\n\tv_mov_b32 v138, %26\n\tv_mov_b32 v139, %27\n\tv_mov_b32 v140, %28\n\tv_mov_b32 v141, %29\n\t"
        "v_mov_b32 v142, %30\n\tv_mov_b32 v143, %31\n\tv_mov_b32 v144, %32\n\tv_mov_b32 v145, %33\n\tv_mov_b32 v146, %34\n\tv_mov_b32 v147, %35\n\t"
        "v_mov_b32 v52, %36\n\tv_mov_b32 v53, %37\n\tv_mov_b32 v150, %38\n\tv_mov_b32 v151, %39\n\tv_mov_b32 v134, %40\n\t"
        "v_mov_b32 v4, %36\n\tv_mov_b32 v123, %37\n\tv_mov_b32 v148, %38\n\tv_mov_b32 v149, %39\n\tv_mov_b32 v43, %40\n\t"
        "v_mov_b32 v92, %24\n\tv_mov_b32 v93, %25\n\tv_mov_b32 v94, %26\n\tv_mov_b32 v95, %27\n\tv_mov_b32 v80, %28\n\tv_mov_b32 v81, %29\n\ts_nop 7\n\t"
        // ---- the block, verbatim ----
        "v_pk_add_f32 v[52:53], v[52:53], v[150:151]\n\t"
        "s_mov_b32 s2, 0x3ca3d70a\n\t"
        "v_pk_mul_f32 v[150:151], v[52:53], s[2:3] op_sel_hi:[1,0]\n\t"
        "v_mov_b32_e32 v135, v134\n\t"
        "v_pk_add_f32 v[52:53], v[60:61], v[150:151] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_sub_f32_e32 v4, v4, v150\n\t"
        "v_pk_fma_f32 v[52:53], v[136:137], v[150:151], v[52:53] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_and_b32_e32 v43, 48, v43\n\t"
        "v_pk_mul_f32 v[60:61], v[134:135], v[52:53]\n\t"
        "v_pk_add_f32 v[52:53], v[62:63], v[150:151] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_fma_f32 v4, -v123, v151, v4\n\t"
        "v_pk_fma_f32 v[52:53], v[138:139], v[150:151], v[52:53] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mul_f32_e32 v4, v134, v4\n\t"
        "v_pk_mul_f32 v[62:63], v[134:135], v[52:53]\n\t"
        "v_pk_add_f32 v[52:53], v[56:57], v[150:151] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, v43\n\t"
        "v_pk_fma_f32 v[52:53], v[140:141], v[150:151], v[52:53] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[96:97], v[92:93], v[136:137], 0 op_sel_hi:[1,1,0]\n\t"
        "v_pk_mul_f32 v[56:57], v[134:135], v[52:53]\n\t"
        "v_pk_add_f32 v[52:53], v[58:59], v[150:151] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 v[98:99], v[94:95], v[138:139], 0 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 v[52:53], v[142:143], v[150:151], v[52:53] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[88:89], v[80:81], v[140:141], 0 op_sel_hi:[1,1,0]\n\t"
        "v_pk_mul_f32 v[58:59], v[134:135], v[52:53]\n\t"
        "v_pk_add_f32 v[52:53], v[54:55], v[150:151] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 v[54:55], v[124:125], v[150:151] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cndmask_b32_e32 v124, 0, v4, vcc\n\t"
        "v_sub_f32_e32 v4, v148, v150\n\t"
        "v_fma_f32 v4, -v149, v151, v4\n\t"
        "v_pk_fma_f32 v[52:53], v[144:145], v[150:151], v[52:53] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[54:55], v[146:147], v[150:151], v[54:55] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mul_f32_e32 v4, v134, v4\n\t"
        "v_pk_fma_f32 v[90:91], v[92:93], v[142:143], 0 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 v[84:85], v[94:95], v[144:145], 0 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 v[86:87], v[80:81], v[146:147], 0 op_sel_hi:[1,1,0]\n\t"
        "v_fma_f32 v130, v128, v123, 0\n\t"
        "v_fma_f32 v131, v129, v149, 0\n\t"
        "v_pk_mul_f32 v[52:53], v[134:135], v[52:53]\n\t"
        "v_pk_mul_f32 v[54:55], v[134:135], v[54:55]\n\t"
        // ---- end ----
        "s_nop 7\n\tv_mov_b32 %0, v60\n\tv_mov_b32 %1, v61\n\tv_mov_b32 %2, v62\n\tv_mov_b32 %3, v63\n\tv_mov_b32 %4, v56\n\tv_mov_b32 %5, v57\n\t"
        "v_mov_b32 %6, v58\n\tv_mov_b32 %7, v59\n\tv_mov_b32 %8, v52\n\tv_mov_b32 %9, v53\n\tv_mov_b32 %10, v54\n\tv_mov_b32 %11, v55\n\t"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
        : "v"(i.dg[0]), "v"(i.dg[1]), "v"(i.dg[2]), "v"(i.dg[3]), "v"(i.dg[4]), "v"(i.dg[5]), "v"(i.dg[6]), "v"(i.dg[7]), "v"(i.dg[8]), "v"(i.dg[9]), "v"(i.dg[10]), "v"(i.dg[11]),
          "v"(i.xh[0]), "v"(i.xh[1]), "v"(i.xh[2]), "v"(i.xh[3]), "v"(i.xh[4]), "v"(i.xh[5]), "v"(i.xh[6]), "v"(i.xh[7]), "v"(i.xh[8]), "v"(i.xh[9]), "v"(i.xh[10]), "v"(i.xh[11]),
          "v"(i.s0), "v"(i.s1), "v"(i.p0), "v"(i.p1), "v"(i.rstd)
        : "v4", "v43", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v80", "v81", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
          "v96", "v97", "v98", "v99", "v123", "v124", "v125", "v128", "v129", "v130", "v131", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151",
          "s2", "vcc");
}
__global__ __launch_bounds__(512) void k_probe(Res* res, int iters, int company, float* sink) {
    __shared__ __attribute__((aligned(16))) __bf16 img[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 512) img[i] = (__bf16)(float)(i & 7);
    __syncthreads();
    unsigned seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 161803u;
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
    if (company == 4 && wave >= 4) return;
    unsigned long long bad = 0, noterm = 0, lo[4] = {0}, hi[4] = {0}, which[12] = {0}, trials = 0;
    for (int it = 0; it < iters; ++it) {
        In in;
        for (int k = 0; k < 12; ++k) { in.dg[k] = frand(seed); in.xh[k] = frand(seed); }
        in.s0 = frand(seed); in.s1 = frand(seed); in.p0 = frand(seed); in.p1 = frand(seed); in.rstd = frand(seed);
        float o[12];
        block(in, o);
        if (company == 4 && blockIdx.x == 0 && threadIdx.x == 0 && it < 4) {
            float* d = sink + 16 + it * 16;
            d[0] = o[0]; d[1] = in.dg[0]; d[2] = in.xh[0]; d[3] = in.s0; d[4] = in.p0; d[5] = in.s1; d[6] = in.p1; d[7] = in.rstd; d[8] = o[1]; d[9] = in.dg[1]; d[10] = in.xh[1];
        }
        const float c1 = __fmul_rn(__fadd_rn(in.s0, in.p0), 0.02f), c2 = __fmul_rn(__fadd_rn(in.s1, in.p1), 0.02f);
        unsigned long long mlo = 0, mhi = 0, mnt = 0;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float t = __fsub_rn(in.dg[k], c1);
            const float e = __fmul_rn(in.rstd, __fmaf_rn(-in.xh[k], c2, t)), nt = __fmul_rn(in.rstd, t);
            const unsigned long long m = __ballot(__float_as_uint(o[k]) != __float_as_uint(e));
            if (m) { which[k] += 1; if (k & 1) mhi |= m; else mlo |= m; mnt |= __ballot(__float_as_uint(o[k]) == __float_as_uint(nt) && __float_as_uint(o[k]) != __float_as_uint(e)); }
        }
        if (mlo | mhi) {
            bad += 1; noterm += mnt != 0;
            for (int g = 0; g < 4; ++g) { lo[g] += ((mlo >> (16 * g)) & 0xFFFFull) != 0; hi[g] += ((mhi >> (16 * g)) & 0xFFFFull) != 0; }
        }
        trials += 1;
    }
    if (lane == 0) {
        Res& r = res[company];
        atomicAdd(&r.trials, trials); atomicAdd(&r.bad, bad); atomicAdd(&r.noterm, noterm);
        for (int g = 0; g < 4; ++g) { atomicAdd(&r.lo[g], lo[g]); atomicAdd(&r.hi[g], hi[g]); }
        for (int k = 0; k < 12; ++k) atomicAdd(&r.which[k], which[k]);
    }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100000;
    Res* res; float* sink;
    (void)hipMalloc(&res, NC * sizeof(Res)); (void)hipMemset(res, 0, NC * sizeof(Res));
    (void)hipMalloc(&sink, 4096); (void)hipMemset(sink, 0, 4096);
    const char* comp[NC] = {"eight testers per CU", "partner: VALU + packed", "partner: MFMA + LDS", "partner: trans + DPP + swap", "one wave per SIMD"};
    for (int c = 0; c < NC; ++c) {
        hipLaunchKernelGGL(k_probe, dim3(512), dim3(512), 0, 0, res, iters, c, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    }
    std::vector<Res> h(NC);
    (void)hipMemcpy(h.data(), res, h.size() * sizeof(Res), hipMemcpyDeviceToHost);
    int rc = 0;
    for (int c = 0; c < NC; ++c) {
        const Res& r = h[c];
        printf("%-28s | the block | wave-trials %llu wrong %llu (of them 'term missing' %llu) | low halves by lane quarter %llu %llu %llu %llu | high halves %llu %llu %llu %llu | per output", comp[c], r.trials, r.bad, r.noterm,
               r.lo[0], r.lo[1], r.lo[2], r.lo[3], r.hi[0], r.hi[1], r.hi[2], r.hi[3]);
        for (int k = 0; k < 12; ++k) printf(" %llu", r.which[k]);
        printf("\n");
        if (r.bad) rc = 1;
    }
    { float d[80]; (void)hipMemcpy(d, sink + 16, sizeof(d), hipMemcpyDeviceToHost);
      for (int k = 0; k < 4; ++k) { const float* q = d + 16 * k; double c1 = (double)(float)(q[3] + q[4]) * 0.02f, c2 = (double)(float)(q[5] + q[6]) * 0.02f;
        printf("dbg o0 %.9g dg0 %.9g xh0 %.9g s0 %.9g p0 %.9g s1 %.9g p1 %.9g rstd %.9g | o1 %.9g dg1 %.9g xh1 %.9g | c1 %.9g c2 %.9g exp0 %.9g exp1 %.9g\n", q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], c1, c2,
               q[7] * (q[1] - c1 - q[2] * c2), q[7] * (q[9] - c1 - q[10] * c2)); } }
    printf(rc ? "MISMATCHES SEEN\n" : "BLOCK EXACT\n");
    return rc;
}
