// Shared device/host helpers for libcastrec (gfx950 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "castrec.h"

int cr_set_error(int code, const char* fmt, ...);
int cr_check_launch(const char* what);

#define CR_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return cr_set_error(CR_ERR_INVALID, __VA_ARGS__); \
    } while (0)

static inline hipStream_t cr_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------------------------------
// f32-in / f32-accumulate MFMA, v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).
// Lane l (li = l & 15, lg = l >> 4):  A[i = li][k = lg],  B[k = lg][j = li],
// accumulator register r holds D[row = 4*lg + r][col = li].
// ---------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#define CR_WAVE 64

// Every 64-byte line of the kernel-argument segment requested at once, at the top of a kernel whose argument block is large.
// The compiler fetches arguments where it first needs them, each fetch behind an s_waitcnt: the one-launch block backward
// (1.2 KB of descriptors) opened with twelve dependent scalar-cache misses -- 3.7 us before its first vector load went out
// (tools/b1_ts.py).  With the lines requested together the misses overlap and the compiler's own loads hit the scalar cache.
// BYTES: the explicit arguments (the hidden ones behind them are requested where the code uses them).
template <int BYTES>
__device__ __forceinline__ void cr_kernarg_touch() {
    static_assert(BYTES > 0 && BYTES <= 20 * 64, "at most 20 lines");
    constexpr int LAST = BYTES - 4;
    const auto p = __builtin_amdgcn_kernarg_segment_ptr();
    unsigned t;
#define CR_KT_O(i) "n"((i) * 64 < LAST ? (i) * 64 : LAST)
    asm volatile(
        "s_load_dword %0, %1, %2\n\ts_load_dword %0, %1, %3\n\ts_load_dword %0, %1, %4\n\ts_load_dword %0, %1, %5\n\t"
        "s_load_dword %0, %1, %6\n\ts_load_dword %0, %1, %7\n\ts_load_dword %0, %1, %8\n\ts_load_dword %0, %1, %9\n\t"
        "s_load_dword %0, %1, %10\n\ts_load_dword %0, %1, %11\n\ts_load_dword %0, %1, %12\n\ts_load_dword %0, %1, %13\n\t"
        "s_load_dword %0, %1, %14\n\ts_load_dword %0, %1, %15\n\ts_load_dword %0, %1, %16\n\ts_load_dword %0, %1, %17\n\t"
        "s_load_dword %0, %1, %18\n\ts_load_dword %0, %1, %19\n\ts_load_dword %0, %1, %20\n\ts_load_dword %0, %1, %21\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(t)
        : "s"(p), CR_KT_O(0), CR_KT_O(1), CR_KT_O(2), CR_KT_O(3), CR_KT_O(4), CR_KT_O(5), CR_KT_O(6), CR_KT_O(7), CR_KT_O(8), CR_KT_O(9),
          CR_KT_O(10), CR_KT_O(11), CR_KT_O(12), CR_KT_O(13), CR_KT_O(14), CR_KT_O(15), CR_KT_O(16), CR_KT_O(17), CR_KT_O(18), CR_KT_O(19)
        : "memory");
#undef CR_KT_O
    (void)t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------
// counter-based dropout generator (restated in numpy by tests/dropout_ref.py)
// ---------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t cr_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ uint32_t cr_site_key(uint32_t seed, uint32_t step, uint32_t site) {
    return cr_fmix32(seed ^ cr_fmix32(step * 0x9E3779B9u + site * 0x85EBCA77u + 0x165667B1u));
}
// Element mixer: x = idx * PHI + key is a Weyl sequence in idx (already equidistributed in its high bits); one
// xor-shift + one multiply decorrelates neighbours, and only the high bits matter for the threshold compare.
// (v_mul_lo_u32 is quarter rate: the 3-multiply fmix32 form cost more issue time than the MFMAs of the
// attention kernels.)  x is linear in idx, so kernels advance it with adds instead of recomputing idx * PHI.
#define CR_PHI 0x9E3779B1u
__host__ __device__ __forceinline__ uint32_t cr_mix(uint32_t x) { return (x ^ (x >> 16)) * 0xD168AAADu; }
__host__ __device__ __forceinline__ bool cr_keep(uint32_t key, uint32_t idx, uint32_t thresh) {
    return cr_mix(idx * CR_PHI + key) >= thresh;
}

// device-side view of cr_rng, resolved once per kernel
struct DropCtx {
    uint32_t key, thresh;
    float scale;
    bool on;
};
// stepv: the value of *r.step.  A kernel whose prologue is on its critical path requests it at its very start (cr_step_request)
// and builds the contexts where they are first needed: `*r.step` at the point of use is a vector load and a full vmcnt(0) wait there
// (0.6-0.7 us at the head of the block backward, tools/b1_ts.py; again in front of each side's attention loop).
__device__ __forceinline__ DropCtx drop_ctx(const cr_rng& r, uint32_t stepv) {
    DropCtx c;
    c.on = r.rate > 0.0f;
    c.key = 0; c.thresh = 0; c.scale = 1.0f;
    if (c.on) {
        c.key = cr_site_key(r.seed, stepv, r.site);
        double t = (double)r.rate * 4294967296.0;
        c.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
        c.scale = 1.0f / (1.0f - r.rate);
    }
    return c;
}
__device__ __forceinline__ uint32_t cr_step_request(const cr_rng& r) { return r.rate > 0.0f ? *r.step : 0u; }
__device__ __forceinline__ DropCtx drop_ctx(const cr_rng& r) { return drop_ctx(r, cr_step_request(r)); }
__device__ __forceinline__ float drop_apply(const DropCtx& c, uint32_t idx, float v) {
    return c.on ? (cr_keep(c.key, idx, c.thresh) ? v * c.scale : 0.0f) : v;
}
// branch-free factor from a pre-multiplied counter x = idx * CR_PHI + key: keep ? scale : 0
__device__ __forceinline__ float drop_factor_x(const DropCtx& c, uint32_t x) { return (cr_mix(x) >= c.thresh) ? c.scale : 0.0f; }

// Sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15) with four DPP adds (quad xor 1, quad xor 2,
// row_half_mirror, row_mirror) instead of four ds_bpermute round trips (__shfl_xor): every lane ends up
// with the row total.  The association differs from a shuffle butterfly only in order.
template <int CTRL>
__device__ __forceinline__ float cr_dpp_add(float v) {
    const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
    return v + __int_as_float(x);
}
__device__ __forceinline__ float cr_row16_sum(float v) {
    v = cr_dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
    v = cr_dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
    v = cr_dpp_add<0x141>(v);    // row_half_mirror
    v = cr_dpp_add<0x140>(v);    // row_mirror
    return v;
}

// rstd * (dg - c1 - xhat * c2): the last statement of a LayerNorm backward, as THREE SCALAR instructions the compiler cannot
// re-form.  Written in C++ it is, per element pair, what hipcc's SLP vectoriser turns into the in-place chain
// v_pk_add_f32 t ; v_pk_fma_f32 t, xhat, c2, t ; v_pk_mul_f32 out, rstd, t  on ONE register pair -- the form whose middle
// instruction's contribution went missing in 111 recorded replays of the one-launch block backward (profiles/r04_flake/README.md:
// the trigger inside the kernel was never isolated, the necessary condition -- that chain -- was).  Inline assembly is opaque to
// every IR pass, so no flag (CASTREC_EXTRA_FLAGS, a future default) can bring the packed chain back for these elements; the
// build also scans the ISA of the register-layout kernels for the pattern (castrec_amd/build.py: check_isa, tests/test_isa.py).
// Same arithmetic as the contracted C++ form: sub, fma with a negated product, mul -- each rounded once.
__device__ __forceinline__ float cr_ln_bwd_tail(float dg, float c1, float xhat, float c2, float rstd) {
    float t;
    asm("v_sub_f32 %0, %1, %2\n\t"
        "v_fma_f32 %0, -%3, %4, %0\n\t"
        "v_mul_f32 %0, %5, %0"
        : "=&v"(t)
        : "v"(dg), "v"(c1), "v"(xhat), "v"(c2), "v"(rstd));
    return t;
}

// After a workgroup's partial sums are in state[0..2]: take a ticket; the LAST workgroup of the grid copies the now
// complete sums and the step counter to state[8..11] (castrec.h, state block) and re-arms the ticket.
// Ordering without __threadfence(): an agent-scope release fence writes the XCD's whole dirty L2 back (this kernel has
// just produced megabytes of gradients: measured +44 us per step).  Only the three float atomics have to be ordered
// before the ticket, they are device-scope atomics (performed at the memory side, not in the L2), and they are issued by
// the same wave as the ticket: waiting for their acknowledgement (vmcnt) is enough.  The last workgroup reads the
// totals back with atomics as well.
// `flag`: an LDS word of the caller's
__device__ __forceinline__ void head_snapshot_at(float* state, unsigned total_workgroups, int* flag) {
    if (threadIdx.x < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            unsigned* ticket = reinterpret_cast<unsigned*>(state + 12);
            *flag = (atomicAdd(ticket, 1u) == total_workgroups - 1u) ? 1 : 0;
        }
    }
    __syncthreads();
    if (!*flag) return;
    if (threadIdx.x < 3) state[8 + threadIdx.x] = atomicAdd(state + threadIdx.x, 0.0f);
    if (threadIdx.x == 3) *reinterpret_cast<unsigned*>(state + 11) = *reinterpret_cast<const unsigned*>(state + 4);
    if (threadIdx.x == 0) *reinterpret_cast<unsigned*>(state + 12) = 0u;
}
// (a kernel that declares all 160 KB of LDS dynamically has no room for this static word: head_snapshot_at)
__device__ __forceinline__ void head_snapshot(float* state, unsigned total_workgroups) {
    __shared__ int last;
    head_snapshot_at(state, total_workgroups, &last);
}

static inline int cr_ceil_div(int a, int b) { return (a + b - 1) / b; }

// Kernels with more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised, and that
// attribute belongs to the CURRENT device: each launcher keeps one mask per kernel, one bit per device ordinal, so a
// process that drives several GPUs gets the attribute on every one of them (a plain "done" flag would skip the
// second device and its launch would fail).  Ordinals >= 64 are simply not cached.
typedef unsigned long long cr_devmask;
static inline int cr_raise_lds_limit(const void* fn, cr_devmask* done) {
    int dev = -1;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    const cr_devmask bit = known ? (1ull << dev) : 0ull;
    if (bit && (__atomic_load_n(done, __ATOMIC_RELAXED) & bit)) return CR_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (bit) __atomic_fetch_or(done, bit, __ATOMIC_RELAXED);
    return CR_OK;
}
