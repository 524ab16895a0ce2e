// Hardware probe for the two primitives cr_attn_bf.hip relies on (run on the GPU box; prints PASS/FAIL):
//  (1) v_mfma_f32_16x16x32_bf16 operand / result lane maps, with asymmetric integer data;
//  (2) ds_read_b64_tr_b16 through tr_frag's addressing on the swizzled [rows][64] image.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf4 lds_bf4;
__device__ __forceinline__ int img_off(int row, int ch) { return row * 64 + ((ch ^ (row & 6)) << 3); }
__device__ __forceinline__ bf8 row_frag(const __bf16* img, int row0, int ks) {
    const int lane = threadIdx.x & 63;
    return *reinterpret_cast<const bf8*>(img + img_off(row0 + (lane & 15), (lane >> 4) + 4 * ks));
}
__device__ __forceinline__ bf8 tr_frag(const __bf16* img, int ra, int rb, int jt) {
    const int lane = threadIdx.x & 63, lg = lane >> 4, idx = lane & 15, q = idx >> 2, p = idx & 3;
    const int ch = 2 * jt + (p >> 1), sub = 4 * (p & 1);
    const bf4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + img_off(ra + 4 * lg + q, ch) + sub));
    const bf4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)(img + img_off(rb + 4 * lg + q, ch) + sub));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}
// A [16][32], B [32][16] row-major floats (small integers) -> D [16][16]
__global__ void k_mfma(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, li = l & 15, lg = l >> 4;
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[li * 32 + 8 * lg + j]; b[j] = (__bf16)B[(8 * lg + j) * 16 + li]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * lg + r) * 16 + li] = c[r];
}
// image of X [32][64]; out[l][j] = tr_frag(img, 0, 16, jt)[j] and row_frag(img, 16, ks)[j]
__global__ void k_tr(const float* X, float* out_tr, float* out_row) {
    __shared__ __attribute__((aligned(16))) __bf16 img[32 * 64];
    const int l = threadIdx.x;
    for (int i = l; i < 32 * 8; i += 64) {
        const int r = i >> 3, ch = i & 7;
        bf8 v;
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)X[r * 64 + 8 * ch + j];
        *reinterpret_cast<bf8*>(img + img_off(r, ch)) = v;
    }
    __syncthreads();
    for (int jt = 0; jt < 4; ++jt) {
        const bf8 t = tr_frag(img, 0, 16, jt);
        for (int j = 0; j < 8; ++j) out_tr[(jt * 64 + l) * 8 + j] = (float)t[j];
    }
    for (int ks = 0; ks < 2; ++ks) {
        const bf8 t = row_frag(img, 16, ks);
        for (int j = 0; j < 8; ++j) out_row[(ks * 64 + l) * 8 + j] = (float)t[j];
    }
}
int main() {
    std::vector<float> A(16 * 32), B(32 * 16), D(256), X(32 * 64), otr(4 * 64 * 8), orow(2 * 64 * 8);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)((k * 2 + j * 7) % 5 - 2);
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 64; ++c) X[r * 64 + c] = (float)(r * 64 + c);   // exact in bf16? only < 256: use small
    for (auto& x : X) x = (float)(((int)x * 37) % 251);                                                // 0..250: exact in bf16
    float *dA, *dB, *dD, *dX, *dT, *dR;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMalloc(&dX, X.size() * 4); hipMalloc(&dT, otr.size() * 4); hipMalloc(&dR, orow.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, dX, dT, dR);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(otr.data(), dT, otr.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(orow.data(), dR, orow.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j];
        if (s != D[i * 16 + j]) { if (bad < 5) printf("mfma mismatch (%d,%d): %g vs %g\n", i, j, D[i * 16 + j], s); ++bad; }
    }
    printf("mfma_f32_16x16x32_bf16 lane maps: %s\n", bad ? "FAIL" : "PASS");
    int bad2 = 0;
    for (int jt = 0; jt < 4; ++jt) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const int lg = l >> 4, li = l & 15;
        const int row = (j < 4 ? 0 : 16) + 4 * lg + (j & 3), col = 16 * jt + li;
        const float want = X[row * 64 + col], got = otr[(jt * 64 + l) * 8 + j];
        if (want != got) { if (bad2 < 8) printf("tr mismatch jt %d lane %d j %d: %g vs %g\n", jt, l, j, got, want); ++bad2; }
    }
    printf("tr_frag (ds_read_b64_tr_b16 on the swizzled image): %s\n", bad2 ? "FAIL" : "PASS");
    int bad3 = 0;
    for (int ks = 0; ks < 2; ++ks) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const float want = X[(16 + (l & 15)) * 64 + 32 * ks + 8 * (l >> 4) + j], got = orow[(ks * 64 + l) * 8 + j];
        if (want != got) { if (bad3 < 8) printf("row mismatch\n"); ++bad3; }
    }
    printf("row_frag: %s\n", bad3 ? "FAIL" : "PASS");
    return (bad || bad2 || bad3) ? 1 : 0;
}
