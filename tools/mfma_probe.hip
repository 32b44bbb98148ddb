// Standalone probe of the MFMA fragment layouts libclipbert_hip assumes (run on the GPU box when a GEMM
// parity test fails): prints PASS/FAIL for v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k_bf16(const __bf16* A, const __bf16* B, float* D) {   // A[16][32], B[16 n][32 k]
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B[(l & 15) * 32 + 8 * (l >> 4) + j]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];   // D[m][n]
}
__global__ void k_f32(const float* A, const float* B, float* D) {      // A[16][4], B[16 n][4 k]
    int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l & 15) * 4 + (l >> 4)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
int main() {
    std::vector<__bf16> A(512), B(512);
    std::vector<float> Af(64), Bf(64), D(256), ref(256);
    for (int i = 0; i < 512; ++i) { A[i] = (__bf16)(float)((i * 7) % 13 - 6); B[i] = (__bf16)(float)((i * 5) % 11 - 5 + i / 32); }
    for (int i = 0; i < 64; ++i) { Af[i] = (float)((i * 7) % 13 - 6); Bf[i] = (float)((i * 5) % 11 - 5 + i / 4); }
    __bf16 *dA, *dB; float *dD, *dAf, *dBf;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024); hipMalloc(&dAf, 256); hipMalloc(&dBf, 256);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dAf, Af.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dBf, Bf.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double e1 = 0, e1t = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        float r = 0; for (int k = 0; k < 32; ++k) r += (float)A[m * 32 + k] * (float)B[n * 32 + k];
        e1 += fabs(r - D[m * 16 + n]); e1t += fabs(r - D[n * 16 + m]);
    }
    printf("mfma_f32_16x16x32_bf16: %s (err %g; transposed-hypothesis err %g)\n", e1 == 0 ? "PASS" : "FAIL", e1, e1t);
    hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dAf, dBf, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double e2 = 0, e2t = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        float r = 0; for (int k = 0; k < 4; ++k) r += Af[m * 4 + k] * Bf[n * 4 + k];
        e2 += fabs(r - D[m * 16 + n]); e2t += fabs(r - D[n * 16 + m]);
    }
    printf("mfma_f32_16x16x4f32: %s (err %g; transposed-hypothesis err %g)\n", e2 == 0 ? "PASS" : "FAIL", e2, e2t);
    return 0;
}
