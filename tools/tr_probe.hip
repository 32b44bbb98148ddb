// Probe of ds_read_b64_tr_b16 semantics on gfx950: prints, for the address pattern used by the GEMM KROW path,
// which (k, r) element of a row-major [k][16] tile every lane/element receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(unsigned short* out, int row_stride_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short T[64 * 64];
    int l = threadIdx.x;
    for (int i = l; i < 64 * 64; i += 64) T[i] = 0xffff;
    __syncthreads();
    for (int kk = 0; kk < 32; ++kk)
        for (int r = l; r < 16; r += 64) T[kk * row_stride_elems + r] = (unsigned short)(kk * 16 + r);
    __syncthreads();
    int g = l >> 4, p = l & 15;
    for (int half = 0; half < 2; ++half) {
        int kbase = 8 * g + 4 * half;
        const unsigned short* addr = &T[(kbase + p / 4) * row_stride_elems + 4 * (p % 4)];
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
        for (int e = 0; e < 4; ++e) out[(l * 2 + half) * 4 + e] = (unsigned short)v[e];
    }
}
int main() {
    for (int rs : {16, 64, 72}) {
        unsigned short* d; hipMalloc(&d, 64 * 8 * 2);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rs);
        std::vector<unsigned short> h(64 * 8);
        hipMemcpy(h.data(), d, 64 * 8 * 2, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
            int want = (8 * (l >> 4) + j) * 16 + (l & 15);      // T[k = 8g + j][r = l & 15]
            if (h[l * 8 + j] != want) ++bad;
        }
        printf("row_stride %d elems: hypothesis H1 %s (%d mismatches)\n", rs, bad ? "FAIL" : "PASS", bad);
        if (bad) for (int l = 0; l < 20; ++l) {
            printf(" lane %2d:", l);
            for (int j = 0; j < 8; ++j) printf(" (k%d,r%d)", h[l * 8 + j] / 16, h[l * 8 + j] % 16);
            printf("\n");
        }
        hipFree(d);
    }
    return 0;
}
