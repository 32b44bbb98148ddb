// Where does the 128x128 GEMM tile lose its time?  Standalone ablation of cb_gemm's inner structure on MI355X:
//   * the tile's LDS images (A, B: [128 rows][128 B], 16-byte segments XOR-swizzled with row & 7, as gemm_impl.h) are filled once;
//   * every block (256 threads = 4 waves, 2x2, wave tile 64x64) then repeats "K tiles" of 2 x (4 + 4 ds_read_b128, 16 MFMA) on them;
//   * variants add back what the real kernel does per K tile: the barrier, the VGPR -> LDS stores (8 ds_write_b128 per thread), the
//     global loads (8 buffer-style 16-byte loads per thread from an L2-resident matrix).
// Reports TFLOP/s (2 blocks per CU resident, 512 blocks) -- the ceiling each ingredient leaves.
//   hipcc --offload-arch=gfx950 -O3 tools/inner_probe.hip -o tools/inner_probe && ./tools/inner_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ int lds_off(int row, int seg) { return row * 128 + ((seg ^ (row & 7)) << 4); }

// V bit 0: barrier per K tile; bit 1: ds_write of the next tile; bit 2: global loads of the next tile (row-contiguous 128-byte lines, as
// the product's loaders); bit 3: s_setprio around MFMAs; bit 4: the next tile arrives by LDS-DMA (global_load_lds, 16 B per lane)
template <int V>
__global__ void __launch_bounds__(256, 2) probe(const u32x4* gsrc, float* out, int ktiles, unsigned gmask) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * 128 * 128];      // double buffer x (A, B)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 2 * 128 * 8; i += 256) {
        u32x4 v = {0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        *reinterpret_cast<u32x4*>(smem + i * 16) = v;
    }
    __syncthreads();
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 st[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = u32x4{1u, 2u, 3u, 4u};
    // this thread's 8 segments of the next tile: rows (tid / 8) + 32 i of a [rows][K] bf16 matrix with K = 768 (96 16-byte words per row)
    unsigned goff = ((blockIdx.x % 20) * 128 + (tid >> 3)) * 96u + (tid & 7);
    for (int t = 0; t < ktiles; ++t) {
        const unsigned char* As = smem + (t & 1) * (2 * 128 * 128);
        const unsigned char* Bs = As + 128 * 128;
        unsigned char* Nx = smem + ((t + 1) & 1) * (2 * 128 * 128);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + lds_off(wm * 64 + i * 16 + (lane & 15), kk * 4 + (lane >> 4)));
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off(wn * 64 + j * 16 + (lane & 15), kk * 4 + (lane >> 4)));
            if (V & 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            if (V & 8) __builtin_amdgcn_s_setprio(0);
            if (kk == 0 && (V & 2)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int idx = tid + i * 256;
                    *reinterpret_cast<u32x4*>(Nx + (idx >> 10) * (128 * 128) + lds_off((idx >> 3) & 127, idx & 7)) = st[i];
                }
            }
            if (kk == 0 && (V & 4)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) st[i] = gsrc[(goff + (unsigned)i * 32u * 96u) & gmask];
                goff += 8u;                                   // next K tile: 128 bytes further along the rows
            }
            if (kk == 0 && (V & 16)) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_global_load_lds(gsrc + ((goff + (unsigned)i * 32u * 96u) & gmask),
                                                     (__attribute__((address_space(3))) void*)(Nx + (i * 4 + wave) * 1024), 16, 0, 0);
                goff += 8u;
            }
        }
        if (V & 1) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)st[i][0];
    if (s == 123.456f) out[0] = s;
}

template <int V> void run(const char* name, const u32x4* g, float* out, unsigned gmask) {
    const int blocks = 512, ktiles = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, g, out, 200, gmask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, g, out, ktiles, gmask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 128 * 128 * 64 * (double)ktiles * blocks;
    printf("%-72s %8.3f ms  %7.1f TFLOP/s  (%4.1f %% of 2500)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0);
}

int main() {
    const size_t n = 1u << 20;                       // 16 MiB of 16-byte words: L2 / Infinity-Cache resident
    u32x4* g; float* out;
    hipMalloc(&g, n * 16); hipMalloc(&out, 4);
    hipMemset(g, 1, n * 16);
    const unsigned gmask = n - 1;
    run<0>("ds_read_b128 + MFMA only", g, out, gmask);
    run<8>("ds_read + MFMA, s_setprio(1) around the MFMAs", g, out, gmask);
    run<1>("+ one barrier per K tile", g, out, gmask);
    run<3>("+ barrier + 8 ds_write_b128 per thread per K tile", g, out, gmask);
    run<5>("+ barrier + 8 global 16-byte loads per thread per K tile (row-contiguous, cached)", g, out, gmask);
    run<7>("+ barrier + ds_write + global loads (the real kernel's per-tile work)", g, out, gmask);
    run<15>("the same with s_setprio(1) around the MFMAs", g, out, gmask);
    run<17>("+ barrier + LDS-DMA of the next tile (no VGPR staging, no ds_write)", g, out, gmask);
    return 0;
}
