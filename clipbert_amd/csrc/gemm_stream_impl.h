// cb_gemm, STREAMING structure (round 4): the HBM-bound products of the path -- the ResNet's 1x1 convolutions with a short reduction
// (res2 / res3 of detectron2's R-50, SURVEY.md a3 / Appendix B: K = 64 / 128 against 10^5 pixel rows; 2*K flop per output element is
// far below the machine balance of ~312 flop/B).
//
// What bounds them with one workgroup per output tile (profiles/r03f_gemm_tuning_cold.json: 200704 x 256 x 64 + residual lands at
// 85-110 us = 2.7 TB/s under EVERY tile shape, 4-wave or 8-wave, and stays there at 8x the rows): a workgroup lives for three
// dependent memory round trips (operands -> residual -> stores) with a single K tile to compute in between, so the bytes a CU keeps in
// flight (~6 KB per resident workgroup) times the ~2.5 us latency is what the kernel streams -- Little's law, not the DRAM.
//
// This structure keeps the bytes in flight instead:
//   * PERSISTENT workgroups (256 threads, <= 2 per CU) walk the M tiles grid-stride; the weights (BN x K, 32 KiB) are loaded into LDS
//     ONCE per workgroup and stay there;
//   * the A tile of the NEXT M tile arrives by LDS-DMA (buffer_load ... lds, two stages) while the current one is multiplied and stored;
//   * the epilogue's M x N operand (the residual) of the current tile is requested into registers BEFORE
//     the wait for its A tile, all 8 chunks per thread at once: one round trip per tile instead of one per staging pass;
//   * no __syncthreads() inside the loop: with LDS-DMA transfers in flight its workgroup-scope fence compiles to s_waitcnt vmcnt(0)
//     (seen in the ISA of the first version) -- every staging pass would wait for the next tile's DMA; the staging passes synchronise with s_waitcnt lgkmcnt(0) + a raw s_barrier (CB_LDS_BARRIER);
//   * one counted wait per tile: s_waitcnt vmcnt(#DMA of the next A tile) -- everything older (this tile's A, its epilogue operands,
//     the previous tile's stores) has retired, only the newest loads may still fly.  Loads retire in order among loads, so the count
//     is exact whatever the stores do (a store that is still outstanding only makes the wait longer, never wrong).
// Each workgroup computes a BM x BN tile with BN = the whole row of C (res2: N = 256) or a quarter of it (res3: N = 512): a tile's
// stores and residual reads are runs of BN * 2 bytes per row over BM consecutive rows.
//
// Waves 2 x 2; accumulators pass through a 16-row LDS staging buffer so that every thread owns 8 consecutive columns of a row
// (16-byte accesses; the same epilogue8 as every other cb_gemm kernel: results are bit-identical to the 4-wave kernels, whose K loop
// adds the same K tiles in the same order).  Forward form only (A, B both k-contiguous).
#pragma once
#include "gemm_impl.h"

namespace cbgemm {

// EPI (compile-time: the prefetch registers exist only where used): 0 = the epilogue reads no M x N operand, 1 = it reads the residual
template <int BM, int BN, int KT, int OCC, int EPI>
__global__ void __launch_bounds__(256, OCC) gemm_stream_kernel(GP p) {
    using T = bf16;
    using LA = RowkDma<BM, false>;
    using LB = RowkDma<BN, false>;
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int TILE_A = BM * 128, TILE_B = BN * 128;            // one K tile (64 k) of each operand
    constexpr int STAGE_A = KT * TILE_A;
    constexpr int SROW = BN * 4 + 16;                               // staging row: BN floats (+16 B: rows staggered over the banks)
    constexpr int CPR = BN / 8;                                     // 8-column chunks per row
    constexpr int NPASS = BM / 16;                                  // epilogue passes of 16 rows
    constexpr int ITER = (16 * CPR + 255) / 256;                    // chunks per thread per pass (BN = 64: half the threads idle)
    constexpr int OFF_A = KT * TILE_B, OFF_STG = OFF_A + 2 * STAGE_A;
    constexpr int SMEM_BYTES = OFF_STG + 16 * SROW;
    constexpr int NDMA_A = KT * LA::NI;                             // DMA instructions per wave per A tile
    static_assert(FM >= 1 && FN >= 1 && 256 % CPR == 0, "tile / thread map");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.y * BN;
    const int ntm = (p.M + BM - 1) / BM;
    int t = blockIdx.x;
    if (t >= ntm) return;
    const int G = gridDim.x;
    const Opnd oa = {p.A, p.a_tab, p.lda, p.a_mode, p.a_bytes};
    const Opnd ob = {p.B, p.b_tab, p.ldb, p.b_mode, p.b_bytes};

    // ---- the weights: once per workgroup
    {
        LB lb;
        lb.init(p, ob, n0, p.N, 0, tid);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) lb.template issue<true>(p, smem + kt * TILE_B, wave);
    }
    // ---- the thread's columns in the epilogue (the same for every tile and pass)
    const int cc = tid % CPR, n = n0 + cc * 8;
    float sc[8], sh[8];
    if (p.scale) load8(p.scale + n, sc);
    if (p.shift) load8(p.shift + n, sh);
    constexpr bool HAS_R = EPI != 0;

    auto issue_a = [&](int tile, int stage) __attribute__((always_inline)) {
        LA la;
        la.init(p, oa, tile * BM, p.M, 0, tid);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) la.template issue<true>(p, smem + OFF_A + stage * STAGE_A + kt * TILE_A, wave);
    };
    issue_a(t, 0);
    int stage = 0;
    for (; t < ntm; t += G) {
        const int m0 = t * BM;
        const bool more = t + G < ntm;
        // epilogue operands of THIS tile: all requested now, consumed after the product
        bf16x8 rpre[HAS_R ? NPASS * ITER : 1];
        if constexpr (HAS_R) {
#pragma unroll
            for (int q = 0; q < NPASS; ++q)
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int id = tid + it * 256;
                    const int m = m0 + q * 16 + id / CPR;
                    const bool ok = id < 16 * CPR && m < p.M;
                    const int64_t mm = ok ? (int64_t)m : 0;         // (rows past M re-read row 0: never used)
                    rpre[q * ITER + it] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(p.residual) + mm * p.ldr + n);
                }
        }
        if (more) {
            issue_a(t + G, stage ^ 1);
            CB_WAIT_VMCNT(NDMA_A);                                  // all but the next tile's DMA: this tile's A (and B) are in LDS
        } else {
            CB_WAIT_VMCNT(0);
        }
        // (the barrier intrinsic is not a memory operation to LLVM: the empty asm statements keep the compiler from moving the fragment
        // ds_reads below above it, or the previous tile's LDS accesses below it -- ADVICE r4)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // ... for every wave's share
        asm volatile("" ::: "memory");

        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
        const unsigned char* As = smem + OFF_A + stage * STAGE_A;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const unsigned char* Ak = As + kt * TILE_A;
            const unsigned char* Bk = smem + kt * TILE_B;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[FM], bfr[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(Ak + lds_off<T>(wm * WM + i * 16 + (lane & 15), kk * 4 + (lane >> 4)));
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    bfr[j] = *reinterpret_cast<const bf16x8*>(Bk + lds_off<T>(wn * WN + j * 16 + (lane & 15), kk * 4 + (lane >> 4)));
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
        }

        // ---- epilogue: 16 rows per pass through the staging buffer
        // (rolled loops, round 5: one copy of the ~10 KB epilogue8 body that stays in the instruction cache across passes AND tiles;
        // unrolled, the NPASS x ITER copies made the per-tile loop larger than the cache, so every tile ran out of instruction misses)
        unsigned char* stg = smem + OFF_STG;
#pragma unroll 1
        for (int q = 0; q < NPASS; ++q) {
            CB_LDS_BARRIER();                                       // the previous pass (or tile) has been read
            if (wm == q / FM) {
#pragma unroll
                for (int f = 0; f < FM; ++f) {
                    if (f == q % FM) {                              // (compile-time accumulator index inside each case)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            *reinterpret_cast<f32x4*>(stg + (lane & 15) * SROW + (wn * WN + j * 16 + 4 * (lane >> 4)) * 4) = acc[f][j];
                    }
                }
            }
            CB_LDS_BARRIER();
#pragma unroll 1
            for (int it = 0; it < ITER; ++it) {
                const int id = tid + it * 256;
                const int rl = id / CPR;
                const int m = m0 + q * 16 + rl;
                bf16x8 rp = {};
                if constexpr (EPI != 0) {
#pragma unroll
                    for (int x = 0; x < NPASS * ITER; ++x)
                        if (x == q * ITER + it) rp = rpre[x];
                }
                if (id < 16 * CPR && m < p.M) {
                    float v[8];
                    load8(reinterpret_cast<const float*>(stg + rl * SROW + cc * 32), v);
                    if constexpr (EPI == 0) epilogue8<T>(p, v, sc, sh, m, (int64_t)m, n);
                    else epilogue8<T, 2>(p, v, sc, sh, m, (int64_t)m, n, rp);
                }
            }
        }
        stage ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------
// host side: which instantiation (if any) covers a prepared problem, and the launch
// ---------------------------------------------------------------------------------------------
//   variant   tile (BM x BN, K tiles)   LDS      workgroups / CU     serves
//   0         64 x 256, 1               65 KiB   2                   res2 conv3 / projection shortcut (K = 64, N = 256)
//   1         64 x 128, 2               73 KiB   2                   res3 conv3 (K = 128, N = 512 in four column blocks)
// Measured on MI355X (profiles/r04c_stream_probe.json, cold caches): 200704 x 256 x 64 + residual 87.0 -> 54.2 us (2.66 -> 4.27 TB/s of
// algorithmic bytes; in the step 63 -> 48 us = 4.9 TB/s), without residual 51.2 -> 34.9 us.  Built, measured and DROPPED (same file):
// 64 x 256 with two K tiles at one workgroup per CU (113 KiB of LDS; res3 conv3 50.1 -> 49.8 us cold, slower in the step), 32 x 64 with
// four K tiles (res2 conv1, K = 256: 28.1 -> 28.9 us -- the 64x64 kernel already streams that shape at 4.5 TB/s), 64 x 64 x 64
// (19.1 -> 26.6 us) and the data-gradient form with the fused ReLU x FrozenBN backward (58.9 -> 78.4 us: the 64x64 kernel with its
// epilogue-operand prefetch runs that one at 5.5 TB/s).
constexpr int STREAM_VARIANTS = 2;
inline int stream_workgroups_per_cu(int) { return 2; }
inline unsigned stream_cus() {                                  // CUs of the current device (MI355X: 256)
    static const unsigned n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            return (unsigned)cus;
        return 256u;
    }();
    const int cap = cb_persistent_max_workgroups(0);            // (tests: CB_PERSISTENT_MAXWG workgroups in all, at least one "CU")
    return cap > 0 ? (unsigned)((cap + 1) / 2) : n;
}

template <int BM, int BN, int KT, int OCC>
int launch_gemm_stream_one(const GP& p, int wg_per_cu, hipStream_t st) {
    const unsigned ntm = (unsigned)((p.M + BM - 1) / BM), ny = (unsigned)(p.N / BN);
    unsigned gx = stream_cus() * (unsigned)wg_per_cu / ny;
    if (gx < 1) gx = 1;
    if (gx > ntm) gx = ntm;
    const dim3 grid(gx, ny);
    if (p.residual) hipLaunchKernelGGL((gemm_stream_kernel<BM, BN, KT, OCC, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_stream_kernel<BM, BN, KT, OCC, 0>), grid, dim3(256), 0, st, p);
    return cb_launch_status("cb_gemm (stream)");
}

int launch_gemm_stream(const GP& p, int variant, hipStream_t st);

}  // namespace cbgemm
