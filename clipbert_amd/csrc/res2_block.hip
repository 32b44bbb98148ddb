// Fused bottleneck block of ResNet-50's res2 stage (detectron2 BottleneckBlock with 64 mid channels, STRIDE_IN_1X1, stride 1), forward
// only, FrozenBN, bf16, NHWC -- SURVEY a3, src/modeling/grid_feat.py:89-105 (the frozen half of the backbone: FREEZE_AT = 2).
//
//     y1 = relu(bn1(conv1x1(x)))      y2 = relu(bn2(conv3x3(y1)))      out = relu(bn3(conv1x1(y2)) + shortcut(x))
//
// Why one kernel.  As three (four) launches the block moves x twice, y1 / y2 once each way and the 256-channel output once: 413 MB for a
// 200704-pixel batch, and runs at 140 us (r05f: 29 + 45 + 65; the stage-entry block with its projection shortcut 179 us) -- every one of
// them HBM-bound (2 * K flop per output element, K <= 576).  Fused, the 64-channel intermediates never leave the CU: x in (with a one-pixel
// halo), out out.
//
// Structure (256 threads = 4 waves, <= 2 workgroups per CU, persistent over 8 x 8-pixel output tiles):
//   * WEIGHTS LIVE IN REGISTERS for the whole kernel: every wave owns a slice of the output channels of each convolution (16 of conv1's
//     and conv2's 64, 64 of conv3's 256) and keeps the MFMA B fragments of its slice -- conv1 16 x CIN, conv2 16 x 9 x 64, conv3 64 x 64
//     (+ projection shortcut 64 x CIN): 136-144 VGPRs, loaded once.
//   * ACTIVATIONS LIVE IN LDS: the 10 x 10 x CIN input tile (halo included), y1 on the same 10 x 10 grid (zeros outside the image: conv2
//     pads y1, not x), y2 on the 8 x 8 tile; rows padded by 16 bytes so that the 16 rows of an A fragment fall into distinct banks.  All
//     four waves read the same A fragments -- each pixel row is multiplied by every wave's channel slice.
//   * v_mfma_f32_16x16x32_bf16, fp32 accumulate; FrozenBN scale / shift, ReLU and the residual in the accumulator layout; the result
//     replaces the residual it consumed in the LDS tile (identity shortcut) or goes to a staging tile (projection), and leaves with
//     16-byte row-contiguous stores.
// Same arithmetic as the unfused path (bf16 storage of y1 / y2, fp32 accumulation in tap-major K order): tests/test_res2_block.py.
#include "common.h"
#include <stdlib.h>
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

namespace {

constexpr int TH = 8, TW = 8, HALO_W = TW + 2, NPIX = (TH + 2) * (TW + 2);      // 100 halo pixels, 64 output pixels
constexpr int MID = 64, COUT = 256;
constexpr int P64 = MID * 2 + 16;                                               // LDS pitch of a 64-channel row
constexpr int POUT = COUT * 2 + 16;

struct Res2P {
    const bf16* x; bf16* out;
    const bf16* w1; const bf16* w2; const bf16* w3; const bf16* wsc;
    const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3; const float* ssc; const float* bsc;
    int N, H, W, tiles_h, tiles_w, ntiles;
    uint32_t x_bytes;
};

typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0)) rsrc_t;
constexpr uint32_t OOB = 0x80000000u;          // a byte offset past every buffer: the range check of the descriptor returns zeros

__device__ __forceinline__ bf16x8 ld8(const bf16* q) { return *reinterpret_cast<const bf16x8*>(q); }
__device__ __forceinline__ bf16x8 lds8(const unsigned char* q) { return *reinterpret_cast<const bf16x8*>(q); }

template <int CIN, bool PROJ>
__global__ void __launch_bounds__(256, 2) res2_block_kernel(Res2P p) {
    // Input tile image.  CIN = 256: rows of exactly 512 B filled by LDS-DMA (buffer_load ... lds: 1 KiB = two rows per wave instruction, the
    // whole 50 KiB tile in flight at once instead of register-staged batches -- the tile load is one memory round trip), 16-byte chunk c of
    // row r stored at chunk c ^ (r & 15): the 16 rows of an A fragment hit 16 different bank groups.  CIN = 64: padded rows, register-staged.
    constexpr bool DMA = CIN == 256;
    constexpr int XP = DMA ? CIN * 2 : CIN * 2 + 16;       // LDS pitch of an input row
    auto xoff = [](int row, int chunk) { return DMA ? row * XP + ((chunk ^ (row & 15)) << 4) : row * XP + (chunk << 4); };
    constexpr int KA = CIN / 32;                           // K steps of conv1 (and of the projection shortcut)
    constexpr int NCH = CIN / 8;                           // 16-byte chunks per input pixel
    constexpr int XS_BYTES = NPIX * XP, Y1_BYTES = NPIX * P64, Y2_BYTES = TH * TW * P64;
    constexpr int OS_BYTES = PROJ ? TH * TW * POUT : 0;
    static_assert(PROJ || CIN == COUT, "identity shortcut: the block keeps its channel count");
    static_assert(2 * (XS_BYTES + Y1_BYTES + Y2_BYTES + OS_BYTES) <= 160 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) unsigned char smem[XS_BYTES + Y1_BYTES + Y2_BYTES + OS_BYTES];
    unsigned char* const Xs = smem;
    unsigned char* const Y1s = Xs + XS_BYTES;
    unsigned char* const Y2s = Y1s + Y1_BYTES;
    unsigned char* const Os = PROJ ? Y2s + Y2_BYTES : Xs;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const uint32_t x_bytes = p.x_bytes;

    // ---- this wave's weight slices, as MFMA B fragments (row n = lr of the slice, 8 consecutive k at 8 * lq): once per kernel
    bf16x8 w1f[KA], w2f[9][2], w3f[4][2], wscf[PROJ ? 4 : 1][PROJ ? KA : 1];
    {
        const int n = 16 * wave + lr;
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) w1f[kk] = ld8(p.w1 + (size_t)n * CIN + kk * 32 + 8 * lq);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) w2f[t][kk] = ld8(p.w2 + (size_t)n * (9 * MID) + t * MID + kk * 32 + 8 * lq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n3 = 64 * wave + 16 * j + lr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) w3f[j][kk] = ld8(p.w3 + (size_t)n3 * MID + kk * 32 + 8 * lq);
            if constexpr (PROJ) {
#pragma unroll
                for (int kk = 0; kk < KA; ++kk) wscf[j][kk] = ld8(p.wsc + (size_t)n3 * CIN + kk * 32 + 8 * lq);
            }
        }
    }
    // FrozenBN constants of the channels this lane holds in the accumulator layout (4 consecutive channels at 4 * lq of a 16-channel fragment)
    const int cA = 16 * wave + 4 * lq;
    const f32x4 s1 = load4(p.s1 + cA), b1 = load4(p.b1 + cA), s2 = load4(p.s2 + cA), b2 = load4(p.b2 + cA);

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int tw = tile % p.tiles_w, rest = tile / p.tiles_w;
        const int th = rest % p.tiles_h, n = rest / p.tiles_h;
        const int h0 = th * TH, w0 = tw * TW;
        // ---- 1. the input tile with its halo -> LDS (pixels outside the image: zeros)
        if constexpr (DMA) {
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.x), (short)0, (int)x_bytes, 0x00020000);
#pragma unroll
            for (int k = 0; k < (NPIX / 2 + 3) / 4; ++k) {
                const int inst = wave + 4 * k;                                  // rows 2 * inst, 2 * inst + 1
                if (inst < NPIX / 2) {
                    const int row = 2 * inst + (lane >> 5), pc = lane & 31, lc = pc ^ (row & 15);
                    const int gh = h0 - 1 + row / HALO_W, gw = w0 - 1 + row % HALO_W;
                    const bool ok = (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
                    const uint32_t off = ok ? (uint32_t)(((((size_t)n * p.H + gh) * p.W + gw) * CIN + lc * 8) * 2) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Xs + inst * 1024), 16, off, 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);                                 // vmcnt(0): the tile has landed (this wave's share)
        } else {
            constexpr int TOTAL = NPIX * NCH, ITERS = (TOTAL + 255) / 256;
            u32x4 v[ITERS];
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = tid + it * 256;
                const int pix = idx / NCH, ch = idx % NCH;
                const int gh = h0 - 1 + pix / HALO_W, gw = w0 - 1 + pix % HALO_W;
                const bool ok = idx < TOTAL && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
                u32x4 z = {0u, 0u, 0u, 0u};
                v[it] = z;
                if (ok) v[it] = *reinterpret_cast<const u32x4*>(p.x + (((size_t)n * p.H + gh) * p.W + gw) * CIN + ch * 8);
            }
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = tid + it * 256;
                if (idx < TOTAL) *reinterpret_cast<u32x4*>(Xs + xoff(idx / NCH, idx % NCH)) = v[it];
            }
        }
        __syncthreads();
        // ---- 2. conv1 (1x1, CIN -> 64) on all 100 halo pixels: this wave's 16 channels; y1 = 0 outside the image (conv2 pads y1)
#pragma unroll 1
        for (int i0 = 0; i0 < 8; i0 += 2) {                                   // two row fragments per trip (independent accumulators), rolled:
            f32x4 acc[2];                                                      // the weight fragments own the register file
            int rr[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = (i0 + u) * 16 + lr;
                rr[u] = row < NPIX ? row : NPIX - 1;                           // (fragment rows past the tile: discarded below)
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                acc[u] = z;
            }
#pragma unroll
            for (int kk = 0; kk < KA; ++kk) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[kk], lds8(Xs + xoff(rr[u], kk * 4 + lq)), acc[u], 0, 0, 0);
                if (kk % 2 == 1) asm volatile("" ::: "memory");               // (keeps the compiler from hoisting all 2 * KA fragment reads: registers)
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = (i0 + u) * 16 + lr;
                if (row < NPIX) {
                    const int gh = h0 - 1 + row / HALO_W, gw = w0 - 1 + row % HALO_W;
                    const bool in = (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = __builtin_fmaf(acc[u][e], s1[e], b1[e]);
                        o[e] = (bf16)((in && y > 0.f) ? y : 0.f);
                    }
                    *reinterpret_cast<bf16x4*>(Y1s + row * P64 + cA * 2) = o;
                }
            }
        }
        __syncthreads();
        // ---- 3. conv2 (3x3, pad 1, 64 -> 64) on the 64 output pixels: K = 9 taps x 64 channels, tap-major
#pragma unroll 1
        for (int i0 = 0; i0 < 4; i0 += 2) {
            f32x4 acc[2];
            const unsigned char* base[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pix = (i0 + u) * 16 + lr;
                base[u] = Y1s + ((pix >> 3) * HALO_W + (pix & 7)) * P64 + lq * 16;
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                acc[u] = z;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t][kk], lds8(base[u] + ((t / 3) * HALO_W + t % 3) * P64 + kk * 64), acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pix = (i0 + u) * 16 + lr;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = __builtin_fmaf(acc[u][e], s2[e], b2[e]);
                    o[e] = (bf16)(y > 0.f ? y : 0.f);
                }
                *reinterpret_cast<bf16x4*>(Y2s + pix * P64 + cA * 2) = o;
            }
        }
        __syncthreads();
        // ---- 4. conv3 (1x1, 64 -> 256) + FrozenBN + shortcut + ReLU: this wave's 64 channels
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                          // one 16-channel fragment at a time: small live set next to the weights
            const int ch = 64 * wave + 16 * j + 4 * lq;
            const f32x4 s3 = load4(p.s3 + ch), b3 = load4(p.b3 + ch);
            f32x4 ss = {0.f, 0.f, 0.f, 0.f}, bs = {0.f, 0.f, 0.f, 0.f};
            if constexpr (PROJ) { ss = load4(p.ssc + ch); bs = load4(p.bsc + ch); }
#pragma unroll 2
            for (int i = 0; i < 4; ++i) {
                const int pix = i * 16 + lr;
                const int hrow = ((pix >> 3) + 1) * HALO_W + (pix & 7) + 1;   // the pixel's row in the halo tile
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[j][0], lds8(Y2s + pix * P64 + lq * 16), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[j][1], lds8(Y2s + pix * P64 + 64 + lq * 16), acc, 0, 0, 0);
                bf16x4 o;
                if constexpr (PROJ) {
                    f32x4 asc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < KA; ++kk)
                        asc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wscf[j][kk], lds8(Xs + xoff(hrow, kk * 4 + lq)), asc, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // (the unfused path stores the shortcut branch in bf16 before the residual add: same rounding point)
                        const float sc = (float)(bf16)__builtin_fmaf(asc[e], ss[e], bs[e]);
                        const float y = __builtin_fmaf(acc[e], s3[e], b3[e]) + sc;
                        o[e] = (bf16)(y > 0.f ? y : 0.f);
                    }
                    *reinterpret_cast<bf16x4*>(Os + pix * POUT + ch * 2) = o;
                } else {
                    unsigned char* q = Xs + xoff(hrow, ch >> 3) + (ch & 7) * 2;                 // the residual; replaced by the result
                    const bf16x4 res = *reinterpret_cast<const bf16x4*>(q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = __builtin_fmaf(acc[e], s3[e], b3[e]) + (float)res[e];
                        o[e] = (bf16)(y > 0.f ? y : 0.f);
                    }
                    *reinterpret_cast<bf16x4*>(q) = o;
                }
            }
        }
        __syncthreads();
        // ---- 5. the 64 x 256 result: LDS -> global, 16 bytes per lane, 512-byte runs per pixel
#pragma unroll
        for (int it = 0; it < TH * TW * (COUT / 8) / 256; ++it) {
            const int idx = tid + it * 256;
            const int pix = idx / (COUT / 8), ch = idx % (COUT / 8);
            const unsigned char* src = PROJ ? Os + pix * POUT + ch * 16 : Xs + xoff(((pix >> 3) + 1) * HALO_W + (pix & 7) + 1, ch);
            const int gh = h0 + (pix >> 3), gw = w0 + (pix & 7);
            if (gh < p.H && gw < p.W)
                *reinterpret_cast<u32x4*>(p.out + (((size_t)n * p.H + gh) * p.W + gw) * COUT + ch * 8) = *reinterpret_cast<const u32x4*>(src);
        }
        __syncthreads();                                     // the next tile overwrites the LDS images
    }
}

}  // namespace

extern "C" int cb_res2_block(const cb_res2_desc* d, void* stream) {
    CB_REQUIRE(d != nullptr, "cb_res2_block: null descriptor");
    CB_REQUIRE(d->x && d->out && d->w1 && d->w2 && d->w3 && d->scale1 && d->shift1 && d->scale2 && d->shift2 && d->scale3 && d->shift3,
               "cb_res2_block: null operand");
    CB_REQUIRE(d->cin == 64 || d->cin == 256, "cb_res2_block: input channels %d (the res2 stage has 64 or 256)", d->cin);
    CB_REQUIRE((d->wsc != nullptr) == (d->cin == 64), "cb_res2_block: the 64-channel entry block takes a projection shortcut, the others none");
    CB_REQUIRE(!d->wsc || (d->scale_sc && d->shift_sc), "cb_res2_block: projection shortcut without its FrozenBN");
    CB_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "cb_res2_block: empty input");
    for (const void* q : {d->x, (const void*)d->out, d->w1, d->w2, d->w3, d->wsc, (const void*)d->scale1, (const void*)d->shift1, (const void*)d->scale2,
                          (const void*)d->shift2, (const void*)d->scale3, (const void*)d->shift3, (const void*)d->scale_sc, (const void*)d->shift_sc})
        CB_REQUIRE(q == nullptr || aligned16(q), "cb_res2_block: operands must be 16-byte aligned");
    Res2P p;
    p.x = (const bf16*)d->x; p.out = (bf16*)d->out;
    p.w1 = (const bf16*)d->w1; p.w2 = (const bf16*)d->w2; p.w3 = (const bf16*)d->w3; p.wsc = (const bf16*)d->wsc;
    p.s1 = d->scale1; p.b1 = d->shift1; p.s2 = d->scale2; p.b2 = d->shift2; p.s3 = d->scale3; p.b3 = d->shift3; p.ssc = d->scale_sc; p.bsc = d->shift_sc;
    p.N = d->N; p.H = d->H; p.W = d->W;
    p.tiles_h = (d->H + TH - 1) / TH; p.tiles_w = (d->W + TW - 1) / TW;
    const int64_t nt = (int64_t)d->N * p.tiles_h * p.tiles_w;
    CB_REQUIRE(nt < (1ll << 31), "cb_res2_block: too many tiles");
    p.ntiles = (int)nt;
    const int64_t xb = (int64_t)d->N * d->H * d->W * d->cin * 2;
    CB_REQUIRE(xb < 0x7fffffffll, "cb_res2_block: input larger than 2 GiB");
    p.x_bytes = (uint32_t)xb;
    const int max_wg = cb_persistent_max_workgroups(512);                                                                           // two per CU
    const unsigned grid = (unsigned)(nt < max_wg ? nt : max_wg);
    hipStream_t st = cb_stream(stream);
    if (d->cin == 64) hipLaunchKernelGGL((res2_block_kernel<64, true>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((res2_block_kernel<256, false>), dim3(grid), dim3(256), 0, st, p);
    return cb_launch_status("cb_res2_block");
}
