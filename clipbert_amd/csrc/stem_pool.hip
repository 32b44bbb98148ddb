// Fused ResNet-50 stem of the grid backbone, forward only: 7x7 stride-2 convolution (3 -> 64) + FrozenBN + ReLU + 3x3 stride-2 max-pool
// (detectron2 BasicStem; src/modeling/grid_feat.py:89-105) on the packed NHWC4 image cb_stem_pack writes, bf16.
//
// Why one kernel.  As cb_gemm + cb_maxpool_fwd the 112 x 112 x 64 convolution output (103 MB for 64 frames) is written, read back by the
// pooling kernel and thrown away: 120 + 36 us (r05f), the convolution issue-bound on its 7-row gather addressing and the pool at a 19 %
// L2 hit rate (profiles/r04k).  Here a workgroup owns an 8 x 8 tile of POOLED pixels: the 17 x 17 convolution outputs it needs are
// computed from a 39 x 40-pixel input tile in LDS (12 KB), stay in LDS as bf16 (41 KB) and are pooled from there -- 27 MB in (x 1.5 of
// halo), 26 MB out.
//
// Structure (256 threads = 4 waves, persistent over the tiles, like cb_res2_block): the filter lives in registers -- every wave owns 16 of
// the 64 output channels as 7 MFMA B fragments, one per filter ROW (K = 7 rows x (8 taps x 4 channels), tap 7 and channel 3 zero: the image
// of _stem_weight / cb_gemm's stem form, so the fp32 summation order is the one of the unfused path); an A fragment is 16 convolution
// pixels x one filter row = 16 bytes per lane straight from the input tile (pixel (2 oy + r, 2 ox + 2 q), 2 pixels x 4 channels).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int PT = 8;                        // pooled tile
constexpr int CT = 2 * PT + 1;               // 17 x 17 convolution outputs feed it (3x3 windows, stride 2, pad 1)
constexpr int NCONV = CT * CT;               // 289
constexpr int NFRAG = (NCONV + 15) / 16;     // 19 row fragments (the last one partial)
constexpr int IR = 2 * CT + 5;               // 39 input rows (7-row filter, stride 2)
constexpr int ICH = 20;                      // 16-byte chunks (2 pixels x 4 channels) per input row: 40 pixels, 39 used
constexpr int IP = ICH * 16;                 // 320 B
constexpr int PC = 64 * 2 + 16;              // LDS pitch of a 64-channel row

struct StemP {
    const bf16* img; bf16* out; const bf16* w; const float* scale; const float* shift;
    int N, Hp, Wp, OH, OW, PH, PW, tiles_h, tiles_w, ntiles;
    // U8 (cb_stem_pool_u8, round 6): the frames themselves, (N, 3, H, W) uint8 RGB planes -- ImageNorm, the RGB -> BGR flip and the zero padding
    // of cb_stem_pack happen while the input tile moves into LDS (same arithmetic, same bits); Hp / Wp are then the PADDED extents H + 6, W + 8
    const uint8_t* src8; int H, W; f32x4 mean, istd;
};

// NT threads: 256 (4 waves: each owns 16 channels and all 19 row fragments) or 512 (8 waves: the two wave groups split the row fragments
// -- twice the waves per SIMD to hide the LDS / MFMA latency chains of a tile; same LDS footprint).
template <int NT, bool U8 = false>
__global__ void __launch_bounds__(NT, 2) stem_pool_kernel(StemP p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[IR * IP + NFRAG * 16 * PC];
    unsigned char* const Is = smem;
    unsigned char* const Cs = smem + IR * IP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_id & 3, half = wave_id >> 2;         // channel slice; row-fragment half (NT = 512)
    constexpr int HALF_FRAGS = NT == 512 ? 10 : NFRAG + 1;     // fragments per wave group (even: two per trip)
    const int lr = lane & 15, lq = lane >> 4;
    bf16x8 wf[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) wf[r] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(16 * wave + lr) * 224 + r * 32 + 8 * lq);
    const int cA = 16 * wave + 4 * lq;
    const f32x4 sc = load4(p.scale + cA), sh = load4(p.shift + cA);

    // input tile of `tile` -> registers (rows / pixels outside the packed image: zeros; they only feed convolution pixels outside the map)
    constexpr int TOTAL = IR * ICH, ITERS = (TOTAL + NT - 1) / NT;
    auto fetch = [&](int tile, u32x4 (&v)[ITERS]) __attribute__((always_inline)) {
        const int tw = tile % p.tiles_w, rest = tile / p.tiles_w;
        const int th = rest % p.tiles_h, n = rest / p.tiles_h;
        const int iy0 = 2 * (2 * th * PT - 1), ix0 = 2 * (2 * tw * PT - 1);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = tid + it * NT;
            const int gy = iy0 + idx / ICH, gx = ix0 + 2 * (idx % ICH);
            const bool ok = tile < p.ntiles && idx < TOTAL && (unsigned)gy < (unsigned)p.Hp && gx >= 0 && gx + 1 < p.Wp;
            u32x4 z = {0u, 0u, 0u, 0u};
            v[it] = z;
            if (ok) v[it] = *reinterpret_cast<const u32x4*>(p.img + (((size_t)n * p.Hp + gy) * p.Wp + gx) * 4);
        }
    };
    auto stash = [&](const u32x4 (&v)[ITERS]) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = tid + it * NT;
            if (idx < TOTAL) {
                const u32x4 o = v[it];
                *reinterpret_cast<u32x4*>(Is + (idx / ICH) * IP + (idx % ICH) * 16) = o;
            }
        }
    };
    // U8: a thread owns 8 consecutive source pixels of one tile row (4 chunks: 39 rows x 5 runs = 195 tasks): three 8-byte loads, one per
    // colour plane, when the whole run lies inside the frame (unaligned: the run starts 3 pixels left of an even column), guarded byte loads
    // on the frame's border.  The raw bytes stay in registers while the current tile is convolved; stash8 applies ImageNorm + BGR0 exactly as
    // cb_stem_pack does -- ((float)byte - mean) * (1 / std), channel c <- plane 2 - c, zeros outside the frame -- on the way into LDS.
    constexpr int RUNS = ICH / 4;
    static_assert(ICH % 4 == 0 && IR * RUNS <= NT, "one run of 8 pixels per thread");
    auto fetch8 = [&](int tile, uint32_t (&raw)[7]) __attribute__((always_inline)) {
        const int tw = tile % p.tiles_w, rest = tile / p.tiles_w;
        const int th = rest % p.tiles_h, n = rest / p.tiles_h;
        const int r = tid / RUNS, g = tid % RUNS;
        const int sy = 2 * (2 * th * PT - 1) + r - 3, sx0 = 2 * (2 * tw * PT - 1) + 8 * g - 3;
        const bool row = tile < p.ntiles && tid < IR * RUNS && (unsigned)sy < (unsigned)p.H;
        uint32_t mask = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) mask |= (row && (unsigned)(sx0 + j) < (unsigned)p.W) ? (1u << j) : 0u;
#pragma unroll
        for (int i = 0; i < 6; ++i) raw[i] = 0u;
        raw[6] = mask;
        if (mask == 0u) return;
        const size_t plane = (size_t)p.H * p.W;
        const uint8_t* q = p.src8 + ((size_t)n * 3 * p.H + sy) * p.W + sx0;
        if (mask == 0xffu) {
#pragma unroll
            for (int c = 0; c < 3; ++c) __builtin_memcpy(&raw[2 * c], q + c * plane, 8);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if ((mask >> j) & 1u) raw[2 * c + (j >> 2)] |= (uint32_t)q[c * plane + j] << (8 * (j & 3));
        }
    };
    auto stash8 = [&](const uint32_t (&raw)[7]) __attribute__((always_inline)) {
        if (tid >= IR * RUNS) return;
        const int r = tid / RUNS, g = tid % RUNS;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            union { bf16x8 h; u32x4 r; } u;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * k + jj;
                const bool ok = (raw[6] >> j) & 1u;
                const float rr = (float)((raw[0 + (j >> 2)] >> (8 * (j & 3))) & 255u), gg = (float)((raw[2 + (j >> 2)] >> (8 * (j & 3))) & 255u),
                            bb = (float)((raw[4 + (j >> 2)] >> (8 * (j & 3))) & 255u);
                u.h[4 * jj + 0] = (bf16)(ok ? (bb - p.mean[2]) * p.istd[2] : 0.f);
                u.h[4 * jj + 1] = (bf16)(ok ? (gg - p.mean[1]) * p.istd[1] : 0.f);
                u.h[4 * jj + 2] = (bf16)(ok ? (rr - p.mean[0]) * p.istd[0] : 0.f);
                u.h[4 * jj + 3] = (bf16)0.f;
            }
            *reinterpret_cast<u32x4*>(Is + r * IP + (4 * g + k) * 16) = u.r;
        }
    };
    u32x4 nxt[ITERS];
    uint32_t raw8[7];
    if constexpr (U8) { fetch8(blockIdx.x, raw8); stash8(raw8); }
    else { fetch(blockIdx.x, nxt); stash(nxt); }
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int tw = tile % p.tiles_w, rest = tile / p.tiles_w;
        const int th = rest % p.tiles_h, n = rest / p.tiles_h;
        const int ph0 = th * PT, pw0 = tw * PT;
        const int cy0 = 2 * ph0 - 1, cx0 = 2 * pw0 - 1;          // convolution pixel of tile position (0, 0)
        // ---- 1. the input tile is in LDS (stashed behind the previous tile's convolution); the NEXT tile's loads go out now and land
        //         under this tile's convolution
        if constexpr (U8) fetch8(tile + gridDim.x, raw8);
        else fetch(tile + gridDim.x, nxt);
        __syncthreads();
        // ---- 2. the 17 x 17 convolution outputs of this wave's 16 channels: FrozenBN + ReLU, zero outside the map, bf16 -> LDS
#pragma unroll 1
        for (int i0 = half * HALF_FRAGS; i0 < (half + 1) * HALF_FRAGS && i0 < NFRAG; i0 += 2) {
            f32x4 acc[2];
            const unsigned char* base[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int pix = (i0 + u) * 16 + lr;
                pix = pix < NCONV ? pix : NCONV - 1;                  // (fragment rows past the tile: discarded below)
                base[u] = Is + (2 * (pix / CT)) * IP + (2 * (pix % CT) + 2 * lq) * 8;
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                acc[u] = z;
            }
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[r], *reinterpret_cast<const bf16x8*>(base[u] + r * IP), acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pix = (i0 + u) * 16 + lr;
                if (i0 + u < NFRAG && pix < NCONV) {
                    const int gy = cy0 + pix / CT, gx = cx0 + pix % CT;
                    const bool in = (unsigned)gy < (unsigned)p.OH && (unsigned)gx < (unsigned)p.OW;
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = __builtin_fmaf(acc[u][e], sc[e], sh[e]);
                        o[e] = (bf16)((in && y > 0.f) ? y : 0.f);     // (>= 0 everywhere: a zero stands in for the pool's -inf padding)
                    }
                    *reinterpret_cast<bf16x4*>(Cs + pix * PC + cA * 2) = o;
                }
            }
        }
        __syncthreads();
        if constexpr (U8) stash8(raw8);                          // (the convolution is done with the input tile: the next one moves in)
        else stash(nxt);
        // ---- 3. 3 x 3 / stride 2 max-pool out of LDS: thread <-> (pooled pixel, 8 channels), 16-byte stores, 128-byte runs per pixel
#pragma unroll
        for (int it = 0; it < PT * PT * 8 / NT; ++it) {
            const int idx = tid + it * NT;
            const int pix = idx >> 3, ch = idx & 7;
            const int py = pix >> 3, px = pix & 7;
            float m[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(Cs + ((2 * py + t / 3) * CT + 2 * px + t % 3) * PC + ch * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; m[e] = f > m[e] ? f : m[e]; }
            }
            const int gy = ph0 + py, gx = pw0 + px;
            if (gy < p.PH && gx < p.PW) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)m[e];
                *reinterpret_cast<bf16x8*>(p.out + (((size_t)n * p.PH + gy) * p.PW + gx) * 64 + ch * 8) = o;
            }
        }
        __syncthreads();
    }
}

}  // namespace

// cb_stem_pool with cb_stem_pack folded in: uint8 frames -> pooled stem output, one launch (the input pipeline ends in the first
// convolution's tile loader)
extern "C" int cb_stem_pool_u8(const uint8_t* frames, const float* mean3, const float* std3, const void* weight, const float* scale,
                               const float* shift, void* out, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t PH, int32_t PW,
                               void* stream) {
    CB_REQUIRE(frames && mean3 && std3 && weight && scale && shift && out, "cb_stem_pool_u8: null operand");
    CB_REQUIRE(N > 0 && H > 0 && W > 0 && OH == (H + 6 - 7) / 2 + 1 && OW == (W + 6 - 7) / 2 + 1, "cb_stem_pool_u8: %d x %d frames give a %d x %d convolution output, not %d x %d",
               H, W, (H + 6 - 7) / 2 + 1, (W + 6 - 7) / 2 + 1, OH, OW);
    CB_REQUIRE(PH == (OH + 2 - 3) / 2 + 1 && PW == (OW + 2 - 3) / 2 + 1, "cb_stem_pool_u8: pooled size %d x %d does not follow from %d x %d", PH, PW, OH, OW);
    for (const void* q : {weight, (const void*)scale, (const void*)shift, (const void*)out})
        CB_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15) == 0, "cb_stem_pool_u8: weight / scale / shift / out must be 16-byte aligned");
    CB_REQUIRE((int64_t)N * 3 * H * W < (1ll << 40), "cb_stem_pool_u8: too many pixels");
    StemP p{};
    p.src8 = frames; p.H = H; p.W = W; p.img = nullptr;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.istd[c] = 1.0f / std3[c]; }
    p.mean[3] = 0.f; p.istd[3] = 1.f;
    p.out = (bf16*)out; p.w = (const bf16*)weight; p.scale = scale; p.shift = shift;
    p.N = N; p.Hp = H + 6; p.Wp = W + 8; p.OH = OH; p.OW = OW; p.PH = PH; p.PW = PW;
    p.tiles_h = (PH + PT - 1) / PT; p.tiles_w = (PW + PT - 1) / PT;
    const int64_t nt = (int64_t)N * p.tiles_h * p.tiles_w;
    CB_REQUIRE(nt < (1ll << 31), "cb_stem_pool_u8: too many tiles");
    p.ntiles = (int)nt;
    const int max_wg = cb_persistent_max_workgroups(512);
    hipLaunchKernelGGL((stem_pool_kernel<512, true>), dim3((unsigned)(nt < max_wg ? nt : max_wg)), dim3(512), 0, cb_stream(stream), p);
    return cb_launch_status("cb_stem_pool_u8");
}

extern "C" int cb_stem_pool(const void* packed, const void* weight, const float* scale, const float* shift, void* out, int32_t N,
                            int32_t Hp, int32_t Wp, int32_t OH, int32_t OW, int32_t PH, int32_t PW, void* stream) {
    CB_REQUIRE(packed && weight && scale && shift && out, "cb_stem_pool: null operand");
    CB_REQUIRE(N > 0 && OH > 0 && OW > 0 && PH == (OH + 2 - 3) / 2 + 1 && PW == (OW + 2 - 3) / 2 + 1, "cb_stem_pool: pooled size %d x %d does not follow from %d x %d", PH, PW, OH, OW);
    CB_REQUIRE(Hp >= 2 * (OH - 1) + 7 && Wp >= 2 * (OW - 1) + 8 && Wp % 2 == 0, "cb_stem_pool: packed image %d x %d too small for a %d x %d convolution (cb_stem_pack: pad 3, even width)", Hp, Wp, OH, OW);
    for (const void* q : {packed, weight, (const void*)scale, (const void*)shift, (const void*)out})
        CB_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15) == 0, "cb_stem_pool: operands must be 16-byte aligned");
    StemP p{};
    p.img = (const bf16*)packed; p.out = (bf16*)out; p.w = (const bf16*)weight; p.scale = scale; p.shift = shift;
    p.N = N; p.Hp = Hp; p.Wp = Wp; p.OH = OH; p.OW = OW; p.PH = PH; p.PW = PW;
    p.tiles_h = (PH + PT - 1) / PT; p.tiles_w = (PW + PT - 1) / PT;
    const int64_t nt = (int64_t)N * p.tiles_h * p.tiles_w;
    CB_REQUIRE(nt < (1ll << 31), "cb_stem_pool: too many tiles");
    p.ntiles = (int)nt;
    const int max_wg = cb_persistent_max_workgroups(512);
    hipLaunchKernelGGL((stem_pool_kernel<512>), dim3((unsigned)(nt < max_wg ? nt : max_wg)), dim3(512), 0, cb_stream(stream), p);
    return cb_launch_status("cb_stem_pool");
}
