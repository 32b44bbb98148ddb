// HBM-bound NHWC helpers around the convolution stack: stem input pack (fused ImageNorm + BGR flip +
// zero padding), max pooling forward/backward, ReLU+FrozenBN backward.  All are one pass over their
// tensors with 8/16-byte accesses along the channel dimension.
#include "common.h"

namespace {

// dst (N, Hp, Wp, 4): pixel (hp, wp) <- src pixel (hp - pad, wp - pad), channels B,G,R,0.
// One thread per destination pixel; consecutive threads walk w, so the three NCHW plane reads and the
// 8/16-byte destination store are coalesced.
template <typename T, typename S>
__global__ void __launch_bounds__(256) stem_pack_kernel(const S* src, T* dst, int N, int H, int W, int Hp, int Wp,
                                                        int pad, f32x4 mean, f32x4 istd) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)N * Hp * Wp;
    if (idx >= total) return;
    int wp = (int)(idx % Wp);
    int64_t t = idx / Wp;
    int hp = (int)(t % Hp);
    int n = (int)(t / Hp);
    int h = hp - pad, w = wp - pad;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
        int64_t plane = (int64_t)H * W;
        const S* p = src + ((int64_t)n * 3) * plane + (int64_t)h * W + w;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int sc = 2 - c;   // RGB -> BGR (src/modeling/grid_feat.py:92-94)
            v[c] = ((float)p[sc * plane] - mean[sc]) * istd[sc];
        }
    }
    store4(dst + idx * 4, v);
}

__global__ void __launch_bounds__(256) image_norm_kernel(const uint8_t* src, float* dst, f32x4 mean, f32x4 istd,
                                                         int64_t total, int64_t hw) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int c = (int)((i / hw) % 3);
    dst[i] = ((float)src[i] - mean[c]) * istd[c];
}

template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T* x, T* y, int N, int H, int W, int C, int OH, int OW,
                                                          int k, int stride, int pad, int relu) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int C4 = C >> 2;
    int64_t total = (int64_t)N * OH * OW * C4;
    if (idx >= total) return;
    int c = (int)(idx % C4) * 4;
    int64_t t = idx / C4;
    int ow = (int)(t % OW); t /= OW;
    int oh = (int)(t % OH);
    int n = (int)(t / OH);
    const float ninf = -3.0e38f;
    f32x4 m = {ninf, ninf, ninf, ninf};
    for (int r = 0; r < k; ++r) {
        int ih = oh * stride - pad + r;
        if ((unsigned)ih >= (unsigned)H) continue;
        for (int s = 0; s < k; ++s) {
            int iw = ow * stride - pad + s;
            if ((unsigned)iw >= (unsigned)W) continue;
            f32x4 v = load4(x + (((int64_t)n * H + ih) * W + iw) * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
        }
    }
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], 0.f);
    }
    store4(y + (((int64_t)n * OH + oh) * OW + ow) * C + c, m);
}

// k = 2, stride 2, pad 0 (floor): windows do not overlap.  The grid walks ceil(H/2) x ceil(W/2) patches so that the kernel itself
// writes the zeros of the pixels no window covers (odd H / W: the last row / column) -- no memset in front of it: a hipMemsetAsync
// issued from inside a hipGraph capture did NOT clear the buffer on replays of the captured step (round 5, tests/test_bench_step.py:
// replay 0 correct on fresh memory, every later replay fed the stale bytes of the uncovered pixels into the whole ResNet backward).
template <typename T>
__global__ void __launch_bounds__(256) maxpool2_bwd_kernel(const T* x, const T* y, const T* dy, T* dx, int N, int H, int W,
                                                           int C, int OH, int OW, int relu) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int C4 = C >> 2;
    const int PH = (H + 1) >> 1, PW = (W + 1) >> 1;          // 2x2 patches, the last ones possibly cut by the border
    int64_t total = (int64_t)N * PH * PW * C4;
    if (idx >= total) return;
    int c = (int)(idx % C4) * 4;
    int64_t t = idx / C4;
    int ow = (int)(t % PW); t /= PW;
    int oh = (int)(t % PH);
    int n = (int)(t / PH);
    if (oh >= OH || ow >= OW) {                             // not a pooling window: its (in-range) pixels receive no gradient
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ih = oh * 2 + (q >> 1), iw = ow * 2 + (q & 1);
            if (ih < H && iw < W) store4(dx + (((int64_t)n * H + ih) * W + iw) * C + c, z);
        }
        return;
    }
    int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c;
    f32x4 g = load4(dy + o);
    if (relu) {
        f32x4 yy = load4(y + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
    }
    f32x4 v[4];
    int64_t base[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        base[q] = (((int64_t)n * H + oh * 2 + (q >> 1)) * W + ow * 2 + (q & 1)) * C + c;
        v[q] = load4(x + base[q]);
    }
    f32x4 out[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int best = 0;
        float bv = v[0][e];
#pragma unroll
        for (int q = 1; q < 4; ++q) if (v[q][e] > bv) { bv = v[q][e]; best = q; }   // first max wins (PyTorch order)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q][e] = (q == best) ? g[e] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) store4(dx + base[q], out[q]);
}

template <typename T>
__global__ void __launch_bounds__(256) relu_scale_bwd_kernel(const T* dy, const T* y, const float* scale, T* g, T* dz,
                                                             const float* scale2, T* g2, int64_t total4, int C) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    int64_t e0 = i * 4;
    int c = (int)(e0 % C);
    f32x4 d = load4(dy + e0), yy = load4(y + e0);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = yy[e] > 0.f ? d[e] : 0.f;
    if (dz) store4(dz + e0, d);
    if (g) store4(g + e0, d * load4(scale + c));
    if (g2) store4(g2 + e0, d * load4(scale2 + c));
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int cb_stem_pack(int32_t dtype, const void* src, int32_t src_u8, const float* mean3, const float* std3,
                            void* dst, int32_t N, int32_t H, int32_t W, int32_t Hp, int32_t Wp, int32_t pad,
                            void* stream) {
    CB_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && Hp >= H + pad && Wp >= W + pad, "cb_stem_pack: bad arguments");
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, istd = {1.f, 1.f, 1.f, 1.f};
    if (src_u8) {
        CB_REQUIRE(mean3 && std3, "cb_stem_pack: uint8 input needs host mean/std arrays");
        for (int c = 0; c < 3; ++c) { mean[c] = mean3[c]; istd[c] = 1.0f / std3[c]; }
    }
    int64_t total = (int64_t)N * Hp * Wp;
    dim3 g(nblk(total)), b(256);
    hipStream_t st = cb_stream(stream);
    if (dtype == CB_BF16) {
        if (src_u8) hipLaunchKernelGGL((stem_pack_kernel<bf16, uint8_t>), g, b, 0, st, (const uint8_t*)src, (bf16*)dst, N, H, W, Hp, Wp, pad, mean, istd);
        else hipLaunchKernelGGL((stem_pack_kernel<bf16, float>), g, b, 0, st, (const float*)src, (bf16*)dst, N, H, W, Hp, Wp, pad, mean, istd);
    } else if (dtype == CB_F32) {
        if (src_u8) hipLaunchKernelGGL((stem_pack_kernel<float, uint8_t>), g, b, 0, st, (const uint8_t*)src, (float*)dst, N, H, W, Hp, Wp, pad, mean, istd);
        else hipLaunchKernelGGL((stem_pack_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, N, H, W, Hp, Wp, pad, mean, istd);
    } else return cb_fail("cb_stem_pack: bad dtype");
    return cb_launch_status("cb_stem_pack");
}

extern "C" int cb_image_norm(const uint8_t* src, float* dst, const float* mean3, const float* std3, int64_t n_images,
                             int64_t hw, void* stream) {
    CB_REQUIRE(src && dst && mean3 && std3 && n_images > 0 && hw > 0, "cb_image_norm: bad arguments");
    f32x4 mean = {mean3[0], mean3[1], mean3[2], 0.f}, istd = {1.f / std3[0], 1.f / std3[1], 1.f / std3[2], 1.f};
    int64_t total = n_images * 3 * hw;
    hipLaunchKernelGGL(image_norm_kernel, dim3(nblk(total)), dim3(256), 0, cb_stream(stream), src, dst, mean, istd, total, hw);
    return cb_launch_status("cb_image_norm");
}

extern "C" int cb_maxpool_fwd(int32_t dtype, const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH,
                              int32_t OW, int32_t k, int32_t stride, int32_t pad, int32_t relu, void* stream) {
    CB_REQUIRE(x && y && C % 4 == 0 && k > 0 && stride > 0, "cb_maxpool_fwd: bad arguments (C must be a multiple of 4)");
    int64_t total = (int64_t)N * OH * OW * (C / 4);
    if (total == 0) return 0;
    dim3 g(nblk(total)), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((maxpool_fwd_kernel<bf16>), g, b, 0, cb_stream(stream), (const bf16*)x, (bf16*)y, N, H, W, C, OH, OW, k, stride, pad, relu);
    else if (dtype == CB_F32) hipLaunchKernelGGL((maxpool_fwd_kernel<float>), g, b, 0, cb_stream(stream), (const float*)x, (float*)y, N, H, W, C, OH, OW, k, stride, pad, relu);
    else return cb_fail("cb_maxpool_fwd: bad dtype");
    return cb_launch_status("cb_maxpool_fwd");
}

extern "C" int cb_maxpool2_bwd(int32_t dtype, const void* x, const void* y, const void* dy, void* dx, int32_t N, int32_t H,
                               int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t relu, void* stream) {
    CB_REQUIRE(x && y && dy && dx && C % 4 == 0 && OH * 2 <= H && OW * 2 <= W, "cb_maxpool2_bwd: bad arguments");
    int esz = dtype == CB_BF16 ? 2 : 4;
    hipStream_t st = cb_stream(stream);
    (void)esz;
    int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);       // every input pixel belongs to exactly one (possibly cut) patch
    if (total == 0) return 0;
    dim3 g(nblk(total)), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((maxpool2_bwd_kernel<bf16>), g, b, 0, st, (const bf16*)x, (const bf16*)y, (const bf16*)dy, (bf16*)dx, N, H, W, C, OH, OW, relu);
    else if (dtype == CB_F32) hipLaunchKernelGGL((maxpool2_bwd_kernel<float>), g, b, 0, st, (const float*)x, (const float*)y, (const float*)dy, (float*)dx, N, H, W, C, OH, OW, relu);
    else return cb_fail("cb_maxpool2_bwd: bad dtype");
    return cb_launch_status("cb_maxpool2_bwd");
}

extern "C" int cb_relu_scale_bwd(int32_t dtype, const void* dy, const void* y, const float* scale, void* g, void* dz,
                                 const float* scale2, void* g2, int64_t rows, int32_t C, void* stream) {
    CB_REQUIRE(dy && y && C % 4 == 0, "cb_relu_scale_bwd: bad arguments");
    CB_REQUIRE((!g || scale) && (!g2 || scale2), "cb_relu_scale_bwd: scale missing");
    int64_t total4 = rows * C / 4;
    if (total4 == 0) return 0;
    dim3 gr(nblk(total4)), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((relu_scale_bwd_kernel<bf16>), gr, b, 0, cb_stream(stream), (const bf16*)dy, (const bf16*)y, scale, (bf16*)g, (bf16*)dz, scale2, (bf16*)g2, total4, C);
    else if (dtype == CB_F32) hipLaunchKernelGGL((relu_scale_bwd_kernel<float>), gr, b, 0, cb_stream(stream), (const float*)dy, (const float*)y, scale, (float*)g, (float*)dz, scale2, (float*)g2, total4, C);
    else return cb_fail("cb_relu_scale_bwd: bad dtype");
    return cb_launch_status("cb_relu_scale_bwd");
}
