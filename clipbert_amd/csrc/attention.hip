// Self-attention core of the cross-modal encoder (BertSelfAttention, transformers.py:257-282 of the
// reference) on the fused QKV activation (B*L, 3*H*64).
//
// Sequence lengths on this path are tiny (L = Lt + Lv = 29..174; 41 at the headline shape), so one
// (batch, head) needs L*64 elements of K and V: attention is < 2 % of the encoder FLOPs and is
// latency-, not MFMA-bound.  Round-1 design: one 64-lane wave per 64 query (or key) rows, one ROW PER
// LANE, K/V (or Q/dO) streamed through LDS in blocks of 16/32 rows and read as wave-wide broadcasts;
// fp32 online softmax; everything stays in registers/LDS, nothing L x L ever touches HBM.
//   fwd : ctx, lse                    (thread = query row)
//   bwd1: D = rowsum(dO*O), dQ        (thread = query row)
//   bwd2: dK, dV                      (thread = key row; recomputes P from lse)
#include "common.h"

namespace {

constexpr int DH = 64;       // head size (hidden 768 / 12 heads, src/configs/base_model.json)
constexpr int KB = 32;       // rows per broadcast block
constexpr float MASK_NEG = -10000.0f;   // HF-2.11 extended attention mask

template <typename T> __device__ __forceinline__ void load_row64(const T* p, float (&v)[DH]) {
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        f32x4 t = load4(p + d);
        v[d] = t[0]; v[d + 1] = t[1]; v[d + 2] = t[2]; v[d + 3] = t[3];
    }
}

// cooperative load of `n` rows x 64 (from a strided global matrix) into LDS [KB][DH] as fp32
template <typename T>
__device__ __forceinline__ void stage_rows(const T* base, int64_t stride, int row0, int nrows_total, float (*dst)[DH], int lane) {
    // 64 lanes x 8 iterations x 4 elements = KB*DH
#pragma unroll
    for (int it = 0; it < KB * DH / (64 * 4); ++it) {
        int idx = (it * 64 + lane) * 4;
        int r = idx / DH, d = idx % DH;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row0 + r < nrows_total) v = load4(base + (int64_t)(row0 + r) * stride + d);
        *reinterpret_cast<f32x4*>(&dst[r][d]) = v;
    }
}

template <typename T>
__global__ void __launch_bounds__(64) attention_fwd_kernel(const T* qkv, const float* key_mask, T* ctx, float* lse, int B, int L,
                                                           int H, float drop_p, uint64_t seed, const uint64_t* seed_ptr) {
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    __shared__ __attribute__((aligned(16))) float Ks[KB][DH];
    __shared__ __attribute__((aligned(16))) float Vs[KB][DH];
    __shared__ float Ms[KB];
    const int lane = threadIdx.x;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = blockIdx.y * 64 + lane;
    const bool qok = qi < L;
    const int64_t stride = 3 * H * DH;
    const T* qbase = qkv + (int64_t)b * L * stride + h * DH;
    const T* kbase = qbase + H * DH;
    const T* vbase = qbase + 2 * H * DH;
    float q[DH], o[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) { q[d] = 0.f; o[d] = 0.f; }
    if (qok) load_row64(qbase + (int64_t)qi * stride, q);
    float m = -3.0e38f, l = 0.f;
    for (int j0 = 0; j0 < L; j0 += KB) {
        __syncthreads();
        stage_rows(kbase, stride, j0, L, Ks, lane);
        stage_rows(vbase, stride, j0, L, Vs, lane);
        if (lane < KB) Ms[lane] = (j0 + lane < L) ? (1.0f - key_mask[(int64_t)b * L + j0 + lane]) * MASK_NEG : 0.f;
        __syncthreads();
        const int nk = (L - j0 < KB) ? L - j0 : KB;
        float s[KB];
        float bm = -3.0e38f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) acc += q[d] * Ks[j][d];
            s[j] = (j < nk) ? acc * 0.125f + Ms[j] : -3.0e38f;
            bm = fmaxf(bm, s[j]);
        }
        const float mn = fmaxf(m, bm);
        const float resc = __expf(m - mn);
        l *= resc;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] *= resc;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (j < nk) {
                float pj = __expf(s[j] - mn);
                l += pj;
                if (drop_p > 0.f) pj *= dropout_mult(seed, ((uint64_t)bh * L + qi) * L + j0 + j, drop_p);
#pragma unroll
                for (int d = 0; d < DH; ++d) o[d] += pj * Vs[j][d];
            }
        }
        m = mn;
    }
    if (qok) {
        const float inv = 1.0f / l;
        T* dst = ctx + ((int64_t)b * L + qi) * (H * DH) + h * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            f32x4 v = {o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
            store4(dst + d, v);
        }
        if (lse) lse[(int64_t)bh * L + qi] = m + __logf(l);
    }
}

// thread = query row: D_i, dQ_i.  dqkv layout = qkv layout; writes the Q third.
template <typename T>
__global__ void __launch_bounds__(64) attention_bwd_q_kernel(const T* qkv, const float* key_mask, const T* ctx, const T* dctx,
                                                             const float* lse, float* dsum, T* dqkv, int B, int L, int H,
                                                             float drop_p, uint64_t seed, const uint64_t* seed_ptr) {
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    __shared__ __attribute__((aligned(16))) float Ks[KB][DH];
    __shared__ __attribute__((aligned(16))) float Vs[KB][DH];
    __shared__ float Ms[KB];
    const int lane = threadIdx.x;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = blockIdx.y * 64 + lane;
    const bool qok = qi < L;
    const int64_t stride = 3 * H * DH;
    const T* qbase = qkv + (int64_t)b * L * stride + h * DH;
    const T* kbase = qbase + H * DH;
    const T* vbase = qbase + 2 * H * DH;
    float q[DH], go[DH], dq[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) { q[d] = 0.f; go[d] = 0.f; dq[d] = 0.f; }
    float Di = 0.f, lse_i = 0.f;
    if (qok) {
        load_row64(qbase + (int64_t)qi * stride, q);
        const int64_t crow = ((int64_t)b * L + qi) * (H * DH) + h * DH;
        load_row64(dctx + crow, go);
        float oo[DH];
        load_row64(ctx + crow, oo);
#pragma unroll
        for (int d = 0; d < DH; ++d) Di += go[d] * oo[d];
        lse_i = lse[(int64_t)bh * L + qi];
        dsum[(int64_t)bh * L + qi] = Di;
    }
    for (int j0 = 0; j0 < L; j0 += KB) {
        __syncthreads();
        stage_rows(kbase, stride, j0, L, Ks, lane);
        stage_rows(vbase, stride, j0, L, Vs, lane);
        if (lane < KB) Ms[lane] = (j0 + lane < L) ? (1.0f - key_mask[(int64_t)b * L + j0 + lane]) * MASK_NEG : 0.f;
        __syncthreads();
        const int nk = (L - j0 < KB) ? L - j0 : KB;
#pragma unroll 4
        for (int j = 0; j < KB; ++j) {
            if (j < nk) {
                float sc = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) { sc += q[d] * Ks[j][d]; dp += go[d] * Vs[j][d]; }
                float p = __expf(sc * 0.125f + Ms[j] - lse_i);
                if (drop_p > 0.f) dp *= dropout_mult(seed, ((uint64_t)bh * L + qi) * L + j0 + j, drop_p);
                float ds = p * (dp - Di) * 0.125f;
#pragma unroll
                for (int d = 0; d < DH; ++d) dq[d] += ds * Ks[j][d];
            }
        }
    }
    if (qok) {
        T* dst = dqkv + ((int64_t)b * L + qi) * stride + h * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            f32x4 v = {dq[d], dq[d + 1], dq[d + 2], dq[d + 3]};
            store4(dst + d, v);
        }
    }
}

// thread = key row: dK_j, dV_j; queries (Q, dO, lse, D) streamed through LDS.
template <typename T>
__global__ void __launch_bounds__(64) attention_bwd_kv_kernel(const T* qkv, const float* key_mask, const T* dctx, const float* lse,
                                                              const float* dsum, T* dqkv, int B, int L, int H, float drop_p,
                                                              uint64_t seed, const uint64_t* seed_ptr) {
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    __shared__ __attribute__((aligned(16))) float Qs[KB][DH];
    __shared__ __attribute__((aligned(16))) float Gs[KB][DH];
    __shared__ float Ls[KB], Ds[KB];
    const int lane = threadIdx.x;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int kj = blockIdx.y * 64 + lane;
    const bool kok = kj < L;
    const int64_t stride = 3 * H * DH;
    const T* qbase = qkv + (int64_t)b * L * stride + h * DH;
    const T* kbase = qbase + H * DH;
    const T* vbase = qbase + 2 * H * DH;
    const T* gbase = dctx + (int64_t)b * L * (H * DH) + h * DH;
    float k[DH], v[DH], dk[DH], dv[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) { k[d] = 0.f; v[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    float madd = 0.f;
    if (kok) {
        load_row64(kbase + (int64_t)kj * stride, k);
        load_row64(vbase + (int64_t)kj * stride, v);
        madd = (1.0f - key_mask[(int64_t)b * L + kj]) * MASK_NEG;
    }
    for (int i0 = 0; i0 < L; i0 += KB) {
        __syncthreads();
        stage_rows(qbase, stride, i0, L, Qs, lane);
        stage_rows(gbase, (int64_t)H * DH, i0, L, Gs, lane);
        if (lane < KB) {
            bool ok = i0 + lane < L;
            Ls[lane] = ok ? lse[(int64_t)bh * L + i0 + lane] : 0.f;
            Ds[lane] = ok ? dsum[(int64_t)bh * L + i0 + lane] : 0.f;
        }
        __syncthreads();
        const int nq = (L - i0 < KB) ? L - i0 : KB;
#pragma unroll 2
        for (int i = 0; i < KB; ++i) {
            if (i < nq) {
                float sc = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) { sc += Qs[i][d] * k[d]; dp += Gs[i][d] * v[d]; }
                float p = __expf(sc * 0.125f + madd - Ls[i]);
                float mult = 1.0f;
                if (drop_p > 0.f) mult = dropout_mult(seed, ((uint64_t)bh * L + i0 + i) * L + kj, drop_p);
                float pd = p * mult;
                float ds = p * (dp * mult - Ds[i]) * 0.125f;
#pragma unroll
                for (int d = 0; d < DH; ++d) { dv[d] += pd * Gs[i][d]; dk[d] += ds * Qs[i][d]; }
            }
        }
    }
    if (kok) {
        T* dkp = dqkv + ((int64_t)b * L + kj) * stride + H * DH + h * DH;
        T* dvp = dkp + H * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            f32x4 a = {dk[d], dk[d + 1], dk[d + 2], dk[d + 3]};
            f32x4 c = {dv[d], dv[d + 1], dv[d + 2], dv[d + 3]};
            store4(dkp + d, a);
            store4(dvp + d, c);
        }
    }
}

}  // namespace

extern "C" int cb_attention_fwd(int32_t dtype, const void* qkv, const float* key_mask, void* ctx, float* lse, int32_t B, int32_t L,
                                int32_t H, float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_ptr, void* stream) {
    CB_REQUIRE(qkv && key_mask && ctx && B > 0 && L > 0 && H > 0, "cb_attention_fwd: bad arguments");
    dim3 g(B * H, (L + 63) / 64), b(64);
    if (dtype == CB_BF16) hipLaunchKernelGGL((attention_fwd_kernel<bf16>), g, b, 0, cb_stream(stream), (const bf16*)qkv, key_mask, (bf16*)ctx, lse, B, L, H, dropout_p, dropout_seed, dropout_seed_ptr);
    else if (dtype == CB_F32) hipLaunchKernelGGL((attention_fwd_kernel<float>), g, b, 0, cb_stream(stream), (const float*)qkv, key_mask, (float*)ctx, lse, B, L, H, dropout_p, dropout_seed, dropout_seed_ptr);
    else return cb_fail("cb_attention_fwd: bad dtype");
    return cb_launch_status("cb_attention_fwd");
}

extern "C" int cb_attention_bwd(int32_t dtype, const void* qkv, const float* key_mask, const void* ctx, const void* dctx,
                                const float* lse, float* dsum_ws, void* dqkv, int32_t B, int32_t L, int32_t H, float dropout_p,
                                uint64_t dropout_seed, const uint64_t* dropout_seed_ptr, void* stream) {
    CB_REQUIRE(qkv && key_mask && ctx && dctx && lse && dsum_ws && dqkv && B > 0 && L > 0 && H > 0, "cb_attention_bwd: bad arguments");
    dim3 g(B * H, (L + 63) / 64), b(64);
    hipStream_t st = cb_stream(stream);
    if (dtype == CB_BF16) {
        hipLaunchKernelGGL((attention_bwd_q_kernel<bf16>), g, b, 0, st, (const bf16*)qkv, key_mask, (const bf16*)ctx, (const bf16*)dctx, lse, dsum_ws, (bf16*)dqkv, B, L, H, dropout_p, dropout_seed, dropout_seed_ptr);
        hipLaunchKernelGGL((attention_bwd_kv_kernel<bf16>), g, b, 0, st, (const bf16*)qkv, key_mask, (const bf16*)dctx, lse, dsum_ws, (bf16*)dqkv, B, L, H, dropout_p, dropout_seed, dropout_seed_ptr);
    } else if (dtype == CB_F32) {
        hipLaunchKernelGGL((attention_bwd_q_kernel<float>), g, b, 0, st, (const float*)qkv, key_mask, (const float*)ctx, (const float*)dctx, lse, dsum_ws, (float*)dqkv, B, L, H, dropout_p, dropout_seed, dropout_seed_ptr);
        hipLaunchKernelGGL((attention_bwd_kv_kernel<float>), g, b, 0, st, (const float*)qkv, key_mask, (const float*)dctx, lse, dsum_ws, (float*)dqkv, B, L, H, dropout_p, dropout_seed, dropout_seed_ptr);
    } else return cb_fail("cb_attention_bwd: bad dtype");
    return cb_launch_status("cb_attention_bwd");
}
