// Self-attention core of the cross-modal encoder (BertSelfAttention, transformers.py:257-282 of the
// reference) on the fused QKV activation (B*L, 3*H*64).
//
// Sequence lengths on this path are tiny (L = Lt + Lv = 29..174; 41 at the headline shape), so one
// (batch, head) needs L*64 elements of K and V: attention is < 2 % of the encoder FLOPs and is
// latency-, not MFMA-bound.  Round-1 design: a 256-thread block owns 64 query (or key) rows of one (batch, head);
// each row is shared by 4 adjacent lanes (16 of the 64 head dims each, dot products completed with two
// quad shuffles), K/V (or Q/dO) stream through LDS in blocks of 32 rows and are read as broadcasts; fp32 online
// softmax; nothing L x L ever touches HBM.
//   fwd : ctx, lse                    (lane quad = query row)
//   bwd1: D = rowsum(dO*O), dQ        (lane quad = query row)
//   bwd2: dK, dV                      (lane quad = key row; recomputes P from lse)
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int DH = 64;       // head size (hidden 768 / 12 heads, src/configs/base_model.json)
constexpr int PD = 16;       // head dims per lane (4 lanes per row)
constexpr int KB = 32;       // rows per broadcast block
constexpr int RPB = 64;      // rows per block
constexpr float MASK_NEG = -10000.0f;   // HF-2.11 extended attention mask

template <typename T> __device__ __forceinline__ void load_part(const T* p, float (&v)[PD]) {
#pragma unroll
    for (int d = 0; d < PD; d += 4) {
        f32x4 t = load4(p + d);
        v[d] = t[0]; v[d + 1] = t[1]; v[d + 2] = t[2]; v[d + 3] = t[3];
    }
}
template <typename T> __device__ __forceinline__ void store_part(T* p, const float (&v)[PD]) {
#pragma unroll
    for (int d = 0; d < PD; d += 4) {
        f32x4 t = {v[d], v[d + 1], v[d + 2], v[d + 3]};
        store4(p + d, t);
    }
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v;
}

// cooperative load of KB rows x 64 (from a strided global matrix) into LDS [KB][DH] as fp32 (256 threads)
template <typename T>
__device__ __forceinline__ void stage_rows(const T* base, int64_t stride, int row0, int nrows_total, float (*dst)[DH], int tid) {
#pragma unroll
    for (int it = 0; it < KB * DH / (256 * 4); ++it) {
        int idx = (it * 256 + tid) * 4;
        int r = idx / DH, d = idx % DH;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row0 + r < nrows_total) v = load4(base + (int64_t)(row0 + r) * stride + d);
        *reinterpret_cast<f32x4*>(&dst[r][d]) = v;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) attention_fwd_kernel(const T* qkv, const float* key_mask, T* ctx, float* lse, int B, int L,
                                                            int H, float drop_p, uint64_t seed, const uint64_t* seed_ptr) {
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    __shared__ __attribute__((aligned(16))) float Ks[KB][DH];
    __shared__ __attribute__((aligned(16))) float Vs[KB][DH];
    __shared__ float Ms[KB];
    const int tid = threadIdx.x, part = tid & 3;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = blockIdx.y * RPB + (tid >> 2);
    const bool qok = qi < L;
    const int64_t stride = 3 * H * DH;
    const T* qbase = qkv + (int64_t)b * L * stride + h * DH;
    const T* kbase = qbase + H * DH;
    const T* vbase = qbase + 2 * H * DH;
    float q[PD], o[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d) { q[d] = 0.f; o[d] = 0.f; }
    if (qok) load_part(qbase + (int64_t)qi * stride + part * PD, q);
    float m = -3.0e38f, l = 0.f;
    for (int j0 = 0; j0 < L; j0 += KB) {
        __syncthreads();
        stage_rows(kbase, stride, j0, L, Ks, tid);
        stage_rows(vbase, stride, j0, L, Vs, tid);
        if (tid < KB) Ms[tid] = (j0 + tid < L) ? (1.0f - key_mask[(int64_t)b * L + j0 + tid]) * MASK_NEG : 0.f;
        __syncthreads();
        const int nk = (L - j0 < KB) ? L - j0 : KB;
        float s[KB];
        float bm = -3.0e38f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < PD; ++d) acc += q[d] * Ks[j][part * PD + d];
            acc = quad_sum(acc);
            s[j] = (j < nk) ? acc * 0.125f + Ms[j] : -3.0e38f;
            bm = fmaxf(bm, s[j]);
        }
        const float mn = fmaxf(m, bm);
        const float resc = __expf(m - mn);
        l *= resc;
#pragma unroll
        for (int d = 0; d < PD; ++d) o[d] *= resc;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (j < nk) {
                float pj = __expf(s[j] - mn);
                l += pj;
                if (drop_p > 0.f) pj *= dropout_mult1(seed, ((uint64_t)bh * L + qi) * ((L + 3) >> 2) + ((j0 + j) >> 2), (j0 + j) & 3, drop_p);
#pragma unroll
                for (int d = 0; d < PD; ++d) o[d] += pj * Vs[j][part * PD + d];
            }
        }
        m = mn;
    }
    if (qok) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < PD; ++d) o[d] *= inv;
        store_part(ctx + ((int64_t)b * L + qi) * (H * DH) + h * DH + part * PD, o);
        if (lse && part == 0) lse[(int64_t)bh * L + qi] = m + __logf(l);
    }
}

// lane quad = query row: D_i, dQ_i.  dqkv layout = qkv layout; writes the Q third.
template <typename T>
__global__ void __launch_bounds__(256) attention_bwd_q_kernel(const T* qkv, const float* key_mask, const T* ctx, const T* dctx,
                                                              const float* lse, float* dsum, T* dqkv, int B, int L, int H,
                                                              float drop_p, uint64_t seed, const uint64_t* seed_ptr) {
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    __shared__ __attribute__((aligned(16))) float Ks[KB][DH];
    __shared__ __attribute__((aligned(16))) float Vs[KB][DH];
    __shared__ float Ms[KB];
    const int tid = threadIdx.x, part = tid & 3;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = blockIdx.y * RPB + (tid >> 2);
    const bool qok = qi < L;
    const int64_t stride = 3 * H * DH;
    const T* qbase = qkv + (int64_t)b * L * stride + h * DH;
    const T* kbase = qbase + H * DH;
    const T* vbase = qbase + 2 * H * DH;
    float q[PD], go[PD], dq[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d) { q[d] = 0.f; go[d] = 0.f; dq[d] = 0.f; }
    float Di = 0.f, lse_i = 0.f;
    if (qok) {
        load_part(qbase + (int64_t)qi * stride + part * PD, q);
        const int64_t crow = ((int64_t)b * L + qi) * (H * DH) + h * DH + part * PD;
        load_part(dctx + crow, go);
        float oo[PD];
        load_part(ctx + crow, oo);
#pragma unroll
        for (int d = 0; d < PD; ++d) Di += go[d] * oo[d];
        lse_i = lse[(int64_t)bh * L + qi];
    }
    Di = quad_sum(Di);
    if (qok && part == 0) dsum[(int64_t)bh * L + qi] = Di;
    for (int j0 = 0; j0 < L; j0 += KB) {
        __syncthreads();
        stage_rows(kbase, stride, j0, L, Ks, tid);
        stage_rows(vbase, stride, j0, L, Vs, tid);
        if (tid < KB) Ms[tid] = (j0 + tid < L) ? (1.0f - key_mask[(int64_t)b * L + j0 + tid]) * MASK_NEG : 0.f;
        __syncthreads();
        const int nk = (L - j0 < KB) ? L - j0 : KB;
#pragma unroll 8
        for (int j = 0; j < KB; ++j) {
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < PD; ++d) { sc += q[d] * Ks[j][part * PD + d]; dp += go[d] * Vs[j][part * PD + d]; }
            sc = quad_sum(sc);
            dp = quad_sum(dp);
            if (j < nk) {
                float p = __expf(sc * 0.125f + Ms[j] - lse_i);
                if (drop_p > 0.f) dp *= dropout_mult1(seed, ((uint64_t)bh * L + qi) * ((L + 3) >> 2) + ((j0 + j) >> 2), (j0 + j) & 3, drop_p);
                float ds = p * (dp - Di) * 0.125f;
#pragma unroll
                for (int d = 0; d < PD; ++d) dq[d] += ds * Ks[j][part * PD + d];
            }
        }
    }
    if (qok) store_part(dqkv + ((int64_t)b * L + qi) * stride + h * DH + part * PD, dq);
}

// lane quad = key row: dK_j, dV_j; queries (Q, dO, lse, D) streamed through LDS.
template <typename T>
__global__ void __launch_bounds__(256) attention_bwd_kv_kernel(const T* qkv, const float* key_mask, const T* dctx, const float* lse,
                                                               const float* dsum, T* dqkv, int B, int L, int H, float drop_p,
                                                               uint64_t seed, const uint64_t* seed_ptr) {
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    __shared__ __attribute__((aligned(16))) float Qs[KB][DH];
    __shared__ __attribute__((aligned(16))) float Gs[KB][DH];
    __shared__ float Ls[KB], Ds[KB];
    const int tid = threadIdx.x, part = tid & 3;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int kj = blockIdx.y * RPB + (tid >> 2);
    const bool kok = kj < L;
    const int64_t stride = 3 * H * DH;
    const T* qbase = qkv + (int64_t)b * L * stride + h * DH;
    const T* kbase = qbase + H * DH;
    const T* vbase = qbase + 2 * H * DH;
    const T* gbase = dctx + (int64_t)b * L * (H * DH) + h * DH;
    float k[PD], v[PD], dk[PD], dv[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d) { k[d] = 0.f; v[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    float madd = 0.f;
    if (kok) {
        load_part(kbase + (int64_t)kj * stride + part * PD, k);
        load_part(vbase + (int64_t)kj * stride + part * PD, v);
        madd = (1.0f - key_mask[(int64_t)b * L + kj]) * MASK_NEG;
    }
    for (int i0 = 0; i0 < L; i0 += KB) {
        __syncthreads();
        stage_rows(qbase, stride, i0, L, Qs, tid);
        stage_rows(gbase, (int64_t)H * DH, i0, L, Gs, tid);
        if (tid < KB) {
            bool ok = i0 + tid < L;
            Ls[tid] = ok ? lse[(int64_t)bh * L + i0 + tid] : 0.f;
            Ds[tid] = ok ? dsum[(int64_t)bh * L + i0 + tid] : 0.f;
        }
        __syncthreads();
        const int nq = (L - i0 < KB) ? L - i0 : KB;
#pragma unroll 8
        for (int i = 0; i < KB; ++i) {
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < PD; ++d) { sc += Qs[i][part * PD + d] * k[d]; dp += Gs[i][part * PD + d] * v[d]; }
            sc = quad_sum(sc);
            dp = quad_sum(dp);
            if (i < nq) {
                float p = __expf(sc * 0.125f + madd - Ls[i]);
                float mult = 1.0f;
                if (drop_p > 0.f) mult = dropout_mult1(seed, ((uint64_t)bh * L + i0 + i) * ((L + 3) >> 2) + (kj >> 2), kj & 3, drop_p);
                float pd = p * mult;
                float ds = p * (dp * mult - Ds[i]) * 0.125f;
#pragma unroll
                for (int d = 0; d < PD; ++d) { dv[d] += pd * Gs[i][part * PD + d]; dk[d] += ds * Qs[i][part * PD + d]; }
            }
        }
    }
    if (kok) {
        T* dkp = dqkv + ((int64_t)b * L + kj) * stride + H * DH + h * DH + part * PD;
        store_part(dkp, dk);
        store_part(dkp + H * DH, dv);
    }
}

}  // namespace

// attention_mfma.hip: one-wave-per-head MFMA kernels for short bf16 sequences
bool cb_attention_mfma_ok(int32_t dtype, const void* qkv, const void* ctx, const void* other, int32_t L);
int cb_attention_fwd_mfma(const void* qkv, const float* key_mask, void* ctx, float* lse, int32_t B, int32_t L, int32_t H, float p,
                          uint64_t seed, const uint64_t* seed_ptr, hipStream_t st);
int cb_attention_bwd_mfma(const void* qkv, const float* key_mask, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                          int32_t B, int32_t L, int32_t H, float p, uint64_t seed, const uint64_t* seed_ptr, hipStream_t st);
static bool use_mfma() { static const bool off = getenv("CB_ATTENTION_NO_MFMA") != nullptr; return !off; }

extern "C" int cb_attention_fwd(int32_t dtype, const void* qkv, const float* key_mask, void* ctx, float* lse, int32_t B, int32_t L,
                                int32_t H, float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_ptr, void* stream) {
    CB_REQUIRE(qkv && key_mask && ctx && B > 0 && L > 0 && H > 0, "cb_attention_fwd: bad arguments");
    if (use_mfma() && cb_attention_mfma_ok(dtype, qkv, ctx, nullptr, L))
        return cb_attention_fwd_mfma(qkv, key_mask, ctx, lse, B, L, H, dropout_p, dropout_seed, dropout_seed_ptr, cb_stream(stream));
    dim3 g(B * H, (L + 63) / 64), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((attention_fwd_kernel<bf16>), g, b, 0, cb_stream(stream), (const bf16*)qkv, key_mask, (bf16*)ctx, lse, B, L, H, dropout_p, dropout_seed, dropout_seed_ptr);
    else if (dtype == CB_F32) hipLaunchKernelGGL((attention_fwd_kernel<float>), g, b, 0, cb_stream(stream), (const float*)qkv, key_mask, (float*)ctx, lse, B, L, H, dropout_p, dropout_seed, dropout_seed_ptr);
    else return cb_fail("cb_attention_fwd: bad dtype");
    return cb_launch_status("cb_attention_fwd");
}

extern "C" int cb_attention_bwd(int32_t dtype, const void* qkv, const float* key_mask, const void* ctx, const void* dctx,
                                const float* lse, float* dsum_ws, void* dqkv, int32_t B, int32_t L, int32_t H, float dropout_p,
                                uint64_t dropout_seed, const uint64_t* dropout_seed_ptr, void* stream) {
    CB_REQUIRE(qkv && key_mask && ctx && dctx && lse && dsum_ws && dqkv && B > 0 && L > 0 && H > 0, "cb_attention_bwd: bad arguments");
    if (use_mfma() && cb_attention_mfma_ok(dtype, qkv, ctx, dctx, L) && (reinterpret_cast<uintptr_t>(dqkv) & 15) == 0)
        return cb_attention_bwd_mfma(qkv, key_mask, ctx, dctx, lse, dqkv, B, L, H, dropout_p, dropout_seed, dropout_seed_ptr, cb_stream(stream));
    dim3 g(B * H, (L + 63) / 64), b(256);
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
