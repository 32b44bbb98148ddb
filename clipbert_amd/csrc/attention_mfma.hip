// MFMA self-attention for short sequences (L <= 64, head size 64, bf16): ONE wave owns one (batch, head).
//
// The cross-modal encoder of the hot path sees L = Lt + Lv = 41 tokens (29..57 across the reference's tasks), so a
// whole head is 3x3 (at most 4x4) MFMA tiles: Q, K, V, dO fragments are loaded straight from the fused QKV
// activation in MFMA operand layout (16 B per lane, k = head dim), the score tiles live in accumulators, softmax
// reduces over the lane's 4 key rows x tiles and two cross-group shuffles, and probabilities feed the second GEMM
// as k-slots directly from the accumulator registers (the k permutation is arbitrary as long as both operands use
// it).  Operands that must be contracted over their ROW index (V in P.V, K in dS.K, Q / dO in dS^T.Q / P^T.dO) are
// staged once in LDS in their natural [row][64] image and read with ds_read_b64_tr_b16 (transpose read).
// Nothing L x L touches HBM; forward = 1 launch, backward = 1 launch (was 2) and both are latency-, not FLOP-bound:
// ~45 / ~150 MFMAs per head.
//
// Semantics = attention.hip (BertSelfAttention, src/modeling/transformers.py:257-282 of the reference): scores/8 +
// (1-mask)*-10000, softmax, inverted dropout on the probabilities (same hash stream, index ((b*H+h)*L+i)*L+j).
#include "common.h"

namespace {

constexpr int DH = 64;
constexpr int RS = 144;                 // LDS row stride in bytes (128 B of data + 16 B stagger)
constexpr float MASK_NEG = -10000.0f;   // HF-2.11 extended attention mask
constexpr float NEG_BIG = -3.0e38f;

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ s16x4 lds_read_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// MFMA operand fragment: row `row` of a [rows][stride] bf16 matrix, head dims d..d+7 (zeros past L)
__device__ __forceinline__ bf16x8 ld_frag(const bf16* base, int64_t stride, int row, int L, int d) {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
    if (row < L) z = *reinterpret_cast<const bf16x8*>(base + (int64_t)row * stride + d);
    return z;
}

// natural [row][64] image of a strided matrix in LDS, rows [L, rows) zero-filled (one wave)
__device__ __forceinline__ void stage(const bf16* base, int64_t stride, int L, int rows, unsigned char* dst, int lane) {
    for (int idx = lane; idx < rows * 8; idx += 64) {
        const int r = idx >> 3, seg = idx & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < L) v = *reinterpret_cast<const u32x4*>(base + (int64_t)r * stride + seg * 8);
        *reinterpret_cast<u32x4*>(dst + r * RS + seg * 16) = v;
    }
}

// Transposed operand of a staged matrix X[row][d]: fragment whose MFMA row a <-> d = 16*(a>>2) + 4*dt + (a&3) and
// whose k-slots (group g, e) <-> row = 32*pair + 16*(e>>2) + 4*g + (e&3) -- the slot order of pack_pair() below.
// With it as the first MFMA operand the lane's 4 accumulator rows are d = 16*g + 4*dt + 0..3: 16 consecutive head
// dims per lane over dt = 0..3.
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* mat, int pair, int dt, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const unsigned char* a0 = mat + (32 * pair + 4 * g + (p >> 2)) * RS + (16 * (p & 3) + 4 * dt) * 2;
    union { struct { s16x4 lo, hi; } h; bf16x8 v; } u;
    u.h.lo = lds_read_tr(a0);
    u.h.hi = lds_read_tr(a0 + 16 * RS);
    return u.v;
}

// two accumulator tiles (rows 4g+r of tiles 2*pair, 2*pair+1) -> 8 bf16 k-slots
__device__ __forceinline__ bf16x8 pack_pair(const f32x4& lo, const f32x4& hi) {
    bf16x8 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = (bf16)lo[r]; v[4 + r] = (bf16)hi[r]; }
    return v;
}

__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

template <int NT>
__global__ void __launch_bounds__(64) attn_fwd_mfma_kernel(const bf16* qkv, const float* key_mask, bf16* ctx, float* lse, int B, int L,
                                                           int H, float drop_p, uint64_t seed, const uint64_t* seed_ptr) {
    constexpr int NP = (NT + 1) / 2;
    __shared__ __attribute__((aligned(16))) unsigned char Vs[NP * 32 * RS];
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t stride = 3 * H * DH;
    const bf16* qb = qkv + (int64_t)b * L * stride + h * DH;
    const bf16* kb = qb + H * DH;
    const bf16* vb = qb + 2 * H * DH;
    stage(vb, stride, L, NP * 32, Vs, lane);

    bf16x8 qf[NT][2], kf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[t][ks] = ld_frag(qb, stride, 16 * t + c, L, 32 * ks + 8 * g);
            kf[t][ks] = ld_frag(kb, stride, 16 * t + c, L, 32 * ks + 8 * g);
        }
    float madd[NT][4];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * g + r;
            madd[jt][r] = j < L ? (1.0f - key_mask[(int64_t)b * L + j]) * MASK_NEG : NEG_BIG;
        }
    __syncthreads();
    bf16x8 vf[NP][4];
#pragma unroll
    for (int jp = 0; jp < NP; ++jp)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vf[jp][dt] = tr_frag(Vs, jp, dt, lane);

#pragma unroll
    for (int it = 0; it < NT; ++it) {
        const int i = 16 * it + c;                      // this lane's query (accumulator column)
        f32x4 s[NP * 2];
        float m = NEG_BIG;
#pragma unroll
        for (int jt = 0; jt < NP * 2; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (jt < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][ks], qf[it][ks], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[r] = acc[r] * 0.125f + madd[jt][r];
                    m = fmaxf(m, acc[r]);
                }
            }
            s[jt] = acc;
        }
        m = group_max(m);
        float l = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[jt][r] = __expf(s[jt][r] - m);
                l += s[jt][r];
            }
        l = group_sum(l);
        const float inv = 1.0f / l;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = s[jt][r] * inv;
                if (drop_p > 0.f) p *= dropout_mult(seed, ((uint64_t)bh * L + i) * L + 16 * jt + 4 * g + r, drop_p);
                s[jt][r] = p;
            }
        if (lse && g == 0 && i < L) lse[(int64_t)bh * L + i] = m + __logf(l);
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            const bf16x8 pf = pack_pair(s[2 * jp], s[2 * jp + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[jp][dt], pf, o[dt], 0, 0, 0);
        }
        if (i < L) {
            bf16* dst = ctx + ((int64_t)b * L + i) * (H * DH) + h * DH + 16 * g;
            bf16x8 lo = pack_pair(o[0], o[1]), hi = pack_pair(o[2], o[3]);
            *reinterpret_cast<bf16x8*>(dst) = lo;
            *reinterpret_cast<bf16x8*>(dst + 8) = hi;
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(64) attn_bwd_mfma_kernel(const bf16* qkv, const float* key_mask, const bf16* ctx, const bf16* dctx,
                                                           const float* lse, bf16* dqkv, int B, int L, int H, float drop_p,
                                                           uint64_t seed, const uint64_t* seed_ptr) {
    constexpr int NP = (NT + 1) / 2;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[NP * 32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char Qs[NP * 32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[NP * 32 * RS];
    __shared__ __attribute__((aligned(16))) float Dl[NP * 32];
    __shared__ __attribute__((aligned(16))) float Ll[NP * 32];
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t stride = 3 * H * DH, cstride = (int64_t)H * DH;
    const bf16* qb = qkv + (int64_t)b * L * stride + h * DH;
    const bf16* kb = qb + H * DH;
    const bf16* vb = qb + 2 * H * DH;
    const bf16* ob = ctx + (int64_t)b * L * cstride + h * DH;
    const bf16* gb = dctx + (int64_t)b * L * cstride + h * DH;
    bf16* dqb = dqkv + (int64_t)b * L * stride + h * DH;
    stage(kb, stride, L, NP * 32, Ks, lane);
    stage(qb, stride, L, NP * 32, Qs, lane);
    stage(gb, cstride, L, NP * 32, Gs, lane);

    bf16x8 qf[NT][2], kf[NT][2], vf[NT][2], gf[NT][2];
    float Dv[NT], Lv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int d = 32 * ks + 8 * g;
            qf[t][ks] = ld_frag(qb, stride, 16 * t + c, L, d);
            kf[t][ks] = ld_frag(kb, stride, 16 * t + c, L, d);
            vf[t][ks] = ld_frag(vb, stride, 16 * t + c, L, d);
            gf[t][ks] = ld_frag(gb, cstride, 16 * t + c, L, d);
            const bf16x8 of = ld_frag(ob, cstride, 16 * t + c, L, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) part += (float)gf[t][ks][e] * (float)of[e];
        }
        Dv[t] = group_sum(part);                                        // D_i = rowsum(dO * O), i = 16t + c
        Lv[t] = (16 * t + c < L) ? lse[(int64_t)bh * L + 16 * t + c] : 0.f;
        if (g == 0) { Dl[16 * t + c] = Dv[t]; Ll[16 * t + c] = Lv[t]; }
    }
    if (NT & 1) {                                                       // zero the padding tile of the last pair
        if (g == 0) { Dl[16 * NT + c] = 0.f; Ll[16 * NT + c] = 0.f; }
    }
    __syncthreads();

    // ---- orientation X: lane = query i, k-slots = keys j  ->  dQ -----------------------------------------
    {
        float madd[NT][4];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * g + r;
                madd[jt][r] = j < L ? (1.0f - key_mask[(int64_t)b * L + j]) * MASK_NEG : NEG_BIG;
            }
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int i = 16 * it + c;
            f32x4 ds[NP * 2];
#pragma unroll
            for (int jt = 0; jt < NP * 2; ++jt) {
                f32x4 sT = {0.f, 0.f, 0.f, 0.f}, dpT = {0.f, 0.f, 0.f, 0.f};
                if (jt < NT) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        sT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][ks], qf[it][ks], sT, 0, 0, 0);
                        dpT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[jt][ks], gf[it][ks], dpT, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __expf(sT[r] * 0.125f + madd[jt][r] - Lv[it]);
                        float dp = dpT[r];
                        if (drop_p > 0.f) dp *= dropout_mult(seed, ((uint64_t)bh * L + i) * L + 16 * jt + 4 * g + r, drop_p);
                        sT[r] = p * (dp - Dv[it]) * 0.125f;
                    }
                }
                ds[jt] = sT;
            }
            f32x4 dq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) {
                const bf16x8 dsf = pack_pair(ds[2 * jp], ds[2 * jp + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Ks, jp, dt, lane), dsf, dq[dt], 0, 0, 0);
            }
            if (i < L) {
                bf16* dst = dqb + (int64_t)i * stride + 16 * g;
                *reinterpret_cast<bf16x8*>(dst) = pack_pair(dq[0], dq[1]);
                *reinterpret_cast<bf16x8*>(dst + 8) = pack_pair(dq[2], dq[3]);
            }
        }
    }

    // ---- orientation Y: lane = key j, k-slots = queries i  ->  dK, dV -----------------------------------
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const int j = 16 * jt + c;
        const float madd = j < L ? (1.0f - key_mask[(int64_t)b * L + j]) * MASK_NEG : NEG_BIG;
        f32x4 pd[NP * 2], ds[NP * 2];
#pragma unroll
        for (int it = 0; it < NP * 2; ++it) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            f32x4 pdv = {0.f, 0.f, 0.f, 0.f};
            if (it < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[it][ks], kf[jt][ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[it][ks], vf[jt][ks], dp, 0, 0, 0);
                }
                const f32x4 li = *reinterpret_cast<const f32x4*>(&Ll[16 * it + 4 * g]);
                const f32x4 di = *reinterpret_cast<const f32x4*>(&Dl[16 * it + 4 * g]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + 4 * g + r;
                    const float p = i < L ? __expf(s[r] * 0.125f + madd - li[r]) : 0.f;
                    float mult = 1.0f;
                    if (drop_p > 0.f) mult = dropout_mult(seed, ((uint64_t)bh * L + i) * L + j, drop_p);
                    pdv[r] = p * mult;
                    s[r] = p * (dp[r] * mult - di[r]) * 0.125f;
                }
            }
            pd[it] = pdv;
            ds[it] = s;
        }
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ip = 0; ip < NP; ++ip) {
            const bf16x8 pdf = pack_pair(pd[2 * ip], pd[2 * ip + 1]);
            const bf16x8 dsf = pack_pair(ds[2 * ip], ds[2 * ip + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Gs, ip, dt, lane), pdf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Qs, ip, dt, lane), dsf, dk[dt], 0, 0, 0);
            }
        }
        if (j < L) {
            bf16* dst = dqb + (int64_t)j * stride + H * DH + 16 * g;
            *reinterpret_cast<bf16x8*>(dst) = pack_pair(dk[0], dk[1]);
            *reinterpret_cast<bf16x8*>(dst + 8) = pack_pair(dk[2], dk[3]);
            dst += H * DH;
            *reinterpret_cast<bf16x8*>(dst) = pack_pair(dv[0], dv[1]);
            *reinterpret_cast<bf16x8*>(dst + 8) = pack_pair(dv[2], dv[3]);
        }
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Used by cb_attention_fwd / cb_attention_bwd (attention.hip) for bf16, L <= 64, 16-byte aligned operands.
bool cb_attention_mfma_ok(int32_t dtype, const void* qkv, const void* ctx, const void* other, int32_t L) {
    return dtype == CB_BF16 && L <= 64 && al16(qkv) && al16(ctx) && (!other || al16(other));
}

int cb_attention_fwd_mfma(const void* qkv, const float* key_mask, void* ctx, float* lse, int32_t B, int32_t L, int32_t H,
                          float p, uint64_t seed, const uint64_t* seed_ptr, hipStream_t st) {
    dim3 g(B * H), b(64);
    const bf16* q = (const bf16*)qkv;
    bf16* c = (bf16*)ctx;
    switch ((L + 15) / 16) {
        case 1: hipLaunchKernelGGL((attn_fwd_mfma_kernel<1>), g, b, 0, st, q, key_mask, c, lse, B, L, H, p, seed, seed_ptr); break;
        case 2: hipLaunchKernelGGL((attn_fwd_mfma_kernel<2>), g, b, 0, st, q, key_mask, c, lse, B, L, H, p, seed, seed_ptr); break;
        case 3: hipLaunchKernelGGL((attn_fwd_mfma_kernel<3>), g, b, 0, st, q, key_mask, c, lse, B, L, H, p, seed, seed_ptr); break;
        default: hipLaunchKernelGGL((attn_fwd_mfma_kernel<4>), g, b, 0, st, q, key_mask, c, lse, B, L, H, p, seed, seed_ptr); break;
    }
    return cb_launch_status("cb_attention_fwd");
}

int cb_attention_bwd_mfma(const void* qkv, const float* key_mask, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                          int32_t B, int32_t L, int32_t H, float p, uint64_t seed, const uint64_t* seed_ptr, hipStream_t st) {
    dim3 g(B * H), b(64);
    const bf16 *q = (const bf16*)qkv, *c = (const bf16*)ctx, *d = (const bf16*)dctx;
    bf16* o = (bf16*)dqkv;
    switch ((L + 15) / 16) {
        case 1: hipLaunchKernelGGL((attn_bwd_mfma_kernel<1>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
        case 2: hipLaunchKernelGGL((attn_bwd_mfma_kernel<2>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
        case 3: hipLaunchKernelGGL((attn_bwd_mfma_kernel<3>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
        default: hipLaunchKernelGGL((attn_bwd_mfma_kernel<4>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
    }
    return cb_launch_status("cb_attention_bwd");
}
