// MFMA self-attention for short sequences (L <= 192, head size 64, bf16; the register-resident kernels below serve L <= 64,
// the LDS-resident *_big kernels further down 64 < L <= 192): one block of ceil(L/16) waves owns one
// (batch, head); each wave owns 16 queries (and, in the backward's second half, 16 keys).
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
// (1-mask)*-10000, softmax, inverted dropout on the probabilities (same mask stream: row (b*H+h)*L+i, column j; common.h).
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

// Natural [row][64] images of NMAT strided matrices in LDS, rows [L, ROWS) zero-filled (whole block), in two phases: ALL global
// loads of the block are issued before the first LDS write, so the staging costs one memory round trip instead of one per
// matrix and loop iteration (the kernels are latency-bound: a head is a few dozen MFMAs).
template <int NMAT, int ROWS, int NTHR>
__device__ __forceinline__ void stage_all(const bf16* const (&base)[NMAT], const int64_t (&stride)[NMAT], unsigned char* const (&dst)[NMAT],
                                          int L, int tid) {
    constexpr int ITER = (ROWS * 8 + NTHR - 1) / NTHR;
    u32x4 v[NMAT][ITER];
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * NTHR, r = idx >> 3, seg = idx & 7;
            u32x4 z = {0u, 0u, 0u, 0u};
            v[m][it] = (idx < ROWS * 8 && r < L) ? *reinterpret_cast<const u32x4*>(base[m] + (int64_t)r * stride[m] + seg * 8) : z;
        }
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * NTHR, r = idx >> 3, seg = idx & 7;
            if (idx < ROWS * 8) *reinterpret_cast<u32x4*>(dst[m] + r * RS + seg * 16) = v[m][it];
        }
}

// k-contiguous MFMA fragment (row `row`, head dims d..d+7) of a staged [row][64] image: one conflict-free ds_read_b128
__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* mat, int row, int d) {
    return *reinterpret_cast<const bf16x8*>(mat + row * RS + d * 2);
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
__global__ void __launch_bounds__(64 * NT) attn_fwd_mfma_kernel(const bf16* qkv, const float* key_mask, bf16* ctx, float* lse, int B, int L,
                                                           int H, float drop_p, uint64_t seed, const uint64_t* seed_ptr) {
    constexpr int NP = (NT + 1) / 2;
    __shared__ __attribute__((aligned(16))) unsigned char Vs[NP * 32 * RS];
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int it = threadIdx.x >> 6;                    // this wave's 16-query tile
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t stride = 3 * H * DH;
    const bf16* qb = qkv + (int64_t)b * L * stride + h * DH;
    const bf16* kb = qb + H * DH;
    const bf16* vb = qb + 2 * H * DH;

    bf16x8 qf[2], kf[NT][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = ld_frag(qb, stride, 16 * it + c, L, 32 * ks + 8 * g);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[t][ks] = ld_frag(kb, stride, 16 * t + c, L, 32 * ks + 8 * g);
    {
        const bf16* const bases[1] = {vb};
        const int64_t strides[1] = {stride};
        unsigned char* const dsts[1] = {Vs};
        stage_all<1, NP * 32, 64 * NT>(bases, strides, dsts, L, threadIdx.x);
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

    {
        const int i = 16 * it + c;                      // this lane's query (accumulator column)
        f32x4 s[NP * 2];
        float m = NEG_BIG;
#pragma unroll
        for (int jt = 0; jt < NP * 2; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (jt < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][ks], qf[ks], acc, 0, 0, 0);
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
        for (int jt = 0; jt < NT; ++jt) {
            s[jt] = s[jt] * inv;
            // keys 16jt + 4g + 0..3 are one 4-element group of the row's mask: one hash per accumulator register quad
            if (drop_p > 0.f) s[jt] = s[jt] * dropout_mult4(seed, ((uint64_t)bh * L + i) * ((L + 3) >> 2) + 4 * jt + g, drop_p);
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

// Backward: a block of 2*NT waves per (batch, head).  Waves [0, NT) ("X") own 16 queries each and produce dQ; waves
// [NT, 2NT) ("Y") own 16 keys each and produce dK, dV.  Both halves are the same computation with the roles of the
// matrices swapped:  tiles(rows of PA) x own columns of OA,
//   X: PA = K, PB = V, OA = Q, OB = dO  ->  S^T = K.Q^T,  dP^T = V.dO^T,  dQ^T = K^T.dS^T
//   Y: PA = Q, PB = dO, OA = K, OB = V  ->  S   = Q.K^T,  dP   = dO.V^T,  dK^T = Q^T.dS,  dV^T = dO^T.P_drop
// Occupancy: the cross-modal batches give ~3 (batch, head) blocks per CU (64 pairs x 12 heads = 768 blocks); with NT = 3 a block is 6
// waves, so all of them are resident at once only if 18 waves fit a CU (5 per SIMD: <= 96 registers).  The tiles' row operands
// (K / V for the query-owning waves, Q / dO for the key-owning ones) are therefore read from the staged LDS images where they are
// used instead of being preloaded from global memory into 16 * NT registers (104 registers: two blocks per CU, two rounds).
template <int NT>
__global__ void __launch_bounds__(128 * NT, (NT == 3 ? 5 : 1)) attn_bwd_mfma_kernel(const bf16* qkv, const float* key_mask, const bf16* ctx, const bf16* dctx,
                                                            const float* lse, bf16* dqkv, int B, int L, int H, float drop_p,
                                                            uint64_t seed, const uint64_t* seed_ptr) {
    constexpr int NP = (NT + 1) / 2;
    constexpr int NTHR = 128 * NT;
    constexpr float POS_BIG = 3.0e38f;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[NP * 32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char Qs[NP * 32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[NP * 32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[NP * 32 * RS];
    __shared__ __attribute__((aligned(16))) float Dl[NP * 32];      // D_i = rowsum(dO * O)
    __shared__ __attribute__((aligned(16))) float Ll[NP * 32];      // lse_i (+big past L: probabilities of padding rows = 0)
    __shared__ __attribute__((aligned(16))) float Ml[NP * 32];      // additive key mask (requested with everything else, BEFORE the barrier:
                                                                    // round 6 -- it used to be a dependent global round trip behind it)
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int wave = threadIdx.x >> 6;
    const bool yph = wave >= NT;
    const int wt = yph ? wave - NT : wave;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t stride = 3 * H * DH, cstride = (int64_t)H * DH;
    const bf16* qb = qkv + (int64_t)b * L * stride + h * DH;
    const bf16* kb = qb + H * DH;
    const bf16* vb = qb + 2 * H * DH;
    const bf16* ob = ctx + (int64_t)b * L * cstride + h * DH;
    const bf16* gb = dctx + (int64_t)b * L * cstride + h * DH;
    bf16* dqb = dqkv + (int64_t)b * L * stride + h * DH;
    const bf16* oa = yph ? kb : qb;
    const bf16* oo = yph ? vb : gb;
    const int64_t soo = yph ? stride : cstride;
    const unsigned char* PA = yph ? Qs : Ks;                        // LDS images of the tiles' row operands
    const unsigned char* PB = yph ? Gs : Vs;
    const int col = 16 * wt + c;                                    // own query (X) / key (Y)
    bf16x8 oaf[2], oof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int d = 32 * ks + 8 * g;
        oaf[ks] = ld_frag(oa, stride, col, L, d);
        oof[ks] = ld_frag(oo, soo, col, L, d);
    }
    bf16x8 ofr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) ofr[ks] = ld_frag(ob, cstride, col, L, 32 * ks + 8 * g);     // (X waves use it below)
    {   // all global loads of the block are in flight before the first wait
        const bf16* const bases[4] = {kb, qb, gb, vb};
        const int64_t strides[4] = {stride, stride, cstride, stride};
        unsigned char* const dsts[4] = {Ks, Qs, Gs, Vs};
        stage_all<4, NP * 32, NTHR>(bases, strides, dsts, L, threadIdx.x);
    }
    if (threadIdx.x < NP * 32) Ml[threadIdx.x] = (int)threadIdx.x < L ? (1.0f - key_mask[(int64_t)(blockIdx.x / H) * L + threadIdx.x]) * MASK_NEG : NEG_BIG;
    if (!yph) {
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 of = ofr[ks];
#pragma unroll
            for (int e = 0; e < 8; ++e) part += (float)oof[ks][e] * (float)of[e];
        }
        part = group_sum(part);
        if (g == 0) {
            Dl[col] = part;
            Ll[col] = col < L ? lse[(int64_t)bh * L + col] : POS_BIG;
        }
        if ((NT & 1) && wt == 0 && g == 1) { Dl[16 * NT + c] = 0.f; Ll[16 * NT + c] = POS_BIG; }   // padding tile of the last pair
    }
    __syncthreads();

    const float cmadd = Ml[col];                                                                     // Y: own key
    const float clse = Ll[col], cD = Dl[col];                                                        // X: own query
    f32x4 pd[NP * 2], ds[NP * 2];
#pragma unroll
    for (int t = 0; t < NP * 2; ++t) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f}, pdv = {0.f, 0.f, 0.f, 0.f};
        if (t < NT) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(PA, 16 * t + c, 32 * ks + 8 * g), oaf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(PB, 16 * t + c, 32 * ks + 8 * g), oof[ks], dp, 0, 0, 0);
            }
            const int row0 = 16 * t + 4 * g;                        // rows row0..row0+3: keys (X) / queries (Y)
            f32x4 rmadd, rlse, rD;
            if (yph) {
                rlse = *reinterpret_cast<const f32x4*>(&Ll[row0]);
                rD = *reinterpret_cast<const f32x4*>(&Dl[row0]);
                rmadd = f32x4{cmadd, cmadd, cmadd, cmadd};
            } else {
                rmadd = *reinterpret_cast<const f32x4*>(&Ml[row0]);
                rlse = f32x4{clse, clse, clse, clse};
                rD = f32x4{cD, cD, cD, cD};
            }
            f32x4 mult = {1.0f, 1.0f, 1.0f, 1.0f};
            if (drop_p > 0.f) {
                const uint64_t ng = (uint64_t)((L + 3) >> 2);
                if (yph) {                                   // 4 queries x own key: 4 rows of the mask, one element each
#pragma unroll
                    for (int r = 0; r < 4; ++r) mult[r] = dropout_mult1(seed, ((uint64_t)bh * L + row0 + r) * ng + (col >> 2), col & 3, drop_p);
                } else {                                     // own query x keys row0..row0+3: one group of the mask row
                    mult = dropout_mult4(seed, ((uint64_t)bh * L + col) * ng + (row0 >> 2), drop_p);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[r] * 0.125f + rmadd[r] - rlse[r]);
                pdv[r] = p * mult[r];
                s[r] = p * (dp[r] * mult[r] - rD[r]) * 0.125f;
            }
        }
        pd[t] = pdv;
        ds[t] = s;
    }
    const unsigned char* tra = yph ? Qs : Ks;
    f32x4 a1[4], a2[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { a1[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; a2[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
        const bf16x8 dsf = pack_pair(ds[2 * pr], ds[2 * pr + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) a1[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(tra, pr, dt, lane), dsf, a1[dt], 0, 0, 0);
        if (yph) {
            const bf16x8 pdf = pack_pair(pd[2 * pr], pd[2 * pr + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) a2[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Gs, pr, dt, lane), pdf, a2[dt], 0, 0, 0);
        }
    }
    if (col < L) {
        bf16* dst = dqb + (int64_t)col * stride + (yph ? H * DH : 0) + 16 * g;      // dQ third (X) / dK third (Y)
        *reinterpret_cast<bf16x8*>(dst) = pack_pair(a1[0], a1[1]);
        *reinterpret_cast<bf16x8*>(dst + 8) = pack_pair(a1[2], a1[3]);
        if (yph) {
            dst += H * DH;                                                          // dV third
            *reinterpret_cast<bf16x8*>(dst) = pack_pair(a2[0], a2[1]);
            *reinterpret_cast<bf16x8*>(dst + 8) = pack_pair(a2[2], a2[3]);
        }
    }
}

// =================================================================================================================
// 64 < L <= 192 (the reference's native sizes: 448 px + 20 tokens -> L = 69, 768 px + 25 tokens -> L = 169): the same
// algorithm with NT = ceil(L/16) rounded up to even <= 12 key tiles.  A whole head still fits on chip (K, V: 27 KiB each at
// NT = 12), so softmax stays single-pass -- no running max / rescale -- but the operand fragments no longer fit in
// registers next to NT score tiles: K, V (forward) and Q, K, V, dO (backward) are staged once in LDS in their natural
// [row][64] image (144-byte rows: conflict-free ds_read_b128 of a fragment) and every fragment is read where it is used,
// k-contiguous ones with ds_read_b128, row-contracted ones with ds_read_b64_tr_b16 (tr_frag above).  One wave per 16
// queries; the backward runs its query-owning (dQ) and key-owning (dK, dV) halves one after the other on the same NT waves
// (2*NT waves would exceed 1024 threads).
// =================================================================================================================

template <int NT>
__global__ void __launch_bounds__(64 * NT) attn_fwd_mfma_big_kernel(const bf16* qkv, const float* key_mask, bf16* ctx, float* lse, int B, int L,
                                                               int H, float drop_p, uint64_t seed, const uint64_t* seed_ptr) {
    static_assert(NT % 2 == 0, "key tiles are consumed in pairs");
    constexpr int NP = NT / 2, ROWS = NT * 16;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[ROWS * RS];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[ROWS * RS];
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int it = threadIdx.x >> 6;                    // this wave's 16-query tile
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t stride = 3 * H * DH;
    const bf16* qb = qkv + (int64_t)b * L * stride + h * DH;
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = ld_frag(qb, stride, 16 * it + c, L, 32 * ks + 8 * g);
    {
        const bf16* const bases[2] = {qb + H * DH, qb + 2 * H * DH};
        const int64_t strides[2] = {stride, stride};
        unsigned char* const dsts[2] = {Ks, Vs};
        stage_all<2, ROWS, 64 * NT>(bases, strides, dsts, L, threadIdx.x);
    }
    __syncthreads();
    if (16 * it >= L) return;                           // (no barrier below)
    const int i = 16 * it + c;                          // this lane's query (accumulator column)
    f32x4 s[NT];
    float m = NEG_BIG;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(Ks, 16 * jt + c, 32 * ks + 8 * g), qf[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * g + r;
            const float madd = j < L ? (1.0f - key_mask[(int64_t)b * L + j]) * MASK_NEG : NEG_BIG;
            acc[r] = acc[r] * 0.125f + madd;
            m = fmaxf(m, acc[r]);
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
    for (int jt = 0; jt < NT; ++jt) {
        s[jt] = s[jt] * inv;
        if (drop_p > 0.f) s[jt] = s[jt] * dropout_mult4(seed, ((uint64_t)bh * L + i) * ((L + 3) >> 2) + 4 * jt + g, drop_p);
    }
    if (lse && g == 0 && i < L) lse[(int64_t)bh * L + i] = m + __logf(l);
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
        const bf16x8 pf = pack_pair(s[2 * jp], s[2 * jp + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Vs, jp, dt, lane), pf, o[dt], 0, 0, 0);
    }
    if (i < L) {
        bf16* dst = ctx + ((int64_t)b * L + i) * (H * DH) + h * DH + 16 * g;
        *reinterpret_cast<bf16x8*>(dst) = pack_pair(o[0], o[1]);
        *reinterpret_cast<bf16x8*>(dst + 8) = pack_pair(o[2], o[3]);
    }
}

// Backward, 64 < L <= 192: phase X (own 16 queries -> dQ) then phase Y (own 16 keys -> dK, dV) on the same wave; see the
// role table above attn_bwd_mfma_kernel.
template <int NT>
__global__ void __launch_bounds__(64 * NT) attn_bwd_mfma_big_kernel(const bf16* qkv, const float* key_mask, const bf16* ctx, const bf16* dctx,
                                                               const float* lse, bf16* dqkv, int B, int L, int H, float drop_p,
                                                               uint64_t seed, const uint64_t* seed_ptr) {
    static_assert(NT % 2 == 0, "tiles are consumed in pairs");
    constexpr int NP = NT / 2, ROWS = NT * 16, NTHR = 64 * NT;
    constexpr float POS_BIG = 3.0e38f;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[ROWS * RS];
    __shared__ __attribute__((aligned(16))) unsigned char Qs[ROWS * RS];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[ROWS * RS];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[ROWS * RS];
    __shared__ __attribute__((aligned(16))) float Dl[ROWS];      // D_i = rowsum(dO * O)
    __shared__ __attribute__((aligned(16))) float Ll[ROWS];      // lse_i (+big past L: probabilities of padding rows = 0)
    __shared__ __attribute__((aligned(16))) float Ml[ROWS];      // additive key mask (-big past L)
    if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int wt = threadIdx.x >> 6;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t stride = 3 * H * DH, cstride = (int64_t)H * DH;
    const bf16* qb = qkv + (int64_t)b * L * stride + h * DH;
    const bf16* ob = ctx + (int64_t)b * L * cstride + h * DH;
    const bf16* gb = dctx + (int64_t)b * L * cstride + h * DH;
    bf16* dqb = dqkv + (int64_t)b * L * stride + h * DH;
    const int col = 16 * wt + c;                                    // own query (X) / key (Y)
    {
        bf16x8 ofr[2], gfr[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { ofr[ks] = ld_frag(ob, cstride, col, L, 32 * ks + 8 * g); gfr[ks] = ld_frag(gb, cstride, col, L, 32 * ks + 8 * g); }
        {   // all global loads of the block are in flight before the first wait
            const bf16* const bases[4] = {qb, qb + H * DH, qb + 2 * H * DH, gb};
            const int64_t strides[4] = {stride, stride, stride, cstride};
            unsigned char* const dsts[4] = {Qs, Ks, Vs, Gs};
            stage_all<4, ROWS, NTHR>(bases, strides, dsts, L, threadIdx.x);
        }
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 of = ofr[ks], gf = gfr[ks];
#pragma unroll
            for (int e = 0; e < 8; ++e) part += (float)gf[e] * (float)of[e];
        }
        part = group_sum(part);
        if (g == 0) {
            Dl[col] = part;
            Ll[col] = col < L ? lse[(int64_t)bh * L + col] : POS_BIG;
            Ml[col] = col < L ? (1.0f - key_mask[(int64_t)b * L + col]) * MASK_NEG : NEG_BIG;
        }
    }
    __syncthreads();
    if (16 * wt >= L) return;                                       // (no barrier below)
    const uint64_t ng = (uint64_t)((L + 3) >> 2);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        const bool yph = ph == 1;
        const unsigned char* PA = yph ? Qs : Ks;                    // tiles' rows
        const unsigned char* PB = yph ? Gs : Vs;
        const unsigned char* OA = yph ? Ks : Qs;                    // own column
        const unsigned char* OO = yph ? Vs : Gs;
        bf16x8 oaf[2], oof[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { oaf[ks] = lds_frag(OA, col, 32 * ks + 8 * g); oof[ks] = lds_frag(OO, col, 32 * ks + 8 * g); }
        const float cmadd = Ml[col], clse = Ll[col], cD = Dl[col];
        f32x4 pd[NT], ds[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(PA, 16 * t + c, 32 * ks + 8 * g), oaf[ks], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(PB, 16 * t + c, 32 * ks + 8 * g), oof[ks], dp, 0, 0, 0);
            }
            const int row0 = 16 * t + 4 * g;                        // rows row0..row0+3: keys (X) / queries (Y)
            f32x4 rmadd, rlse, rD;
            if (yph) {
                rlse = *reinterpret_cast<const f32x4*>(&Ll[row0]);
                rD = *reinterpret_cast<const f32x4*>(&Dl[row0]);
                rmadd = f32x4{cmadd, cmadd, cmadd, cmadd};
            } else {
                rmadd = *reinterpret_cast<const f32x4*>(&Ml[row0]);
                rlse = f32x4{clse, clse, clse, clse};
                rD = f32x4{cD, cD, cD, cD};
            }
            f32x4 mult = {1.0f, 1.0f, 1.0f, 1.0f};
            if (drop_p > 0.f) {
                if (yph) {                                   // 4 queries x own key: 4 rows of the mask, one element each
#pragma unroll
                    for (int r = 0; r < 4; ++r) mult[r] = dropout_mult1(seed, ((uint64_t)bh * L + row0 + r) * ng + (col >> 2), col & 3, drop_p);
                } else {                                     // own query x keys row0..row0+3: one group of the mask row
                    mult = dropout_mult4(seed, ((uint64_t)bh * L + col) * ng + (row0 >> 2), drop_p);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(sc[r] * 0.125f + rmadd[r] - rlse[r]);
                pd[t][r] = p * mult[r];
                ds[t][r] = p * (dp[r] * mult[r] - rD[r]) * 0.125f;
            }
        }
        f32x4 a1[4], a2[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { a1[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; a2[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            const bf16x8 dsf = pack_pair(ds[2 * pr], ds[2 * pr + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) a1[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(PA, pr, dt, lane), dsf, a1[dt], 0, 0, 0);
            if (yph) {
                const bf16x8 pdf = pack_pair(pd[2 * pr], pd[2 * pr + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) a2[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Gs, pr, dt, lane), pdf, a2[dt], 0, 0, 0);
            }
        }
        if (col < L) {
            bf16* dst = dqb + (int64_t)col * stride + (yph ? H * DH : 0) + 16 * g;      // dQ third (X) / dK third (Y)
            *reinterpret_cast<bf16x8*>(dst) = pack_pair(a1[0], a1[1]);
            *reinterpret_cast<bf16x8*>(dst + 8) = pack_pair(a1[2], a1[3]);
            if (yph) {
                dst += H * DH;                                                          // dV third
                *reinterpret_cast<bf16x8*>(dst) = pack_pair(a2[0], a2[1]);
                *reinterpret_cast<bf16x8*>(dst + 8) = pack_pair(a2[2], a2[3]);
            }
        }
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Used by cb_attention_fwd / cb_attention_bwd (attention.hip) for bf16, L <= 192, 16-byte aligned operands.
bool cb_attention_mfma_ok(int32_t dtype, const void* qkv, const void* ctx, const void* other, int32_t L) {
    return dtype == CB_BF16 && L <= 192 && al16(qkv) && al16(ctx) && (!other || al16(other));
}

int cb_attention_fwd_mfma(const void* qkv, const float* key_mask, void* ctx, float* lse, int32_t B, int32_t L, int32_t H,
                          float p, uint64_t seed, const uint64_t* seed_ptr, hipStream_t st) {
    const int nt = (L + 15) / 16;
    dim3 g(B * H), b(64 * nt);
    const bf16* q = (const bf16*)qkv;
    bf16* c = (bf16*)ctx;
    if (nt > 4) {
        const int ne = (nt + 1) / 2 * 2;
        b = dim3(64 * ne);
        switch (ne) {
            case 6: hipLaunchKernelGGL((attn_fwd_mfma_big_kernel<6>), g, b, 0, st, q, key_mask, c, lse, B, L, H, p, seed, seed_ptr); break;
            case 8: hipLaunchKernelGGL((attn_fwd_mfma_big_kernel<8>), g, b, 0, st, q, key_mask, c, lse, B, L, H, p, seed, seed_ptr); break;
            case 10: hipLaunchKernelGGL((attn_fwd_mfma_big_kernel<10>), g, b, 0, st, q, key_mask, c, lse, B, L, H, p, seed, seed_ptr); break;
            default: hipLaunchKernelGGL((attn_fwd_mfma_big_kernel<12>), g, b, 0, st, q, key_mask, c, lse, B, L, H, p, seed, seed_ptr); break;
        }
        return cb_launch_status("cb_attention_fwd");
    }
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
    const int nt = (L + 15) / 16;
    dim3 g(B * H), b(128 * nt);
    const bf16 *q = (const bf16*)qkv, *c = (const bf16*)ctx, *d = (const bf16*)dctx;
    bf16* o = (bf16*)dqkv;
    if (nt > 4) {
        const int ne = (nt + 1) / 2 * 2;
        b = dim3(64 * ne);
        switch (ne) {
            case 6: hipLaunchKernelGGL((attn_bwd_mfma_big_kernel<6>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
            case 8: hipLaunchKernelGGL((attn_bwd_mfma_big_kernel<8>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
            case 10: hipLaunchKernelGGL((attn_bwd_mfma_big_kernel<10>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
            default: hipLaunchKernelGGL((attn_bwd_mfma_big_kernel<12>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
        }
        return cb_launch_status("cb_attention_bwd");
    }
    switch ((L + 15) / 16) {
        case 1: hipLaunchKernelGGL((attn_bwd_mfma_kernel<1>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
        case 2: hipLaunchKernelGGL((attn_bwd_mfma_kernel<2>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
        case 3: hipLaunchKernelGGL((attn_bwd_mfma_kernel<3>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
        default: hipLaunchKernelGGL((attn_bwd_mfma_kernel<4>), g, b, 0, st, q, key_mask, c, d, lse, o, B, L, H, p, seed, seed_ptr); break;
    }
    return cb_launch_status("cb_attention_bwd");
}
