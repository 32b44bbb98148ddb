// LayerNorm (fp32 statistics) forward/backward and the two embedding front-ends of the cross-modal
// BERT.  One 64-lane wave owns one row of D elements (D % 4 == 0, D <= 2048): the row lives in
// registers (4 elements per lane per chunk of 256), statistics are two-pass in fp32 as apex
// FusedLayerNorm does, and every global access is an 8/16-byte vector.  These kernels are HBM-bound.
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int MAXD = 2048;  // NCH chunks of 256 elements, NCH in {3, 4, 8}

// loads row elements (lane + 64*c)*4 .. +3
template <int MAXCH, typename T>
__device__ __forceinline__ void load_row(const T* p, int D, int lane, f32x4 (&v)[MAXCH]) {
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        int e = (lane + 64 * c) * 4;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v[c] = (e < D) ? load4(p + e) : z;
    }
}

template <int MAXCH>
__device__ __forceinline__ void row_stats(const f32x4 (&v)[MAXCH], int D, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        int e = (lane + 64 * c) * 4;
        if (e < D) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { float d = v[c][i] - mean; q += d * d; }
        }
    }
    rstd = rsqrtf(wave_sum(q) / (float)D + eps);
}

template <int MAXCH, typename T>
__device__ __forceinline__ void norm_store(const f32x4 (&v)[MAXCH], const float* gamma, const float* beta, T* y, int D,
                                           int lane, float mean, float rstd) {
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        int e = (lane + 64 * c) * 4;
        if (e < D) {
            f32x4 g = load4(gamma + e), b = load4(beta + e), o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (v[c][i] - mean) * rstd * g[i] + b[i];
            store4(y + e, o);
        }
    }
}

// FULL: D == MAXCH * 256 (the encoder's 768): no column guards, and gamma / beta are requested WITH the row (they used to be a second,
// dependent round trip behind the two reductions).
template <typename T, int MAXCH, bool FULL = false>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const T* x, const float* gamma, const float* beta, T* y,
                                                            float* mean_out, float* rstd_out, int64_t rows, int D, float eps,
                                                            int seg_len, int seg_stride, int seg_off) {
    const int lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (seg_len > 0) row = (row / seg_len) * seg_stride + seg_off + row % seg_len;
    f32x4 v[MAXCH];
    if constexpr (FULL) {
        f32x4 gm[MAXCH], bt[MAXCH];
        load_row(x + row * D, MAXCH * 256, lane, v);
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) { gm[c] = load4(gamma + (lane + 64 * c) * 4); bt[c] = load4(beta + (lane + 64 * c) * 4); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
        const float inv_d = 1.0f / (float)(MAXCH * 256);
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = v[c][i] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (v[c][i] - mean) * rstd * gm[c][i] + bt[c][i];
            store4(y + row * D + (lane + 64 * c) * 4, o);
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
        return;
    }
    load_row(x + row * D, D, lane, v);
    float mean, rstd;
    row_stats(v, D, lane, eps, mean, rstd);
    norm_store(v, gamma, beta, y + row * D, D, lane, mean, rstd);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// Each wave walks rows with stride gridDim*NW (next row prefetched); dgamma/dbeta partials stay in registers, are combined across the
// block's NW waves through LDS and flushed with one atomic (or one partial-sum store) per element per block.  Since round 6 only rows
// wider than 1024 elements (8 chunks: too many registers for the rows-in-flight kernel below) run it, with NW = 4.
template <typename T, int MAXCH, int NW>
__global__ void __launch_bounds__(NW * 64) layernorm_bwd_kernel(const T* dy, const T* x, const float* gamma, const float* mean,
                                                               const float* rstd, T* dx, float* dgamma, float* dbeta,
                                                               int64_t rows, int D, T* dx2, float drop_p, uint64_t seed,
                                                               const uint64_t* seed_ptr, int seg_len, int seg_stride, int seg_off,
                                                               float* part) {
    // part != nullptr: the block's sums go to part[blockIdx.x][gamma|beta][D] with plain stores (no atomics: the 2*D same-address
    // memory-side atomics of ~160 blocks are the kernel's tail) and cb_ln_partials_reduce adds them up later -- for all the
    // encoder's LayerNorms in one launch, and in a fixed order (deterministic parameter gradients).
    __shared__ float red[2][NW][256];          // [gamma|beta][wave][one 256-element chunk]
    if (dx2 && seed_ptr) seed += *seed_ptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 ag[MAXCH], ab[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; ag[c] = z; ab[c] = z; }
    const int64_t stride = (int64_t)gridDim.x * NW;
    auto phys = [&](int64_t lrow) { return seg_len > 0 ? (lrow / seg_len) * seg_stride + seg_off + lrow % seg_len : lrow; };
    int64_t lrow = (int64_t)blockIdx.x * NW + wave;
    f32x4 g[MAXCH], xv[MAXCH], gn[MAXCH], xn[MAXCH];
    float mu = 0.f, rs = 0.f, mun = 0.f, rsn = 0.f;
    int64_t row = 0, rown = 0;
    if (lrow < rows) {
        row = phys(lrow);
        load_row(dy + row * D, D, lane, g);
        load_row(x + row * D, D, lane, xv);
        mu = mean[row]; rs = rstd[row];
    }
    for (; lrow < rows; lrow += stride) {
        const bool more = lrow + stride < rows;
        if (more) {                               // software prefetch of the wave's next row
            rown = phys(lrow + stride);
            load_row(dy + rown * D, D, lane, gn);
            load_row(x + rown * D, D, lane, xn);
            mun = mean[rown]; rsn = rstd[rown];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            int e = (lane + 64 * c) * 4;
            if (e < D) {
                f32x4 gm = load4(gamma + e);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float xh = (xv[c][i] - mu) * rs;
                    float dg = g[c][i] * gm[i];
                    ag[c][i] += g[c][i] * xh;
                    ab[c][i] += g[c][i];
                    s1 += dg; s2 += dg * xh;
                    xv[c][i] = xh; g[c][i] = dg;
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            int e = (lane + 64 * c) * 4;
            if (e < D) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = rs * (g[c][i] - c1 - xv[c][i] * c2);
                store4(dx + row * D + e, o);
                if (dx2) {
                    o = o * dropout_mult4(seed, ((uint64_t)row * D + e) >> 2, drop_p);            // D % 4 == 0, e % 4 == 0
                    store4(dx2 + row * D + e, o);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int c = 0; c < MAXCH; ++c) { g[c] = gn[c]; xv[c] = xn[c]; }
            mu = mun; rs = rsn; row = rown;
        }
    }
    // block reduction of the parameter gradients, chunk by chunk (LDS: 2 x NW waves x 256 floats)
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        if (c * 256 >= D) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[0][wave][lane * 4 + i] = ag[c][i];
            red[1][wave][lane * 4 + i] = ab[c][i];
        }
        __syncthreads();
        if constexpr (NW * 64 >= 512) {                 // threads 0..511 <-> (kind, element): one atomic per element per block
            if (threadIdx.x < 512) {
                const int e = threadIdx.x & 255, kind = threadIdx.x >> 8;
                const int col = c * 256 + e;
                if (col < D) {
                    float sacc = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sacc += red[kind][w][e];
                    if (part) part[((int64_t)blockIdx.x * 2 + kind) * D + col] = sacc;
                    else atomicAdd((kind ? dbeta : dgamma) + col, sacc);
                }
            }
        } else {
            const int e = threadIdx.x, col = c * 256 + e;
            if (col < D) {
                float sg = 0.f, sb = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) { sg += red[0][w][e]; sb += red[1][w][e]; }
                if (part) {
                    part[((int64_t)blockIdx.x * 2 + 0) * D + col] = sg;
                    part[((int64_t)blockIdx.x * 2 + 1) * D + col] = sb;
                } else {
                    atomicAdd(dgamma + col, sg);
                    atomicAdd(dbeta + col, sb);
                }
            }
        }
    }
}

// Round 6: the same backward with EVERY row of a wave in flight at once.  The kernel above gives a wave one row at a time (next row
// prefetched): at the encoder's 2624 rows that is one row per wave on 164 sixteen-wave blocks -- 12.2 us in the step against 5.8 us for the
// forward (profiles/r05z_train_step.md): a launch of one dependent chain {loads -> two 64-lane reductions -> stores -> a 16-way LDS
// reduction of 2 * D sums behind six barriers} with ten waves per CU to hide it.  Here a block is NW waves x RPW rows: a wave requests the
// dy / x rows of all its RPW rows up front (RPW x the bytes in flight per wave), forms the 2 * RPW row sums and reduces them TOGETHER
// (independent shuffles interleave), stores, and keeps the column sums of its rows in registers -- the cross-wave reduction is NW-way
// (4 instead of 16) behind ONE barrier pair per 1024 columns.  Same arithmetic per element and the same order of the column additions
// inside a wave's rows; the partial-sum layout part[block][gamma | beta][D] and cb_ln_partials_reduce are unchanged.
// FULL: D == MAXCH * 256 (the encoder's 768): no column guards anywhere -- every `e < D` test is an exec-mask branch per chunk otherwise.
template <typename T, int MAXCH, int NW, int RPW, bool FULL>
__global__ void __launch_bounds__(NW * 64) layernorm_bwd_rows_kernel(const T* dy, const T* x, const float* gamma, const float* mean,
                                                                    const float* rstd, T* dx, float* dgamma, float* dbeta,
                                                                    int64_t rows, int D, T* dx2, float drop_p, uint64_t seed,
                                                                    const uint64_t* seed_ptr, int seg_len, int seg_stride, int seg_off,
                                                                    float* part) {
    constexpr int GC = MAXCH < 4 ? MAXCH : 4;                     // chunks (of 256 columns) reduced across the waves per barrier pair
    __shared__ float red[2][NW][GC * 256];
    if (dx2 && seed_ptr) seed += *seed_ptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 ag[MAXCH], ab[MAXCH], gm[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ag[c] = z; ab[c] = z;
        const int e = (lane + 64 * c) * 4;
        gm[c] = (FULL || e < D) ? load4(gamma + e) : z;
    }
    auto phys = [&](int64_t lrow) { return seg_len > 0 ? (lrow / seg_len) * seg_stride + seg_off + lrow % seg_len : lrow; };
    constexpr int RPB = NW * RPW;
    const float inv_d = 1.0f / (float)D;
    for (int64_t base = (int64_t)blockIdx.x * RPB; base < rows; base += (int64_t)gridDim.x * RPB) {
        f32x4 g[RPW][MAXCH], xv[RPW][MAXCH];
        float mu[RPW], rs[RPW];
        int64_t row[RPW];
        bool ok[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {                           // wave w owns rows base + w, + NW, ...: a short last block stays spread
            const int64_t lrow = base + wave + NW * j;
            ok[j] = lrow < rows;
            row[j] = ok[j] ? phys(lrow) : 0;
            if (ok[j]) {
                load_row(dy + row[j] * D, FULL ? MAXCH * 256 : D, lane, g[j]);
                load_row(x + row[j] * D, FULL ? MAXCH * 256 : D, lane, xv[j]);
                mu[j] = mean[row[j]]; rs[j] = rstd[row[j]];
            } else {
#pragma unroll
                for (int c = 0; c < MAXCH; ++c) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; g[j][c] = z; xv[j][c] = z; }
                mu[j] = 0.f; rs[j] = 0.f;
            }
        }
        float s1[RPW], s2[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            s1[j] = 0.f; s2[j] = 0.f;
#pragma unroll
            for (int c = 0; c < MAXCH; ++c) {
                const int e = (lane + 64 * c) * 4;
                if (FULL || e < D) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xh = (xv[j][c][i] - mu[j]) * rs[j];
                        const float dgv = g[j][c][i] * gm[c][i];
                        ag[c][i] += g[j][c][i] * xh;
                        ab[c][i] += g[j][c][i];
                        s1[j] += dgv; s2[j] += dgv * xh;
                        xv[j][c][i] = xh; g[j][c][i] = dgv;
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {                        // the 2 * RPW reductions side by side
#pragma unroll
            for (int j = 0; j < RPW; ++j) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
        }
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            if (!ok[j]) continue;
            const float c1 = s1[j] * inv_d, c2 = s2[j] * inv_d;
#pragma unroll
            for (int c = 0; c < MAXCH; ++c) {
                const int e = (lane + 64 * c) * 4;
                if (FULL || e < D) {
                    f32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = rs[j] * (g[j][c][i] - c1 - xv[j][c][i] * c2);
                    store4(dx + row[j] * D + e, o);
                    if (dx2) {
                        o = o * dropout_mult4(seed, ((uint64_t)row[j] * D + e) >> 2, drop_p);
                        store4(dx2 + row[j] * D + e, o);
                    }
                }
            }
        }
    }
    // cross-wave reduction of the column sums: GC chunks per barrier pair, waves added in wave order (fixed: deterministic)
#pragma unroll
    for (int c0 = 0; c0 < MAXCH; c0 += GC) {
        if (c0 * 256 >= D) break;
        if (c0 > 0) __syncthreads();
#pragma unroll
        for (int c = 0; c < GC; ++c) {
            if (c0 + c < MAXCH) {
                *reinterpret_cast<f32x4*>(&red[0][wave][c * 256 + lane * 4]) = ag[c0 + c];
                *reinterpret_cast<f32x4*>(&red[1][wave][c * 256 + lane * 4]) = ab[c0 + c];
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 2 * GC * 256; idx += NW * 64) {
            const int kind = idx / (GC * 256), e = idx - kind * (GC * 256);
            const int col = c0 * 256 + e;
            if (col < D) {
                float sacc = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sacc += red[kind][w][e];
                if (part) part[((int64_t)blockIdx.x * 2 + kind) * D + col] = sacc;
                else atomicAdd((kind ? dbeta : dgamma) + col, sacc);
            }
        }
    }
}

// grad[off[kind][job] + col] += sum_b part[job][b][kind][col].  Block = (256-column chunk, kind, job), 16 waves: wave w sums the partial
// rows b = w, w + 16, ... (lane = 4 consecutive columns, 1 KiB per wave per row, every load of a wave's rows in flight together), the 16
// wave sums are then added in wave order through LDS -- a fixed order: deterministic.
__global__ void __launch_bounds__(1024) ln_partials_reduce_kernel(const float* part, float* grad, const int64_t* off_gamma,
                                                                  const int64_t* off_beta, int nblocks, int D) {
    __shared__ float red[16][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kind = blockIdx.y, job = blockIdx.z;
    const int col = blockIdx.x * 256 + lane * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < D) {                                   // D % 4 == 0
        const float* q = part + ((int64_t)job * nblocks * 2 + kind) * D + col;
        int b = wave;
        for (; b + 48 < nblocks; b += 64) {
            f32x4 v0 = load4(q + (int64_t)(b + 0) * 2 * D), v1 = load4(q + (int64_t)(b + 16) * 2 * D);
            f32x4 v2 = load4(q + (int64_t)(b + 32) * 2 * D), v3 = load4(q + (int64_t)(b + 48) * 2 * D);
            s = s + ((v0 + v1) + (v2 + v3));
        }
        for (; b < nblocks; b += 16) s = s + load4(q + (int64_t)b * 2 * D);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][lane * 4 + i] = s[i];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c < D) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) t += red[w][threadIdx.x];
            grad[(kind ? off_beta : off_gamma)[job] + c] += t;
        }
    }
}

// ---- embeddings -----------------------------------------------------------------------------------
template <typename T, int MAXCH>
__global__ void __launch_bounds__(256) text_embed_fwd_kernel(const int64_t* ids, const T* word, const T* pos, const T* type0,
                                                             const float* gamma, const float* beta, T* out, T* pre,
                                                             float* mean_out, float* rstd_out, int B, int Lt, int Ltot, int D,
                                                             float eps, const int64_t* attn_mask, float* key_mask, int period) {
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= (int64_t)B * Lt) return;
    const int b = (int)(tok / Lt), t = (int)(tok % Lt);
    const int64_t orow = (int64_t)b * Ltot + t;
    const int64_t irow = (int64_t)(b % period) * Lt + t;         // the text batch repeats with this period (folded clips)
    const int64_t id = ids[irow];
    if (key_mask && lane == 0) key_mask[orow] = attn_mask ? (float)attn_mask[irow] : 1.0f;
    f32x4 v[MAXCH], a[MAXCH];
    load_row(word + id * D, D, lane, v);
    load_row(pos + (int64_t)t * D, D, lane, a);
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) v[c] = v[c] + a[c];
    load_row(type0, D, lane, a);
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) v[c] = v[c] + a[c];
    if (pre) {
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) { int e = (lane + 64 * c) * 4; if (e < D) store4(pre + orow * D + e, v[c]); }
    }
    float mean, rstd;
    row_stats(v, D, lane, eps, mean, rstd);
    norm_store(v, gamma, beta, out + orow * D, D, lane, mean, rstd);
    if (lane == 0) {
        if (mean_out) mean_out[orow] = mean;
        if (rstd_out) rstd_out[orow] = rstd;
    }
}

template <typename T, int MAXCH>
__global__ void __launch_bounds__(256) visual_embed_fwd_kernel(const T* grid, const int32_t* src_row, const int32_t* sel,
                                                               const T* row_emb, const T* col_emb, const T* type0,
                                                               const float* gamma, const float* beta, T* out, T* pre,
                                                               float* mean_out, float* rstd_out, int B, int Tf, int Hg, int Wg,
                                                               int Lv, int Lt, int Ltot, int D, float eps, float* key_mask) {
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= (int64_t)B * Lv) return;
    const int b = (int)(tok / Lv), pidx = (int)(tok % Lv);
    const int q = sel ? sel[pidx] : pidx;
    const int h = q / Wg, w = q % Wg;
    const int64_t src = src_row ? src_row[b] : b;
    const int64_t orow = (int64_t)b * Ltot + Lt + pidx;
    if (key_mask && lane == 0) key_mask[orow] = 1.0f;            // visual tokens are never masked (modeling.py:217-220)
    f32x4 v[MAXCH], a[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; v[c] = z; }
    for (int t = 0; t < Tf; ++t) {
        load_row(grid + (((src * Tf + t) * Hg + h) * Wg + w) * D, D, lane, a);
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) v[c] = v[c] + a[c];
    }
    const float inv = 1.0f / (float)Tf;
    load_row(row_emb + (int64_t)h * D, D, lane, a);
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) v[c] = v[c] * inv + a[c];
    load_row(col_emb + (int64_t)w * D, D, lane, a);
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) v[c] = v[c] + a[c];
    load_row(type0, D, lane, a);
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) v[c] = v[c] + a[c];
    if (pre) {
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) { int e = (lane + 64 * c) * 4; if (e < D) store4(pre + orow * D + e, v[c]); }
    }
    float mean, rstd;
    row_stats(v, D, lane, eps, mean, rstd);
    norm_store(v, gamma, beta, out + orow * D, D, lane, mean, rstd);
    if (lane == 0) {
        if (mean_out) mean_out[orow] = mean;
        if (rstd_out) rstd_out[orow] = rstd;
    }
}

// Embedding-table gradients.  One launch, two kinds of blocks:
//  * "sum" blocks, one per sequence position: the position / token-type (/ grid row / column) gradients are sums over the batch -- the
//    block adds its B rows in registers (independent loads, one memory round trip) and reaches memory with ONE atomic per element
//    (the naive one-atomic-per-token form serialises B*L updates on the same few rows: 120 us for the 32x32-token bench batch);
//  * "scatter" blocks, one WAVE per token: the word-table row (/ the frames' grid cells) of that token takes the token's gradient
//    (distinct rows, little contention), B*L/4 blocks in flight instead of a loop over the batch inside L blocks.
// Round 6: the two roles used to share a block that walked a quarter of the batch in a dependent loop (30 / 21 us for the bench batch).
template <typename T>
__global__ void __launch_bounds__(256) text_embed_bwd_kernel(const T* dpre, const int64_t* ids, float* dword, float* dpos,
                                                             float* dtype0, int B, int Lt, int Ltot, int D, int64_t pad_id, int period) {
    if ((int)blockIdx.x < Lt) {                                      // ---- sum block of position t
        const int t = blockIdx.x;
        for (int e = threadIdx.x * 4; e < D; e += blockDim.x * 4) {
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int b = 0; b < B; ++b) sum = sum + load4(dpre + ((int64_t)b * Ltot + t) * D + e);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                atomicAdd(dpos + (int64_t)t * D + e + i, sum[i]);
                atomicAdd(dtype0 + e + i, sum[i]);
            }
        }
        return;
    }
    const int tok = ((int)blockIdx.x - Lt) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // ---- scatter: one wave per token
    if (tok >= B * Lt) return;
    const int b = tok / Lt, t = tok - b * Lt;
    const int64_t id = ids[(int64_t)(b % period) * Lt + t];
    if (id == pad_id) return;
    for (int e = lane * 4; e < D; e += 256) {
        const f32x4 g = load4(dpre + ((int64_t)b * Ltot + t) * D + e);
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(dword + id * D + e + i, g[i]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) visual_embed_bwd_kernel(const T* dpre, const int32_t* src_row, const int32_t* sel,
                                                               float* dgrid, float* drow, float* dcol, float* dtype0, int B,
                                                               int Tf, int Hg, int Wg, int Lv, int Lt, int Ltot, int D) {
    if ((int)blockIdx.x < Lv) {                                      // ---- sum block of visual position pidx
        const int pidx = blockIdx.x;
        const int q = sel ? sel[pidx] : pidx;
        const int h = q / Wg, w = q % Wg;
        for (int e = threadIdx.x * 4; e < D; e += blockDim.x * 4) {
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int b = 0; b < B; ++b) sum = sum + load4(dpre + ((int64_t)b * Ltot + Lt + pidx) * D + e);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                atomicAdd(drow + (int64_t)h * D + e + i, sum[i]);
                atomicAdd(dcol + (int64_t)w * D + e + i, sum[i]);
                atomicAdd(dtype0 + e + i, sum[i]);
            }
        }
        return;
    }
    const int tok = ((int)blockIdx.x - Lv) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // ---- scatter: one wave per (b, position)
    if (tok >= B * Lv) return;
    const int b = tok / Lv, pidx = tok - b * Lv;
    const int q = sel ? sel[pidx] : pidx;
    const int h = q / Wg, w = q % Wg;
    const int64_t src = src_row ? src_row[b] : b;
    const float inv = 1.0f / (float)Tf;
    for (int e = lane * 4; e < D; e += 256) {
        const f32x4 g = load4(dpre + ((int64_t)b * Ltot + Lt + pidx) * D + e);
        for (int t = 0; t < Tf; ++t) {
            float* dst = dgrid + (((src * Tf + t) * Hg + h) * Wg + w) * D + e;
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(dst + i, g[i] * inv);
        }
    }
}

inline unsigned nblk(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }
#define CB_D_OK(D) ((D) > 0 && (D) % 4 == 0 && (D) <= MAXD)

// dispatch FUNC<T, NCH>(args...) on dtype and on the number of 256-element chunks a row needs
#define CB_DISPATCH(FUNC, ...)                                                        \
    do {                                                                              \
        const int nch_ = D <= 768 ? 3 : (D <= 1024 ? 4 : 8);                          \
        if (dtype == CB_BF16) {                                                       \
            if (nch_ == 3) FUNC<bf16, 3>(__VA_ARGS__);                                \
            else if (nch_ == 4) FUNC<bf16, 4>(__VA_ARGS__);                           \
            else FUNC<bf16, 8>(__VA_ARGS__);                                          \
        } else if (dtype == CB_F32) {                                                 \
            if (nch_ == 3) FUNC<float, 3>(__VA_ARGS__);                               \
            else if (nch_ == 4) FUNC<float, 4>(__VA_ARGS__);                          \
            else FUNC<float, 8>(__VA_ARGS__);                                         \
        } else return cb_fail("bad dtype %d", dtype);                                 \
    } while (0)

template <typename T, int NCH>
void run_ln_fwd(hipStream_t st, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                int64_t rows, int D, float eps, int seg_len, int seg_stride, int seg_off) {
    if (D == NCH * 256 && NCH <= 4)
        hipLaunchKernelGGL((layernorm_fwd_kernel<T, (NCH <= 4 ? NCH : 4), true>), dim3(nblk(rows, 4)), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y,
                           mean, rstd, rows, D, eps, seg_len, seg_stride, seg_off);
    else
        hipLaunchKernelGGL((layernorm_fwd_kernel<T, NCH>), dim3(nblk(rows, 4)), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y,
                           mean, rstd, rows, D, eps, seg_len, seg_stride, seg_off);
}
template <typename T, int NCH>
void run_ln_bwd(hipStream_t st, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                float* dgamma, float* dbeta, int64_t rows, int D, void* dx2, float p, uint64_t seed, const uint64_t* seed_ptr,
                int seg_len, int seg_stride, int seg_off, float* part, int part_blocks) {
    // grid cap: every block ends with 2*D atomics, but a block per CU (one row per wave at the encoder's 2624 rows) measured faster
    // next to other kernels than 128 blocks with two rows per wave (tools/small_kernel_probe.py: 13.3 vs 16.5 us)
    constexpr unsigned cap = 256;
    if constexpr (NCH <= 4) {
        // rows-in-flight kernel, 12 waves x 1 row per block: measured best at the encoder's 2624 rows (219 blocks: 7.6 us against 12.8 us for
        // the 16-wave one-row-per-wave kernel it replaces, profiles/r06d_ln_bwd_probe.txt; 8 x 2: 8.3, 16 x 1: 7.8, 4 x 1 on 656 blocks: 8.8)
        unsigned blocks = nblk(rows, 12);
        if (blocks > cap) blocks = cap;
        if (part) blocks = (unsigned)part_blocks;
        if (D == NCH * 256)
            hipLaunchKernelGGL((layernorm_bwd_rows_kernel<T, NCH, 12, 1, true>), dim3(blocks), dim3(12 * 64), 0, st, (const T*)dy, (const T*)x, gamma, mean, rstd,
                               (T*)dx, dgamma, dbeta, rows, D, (T*)dx2, p, seed, seed_ptr, seg_len, seg_stride, seg_off, part);
        else
            hipLaunchKernelGGL((layernorm_bwd_rows_kernel<T, NCH, 12, 1, false>), dim3(blocks), dim3(12 * 64), 0, st, (const T*)dy, (const T*)x, gamma, mean, rstd,
                               (T*)dx, dgamma, dbeta, rows, D, (T*)dx2, p, seed, seed_ptr, seg_len, seg_stride, seg_off, part);
        return;
    }
    unsigned blocks = nblk(rows, 4);
    if (blocks > cap) blocks = cap;          // every block ends with 2*D atomics: keep them few
    if (part) blocks = (unsigned)part_blocks;
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, NCH, 4>), dim3(blocks), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, mean, rstd,
                       (T*)dx, dgamma, dbeta, rows, D, (T*)dx2, p, seed, seed_ptr, seg_len, seg_stride, seg_off, part);
}
template <typename T, int NCH>
void run_text_fwd(hipStream_t st, const int64_t* ids, const void* word, const void* pos, const void* type0, const float* gamma,
                  const float* beta, void* out, void* pre, float* mean, float* rstd, int B, int Lt, int Ltot, int D, float eps,
                  const int64_t* attn_mask, float* key_mask, int period) {
    hipLaunchKernelGGL((text_embed_fwd_kernel<T, NCH>), dim3(nblk((int64_t)B * Lt, 4)), dim3(256), 0, st, ids, (const T*)word,
                       (const T*)pos, (const T*)type0, gamma, beta, (T*)out, (T*)pre, mean, rstd, B, Lt, Ltot, D, eps, attn_mask, key_mask,
                       period);
}
template <typename T, int NCH>
void run_vis_fwd(hipStream_t st, const void* grid, const int32_t* src_row, const int32_t* sel, const void* row_emb,
                 const void* col_emb, const void* type0, const float* gamma, const float* beta, void* out, void* pre, float* mean,
                 float* rstd, int B, int Tf, int Hg, int Wg, int Lv, int Lt, int Ltot, int D, float eps, float* key_mask) {
    hipLaunchKernelGGL((visual_embed_fwd_kernel<T, NCH>), dim3(nblk((int64_t)B * Lv, 4)), dim3(256), 0, st, (const T*)grid, src_row,
                       sel, (const T*)row_emb, (const T*)col_emb, (const T*)type0, gamma, beta, (T*)out, (T*)pre, mean, rstd, B, Tf,
                       Hg, Wg, Lv, Lt, Ltot, D, eps, key_mask);
}

}  // namespace

extern "C" int cb_layernorm_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                float* rstd, int64_t rows, int32_t D, float eps, int32_t seg_len, int32_t seg_stride,
                                int32_t seg_off, void* stream) {
    CB_REQUIRE(x && y && gamma && beta && CB_D_OK(D), "cb_layernorm_fwd: bad arguments (D %% 4 == 0, D <= 2048)");
    if (rows == 0) return 0;
    CB_DISPATCH(run_ln_fwd, cb_stream(stream), x, gamma, beta, y, mean, rstd, rows, D, eps, seg_len, seg_stride, seg_off);
    return cb_launch_status("cb_layernorm_fwd");
}

extern "C" int cb_layernorm_bwd(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                                const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int32_t D, void* dx2,
                                float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_ptr, int32_t seg_len,
                                int32_t seg_stride, int32_t seg_off, void* stream) {
    CB_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && CB_D_OK(D), "cb_layernorm_bwd: bad arguments");
    if (rows == 0) return 0;
    CB_DISPATCH(run_ln_bwd, cb_stream(stream), dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, dx2, dropout_p, dropout_seed,
                dropout_seed_ptr, seg_len, seg_stride, seg_off, nullptr, 0);
    return cb_launch_status("cb_layernorm_bwd");
}

extern "C" int cb_layernorm_bwd_part(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                                     const float* rstd, void* dx, float* part, int32_t nblocks, int64_t rows, int32_t D, void* dx2,
                                     float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_ptr, int32_t seg_len,
                                     int32_t seg_stride, int32_t seg_off, void* stream) {
    CB_REQUIRE(dy && x && gamma && mean && rstd && dx && part && CB_D_OK(D), "cb_layernorm_bwd_part: bad arguments");
    CB_REQUIRE(nblocks >= 1 && nblocks <= 1024 && rows > 0, "cb_layernorm_bwd_part: nblocks %d / rows out of range", nblocks);
    CB_DISPATCH(run_ln_bwd, cb_stream(stream), dy, x, gamma, mean, rstd, dx, nullptr, nullptr, rows, D, dx2, dropout_p, dropout_seed,
                dropout_seed_ptr, seg_len, seg_stride, seg_off, part, nblocks);
    return cb_launch_status("cb_layernorm_bwd_part");
}

extern "C" int cb_ln_partials_reduce(const float* part, float* grad, const int64_t* off_gamma, const int64_t* off_beta,
                                     int32_t njobs, int32_t nblocks, int32_t D, void* stream) {
    CB_REQUIRE(part && grad && off_gamma && off_beta && njobs >= 1 && nblocks >= 1 && D >= 4 && D % 4 == 0, "cb_ln_partials_reduce: bad arguments");
    hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3((D + 255) / 256, 2, njobs), dim3(1024), 0, cb_stream(stream), part, grad,
                       off_gamma, off_beta, nblocks, D);
    return cb_launch_status("cb_ln_partials_reduce");
}

extern "C" int cb_text_embed_fwd(int32_t dtype, const int64_t* ids, const void* word, const void* pos, const void* type0,
                                 const float* gamma, const float* beta, void* out, void* pre, float* mean, float* rstd,
                                 int32_t B, int32_t Lt, int32_t L_total, int32_t D, float eps, const int64_t* attn_mask,
                                 float* key_mask, int32_t rows_period, void* stream) {
    CB_REQUIRE(ids && word && pos && type0 && gamma && beta && out && CB_D_OK(D) && Lt <= L_total, "cb_text_embed_fwd: bad arguments");
    CB_REQUIRE(rows_period >= 0 && rows_period <= B, "cb_text_embed_fwd: rows_period %d out of range", rows_period);
    if ((int64_t)B * Lt == 0) return 0;
    const int period = rows_period > 0 ? rows_period : B;
    CB_DISPATCH(run_text_fwd, cb_stream(stream), ids, word, pos, type0, gamma, beta, out, pre, mean, rstd, B, Lt, L_total, D, eps,
                attn_mask, key_mask, period);
    return cb_launch_status("cb_text_embed_fwd");
}

extern "C" int cb_visual_embed_fwd(int32_t dtype, const void* grid, const int32_t* src_row, const int32_t* sel,
                                   const void* row_emb, const void* col_emb, const void* type0, const float* gamma,
                                   const float* beta, void* out, void* pre, float* mean, float* rstd, int32_t B, int32_t T,
                                   int32_t Hg, int32_t Wg, int32_t Lv, int32_t Lt, int32_t L_total, int32_t D, float eps,
                                   float* key_mask, void* stream) {
    CB_REQUIRE(grid && row_emb && col_emb && type0 && gamma && beta && out && CB_D_OK(D) && T > 0 && Lt + Lv <= L_total,
               "cb_visual_embed_fwd: bad arguments");
    CB_REQUIRE(sel || Lv == Hg * Wg, "cb_visual_embed_fwd: Lv must equal Hg*Wg without a selection");
    if ((int64_t)B * Lv == 0) return 0;
    CB_DISPATCH(run_vis_fwd, cb_stream(stream), grid, src_row, sel, row_emb, col_emb, type0, gamma, beta, out, pre, mean, rstd, B, T,
                Hg, Wg, Lv, Lt, L_total, D, eps, key_mask);
    return cb_launch_status("cb_visual_embed_fwd");
}

extern "C" int cb_text_embed_bwd(int32_t dtype, const void* dpre, const int64_t* ids, float* dword, float* dpos, float* dtype0,
                                 int32_t B, int32_t Lt, int32_t L_total, int32_t D, int64_t pad_id, int32_t rows_period, void* stream) {
    CB_REQUIRE(dpre && ids && dword && dpos && dtype0 && D % 4 == 0 && rows_period >= 0 && rows_period <= B, "cb_text_embed_bwd: bad arguments");
    if ((int64_t)B * Lt == 0) return 0;
    const int period = rows_period > 0 ? rows_period : B;
    dim3 g((unsigned)(Lt + ((int64_t)B * Lt + 3) / 4)), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((text_embed_bwd_kernel<bf16>), g, b, 0, cb_stream(stream), (const bf16*)dpre, ids, dword, dpos, dtype0, B, Lt, L_total, D, pad_id, period);
    else if (dtype == CB_F32) hipLaunchKernelGGL((text_embed_bwd_kernel<float>), g, b, 0, cb_stream(stream), (const float*)dpre, ids, dword, dpos, dtype0, B, Lt, L_total, D, pad_id, period);
    else return cb_fail("cb_text_embed_bwd: bad dtype");
    return cb_launch_status("cb_text_embed_bwd");
}

extern "C" int cb_visual_embed_bwd(int32_t dtype, const void* dpre, const int32_t* src_row, const int32_t* sel, float* dgrid,
                                   float* drow, float* dcol, float* dtype0, int32_t B, int32_t T, int32_t Hg, int32_t Wg,
                                   int32_t Lv, int32_t Lt, int32_t L_total, int32_t D, void* stream) {
    CB_REQUIRE(dpre && dgrid && drow && dcol && dtype0 && D % 4 == 0 && T > 0, "cb_visual_embed_bwd: bad arguments");
    if ((int64_t)B * Lv == 0) return 0;
    dim3 g((unsigned)(Lv + ((int64_t)B * Lv + 3) / 4)), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((visual_embed_bwd_kernel<bf16>), g, b, 0, cb_stream(stream), (const bf16*)dpre, src_row, sel, dgrid, drow, dcol, dtype0, B, T, Hg, Wg, Lv, Lt, L_total, D);
    else if (dtype == CB_F32) hipLaunchKernelGGL((visual_embed_bwd_kernel<float>), g, b, 0, cb_stream(stream), (const float*)dpre, src_row, sel, dgrid, drow, dcol, dtype0, B, T, Hg, Wg, Lv, Lt, L_total, D);
    else return cb_fail("cb_visual_embed_bwd: bad dtype");
    return cb_launch_status("cb_visual_embed_bwd");
}
