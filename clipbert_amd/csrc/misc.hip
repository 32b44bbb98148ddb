// Small fp32 / elementwise kernels around the heads: softmax cross-entropy (+ gradient), column sums
// (bias gradients), dtype casts, activation backward.
#include "common.h"

namespace {

// one wave per row; logits fp32 (rows, C)
__global__ void __launch_bounds__(256) cross_entropy_kernel(const float* logits, int64_t ld, const int64_t* labels, float* loss,
                                                            float* dlogits, const float* dloss, int64_t rows, int C,
                                                            int64_t ignore_index) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = logits + row * ld;
    const int64_t y = labels[row];
    const bool ignored = (y == ignore_index);
    float m = -3.0e38f;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(x[c] - m);
    s = wave_sum(s);
    const float lse = m + __logf(s);
    if (lane == 0 && loss) loss[row] = ignored ? 0.f : (lse - x[y]);
    if (dlogits) {
        const float g = ignored ? 0.f : (dloss ? dloss[row] : 1.0f);
        float* d = dlogits + row * ld;
        for (int c = lane; c < C; c += 64) {
            float p = __expf(x[c] - lse);
            d[c] = g * (p - ((int64_t)c == y ? 1.0f : 0.0f));
        }
    }
}

// out[n] += sum over a slab of rows; thread = column, blockIdx.y = row slab
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* g, int64_t ldg, float* out, int64_t M, int N, int64_t rows_per) {
    int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    int64_t m0 = (int64_t)blockIdx.y * rows_per;
    int64_t m1 = m0 + rows_per < M ? m0 + rows_per : M;
    float s = 0.f;
    for (int64_t m = m0; m < m1; ++m) s += to_f32(g[m * ldg + n]);
    atomicAdd(out + n, s);
}

// vectorised variant: 32 column groups (4 columns each) x 8 row lanes per block; 8/16-byte loads,
// the 8 row lanes are combined through LDS, one atomic per column per block.
template <typename T>
__global__ void __launch_bounds__(256) colsum4_kernel(const T* g, int64_t ldg, float* out, int64_t M, int N, int64_t rows_per) {
    __shared__ float red[8][32 * 4 + 4];
    const int cg = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int n = (blockIdx.x * 32 + cg) * 4;
    int64_t m0 = (int64_t)blockIdx.y * rows_per;
    int64_t m1 = m0 + rows_per < M ? m0 + rows_per : M;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        for (int64_t m = m0 + ry; m < m1; m += 8) s = s + load4(g + m * ldg + n);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ry][cg * 4 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 128) {
        int c = threadIdx.x;
        int col = blockIdx.x * 128 + c;
        if (col < N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += red[r][c];
            atomicAdd(out + col, t);
        }
    }
}

template <typename S, typename D>
__global__ void __launch_bounds__(256) cast_kernel(const S* src, D* dst, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        store4(dst + i, load4(src + i));
    } else {
        for (; i < n; ++i) dst[i] = from_f32<D>(to_f32(src[i]));
    }
}

template <typename T>
__global__ void __launch_bounds__(256) act_bwd_kernel(int act, const T* dy, const T* ref, T* dx, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        f32x4 g = load4(dy + i), r = load4(ref + i), o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d;
            if (act == CB_ACT_GELU) d = gelu_erf_grad(r[e]);          // ref = pre-activation
            else if (act == CB_ACT_RELU) d = r[e] > 0.f ? 1.f : 0.f;  // ref = output
            else if (act == CB_ACT_TANH) d = 1.0f - r[e] * r[e];      // ref = output
            else d = 1.0f;
            o[e] = g[e] * d;
        }
        store4(dx + i, o);
    } else {
        for (; i < n; ++i) {
            float r = to_f32(ref[i]), d;
            if (act == CB_ACT_GELU) d = gelu_erf_grad(r);
            else if (act == CB_ACT_RELU) d = r > 0.f ? 1.f : 0.f;
            else if (act == CB_ACT_TANH) d = 1.0f - r * r;
            else d = 1.0f;
            dx[i] = from_f32<T>(to_f32(dy[i]) * d);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) dropout_kernel(const T* x, T* y, int64_t n, float p, uint64_t seed, const uint64_t* seed_ptr) {
    if (seed_ptr) seed += *seed_ptr;
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        f32x4 v = load4(x + i) * dropout_mult4(seed, (uint64_t)(i >> 2), p);      // i % 4 == 0
        store4(y + i, v);
    } else {
        const uint64_t grp = (uint64_t)(i >> 2);
        for (int e = 0; i < n; ++i, ++e) y[i] = from_f32<T>(to_f32(x[i]) * dropout_mult1(seed, grp, e, p));
    }
}

// ---- ELU + BatchNorm1d of the regression head (ClipBertForRegression.regressor[1:3], src/modeling/modeling.py:461-466) ------
// x: (B, D) pre-activation (output of regressor[0]); one thread per feature column walks the B rows (B = pairs in the batch,
// D = hidden size: a few hundred rows x 768 columns -- latency-, not bandwidth-relevant).  training: batch statistics
// (biased variance for the normalisation, unbiased for the running estimate, torch.nn.BatchNorm1d semantics).
__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expm1f(v); }

template <typename T>
__global__ void __launch_bounds__(256) elu_bn1d_fwd_kernel(const T* x, const float* gamma, const float* beta, float* run_mean, float* run_var,
                                                          T* y, float* save_mean, float* save_invstd, int B, int D, int training,
                                                          float momentum, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    float mean, invstd;
    if (training) {
        float s = 0.f;
        for (int r = 0; r < B; ++r) s += elu1(to_f32(x[(int64_t)r * D + c]));
        mean = s / (float)B;
        float q = 0.f;
        for (int r = 0; r < B; ++r) { const float d = elu1(to_f32(x[(int64_t)r * D + c])) - mean; q += d * d; }
        const float var = q / (float)B;
        invstd = rsqrtf(var + eps);
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (B > 1 ? q / (float)(B - 1) : var);
    } else {
        mean = run_mean[c];
        invstd = rsqrtf(run_var[c] + eps);
    }
    if (save_mean) { save_mean[c] = mean; save_invstd[c] = invstd; }
    const float g = gamma[c] * invstd, b = beta[c];
    for (int r = 0; r < B; ++r) y[(int64_t)r * D + c] = from_f32<T>((elu1(to_f32(x[(int64_t)r * D + c])) - mean) * g + b);
}

template <typename T>
__global__ void __launch_bounds__(256) elu_bn1d_bwd_kernel(const T* dy, const T* x, const float* gamma, const float* save_mean,
                                                          const float* save_invstd, T* dx, float* dgamma, float* dbeta, int B, int D,
                                                          int training) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    const float mean = save_mean[c], invstd = save_invstd[c], g = gamma[c];
    float sdy = 0.f, sdyx = 0.f;
    for (int r = 0; r < B; ++r) {
        const float d = to_f32(dy[(int64_t)r * D + c]);
        sdy += d;
        sdyx += d * (elu1(to_f32(x[(int64_t)r * D + c])) - mean) * invstd;
    }
    dgamma[c] += sdyx;
    dbeta[c] += sdy;
    const float k1 = training ? sdy / (float)B : 0.f, k2 = training ? sdyx / (float)B : 0.f;
    for (int r = 0; r < B; ++r) {
        const float xv = to_f32(x[(int64_t)r * D + c]);
        const float xh = (elu1(xv) - mean) * invstd;
        const float de = g * invstd * (to_f32(dy[(int64_t)r * D + c]) - k1 - xh * k2);
        dx[(int64_t)r * D + c] = from_f32<T>(de * (xv > 0.f ? 1.f : __expf(xv)));
    }
}

// ---- clip aggregation (a20: run_video_retrieval.py:402-418 training, :664-682 inference) -------------------------
// logits are clip-major [N][B*C] fp32 (the stack of the per-clip forward outputs); one thread per (pair, class).
__global__ void __launch_bounds__(256) clip_agg_fwd_kernel(const float* x, int N, int64_t bc, int mode, float* out, int32_t* argmax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= bc) return;
    float m = x[i];
    int am = 0;
    for (int n = 1; n < N; ++n) {
        const float v = x[(int64_t)n * bc + i];
        if (v > m) { m = v; am = n; }
    }
    if (mode == CB_AGG_MEAN) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += x[(int64_t)n * bc + i];
        out[i] = s / (float)N;
    } else if (mode == CB_AGG_MAX) {
        out[i] = m;
        if (argmax) argmax[i] = am;
    } else {                                   // log-sum-exp over the clips (max-shifted)
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += expf(x[(int64_t)n * bc + i] - m);
        out[i] = m + logf(s);
    }
}

__global__ void __launch_bounds__(256) clip_agg_bwd_kernel(const float* dout, const float* x, const float* out, const int32_t* argmax,
                                                           int N, int64_t bc, int mode, float* dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= bc) return;
    const float g = dout[i];
    for (int n = 0; n < N; ++n) {
        float d;
        if (mode == CB_AGG_MEAN) d = g / (float)N;
        else if (mode == CB_AGG_MAX) d = argmax[i] == n ? g : 0.f;
        else d = g * expf(x[(int64_t)n * bc + i] - out[i]);
        dx[(int64_t)n * bc + i] = d;
    }
}

// LSE training loss of one pair b: logsumexp over all (clip, class) - logsumexp over the clips of class label[b]
// (run_video_retrieval.py:415-418); optional gradient dlogits = dloss[b] * (softmax_all - [c == label] softmax_clips).
__global__ void __launch_bounds__(256) lse_loss_kernel(const float* x, const int64_t* labels, int N, int B, int C, float* loss,
                                                       const float* dloss, float* dx) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int64_t bc = (int64_t)B * C;
    const int lab = (int)labels[b];
    float ma = -3.0e38f, ml = -3.0e38f;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float v = x[(int64_t)n * bc + (int64_t)b * C + c];
            ma = fmaxf(ma, v);
            if (c == lab) ml = fmaxf(ml, v);
        }
    float sa = 0.f, sl = 0.f;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float v = x[(int64_t)n * bc + (int64_t)b * C + c];
            sa += expf(v - ma);
            if (c == lab) sl += expf(v - ml);
        }
    const float la = ma + logf(sa), ll = ml + logf(sl);
    if (loss) loss[b] = la - ll;
    if (dx) {
        const float g = dloss ? dloss[b] : 1.0f;
        for (int n = 0; n < N; ++n)
            for (int c = 0; c < C; ++c) {
                const int64_t idx = (int64_t)n * bc + (int64_t)b * C + c;
                float d = expf(x[idx] - la);
                if (c == lab) d -= expf(x[idx] - ll);
                dx[idx] = g * d;
            }
    }
}

inline unsigned nblk(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }


// ---- element-wise head losses and retrieval scores (fp32 logits; a handful of values per step, but part of the path: a16 / a18) ----
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// kind 0 / 1: one thread per element; kind 2 (sigmoid margin ranking): one thread per row of `group` logits (first = the positive)
__global__ void __launch_bounds__(256) head_loss_kernel(int kind, const float* x, const float* y, float* loss, const float* dloss, float* dx,
                                                        int64_t n, int group, float margin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (kind == 2) {
        const int64_t rows = n / group;
        if (i >= rows) return;
        const float* xr = x + i * group;
        const float s0 = sigmoidf_(xr[0]);
        float d0 = 0.f;
        for (int j = 1; j < group; ++j) {
            const float sj = sigmoidf_(xr[j]);
            const float l = margin + sj - s0;
            const bool on = l > 0.f;
            if (loss) loss[i * (group - 1) + j - 1] = on ? l : 0.f;
            if (dx) {
                const float g = on ? (dloss ? dloss[i * (group - 1) + j - 1] : 1.0f) : 0.f;
                dx[i * group + j] = g * sj * (1.0f - sj);
                d0 -= g;
            }
        }
        if (dx) dx[i * group] = d0 * s0 * (1.0f - s0);
        return;
    }
    if (i >= n) return;
    const float xv = x[i], yv = y[i];
    const float g = dx ? (dloss ? dloss[i] : 1.0f) : 0.f;
    if (kind == 0) {
        const float d = xv - yv;
        if (loss) loss[i] = d * d;
        if (dx) dx[i] = 2.0f * d * g;
    } else {
        if (loss) loss[i] = fmaxf(xv, 0.f) - xv * yv + log1pf(__expf(-fabsf(xv)));
        if (dx) dx[i] = (sigmoidf_(xv) - yv) * g;
    }
}

// C == 2: softmax(x)[:, 1] = sigmoid(x1 - x0);  C == 1: sigmoid(x)
__global__ void __launch_bounds__(256) retrieval_scores_kernel(const float* x, float* out, int64_t rows, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    if (C == 2) {
        const float a = x[2 * i], b = x[2 * i + 1], m = fmaxf(a, b);
        const float ea = __expf(a - m), eb = __expf(b - m);
        out[i] = eb / (ea + eb);
    } else {
        out[i] = sigmoidf_(x[i]);
    }
}

}  // namespace

extern "C" int cb_clip_aggregate_fwd(const float* logits, int32_t n_clips, int64_t bc, int32_t mode, float* out, int32_t* argmax,
                                     void* stream) {
    CB_REQUIRE(logits && out && n_clips > 0 && bc >= 0 && mode >= CB_AGG_MEAN && mode <= CB_AGG_LSE, "cb_clip_aggregate_fwd: bad arguments");
    if (bc == 0) return 0;
    hipLaunchKernelGGL(clip_agg_fwd_kernel, dim3(nblk(bc, 256)), dim3(256), 0, cb_stream(stream), logits, n_clips, bc, mode, out, argmax);
    return cb_launch_status("cb_clip_aggregate_fwd");
}

extern "C" int cb_clip_aggregate_bwd(const float* dout, const float* logits, const float* out, const int32_t* argmax, int32_t n_clips,
                                     int64_t bc, int32_t mode, float* dlogits, void* stream) {
    CB_REQUIRE(dout && dlogits && n_clips > 0 && mode >= CB_AGG_MEAN && mode <= CB_AGG_LSE, "cb_clip_aggregate_bwd: bad arguments");
    CB_REQUIRE(mode != CB_AGG_MAX || argmax, "cb_clip_aggregate_bwd: max needs the forward's argmax");
    CB_REQUIRE(mode != CB_AGG_LSE || (logits && out), "cb_clip_aggregate_bwd: lse needs the forward's input and output");
    if (bc == 0) return 0;
    hipLaunchKernelGGL(clip_agg_bwd_kernel, dim3(nblk(bc, 256)), dim3(256), 0, cb_stream(stream), dout, logits, out, argmax, n_clips, bc,
                       mode, dlogits);
    return cb_launch_status("cb_clip_aggregate_bwd");
}

extern "C" int cb_head_loss(int32_t kind, const float* x, const float* y, float* loss, const float* dloss, float* dx, int64_t n, int32_t group,
                            float margin, void* stream) {
    CB_REQUIRE(kind >= 0 && kind <= 2 && x && (loss || dx) && n >= 0, "cb_head_loss: bad arguments");
    CB_REQUIRE(kind == 2 ? (group >= 2 && n % group == 0) : (y != nullptr), "cb_head_loss: kind %d needs %s", kind, kind == 2 ? "group >= 2 dividing n" : "targets");
    if (n == 0) return 0;
    const int64_t work = kind == 2 ? n / group : n;
    hipLaunchKernelGGL(head_loss_kernel, dim3(nblk(work, 256)), dim3(256), 0, cb_stream(stream), kind, x, y, loss, dloss, dx, n, group, margin);
    return cb_launch_status("cb_head_loss");
}

extern "C" int cb_retrieval_scores(const float* logits, float* out, int64_t rows, int32_t C, void* stream) {
    CB_REQUIRE(logits && out && (C == 1 || C == 2) && rows >= 0, "cb_retrieval_scores: bad arguments (C = 1 or 2)");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(retrieval_scores_kernel, dim3(nblk(rows, 256)), dim3(256), 0, cb_stream(stream), logits, out, rows, C);
    return cb_launch_status("cb_retrieval_scores");
}

extern "C" int cb_lse_loss(const float* logits, const int64_t* labels, int32_t n_clips, int32_t B, int32_t C, float* loss,
                           const float* dloss, float* dlogits, void* stream) {
    CB_REQUIRE(logits && labels && n_clips > 0 && B >= 0 && C > 0 && (loss || dlogits), "cb_lse_loss: bad arguments");
    if (B == 0) return 0;
    hipLaunchKernelGGL(lse_loss_kernel, dim3(nblk(B, 256)), dim3(256), 0, cb_stream(stream), logits, labels, n_clips, B, C, loss, dloss, dlogits);
    return cb_launch_status("cb_lse_loss");
}

extern "C" int cb_dropout(int32_t dtype, const void* x, void* y, int64_t n, float p, uint64_t seed, const uint64_t* seed_ptr,
                          void* stream) {
    CB_REQUIRE(x && y && p >= 0.f && p < 1.f, "cb_dropout: bad arguments");
    if (n == 0) return 0;
    dim3 g(nblk(n, 1024)), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((dropout_kernel<bf16>), g, b, 0, cb_stream(stream), (const bf16*)x, (bf16*)y, n, p, seed, seed_ptr);
    else if (dtype == CB_F32) hipLaunchKernelGGL((dropout_kernel<float>), g, b, 0, cb_stream(stream), (const float*)x, (float*)y, n, p, seed, seed_ptr);
    else return cb_fail("cb_dropout: bad dtype");
    return cb_launch_status("cb_dropout");
}

extern "C" int cb_cross_entropy(const float* logits, int64_t ld, const int64_t* labels, float* loss, float* dlogits,
                                const float* dloss, int64_t rows, int32_t C, int64_t ignore_index, void* stream) {
    CB_REQUIRE(logits && labels && C > 0 && ld >= C, "cb_cross_entropy: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(nblk(rows, 4)), dim3(256), 0, cb_stream(stream), logits, ld, labels, loss,
                       dlogits, dloss, rows, C, ignore_index);
    return cb_launch_status("cb_cross_entropy");
}

extern "C" int cb_colsum(int32_t dtype, const void* g, int64_t ldg, float* out, int64_t M, int32_t N, void* stream) {
    CB_REQUIRE(g && out && N > 0, "cb_colsum: bad arguments");
    if (M == 0) return 0;
    const int esz = dtype == CB_BF16 ? 2 : 4;
    if (N % 4 == 0 && ldg % 4 == 0 && (reinterpret_cast<uintptr_t>(g) % (4 * esz)) == 0) {
        unsigned cb = nblk(N, 128);
        int64_t slabs = (M + 63) / 64;
        int64_t want = (1024 + cb - 1) / cb;              // ~1024 blocks in flight
        if (slabs > want) slabs = want;
        int64_t rows_per = ((M + slabs - 1) / slabs + 7) / 8 * 8;
        slabs = (M + rows_per - 1) / rows_per;
        dim3 gr(cb, (unsigned)slabs), b(256);
        if (dtype == CB_BF16) hipLaunchKernelGGL((colsum4_kernel<bf16>), gr, b, 0, cb_stream(stream), (const bf16*)g, ldg, out, M, N, rows_per);
        else if (dtype == CB_F32) hipLaunchKernelGGL((colsum4_kernel<float>), gr, b, 0, cb_stream(stream), (const float*)g, ldg, out, M, N, rows_per);
        else return cb_fail("cb_colsum: bad dtype");
        return cb_launch_status("cb_colsum");
    }
    int64_t slabs = (M + 127) / 128;
    if (slabs > 256) slabs = 256;
    int64_t rows_per = (M + slabs - 1) / slabs;
    dim3 gr(nblk(N, 256), (unsigned)slabs), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((colsum_kernel<bf16>), gr, b, 0, cb_stream(stream), (const bf16*)g, ldg, out, M, N, rows_per);
    else if (dtype == CB_F32) hipLaunchKernelGGL((colsum_kernel<float>), gr, b, 0, cb_stream(stream), (const float*)g, ldg, out, M, N, rows_per);
    else return cb_fail("cb_colsum: bad dtype");
    return cb_launch_status("cb_colsum");
}

extern "C" int cb_cast(int32_t src_dtype, const void* src, int32_t dst_dtype, void* dst, int64_t n, void* stream) {
    CB_REQUIRE(src && dst, "cb_cast: null pointer");
    if (n == 0) return 0;
    dim3 g(nblk(n, 1024)), b(256);
    hipStream_t st = cb_stream(stream);
    if (src_dtype == CB_F32 && dst_dtype == CB_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), g, b, 0, st, (const float*)src, (bf16*)dst, n);
    else if (src_dtype == CB_BF16 && dst_dtype == CB_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), g, b, 0, st, (const bf16*)src, (float*)dst, n);
    else if (src_dtype == CB_F32 && dst_dtype == CB_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, n);
    else if (src_dtype == CB_BF16 && dst_dtype == CB_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), g, b, 0, st, (const bf16*)src, (bf16*)dst, n);
    else return cb_fail("cb_cast: bad dtype");
    return cb_launch_status("cb_cast");
}

extern "C" int cb_act_bwd(int32_t dtype, int32_t act, const void* dy, const void* ref, void* dx, int64_t n, void* stream) {
    CB_REQUIRE(dy && ref && dx, "cb_act_bwd: null pointer");
    if (n == 0) return 0;
    dim3 g(nblk(n, 1024)), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((act_bwd_kernel<bf16>), g, b, 0, cb_stream(stream), act, (const bf16*)dy, (const bf16*)ref, (bf16*)dx, n);
    else if (dtype == CB_F32) hipLaunchKernelGGL((act_bwd_kernel<float>), g, b, 0, cb_stream(stream), act, (const float*)dy, (const float*)ref, (float*)dx, n);
    else return cb_fail("cb_act_bwd: bad dtype");
    return cb_launch_status("cb_act_bwd");
}

extern "C" int cb_elu_bn1d_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                               void* y, float* save_mean, float* save_invstd, int32_t B, int32_t D, int32_t training, float momentum,
                               float eps, void* stream) {
    CB_REQUIRE(x && gamma && beta && running_mean && running_var && y && B > 0 && D > 0, "cb_elu_bn1d_fwd: bad arguments");
    CB_REQUIRE((save_mean == nullptr) == (save_invstd == nullptr), "cb_elu_bn1d_fwd: save_mean and save_invstd go together");
    dim3 g((unsigned)((D + 255) / 256)), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((elu_bn1d_fwd_kernel<bf16>), g, b, 0, cb_stream(stream), (const bf16*)x, gamma, beta, running_mean, running_var, (bf16*)y, save_mean, save_invstd, B, D, training, momentum, eps);
    else if (dtype == CB_F32) hipLaunchKernelGGL((elu_bn1d_fwd_kernel<float>), g, b, 0, cb_stream(stream), (const float*)x, gamma, beta, running_mean, running_var, (float*)y, save_mean, save_invstd, B, D, training, momentum, eps);
    else return cb_fail("cb_elu_bn1d_fwd: bad dtype");
    return cb_launch_status("cb_elu_bn1d_fwd");
}

extern "C" int cb_elu_bn1d_bwd(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* save_mean,
                               const float* save_invstd, void* dx, float* dgamma, float* dbeta, int32_t B, int32_t D, int32_t training,
                               void* stream) {
    CB_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && B > 0 && D > 0, "cb_elu_bn1d_bwd: bad arguments");
    dim3 g((unsigned)((D + 255) / 256)), b(256);
    if (dtype == CB_BF16) hipLaunchKernelGGL((elu_bn1d_bwd_kernel<bf16>), g, b, 0, cb_stream(stream), (const bf16*)dy, (const bf16*)x, gamma, save_mean, save_invstd, (bf16*)dx, dgamma, dbeta, B, D, training);
    else if (dtype == CB_F32) hipLaunchKernelGGL((elu_bn1d_bwd_kernel<float>), g, b, 0, cb_stream(stream), (const float*)dy, (const float*)x, gamma, save_mean, save_invstd, (float*)dx, dgamma, dbeta, B, D, training);
    else return cb_fail("cb_elu_bn1d_bwd: bad dtype");
    return cb_launch_status("cb_elu_bn1d_bwd");
}

// ---- scalar plumbing of a training step (so that a captured step holds no framework-side elementwise kernels) ------------------
namespace {
__global__ void __launch_bounds__(256) mean_fwd_kernel(const float* x, int64_t n, float* out) {
    __shared__ float part[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];           // fixed order per thread + fixed tree: deterministic
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (part[0] + part[1] + part[2] + part[3]) / (float)n;
}
__global__ void __launch_bounds__(256) mean_bwd_kernel(const float* dmean, int64_t n, float* dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = *dmean / (float)n;
}
__global__ void counter_add_kernel(int64_t* c, int64_t inc) { *c += inc; }
// p[0, bytes) = 0: 16-byte stores over the aligned middle (grid-stride), byte stores for the unaligned head / tail
__global__ __launch_bounds__(256) void zero_kernel(unsigned char* p, int64_t bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    int64_t head = (int64_t)((16 - (a & 15)) & 15);
    if (head > bytes) head = bytes;
    const int64_t n16 = (bytes - head) >> 4;
    u32x4* q = reinterpret_cast<u32x4*>(p + head);
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int64_t i = tid; i < n16; i += stride) q[i] = z;
    const int64_t tail0 = head + (n16 << 4);
    if (tid < head) p[tid] = 0;
    if (tid < bytes - tail0) p[tail0 + tid] = 0;
}
// up to four ranges in one launch (cb_zero_ranges): blockIdx.y picks the range, the body is zero_kernel's
struct ZeroRanges { unsigned char* p[4]; int64_t bytes[4]; };
__global__ __launch_bounds__(256) void zero_ranges_kernel(ZeroRanges zr) {
    unsigned char* p = zr.p[blockIdx.y];
    const int64_t bytes = zr.bytes[blockIdx.y];
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    int64_t head = (int64_t)((16 - (a & 15)) & 15);
    if (head > bytes) head = bytes;
    const int64_t n16 = (bytes - head) >> 4;
    u32x4* q = reinterpret_cast<u32x4*>(p + head);
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int64_t i = tid; i < n16; i += stride) q[i] = z;
    const int64_t tail0 = head + (n16 << 4);
    if (tid < head) p[tid] = 0;
    if (tid < bytes - tail0) p[tail0 + tid] = 0;
}
}  // namespace

extern "C" int cb_mean_fwd(const float* x, int64_t n, float* out, void* stream) {
    CB_REQUIRE(x && out && n > 0, "cb_mean_fwd: bad arguments");
    hipLaunchKernelGGL(mean_fwd_kernel, dim3(1), dim3(256), 0, cb_stream(stream), x, n, out);
    return cb_launch_status("cb_mean_fwd");
}
extern "C" int cb_mean_bwd(const float* dmean, int64_t n, float* dx, void* stream) {
    CB_REQUIRE(dmean && dx && n > 0, "cb_mean_bwd: bad arguments");
    hipLaunchKernelGGL(mean_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cb_stream(stream), dmean, n, dx);
    return cb_launch_status("cb_mean_bwd");
}
extern "C" int cb_zero(void* p, int64_t bytes, void* stream) {
    CB_REQUIRE(bytes >= 0 && (p || bytes == 0), "cb_zero: bad arguments");
    if (bytes == 0) return 0;
    const int64_t n16 = bytes >> 4;
    int64_t blocks = (n16 + 256 * 4 - 1) / (256 * 4);                 // ~4 stores per thread; one block covers head + tail bytes
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, cb_stream(stream), static_cast<unsigned char*>(p), bytes);
    return cb_launch_status("cb_zero");
}
extern "C" int cb_zero_ranges(void* const* ptrs, const int64_t* bytes, int32_t n, void* stream) {
    CB_REQUIRE(n >= 0 && n <= 4 && (n == 0 || (ptrs && bytes)), "cb_zero_ranges: 0..4 ranges");
    ZeroRanges zr{};
    int m = 0;
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        CB_REQUIRE(bytes[i] >= 0 && (bytes[i] == 0 || ptrs[i]), "cb_zero_ranges: bad range %d", i);
        if (bytes[i] == 0) continue;
        zr.p[m] = static_cast<unsigned char*>(ptrs[i]); zr.bytes[m] = bytes[i]; ++m;
        most = bytes[i] > most ? bytes[i] : most;
    }
    if (m == 0) return 0;
    int64_t blocks = ((most >> 4) + 256 * 4 - 1) / (256 * 4);
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, cb_stream(stream), zr);
    return cb_launch_status("cb_zero_ranges");
}
extern "C" int cb_counter_add(int64_t* counter, int64_t inc, void* stream) {
    CB_REQUIRE(counter, "cb_counter_add: null counter");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, cb_stream(stream), counter, inc);
    return cb_launch_status("cb_counter_add");
}
