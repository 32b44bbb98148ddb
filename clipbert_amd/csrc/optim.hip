// Fused AdamW over flat fp32 ranges with on-device global-norm clipping (no host sync), optionally
// emitting the bf16 compute copy of the updated weights in the same pass.  HBM-bound:
// 4 reads + 3 writes of fp32 (+ 1 bf16 write) per parameter.
#include "common.h"

#include <math.h>

namespace {

__global__ void __launch_bounds__(256) sq_sum_kernel(const float* g, int64_t n, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            f32x4 v = load4(g + i);
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        } else {
            for (int64_t j = i; j < n; ++j) s += g[j] * g[j];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// Deterministic variant: a FIXED grid writes one partial per block, a second one-block launch adds the partials in index order.
// The result depends on (n, grid) only -- not on the order blocks retire -- so every data-parallel rank, holding bit-identical
// all-reduced gradients, derives the bit-identical clip coefficient and stays bit-identical after the update.
template <typename G>
__global__ void __launch_bounds__(256) sq_sum_partial_kernel(const G* g, int64_t n, float* partial) {
    __shared__ float red[4];
    float s = 0.f;
    int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            f32x4 v = load4(g + i);
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        } else {
            for (int64_t j = i; j < n; ++j) { const float x = to_f32(g[j]); s += x * x; }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) sq_sum_final_kernel(const float* partial, int nb, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] += (red[0] + red[1]) + (red[2] + red[3]);
}

// cb_sq_sum_fold: up to four ranges of g as ONE index space (block b strides over it exactly as sq_sum_partial_kernel does), then the
// partials and the per-tile shares the weight-gradient launches left in `slots`, added in index order by one block.
struct SqSegs { int64_t lo[4], len[4]; int n; };
__global__ void __launch_bounds__(256) sq_sum_segs_kernel(const float* g, SqSegs sg, float* partial) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
#pragma unroll 1
    for (int k = 0; k < sg.n; ++k) {
        const float* q = g + sg.lo[k];
        const int64_t n = sg.len[k];
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
            if (i + 3 < n && ((sg.lo[k] & 3) == 0)) {
                f32x4 v = load4(q + i);
                s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            } else {
                for (int64_t j = i; j < n && j < i + 4; ++j) s += q[j] * q[j];
            }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(1024) sum_two_final_kernel(const float* a, int na, const float* b, int64_t nb, float* out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < na; i += 1024) s += a[i];
    // (eight independent loads per trip: the ~25 k slots of a step used to be a chain of dependent memory round trips, 14 us)
    int64_t i = threadIdx.x;
    for (; i + 7 * 1024 < nb; i += 8 * 1024) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = b[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; i < nb; i += 1024) s += b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] += t;
    }
}

struct PMV { float p, m, v; };
__device__ __forceinline__ PMV adamw_one(float p, float g, float m, float v, float gs, float b1, float b2, float eps,
                                         float step_size, float decay) {
    g *= gs;
    m = m * b1 + (1.0f - b1) * g;
    v = v * b2 + (1.0f - b2) * g * g;
    float denom = sqrtf(v) + eps;
    p = p - step_size * (m / denom);
    p = p - decay * p;          // decoupled weight decay AFTER the Adam update (adamw.py:77-101)
    PMV r = {p, m, v};
    return r;
}

// NT: the fp32 state (p, m, v: read once and written once per step, next touched a step later) moves with non-temporal loads / stores
// so that it does not displace what the next forward reads from the caches; the bf16 weight copy is stored normally.
template <typename G, bool NT>
__global__ void __launch_bounds__(256) adamw_kernel(float* p, const G* g, float* m, float* v, bf16* w16, int64_t n,
                                                    const float* hp, const float* sq_sum) {
    if (hp[CB_HP_SKIP] != 0.f) return;                   // (block-uniform: every thread reads the same word)
    const float lr = hp[CB_HP_LR], b1 = hp[CB_HP_BETA1], b2 = hp[CB_HP_BETA2], eps = hp[CB_HP_EPS];
    const float wd = hp[CB_HP_WD], max_norm = hp[CB_HP_MAX_NORM];
    const float step_size = lr * sqrtf(hp[CB_HP_BC2]) / hp[CB_HP_BC1];
    const float decay = wd > 0.f ? lr * wd : 0.f;
    float gs = hp[CB_HP_GRAD_SCALE];
    if (sq_sum && max_norm > 0.f) {
        float total = sqrtf(*sq_sum) * gs;
        float coef = max_norm / (total + 1e-6f);
        gs *= coef < 1.0f ? coef : 1.0f;
    }
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        f32x4 pp, mm, vv;
        const f32x4 gg = load4(g + i);
        if constexpr (NT) {
            pp = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i));
            mm = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + i));
            vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + i));
        } else {
            pp = load4(p + i); mm = load4(m + i); vv = load4(v + i);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            PMV r = adamw_one(pp[e], gg[e], mm[e], vv[e], gs, b1, b2, eps, step_size, decay);
            pp[e] = r.p; mm[e] = r.m; vv[e] = r.v;
        }
        if constexpr (NT) {
            __builtin_nontemporal_store(pp, reinterpret_cast<f32x4*>(p + i));
            __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m + i));
            __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + i));
        } else {
            store4(p + i, pp); store4(m + i, mm); store4(v + i, vv);
        }
        if (w16) store4(w16 + i, pp);
    } else {
        for (; i < n; ++i) {
            PMV r = adamw_one(p[i], to_f32(g[i]), m[i], v[i], gs, b1, b2, eps, step_size, decay);
            p[i] = r.p; m[i] = r.m; v[i] = r.v;
            if (w16) w16[i] = (bf16)r.p;
        }
    }
}

// non-temporal fp32 state traffic (round 3, profiles/r03p_adamw_nt.txt: 6.09 -> 6.38 TB/s on a 96 M-element range)
constexpr bool adamw_nt() { return true; }
}  // namespace

extern "C" int cb_sq_sum(const float* g, int64_t n, float* out_accum, void* stream) {
    CB_REQUIRE(g && out_accum, "cb_sq_sum: null pointer");
    if (n == 0) return 0;
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sq_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, cb_stream(stream), g, n, out_accum);
    return cb_launch_status("cb_sq_sum");
}

extern "C" int cb_sq_sum_det(const float* g, int64_t n, float* out_accum, float* ws, int32_t ws_floats, void* stream) {
    CB_REQUIRE(g && out_accum && ws && ws_floats >= 1, "cb_sq_sum_det: null pointer / empty workspace");
    if (n == 0) return 0;
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > ws_floats) blocks = ws_floats;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sq_sum_partial_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, cb_stream(stream), g, n, ws);
    hipLaunchKernelGGL(sq_sum_final_kernel, dim3(1), dim3(256), 0, cb_stream(stream), ws, (int)blocks, out_accum);
    return cb_launch_status("cb_sq_sum_det");
}

extern "C" int cb_sq_sum_fold(const float* g, const int64_t* seg_lo_hi, int32_t nseg, const float* slots, int64_t nslots, float* out_accum, float* ws,
                              int32_t ws_floats, void* stream) {
    CB_REQUIRE(out_accum && ws && ws_floats >= 1 && nseg >= 0 && nseg <= 4 && (nseg == 0 || (g && seg_lo_hi)) && nslots >= 0 && (nslots == 0 || slots),
               "cb_sq_sum_fold: bad arguments");
    SqSegs sg{};
    int64_t total = 0;
    for (int i = 0; i < nseg; ++i) {
        CB_REQUIRE(seg_lo_hi[2 * i] >= 0 && seg_lo_hi[2 * i + 1] >= seg_lo_hi[2 * i], "cb_sq_sum_fold: bad segment %d", i);
        if (seg_lo_hi[2 * i + 1] == seg_lo_hi[2 * i]) continue;
        sg.lo[sg.n] = seg_lo_hi[2 * i];
        sg.len[sg.n] = seg_lo_hi[2 * i + 1] - seg_lo_hi[2 * i];
        total = sg.len[sg.n] > total ? sg.len[sg.n] : total;
        ++sg.n;
    }
    int64_t blocks = (total + 1023) / 1024;
    if (blocks > ws_floats) blocks = ws_floats;
    if (blocks > 1024) blocks = 1024;
    if (blocks > 0) hipLaunchKernelGGL(sq_sum_segs_kernel, dim3((unsigned)blocks), dim3(256), 0, cb_stream(stream), g, sg, ws);
    hipLaunchKernelGGL(sum_two_final_kernel, dim3(1), dim3(1024), 0, cb_stream(stream), ws, (int)blocks, slots, nslots, out_accum);
    return cb_launch_status("cb_sq_sum_fold");
}

// bf16 gradients (the data-parallel wire image after the all-reduce): the optimizer consumes them directly, no cast back to fp32
extern "C" int cb_sq_sum_det_bf16(const void* g16, int64_t n, float* out_accum, float* ws, int32_t ws_floats, void* stream) {
    CB_REQUIRE(g16 && out_accum && ws && ws_floats >= 1, "cb_sq_sum_det_bf16: null pointer / empty workspace");
    if (n == 0) return 0;
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > ws_floats) blocks = ws_floats;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sq_sum_partial_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, cb_stream(stream), (const bf16*)g16, n, ws);
    hipLaunchKernelGGL(sq_sum_final_kernel, dim3(1), dim3(256), 0, cb_stream(stream), ws, (int)blocks, out_accum);
    return cb_launch_status("cb_sq_sum_det_bf16");
}

extern "C" int cb_adamw_g16(float* p, const void* g16, float* m, float* v, void* w16, int64_t n, const float* hyper,
                            const float* grad_sq_sum, void* stream) {
    CB_REQUIRE(p && g16 && m && v && hyper, "cb_adamw_g16: bad arguments");
    if (n == 0) return 0;
    if (adamw_nt())
        hipLaunchKernelGGL((adamw_kernel<bf16, true>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, cb_stream(stream), p, (const bf16*)g16, m, v,
                           (bf16*)w16, n, hyper, grad_sq_sum);
    else
        hipLaunchKernelGGL((adamw_kernel<bf16, false>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, cb_stream(stream), p, (const bf16*)g16, m, v,
                           (bf16*)w16, n, hyper, grad_sq_sum);
    return cb_launch_status("cb_adamw_g16");
}

extern "C" int cb_adamw(float* p, const float* g, float* m, float* v, void* w16, int64_t n, const float* hyper,
                        const float* grad_sq_sum, void* stream) {
    CB_REQUIRE(p && g && m && v && hyper, "cb_adamw: bad arguments");
    if (n == 0) return 0;
    if (adamw_nt())
        hipLaunchKernelGGL((adamw_kernel<float, true>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, cb_stream(stream), p, g, m, v, (bf16*)w16, n,
                           hyper, grad_sq_sum);
    else
        hipLaunchKernelGGL((adamw_kernel<float, false>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, cb_stream(stream), p, g, m, v, (bf16*)w16, n,
                           hyper, grad_sq_sum);
    return cb_launch_status("cb_adamw");
}
