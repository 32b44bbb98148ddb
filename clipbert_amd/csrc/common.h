// Shared device/host helpers for libclipbert_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "clipbert_hip.h"

// gfx950 only: the kernels rely on its MFMA shapes, LDS-DMA, transpose reads -- and on its memory behaviour (the slab K split publishes
// partial tiles with sc1 write-through stores / sc1 loads around a relaxed ticket instead of release / acquire fences: measured safe on
// MI355X under uneven load, tests/test_gemm_group.py stale-line test, tools/replay_determinism.py; NOT a HIP memory-model guarantee).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libclipbert_hip is written for gfx950 (MI355X) only"
#endif

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---- error reporting (thread-local message, never throws across the ABI) ----------------------
int cb_fail(const char* fmt, ...);   // records the message, returns -1
int cb_launch_status(const char* what);
#define CB_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) return cb_fail(__VA_ARGS__); \
    } while (0)

static inline hipStream_t cb_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
// CB_PERSISTENT_MAXWG=n (tests only): the persistent kernels (cb_stem_pool, cb_res2_block, the streaming cb_gemm) launch at most n
// workgroups, so that a few workgroups walk many tiles and the tile loops are exercised on small problems.  Read per call.
static inline int cb_persistent_max_workgroups(int dflt) {
    const char* cap = getenv("CB_PERSISTENT_MAXWG");
    return cap && atoi(cap) > 0 ? atoi(cap) : dflt;
}

// ---- scalar conversions ----------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// 4 consecutive elements <-> f32x4
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}

// ---- activations -----------------------------------------------------------------------------
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 round-off level): erfc(z) = q(t) exp(-z^2), t = 1 / (1 + p z), z >= 0.
// One v_exp_f32 + one v_rcp_f32 (1 ulp; an IEEE division is ten instructions) + 6 FMAs instead of libm's branchy erff -- the exact-erf
// GELU sits in the epilogue of the largest GEMM of every layer, where the stamps showed ~30 VALU instructions per element = 7 us of a 30 us
// launch (profiles/r05r).  The GELU derivative shares ONE exponential between the cdf and the pdf: exp(-(x/sqrt 2)^2) = exp(-x^2/2).
__device__ __forceinline__ float as_erfc_poly(float az) {       // erfc(az) / exp(-az^2), az >= 0
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    return p * t;
}
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float r = 1.0f - as_erfc_poly(ax) * __expf(-ax * ax);
    return x < 0.f ? -r : r;
}
__device__ __forceinline__ float gelu_erf(float x) {
    const float hq = 0.5f * as_erfc_poly(fabsf(x) * 0.70710678118654752f) * __expf(-0.5f * x * x);      // erfc(|x| / sqrt 2) / 2
    return x * (x < 0.f ? hq : 1.0f - hq);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float e = __expf(-0.5f * x * x);
    const float hq = 0.5f * as_erfc_poly(fabsf(x) * 0.70710678118654752f) * e;
    return (x < 0.f ? hq : 1.0f - hq) + x * (0.3989422804014327f * e);
}
// GELU and its derivative at once (CB_ACT_GELU_SAVE_GRAD): the forward epilogue already holds exp(-x^2 / 2) and erfc(|x| / sqrt 2) / 2; the
// derivative costs three more instructions there and saves the backward epilogue the whole evaluation (it multiplies by the stored value).
__device__ __forceinline__ void gelu_erf_both(float x, float& y, float& dy) {
    const float e = __expf(-0.5f * x * x);
    const float hq = 0.5f * as_erfc_poly(fabsf(x) * 0.70710678118654752f) * e;
    const float cdf = x < 0.f ? hq : 1.0f - hq;
    y = x * cdf;
    dy = cdf + x * (0.3989422804014327f * e);
}
// Two elements at once in PACKED fp32 (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: one lane-instruction per pair), round 6 -- the
// bf16 performance mode's GELU epilogue (FFN1 spent 8 of its 12-13.6 us of epilogue on ~35 scalar fp32 VALU slots per
// element).  Same A&S 7.1.26 rational as above (the 0.5 of Phi folded into its coefficients), exp as ONE v_exp_f32 on a pre-scaled
// argument (2^(-x^2 log2(e) / 2): __expf costs a multiply more), the branch Phi = x < 0 ? q : 1 - q as 0.5 + copysign(0.5 - q, x).  The fmas
// are written as fmas (no dependence on what -ffp-contract decides per call site): every bf16 kernel evaluates these exact operations.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void gelu_erf_both2(f32x2 x, f32x2& y, f32x2& dy) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 xx = x * x;
    const f32x2 ea = xx * (-0.72134752044448170368f);                                   // -x^2 / 2 * log2(e)
    const f32x2 e = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};     // exp(-x^2 / 2)
    const f32x2 den = pk_fma(ax, f32x2{0.23164188826636f, 0.23164188826636f}, f32x2{1.0f, 1.0f});   // 1 + 0.3275911 |x| / sqrt 2
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2 q = pk_fma(f32x2{0.5307027145f, 0.5307027145f}, t, f32x2{-0.7265760135f, -0.7265760135f});
    q = pk_fma(q, t, f32x2{0.7107068705f, 0.7107068705f});
    q = pk_fma(q, t, f32x2{-0.142248368f, -0.142248368f});
    q = pk_fma(q, t, f32x2{0.127414796f, 0.127414796f});
    q = q * t * e;                                                                      // erfc(|x| / sqrt 2) / 2
    const f32x2 hm = f32x2{0.5f, 0.5f} - q;
    const f32x2 cdf = f32x2{0.5f, 0.5f} + f32x2{__builtin_copysignf(hm[0], x[0]), __builtin_copysignf(hm[1], x[1])};
    y = x * cdf;
    dy = pk_fma(x, e * 0.3989422804014327f, cdf);
}
// scalar faces of the packed evaluation: EVERY bf16-mode GELU of the library goes through gelu_erf_both2, so that kernels with different
// epilogue structures (4-wave / 8-wave, generic / specialised, with or without the stored derivative) produce the same bits
__device__ __forceinline__ float gelu_erf_pk(float x) {
    f32x2 y, dy;
    gelu_erf_both2(f32x2{x, x}, y, dy);
    return y[0];
}
__device__ __forceinline__ void gelu_erf_both_pk(float x, float& y, float& dy) {
    f32x2 yy, dd;
    gelu_erf_both2(f32x2{x, x}, yy, dd);
    y = yy[0]; dy = dd[0];
}
__device__ __forceinline__ float apply_act(int act, float v) {
    switch (act) {
        case CB_ACT_RELU: return v > 0.f ? v : 0.f;
        case CB_ACT_GELU_SAVE_GRAD:                    // (without a second output it is a plain GELU)
        case CB_ACT_GELU: return gelu_erf(v);
        case CB_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// ---- stateless dropout mask ------------------------------------------------------------------------
// One 64-bit hash (splitmix64 finaliser) decides FOUR consecutive elements, 16 bits each (the 64-bit multiplies are the
// expensive part on the VALU; p is resolved to 1/65536).  Elements are addressed as (group, e): element 4*group + e of
// the site's stream; a 2-D site (rows x cols) uses group = row * ceil(cols/4) + col/4, e = col%4, so every producer
// and consumer of one mask (GEMM epilogue <-> LayerNorm backward, attention forward <-> backward, cb_dropout forward
// <-> backward) agrees whatever its own vector width is.
__device__ __forceinline__ uint64_t cb_hash64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t dropout_threshold(float p) { return (uint32_t)(p * 65536.0f); }
// multipliers of the 4 elements of `group`: 0 (dropped) or 1/(1-p) (kept)
__device__ __forceinline__ f32x4 dropout_mult4(uint64_t seed, uint64_t group, float p) {
    const uint64_t z = cb_hash64(seed, group);
    const uint32_t thr = dropout_threshold(p), lo = (uint32_t)z, hi = (uint32_t)(z >> 32);
    const float keep = 1.0f / (1.0f - p);
    f32x4 m;
    m[0] = (lo & 0xffffu) < thr ? 0.f : keep;
    m[1] = (lo >> 16) < thr ? 0.f : keep;
    m[2] = (hi & 0xffffu) < thr ? 0.f : keep;
    m[3] = (hi >> 16) < thr ? 0.f : keep;
    return m;
}
// multiplier of element e (0..3) of `group`
__device__ __forceinline__ float dropout_mult1(uint64_t seed, uint64_t group, int e, float p) {
    const uint64_t z = cb_hash64(seed, group);
    return (uint32_t)((z >> (16 * e)) & 0xffffu) < dropout_threshold(p) ? 0.f : 1.0f / (1.0f - p);
}

// ---- wave reductions (64 lanes) -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
