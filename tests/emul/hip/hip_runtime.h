// Lane-level HOST emulator of the HIP device environment -- TEST INFRASTRUCTURE ONLY.
//
// The product kernels in clipbert_amd/csrc/*.hip are pure gfx950 HIP (no dual paths).  Because
// the build container has no GPU, the CPU test-suite compiles those SAME source files with the host
// clang++ against this header (-I tests/emul shadows <hip/hip_runtime.h>) into
// tests/emul/_build/libclipbert_emul.so and drives them through the same C ABI.  This checks the
// index arithmetic, tiling, boundary handling and numerics of every kernel on CPU.  It is never
// loaded by clipbert_amd/ (the product loader only opens libclipbert_hip.so and fails loudly).
//
// Model: each GPU thread is a cooperative fiber; a block's fibers run round-robin on one OS thread
// and switch only at collective points (__syncthreads, MFMA, shuffles).  Blocks of a grid are spread
// over a few OS threads.  MFMA fragment layouts follow cdna_hip_programming.md section 3.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline const char* hipGetErrorString(hipError_t) { return "emul"; }
static inline hipError_t hipGetLastError() { return 0; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return 0; }      // (an MI355X's CU count)
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? 0 : 2; }
typedef int hipStreamCaptureStatus;
#define hipStreamCaptureStatusNone 0
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return 0; }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    memmove(d, s, n); return 0;
}

namespace emul {
constexpr int kWave = 64;
struct Wave {
    int arrive = 0;
    int gen = 0;
    int nlanes = 64;
    alignas(16) unsigned char buf[2][kWave][64];
};
// one lane's share of an LDS-DMA instruction that was issued and has not been retired by an s_waitcnt vmcnt(N) yet
struct PendingDma { unsigned char* dst; const unsigned char* src; unsigned size; };
constexpr int kMaxDma = 64;
struct Fiber {
    void* sp = nullptr;
    dim3 tid;
    int lane = 0;
    int wave = 0;
    bool done = false;
    PendingDma dma[kMaxDma];
    int dma_head = 0, dma_n = 0;
};
struct Block {
    dim3 bid, bdim, gdim;
    int nthreads = 0;
    int bar_arrive = 0;
    int bar_gen = 0;
    Wave* waves = nullptr;
};
extern thread_local Fiber* cur;
extern thread_local Block* blk;
void yield();
// run `body` once per thread of every block of the grid
void launch(dim3 grid, dim3 block, const std::function<void()>& body);

// EMUL_DMA_LAZY=1: an LDS-DMA's bytes reach LDS only when the issuing lane's s_waitcnt vmcnt(N) retires it (the latest
// moment the hardware allows): a ds_read placed before the covering wait + barrier then sees stale data -- deterministically.
// Default (eager): the bytes land at issue, the earliest moment: exposes a stage that is re-filled while still being read.
inline bool dma_lazy() {
    static const bool v = [] { const char* e = getenv("EMUL_DMA_LAZY"); return e && atoi(e) != 0; }();
    return v;
}
inline void dma_apply(const PendingDma& d) {
    if (d.src) memcpy(d.dst, d.src, d.size); else memset(d.dst, 0, d.size);
}
inline void dma_retire(int keep) {              // oldest first, until at most `keep` are outstanding
    Fiber* f = cur;
    while (f->dma_n > keep) {
        dma_apply(f->dma[f->dma_head]);
        f->dma_head = (f->dma_head + 1) % kMaxDma;
        --f->dma_n;
    }
}
inline void dma_issue(unsigned char* dst, const unsigned char* src, unsigned size) {
    PendingDma d{dst, src, size};
    if (!dma_lazy()) { dma_apply(d); return; }
    Fiber* f = cur;
    if (f->dma_n == kMaxDma) { fprintf(stderr, "emul: more than %d LDS-DMAs in flight (vmcnt is a 6-bit counter)\n", kMaxDma); abort(); }
    f->dma[(f->dma_head + f->dma_n) % kMaxDma] = d;
    ++f->dma_n;
}

inline void block_barrier() {
    Block* b = blk;
    int g = b->bar_gen;
    if (++b->bar_arrive == b->nthreads) { b->bar_arrive = 0; b->bar_gen++; return; }
    while (b->bar_gen == g) yield();
}
// deposit `bytes` (<= 64) for this lane, wait for the whole wave, return the wave's table
inline unsigned char (*wave_collect(const void* mine, int bytes))[64] {
    Wave& w = blk->waves[cur->wave];
    int g = w.gen, par = g & 1;
    memcpy(w.buf[par][cur->lane], mine, bytes);
    if (++w.arrive == w.nlanes) { w.arrive = 0; w.gen++; }
    else while (w.gen == g) yield();
    return w.buf[par];
}
}  // namespace emul

#define threadIdx (emul::cur->tid)
#define blockIdx (emul::blk->bid)
#define blockDim (emul::blk->bdim)
#define gridDim (emul::blk->gdim)
constexpr int warpSize = 64;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

namespace emul {
template <class K, class... A> inline void launch_kernel(dim3 g, dim3 b, K k, A... a) {
    launch(g, b, [&]() { k(a...); });
}
}  // namespace emul
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emul::launch_kernel((grid), (block), (kernel), __VA_ARGS__)

static inline void __syncthreads() { emul::block_barrier(); }
static inline void __builtin_amdgcn_s_barrier() { emul::block_barrier(); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
template <class T> static inline T __builtin_amdgcn_readfirstlane(T v) { return v; }

// ---- shuffles ------------------------------------------------------------------------------
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) <= 64, "");
    auto tab = emul::wave_collect(&v, sizeof(T));
    int lane = emul::cur->lane;
    int base = lane & ~(width - 1);
    T r; memcpy(&r, tab[base + (src & (width - 1))], sizeof(T));
    return r;
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    auto tab = emul::wave_collect(&v, sizeof(T));
    int lane = emul::cur->lane;
    int src = lane ^ mask;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    T r; memcpy(&r, tab[src], sizeof(T));
    return r;
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    auto tab = emul::wave_collect(&v, sizeof(T));
    int lane = emul::cur->lane;
    int src = lane + (int)delta;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    T r; memcpy(&r, tab[src], sizeof(T));
    return r;
}

// ---- MFMA ----------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 emul_bf16x8;
typedef __attribute__((ext_vector_type(4))) float emul_f32x4;

// D = A(16x32) * B(32x16) + C.  A: lane l holds A[l&15][8*(l>>4)+j]; B: lane l holds
// B[8*(l>>4)+j][l&15]; C/D: lane l reg r -> row 4*(l>>4)+r, col l&15.
static inline emul_f32x4 emul_mfma_f32_16x16x32_bf16(emul_bf16x8 a, emul_bf16x8 b, emul_f32x4 c, int, int, int) {
    struct { emul_bf16x8 a, b; } mine{a, b};
    auto tab = emul::wave_collect(&mine, sizeof(mine));
    int l = emul::cur->lane, col = l & 15;
    emul_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k) {
            emul_bf16x8 av, bv;
            memcpy(&av, tab[row + 16 * (k >> 3)], 16);
            memcpy(&bv, tab[col + 16 * (k >> 3)] + 16, 16);
            acc += (float)av[k & 7] * (float)bv[k & 7];
        }
        d[r] += acc;
    }
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emul_mfma_f32_16x16x32_bf16

// f32-input MFMA: A lane l holds A[l&15][l>>4]; B lane l holds B[l>>4][l&15]; k = 0..3.
static inline emul_f32x4 emul_mfma_f32_16x16x4f32(float a, float b, emul_f32x4 c, int, int, int) {
    struct { float a, b; } mine{a, b};
    auto tab = emul::wave_collect(&mine, sizeof(mine));
    int l = emul::cur->lane, col = l & 15;
    emul_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = d[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, tab[row + 16 * k], 4);
            memcpy(&bv, tab[col + 16 * k] + 4, 4);
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emul_mfma_f32_16x16x4f32

// ---- buffer descriptor loads (range-checked: offsets past num_records read zeros) --------------------
struct emul_rsrc { const unsigned char* base; uint32_t n; };
static inline emul_rsrc emul_make_buffer_rsrc(void* p, short, int n, int) { return emul_rsrc{(const unsigned char*)p, (uint32_t)n}; }
typedef __attribute__((ext_vector_type(4))) uint32_t emul_u32x4;
static inline emul_u32x4 emul_raw_buffer_load_b128(emul_rsrc r, uint32_t voff, uint32_t soff, int) {
    emul_u32x4 v = {0u, 0u, 0u, 0u};
    uint64_t o = (uint64_t)voff + soff;
    if (o + 16 <= r.n) memcpy(&v, r.base + o, 16);
    return v;
}
static inline void emul_raw_buffer_store_b128(emul_u32x4 v, emul_rsrc r, uint32_t voff, uint32_t soff, int) {
    uint64_t o = (uint64_t)voff + soff;
    if (o + 16 <= r.n) memcpy(const_cast<unsigned char*>(r.base) + o, &v, 16);       // (out-of-range stores are dropped, as the hardware does)
}
#define __builtin_amdgcn_make_buffer_rsrc emul_make_buffer_rsrc
#define __builtin_amdgcn_raw_buffer_load_b128 emul_raw_buffer_load_b128
#define __builtin_amdgcn_raw_buffer_store_b128 emul_raw_buffer_store_b128

// LDS-DMA: 16 bytes per lane from the (range-checked) buffer to wave-uniform LDS base + lane*16
static inline void emul_buffer_load_lds(emul_rsrc r, void* lds_base, unsigned size, uint32_t voff, uint32_t soff, uint32_t, uint32_t) {
    unsigned char* dst = (unsigned char*)lds_base + emul::cur->lane * size;
    uint64_t o = (uint64_t)voff + soff;
    emul::dma_issue(dst, (o + size <= r.n) ? r.base + o : nullptr, size);
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, p, sz, vo, so, off, aux) emul_buffer_load_lds((r), (void*)(p), (sz), (vo), (so), (off), (aux))
// s_waitcnt: only the vmcnt field matters to the emulator (imm[3:0] | imm[15:14] << 4): retires this lane's oldest LDS-DMAs
static inline void __builtin_amdgcn_s_waitcnt(int imm) { emul::dma_retire((imm & 15) | (((imm >> 14) & 3) << 4)); }

// v_perm_b32: result byte i = byte sel[i] of the 8 bytes {s1 (0..3), s0 (4..7)}
static inline uint32_t emul_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        uint32_t b = (sel >> (8 * i)) & 0xff;
        uint32_t v = b < 8 ? (uint32_t)((src >> (8 * b)) & 0xff) : (b >= 0xd ? 0xffu : 0u);
        r |= v << (8 * i);
    }
    return r;
}
#define __builtin_amdgcn_perm emul_perm

// ds_read_b64_tr_b16: per 16-lane group, lane p supplies 4 consecutive 16-bit elements of row p/4
// (columns 4*(p%4)..+3); lane i receives column i of that 4 x 16 block (verified on gfx950, tools/tr_probe.hip)
typedef __attribute__((ext_vector_type(4))) short emul_s16x4;
static inline emul_s16x4 emul_ds_read_tr16(const void* addr) {
    emul_s16x4 mine;
    memcpy(&mine, addr, 8);
    auto tab = emul::wave_collect(&mine, 8);
    const int lane = emul::cur->lane, base = lane & ~15, i = lane & 15;
    emul_s16x4 r;
    for (int e = 0; e < 4; ++e) {
        emul_s16x4 src;
        memcpy(&src, tab[base + 4 * e + i / 4], 8);
        r[e] = src[i % 4];
    }
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emul_ds_read_tr16((const void*)(p))

// ---- atomics -------------------------------------------------------------------------------
static inline float atomicAdd(float* p, float v) {
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        float f; memcpy(&f, &old, 4); f += v;
        uint32_t nu; memcpy(&nu, &f, 4);
        if (__atomic_compare_exchange_n(u, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            float o; memcpy(&o, &old, 4); return o;
        }
    }
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }

// ---- math ----------------------------------------------------------------------------------
#define __expf(x) expf(x)
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
