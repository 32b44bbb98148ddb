// cb_gemm, few rows (tile 9): the heads' products -- pooler, the classifier MLP and their data gradients: 64 rows x 768..1536 columns over a
// reduction of 768..1536.  One 64x64 tile of the 4-wave kernels walks that reduction as ONE chain of K tiles behind a barrier each
// (10-19 us per launch in the step for ~0.2 GFLOP); here a workgroup owns 32 rows x 64 columns and its four waves SPLIT the reduction:
// wave w takes the 32-deep K steps w, w+4, ..., every MFMA operand of up to six steps is requested at once straight from global memory in
// fragment layout (A and a k-contiguous B: 16 bytes per lane, no LDS; a reduction-major B: its natural [k][64 columns] lines, staged in
// the wave's own LDS slots and read back with the transpose read), and the four partial tiles meet in LDS, where each thread then
// owns 8 consecutive columns of a row and runs the library's generic epilogue (epilogue_vec / epilogue_elem: every option of the
// descriptor means what it means everywhere else).  The sum order differs from the other structures' (4 interleaved partial sums).
#include "gemm_impl.h"

namespace cbgemm {
namespace {
constexpr int SK_BM = 32, SK_BN = 64, SK_NB = 6;          // tile, K steps requested at once per wave
constexpr int SK_STEP_BYTES = 32 * SK_BN * 2;             // one staged K step of a reduction-major B: 32 lines x 128 B

// 8 bf16 of row `row` at k..k+7 (k-contiguous operand): one 16-byte load where the host found everything aligned, guarded elements otherwise
__device__ __forceinline__ bf16x8 sk_load_rowk(const bf16* base, int64_t ld, int row, bool row_ok, int k, int K, bool vec) {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
    if (!row_ok || k >= K) return z;
    const bf16* q = base + (int64_t)row * ld + k;
    if (vec) return *reinterpret_cast<const bf16x8*>(q);
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (k + e < K) z[e] = q[e];
    return z;
}

template <bool BKROW>
__global__ void __launch_bounds__(256) gemm_skinny_kernel(GP p, int a_vec, int b_vec) {
    constexpr int STAGE = BKROW ? 4 * SK_NB * SK_STEP_BYTES : 0, RED = 4 * SK_BM * SK_BN * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE > RED ? STAGE : RED];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * SK_BM, n0 = blockIdx.x * SK_BN;
    const int nsteps = (p.K + 31) / 32, per_wave = (nsteps + 3) / 4;           // (per_wave: the same for every wave -- barriers below)
    const bf16* A = reinterpret_cast<const bf16*>(p.A);
    const bf16* B = reinterpret_cast<const bf16*>(p.B);
    if (p.dropout_p > 0.f && p.seed_ptr) p.seed += *p.seed_ptr;
    const int kq = 8 * (lane >> 4), r16 = lane & 15;

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }

    unsigned char* stage = smem + wave * (SK_NB * SK_STEP_BYTES);
    for (int b0 = 0; b0 < per_wave; b0 += SK_NB) {
        bf16x8 af[SK_NB][2];
        bf16x8 bq[SK_NB][4];           // k-contiguous B: the four column-block fragments | reduction-major B: this lane's four 16-byte line chunks
#pragma unroll
        for (int q = 0; q < SK_NB; ++q) {
            const int s = wave + 4 * (b0 + q);
            const int k = s * 32 + kq;
            const bool on = (b0 + q) < per_wave && s < nsteps;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = m0 + i * 16 + r16;
                af[q][i] = sk_load_rowk(A, p.lda, row, on && row < p.M, k, p.K, a_vec != 0);
            }
            if constexpr (!BKROW) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = n0 + j * 16 + r16;
                    bq[q][j] = sk_load_rowk(B, p.ldb, col, on && col < p.N, k, p.K, b_vec != 0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int idx = lane + 64 * c, chunk = idx & 7, kline = idx >> 3;
                    const int kk = s * 32 + kline, col = n0 + chunk * 8;
                    bf16x8 z;
#pragma unroll
                    for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
                    if (on && kk < p.K && col < p.N) z = *reinterpret_cast<const bf16x8*>(B + (int64_t)kk * p.ldb + col);     // (host: N, ldb multiples of 8)
                    bq[q][c] = z;
                }
            }
        }
        if constexpr (BKROW) {
#pragma unroll
            for (int q = 0; q < SK_NB; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int idx = lane + 64 * c, chunk = idx & 7, kline = idx >> 3;
                    *reinterpret_cast<bf16x8*>(stage + q * SK_STEP_BYTES + kline * (SK_BN * 2) + ((chunk ^ tr_chunk_swz<SK_BN>(kline)) << 4)) = bq[q][c];
                }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < SK_NB; ++q) {
            bf16x8 bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (BKROW) bfr[j] = tr_frag<SK_BN>(stage + q * SK_STEP_BYTES, j * 16, 0, lane);
                else bfr[j] = bq[q][j];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[q][i], acc[i][j], 0, 0, 0);
        }
        if constexpr (BKROW) __syncthreads();
    }

    // the four partial tiles -> LDS; thread t then owns row t / 8, columns 8 * (t % 8) .. + 7
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(red + (wave * SK_BM + i * 16 + r16) * SK_BN + j * 16 + 4 * (lane >> 4)) = acc[i][j];
    __syncthreads();
    const int row = tid >> 3, c8 = (tid & 7) * 8;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {                            // (wave order: a fixed order)
        v0 = v0 + *reinterpret_cast<const f32x4*>(red + (w * SK_BM + row) * SK_BN + c8);
        v1 = v1 + *reinterpret_cast<const f32x4*>(red + (w * SK_BM + row) * SK_BN + c8 + 4);
    }
    const int m = m0 + row, n = n0 + c8;
    if (m >= p.M || n >= p.N) return;
    const int64_t orow = p.c_rowmap ? (int64_t)p.c_rowmap[m] : (int64_t)m;
    if (p.c_vec && n + 8 <= p.N) {
        epilogue_vec<bf16>(p, v0, m, orow, n);
        epilogue_vec<bf16>(p, v1, m, orow, n + 4);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n + e < p.N) epilogue_elem<bf16>(p, v0[e], m, orow, n + e);
            if (n + 4 + e < p.N) epilogue_elem<bf16>(p, v1[e], m, orow, n + 4 + e);
        }
    }
}
}  // namespace

// does tile 9 cover this (prepared) problem?  bf16, plain k-contiguous A, B k-contiguous or reduction-major (aligned lines), one problem, no K split
bool skinny_covers(const cb_gemm_desc* d, const GP& p) {
    if (d->dtype != CB_BF16 || d->a_mode != CB_ROWK || (d->b_mode != CB_ROWK && d->b_mode != CB_KROW)) return false;
    if (p.batch > 1 || p.split_k > 1 || d->zero_fill_pitch || d->a_rowsum || d->sq_slots || d->accumulate == 2) return false;
    if (d->b_mode == CB_KROW && !(d->N % 8 == 0 && d->ldb % 8 == 0 && aligned16(d->B))) return false;
    return true;
}

int launch_gemm_skinny(const cb_gemm_desc* d, GP& p, hipStream_t st) {
    const int a_vec = d->lda % 8 == 0 && d->K % 8 == 0 && aligned16(d->A);
    const int b_vec = d->ldb % 8 == 0 && d->K % 8 == 0 && aligned16(d->B);
    p.fast_epi = 0;
    const dim3 grid((unsigned)((d->N + SK_BN - 1) / SK_BN), (unsigned)((d->M + SK_BM - 1) / SK_BM));
    if (d->b_mode == CB_KROW) hipLaunchKernelGGL((gemm_skinny_kernel<true>), grid, dim3(256), 0, st, p, a_vec, b_vec);
    else hipLaunchKernelGGL((gemm_skinny_kernel<false>), grid, dim3(256), 0, st, p, a_vec, b_vec);
    return cb_launch_status("cb_gemm");
}
}  // namespace cbgemm
