// cb_gemm, 8-wave structure (round 3): 512 threads = two waves per SIMD, tiles 256x256 / 128x256 / 256x128 (bf16).
//
// Why a second structure.  Measured in rounds 1-2 (profiles/r01_gemm_l2_analysis.md, r02_gemm_by_shape.md): every 4-wave tile
// converges to the same global -> LDS delivery rate (~15 TB/s chip-wide), so a tile's ceiling is set by its bytes per flop,
// (BM + BN) / (BM * BN): 64x64 -> ~480 TF, 128x128 -> ~960 TF, 128x256 -> ~1280 TF, 256x256 -> ~1900 TF.  The only way up is a
// larger tile per CU, which needs (a) eight waves to own it (a 128x64 / 64x64 accumulator block per wave: <= 256 registers, two
// waves per SIMD, so that one wave's LDS reads / DMA issue hide behind the other's MFMAs), (b) operands that never pass through
// registers (LDS-DMA: buffer_load ... lds, 1 KiB per wave instruction, whole 128-byte lines), (c) K tiles in flight ACROSS
// barriers (counted s_waitcnt vmcnt(N), raw s_barrier), and (d) enough blocks: the caller splits K (fp32 partial slabs in a
// workspace + cb_gemm's reduce kernel, deterministic) so that large tiles still cover the 256 CUs.
//
// Schedules (template parameter MODE), all over an NST-deep LDS ring of K tiles (BK = 64):
//   0  one barrier per K tile; DMA of tile t+NST-1 issued right behind it; reads + MFMAs compiler-scheduled (the two waves of a
//      SIMD drift apart inside the tile and cover each other);
//   1  the K tile in P phases {ds_read fragments, [DMA slice], barrier, MFMA cluster (s_setprio 1), barrier}: lockstep;
//   2  as 1 with the two wave groups (waves 0-3 / 4-7 = one wave of every SIMD each) offset by one barrier: while one group runs
//      its MFMA cluster the other reads / issues DMA (the ping-pong of cdna_hip_programming.md section 5).
//   (A persistent variant -- one workgroup per CU walking the tiles, next tile's DMA issued before the epilogue -- was built in round 3,
//   measured slower on every shape, profiles/r03r_persistent_probe.txt, and deleted in round 6.)
// Hazards (MODE 1 / 2): a stage is re-filled >= 2 phases after its last ds_read (the groups are one barrier apart and a read is
// only complete at its consumer's lgkmcnt wait); a tile is read one phase AFTER the counted vmcnt + barrier that retire it.
#pragma once
#include "gemm_impl.h"

namespace cbgemm {

constexpr int NT8 = 512;

// ---------------------------------------------------------------------------------------------
// ROWK operand: image [ROWS][128 B], segment s of row r at segment s ^ (r & 7) (lds_off<bf16>).  DMA instruction i of wave w
// fills rows (i*8 + w)*8 .. +7 (1 KiB): lane -> (row & 7 = lane >> 3, physical segment = lane & 7), source = logical segment.
// ---------------------------------------------------------------------------------------------
template <int ROWS, bool GATHER> struct RowkDma8 {
    using X = Tr<bf16>;
    static constexpr bool TR = false;
    static constexpr int NI = ROWS / 64;
    static constexpr int TILE_BYTES = ROWS * 128;
    rsrc_t rs;
    uint32_t voff[NI];
    int ih[GATHER ? NI : 1], iw[GATHER ? NI : 1];
    int c, rr, ss, krem;

    __device__ __forceinline__ void init(const GP& p, const Opnd& o, int row0, int bound, int kt0, int lane, int wave) {
        rs = make_rsrc(o.base, o.bytes);
        const int rin = lane >> 3;
        const int lseg = (lane & 7) ^ rin;
        const int k = kt0 * X::BK + lseg * 8;
        c = k; rr = 0; ss = 0; krem = p.K - k;
        if constexpr (GATHER) {
            const int tap = k / p.Ct;
            c = k - tap * p.Ct;
            rr = tap / p.S; ss = tap - rr * p.S;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = row0 + (i * 8 + wave) * 8 + rin;
            const bool ok = row < bound;
            if constexpr (GATHER) {
                cb_pixel px = {0, 0, 0};
                if (ok) px = o.tab[row];
                ih[i] = px.ih0; iw[i] = px.iw0;
                voff[i] = ok ? (uint32_t)px.off * 2u : OOB;
            } else {
                voff[i] = ok ? ((uint32_t)row * (uint32_t)o.ld + (uint32_t)k) * 2u : OOB;
            }
        }
    }
    // instruction i of the current K tile (i is a compile-time constant after unrolling)
    __device__ __forceinline__ void issue_one(const GP& p, unsigned char* tile, int wave, int i) {
        uint32_t o32;
        if constexpr (GATHER) {
            const bool v = rr < p.R && (unsigned)(ih[i] + rr) < (unsigned)p.H && (unsigned)(iw[i] + ss) < (unsigned)p.W;
            o32 = v ? voff[i] + (uint32_t)(rr * (int)p.sH + ss * (int)p.sW + c) * 2u : OOB;
        } else {
            o32 = krem > 0 ? voff[i] : OOB;                    // (K % 8 == 0 on this path: a segment is all-in or all-out)
        }
        dma16(rs, tile + (i * 8 + wave) * 1024, o32);
    }
    __device__ __forceinline__ void advance(const GP& p) {     // to the next K tile
        if constexpr (GATHER) {
            if (p.Ct >= X::BK) {
                c += X::BK;
                if (c >= p.Ct) { c -= p.Ct; if (++ss == p.S) { ss = 0; ++rr; } }
            } else {
                ss += X::BK / p.Ct;
                while (ss >= p.S) { ss -= p.S; ++rr; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) voff[i] += X::BK * 2u;          // (OOB + n*128 stays out of range: K * 2 < 2 GiB)
            krem -= X::BK;
        }
    }
    __device__ __forceinline__ bf16x8 frag(const unsigned char* tile, int r0, int kk, int lane) const {
        return *reinterpret_cast<const bf16x8*>(tile + lds_off<bf16>(r0 + (lane & 15), kk * 4 + (lane >> 4)));
    }
};

// ---------------------------------------------------------------------------------------------
// KROW operand (reduction index outermost in memory): ROWS / 128 panels, each the natural image [64 k-lines][256 B] of
// KrowTr<128> (16-byte chunk c of line k at chunk c ^ tr_chunk_swz<128>(k)), read with ds_read_b64_tr_b16.  DMA instruction
// i = 2 * panel + sub of wave w fills lines (sub*8 + w)*4 .. +3 of its panel: lane -> (line = lane >> 4, physical chunk = lane & 15).
// ---------------------------------------------------------------------------------------------
template <int ROWS, int KMODE> struct KrowDma8 {
    using X = Tr<bf16>;
    static constexpr bool TR = true;
    static constexpr int NP = ROWS / 128;
    static constexpr int NI = 2 * NP;
    static constexpr int TILE_BYTES = ROWS * 128;
    rsrc_t rs;
    const cb_pixel* tab;
    uint32_t ldb, bound;
    uint32_t voff[NI];
    int kl[2];                               // global k of this lane's line, per sub (the panels share it)
    int co[KMODE == KM_TAPS ? 2 : 1], tap[KMODE == KM_TAPS ? 2 : 1];
    int rr[KMODE == KM_GATHER ? NI : 1], ss[KMODE == KM_GATHER ? NI : 1];
    cb_pixel px[KMODE == KM_GATHER ? 2 : 1];  // GATHER: table entry of this lane's line of the NEXT tile to issue

    __device__ __forceinline__ void init(const GP& p, const Opnd& o, int row0, int bnd, int kt0, int lane, int wave) {
        rs = make_rsrc(o.base, o.bytes);
        tab = o.tab; ldb = (uint32_t)o.ld * 2u; bound = (uint32_t)bnd;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kline = (sub * 8 + wave) * 4 + (lane >> 4);
            kl[sub] = kt0 * X::BK + kline;
            if constexpr (KMODE == KM_TAPS) {
                tap[sub] = kl[sub] / p.Ct;
                co[sub] = kl[sub] - tap[sub] * p.Ct;
            }
            if constexpr (KMODE == KM_GATHER) {
                cb_pixel e = {0, (int16_t)-30000, (int16_t)-30000};
                if (kl[sub] < p.K) e = tab[kl[sub]];
                px[sub] = e;
            }
#pragma unroll
            for (int pn = 0; pn < NP; ++pn) {
                const int i = 2 * pn + sub;
                const int lchunk = (lane & 15) ^ tr_chunk_swz<128>(kline);
                const int row = row0 + pn * 128 + lchunk * 8;
                const bool ok = row < bnd;
                if constexpr (KMODE == KM_PLAIN) {
                    voff[i] = ok ? ((uint32_t)kl[sub] * (uint32_t)o.ld + (uint32_t)row) * 2u : OOB;
                } else if constexpr (KMODE == KM_TAPS) {
                    voff[i] = ok ? (uint32_t)row * 2u : OOB;
                } else {                                     // row = (tap, channel) of the gathered image; 8 rows share a tap
                    const int tp = row / p.Ct, ch = row - tp * p.Ct;
                    rr[i] = tp / p.S; ss[i] = tp - rr[i] * p.S;
                    voff[i] = ok ? (uint32_t)(rr[i] * (int)p.sH + ss[i] * (int)p.sW + ch) * 2u : OOB;
                }
            }
        }
    }
    __device__ __forceinline__ void issue_one(const GP& p, unsigned char* tile, int wave, int i) {
        const int pn = i >> 1, sub = i & 1;
        uint32_t o32;
        if constexpr (KMODE == KM_PLAIN) {
            o32 = (kl[sub] < p.K) ? voff[i] : OOB;
        } else if constexpr (KMODE == KM_TAPS) {
            const int tapw = p.flip ? (p.R * p.S - 1 - tap[sub]) : tap[sub];
            const bool v = tap[sub] < p.R * p.S && voff[i] != OOB;
            o32 = v ? (uint32_t)co[sub] * ldb + (uint32_t)tapw * bound * 2u + voff[i] : OOB;
        } else {
            const cb_pixel e = px[sub];
            const bool v = voff[i] != OOB && (unsigned)(e.ih0 + rr[i]) < (unsigned)p.H && (unsigned)(e.iw0 + ss[i]) < (unsigned)p.W;
            o32 = v ? (uint32_t)e.off * 2u + voff[i] : OOB;
        }
        dma16(rs, tile + pn * 16384 + (sub * 8 + wave) * 1024, o32);
    }
    __device__ __forceinline__ void advance(const GP& p) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            kl[sub] += X::BK;
            if constexpr (KMODE == KM_TAPS) {
                co[sub] += X::BK;
                while (co[sub] >= p.Ct) { co[sub] -= p.Ct; ++tap[sub]; }
            }
            if constexpr (KMODE == KM_GATHER) {
                cb_pixel e = {0, (int16_t)-30000, (int16_t)-30000};
                if (kl[sub] < p.K) e = tab[kl[sub]];
                px[sub] = e;
            }
        }
        if constexpr (KMODE == KM_PLAIN) {
#pragma unroll
            for (int i = 0; i < NI; ++i) voff[i] += X::BK * ldb;       // (OOB + n * BK * ld stays >= 2 GiB: K * ld * 2 < 2 GiB)
        }
    }
    __device__ __forceinline__ bf16x8 frag(const unsigned char* tile, int r0, int kk, int lane) const {
        return tr_frag<128>(tile + (r0 >> 7) * 16384, r0 & 127, kk, lane);
    }
};

// ---------------------------------------------------------------------------------------------
// Epilogue for a WGM x WGN grid of 8 waves.  The fp32 accumulators pass through LDS 64 tile rows at a time so that a thread owns
// 8 consecutive columns of one row (16-byte, line-contiguous global accesses), then either the full cb_gemm epilogue (epilogue8)
// or -- K-split partial products -- plain fp32 stores into this split's slab of the workspace.
// ---------------------------------------------------------------------------------------------
// Code size (round 5, profiles/r05a_stamps.md): the epilogue8 body is ~10 KB of branchy straight-line code (every epilogue option is a
// wave-uniform runtime branch).  Fully unrolled -- BM / PR passes x ITER chunks = 8 copies -- the epilogue was 78 KB that each workgroup
// executes ONCE, i.e. entirely out of instruction-cache misses: ~1.1 us per chunk, 9 us for a bias-only 128x256 tile, 17 us for FFN1's
// GELU + two outputs (longer than its 11 us K loop).  The pass loop and the chunk loop are therefore ROLLED: one copy of the body, warm
// after its first trip.  Only the accumulator -> LDS staging needs compile-time register indices: a switch over the WM / PR row blocks.
template <int BM, int BN, int WGM, int WGN, int SMEM_BYTES, int PR = 64, bool WG = false, int FORM = 0>
__device__ __forceinline__ void tile_epilogue8w(GP& p, f32x4 (&acc)[BM / WGM / 16][BN / WGN / 16], unsigned char* smem, int m0, int n0,
                                                int tid, float* slab, int tile_lin = 0) {
    using T = bf16;
    {
        // (not on the 256x256 forward / data-gradient kernels: with the sixteen bodies compiled in, that instantiation -- 9 k instructions
        // instead of 3 k -- ran EVERY epilogue, the generic one included, 1.5-1.7x slower per launch: 10496 x 2304 x 768 + bias 65 -> 100 us,
        // profiles/r06u_tile5_epilogue_bodies.txt; the metric step has no such launch, the inference row does.  Its weight-gradient
        // instantiation holds one body only, the fp32 store with the norm share.)
        if constexpr (!(BM == 256 && BN == 256) || WG)
        if (!slab && p.fast_epi != 0 && (WG == (p.fast_epi == FAST_EPI_F32)) && fe_in_form(p.fast_epi, FORM)) {                         // specialised body for this call's option combination (gemm_impl.h fast_epilogue)
            constexpr int WM_ = BM / WGM, WN_ = BN / WGN, FN_ = WN_ / 16, SROW_ = BN * 4 + 16, SUB_ = WM_ / PR;
            static_assert(WM_ % PR == 0 && PR * SROW_ <= SMEM_BYTES, "epilogue staging");
            const int lane_ = tid & 63, wave_ = tid >> 6;
            const int wm_ = wave_ / WGN, wn_ = wave_ % WGN;
            if (p.dropout_p > 0.f && p.seed_ptr) p.seed += *p.seed_ptr;
            unsigned char* const sb_ = smem + (lane_ & 15) * SROW_ + (wn_ * WN_ + 4 * (lane_ >> 4)) * 4;
            auto stage = [&](int h) __attribute__((always_inline)) {
                if (wm_ == h / SUB_) {
                    const int sub = h % SUB_;
#pragma unroll
                    for (int sb = 0; sb < SUB_; ++sb) {
                        if (sb == sub) {
#pragma unroll
                            for (int i = 0; i < PR / 16; ++i)
#pragma unroll
                                for (int j = 0; j < FN_; ++j)
                                    *reinterpret_cast<f32x4*>(sb_ + i * 16 * SROW_ + j * 64) = acc[sb * (PR / 16) + i][j];
                        }
                    }
                }
            };
            fast_epilogue_dispatch<NT8, BN, PR, BM / PR, WG, FORM>(p, smem, m0, n0, tid, stage, nullptr, nullptr, tile_lin);
            return;
        }
    }
    constexpr int WM = BM / WGM, WN = BN / WGN, FN = WN / 16;   // PR: tile rows per pass through the staging area
    constexpr int SROW = BN * 4 + 16;
    constexpr int CPR = BN / 8, ITER = PR * CPR / NT8;
    constexpr int NPASS = BM / PR, SUB = WM / PR;              // passes; passes per wave row block
    static_assert(WM % PR == 0 && PR * SROW <= SMEM_BYTES && PR * CPR % NT8 == 0 && NT8 % CPR == 0, "epilogue staging");
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    if (p.dropout_p > 0.f && p.seed_ptr) p.seed += *p.seed_ptr;
    const int cc = tid % CPR, n = n0 + cc * 8;
    const bool nok = n < p.N;
    float sc[8], sh[8];
    if (!slab) {
        if (p.scale && nok) load8(p.scale + n, sc);
        if (p.shift && nok) load8(p.shift + n, sh);
    }
    unsigned char* const stage_base = smem + (lane & 15) * SROW + (wn * WN + 4 * (lane >> 4)) * 4;
#pragma unroll 1
    for (int h = 0; h < NPASS; ++h) {
        __syncthreads();
        if (wm == h / SUB) {
            const int sub = h % SUB;
#pragma unroll
            for (int sb = 0; sb < SUB; ++sb) {
                if (sb == sub) {                                 // (compile-time accumulator indices inside each case)
#pragma unroll
                    for (int i = 0; i < PR / 16; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            *reinterpret_cast<f32x4*>(stage_base + i * 16 * SROW + j * 64) = acc[sb * (PR / 16) + i][j];
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < ITER; ++it) {
            const int rl = (tid + it * NT8) / CPR;
            const int m = m0 + h * PR + rl;
            if (m < p.M && nok) {
                float v[8];
                load8(reinterpret_cast<const float*>(smem + rl * SROW + cc * 32), v);
                if (slab) {
                    store8(slab + (int64_t)m * p.N + n, v);
                } else {
                    const int64_t orow = p.c_rowmap ? (int64_t)p.c_rowmap[m] : (int64_t)m;
                    epilogue8<T>(p, v, sc, sh, m, orow, n);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The kernel.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, int NST, int MODE, typename LA, typename LB, bool RS>
__global__ void __launch_bounds__(NT8, 2) gemm8_kernel(GP p, float* ws) {
    using T = bf16;
    constexpr int BK = 64;
    static_assert(WGM * WGN == 8, "eight waves");
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
    constexpr int TILE_A = BM * 128, TILE_B = BN * 128, STAGE = TILE_A + TILE_B;
    constexpr int SMEM_BYTES = NST * STAGE;
    constexpr int LPT = LA::NI + LB::NI;                       // DMA instructions per wave per K tile
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
    // phases of a K tile (MODE 1 / 2): k32 halves x 64-row halves of the wave's accumulator block
    constexpr int MS = FM > 4 ? 2 : 1, FMH = FM / MS, P = 2 * MS;
    constexpr int D = NST - 1;                                 // K tiles issued ahead of the one being read
    // MODE 1 / 2: the DMA of tile t+D is spread over phases 1 .. PL of tile t.  With a single tile in flight (D == 1) the wait in
    // phase P-1 covers what was just issued, so the last phase issues nothing (the transfers get a whole phase to land)
    constexpr int PL = (D == 1 && P >= 3) ? P - 2 : P - 1;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    CB_STAMP_DECL();
    CB_STAMP(0);
    TileId bid = tile_id(p);
#ifdef CB_STAMPS
    const unsigned stamp_lin = (unsigned)bid.bx + gridDim.x * ((unsigned)bid.by + gridDim.y * (unsigned)bid.bz);
#endif
    const int zsplit = bid.bz % p.split_k, zbatch = bid.bz / p.split_k;      // (before apply_batch rewrites bz)
    apply_batch(p, bid);
    const int m0 = bid.by * BM, n0 = bid.bx * BN;
    const int kt_per = (p.ktiles + p.split_k - 1) / p.split_k;
    const int kt0 = bid.bz * kt_per;
    const int nt = ((kt0 + kt_per < p.ktiles) ? kt0 + kt_per : p.ktiles) - kt0;
    float* slab = ws ? ws + ((int64_t)zbatch * p.split_k + zsplit) * (int64_t)p.M * p.N : nullptr;
    if (nt <= 0 && !slab) return;                              // (a slab must be written even when its K range is empty)

    LA la;
    LB lb;
    {
        Opnd oa = {p.A, p.a_tab, p.lda, p.a_mode, p.a_bytes};
        Opnd ob = {p.B, p.b_tab, p.ldb, p.b_mode, p.b_bytes};
        la.init(p, oa, m0, p.M, kt0, lane, wave);
        lb.init(p, ob, n0, p.N, kt0, lane, wave);
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
    f32x4 accr[RS ? FM : 1];
#pragma unroll
    for (int i = 0; i < (RS ? FM : 1); ++i) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; accr[i] = z; }
    const bool rs_on = RS && p.a_rowsum != nullptr && bid.bx == 0 && wn == 0;      // wave-uniform
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    // DMA instructions [lo, hi) of the next K tile to issue, into `stage` (A's instructions first, then B's)
    auto issue_range = [&](int stage, int lo, int hi) __attribute__((always_inline)) {
        unsigned char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            if (i >= lo && i < hi) {
                if (i < LA::NI) {
                    la.issue_one(p, base, wave, i);
                    if (i == LA::NI - 1) la.advance(p);
                } else {
                    lb.issue_one(p, base + TILE_A, wave, i - LA::NI);
                    if (i == LPT - 1) lb.advance(p);
                }
            }
        }
    };

    if constexpr (MODE == 0) {
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (s < nt) issue_range(s, 0, LPT);
        int stage = 0;
        for (int t = 0; t < nt; ++t) {
            if (nt - 1 - t >= D - 1) { CB_WAIT_VMCNT(LPT * (D - 1)); }
            else { CB_WAIT_VMCNT(0); }
            __builtin_amdgcn_s_barrier();             // every wave's share of tile t landed; everyone is done reading tile t-1
#ifdef CB_STAMPS
            if (t == 0) CB_STAMP(1);
#endif
            if (t + D < nt) {
                int ns = stage + D;
                if (ns >= NST) ns -= NST;
                issue_range(ns, 0, LPT);
            }
            const unsigned char* As = smem + stage * STAGE;
            const unsigned char* Bs = As + TILE_A;
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                bf16x8 af[FM], bfr[FN];
#pragma unroll
                for (int j = 0; j < FN; ++j) bfr[j] = lb.frag(Bs, wn * WN + j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < FM; ++i) af[i] = la.frag(As, wm * WM + i * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                if constexpr (RS) {
                    if (rs_on) {
#pragma unroll
                        for (int i = 0; i < FM; ++i) accr[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accr[i], 0, 0, 0);
                    }
                }
            }
            if (++stage == NST) stage = 0;
        }
    } else {
        const bool late = (MODE == 2) && wave >= 4;      // the wave group that runs one barrier behind
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (s < nt) issue_range(s, 0, LPT);
        if (nt - 1 >= D - 1) { CB_WAIT_VMCNT(LPT * (D - 1)); }
        else { CB_WAIT_VMCNT(0); }
        __builtin_amdgcn_s_barrier();                     // tile 0 landed
        CB_STAMP(1);
        if (late) __builtin_amdgcn_s_barrier();
        int stage = 0;
        for (int t = 0; t < nt; ++t) {
            const unsigned char* As = smem + stage * STAGE;
            const unsigned char* Bs = As + TILE_A;
            int ns = stage + D;
            if (ns >= NST) ns -= NST;
            const bool more = t + D < nt;
            bf16x8 af[FMH], bfr[FN];
#pragma unroll
            for (int ph = 0; ph < P; ++ph) {
                const int kk = ph / MS, mh = ph % MS;
                if (mh == 0) {
#pragma unroll
                    for (int j = 0; j < FN; ++j) bfr[j] = lb.frag(Bs, wn * WN + j * 16, kk, lane);
                }
#pragma unroll
                for (int i = 0; i < FMH; ++i) af[i] = la.frag(As, wm * WM + (mh * FMH + i) * 16, kk, lane);
                // the stage being re-filled held tile t-1, last read in its phase P-1: two phases ago from phase 1 on
                if (ph >= 1 && ph <= PL && more) issue_range(ns, (ph - 1) * LPT / PL, ph * LPT / PL);
                if (ph == P - 1) {                         // tile t+1 (issued during tile t-1) must have landed before phase 0 of t+1
                    if (more) { CB_WAIT_VMCNT(LPT * (D - 1)); }
                    else { CB_WAIT_VMCNT(0); }
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < FMH; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[mh * FMH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[mh * FMH + i][j], 0, 0, 0);
                if constexpr (RS) {
                    if (rs_on) {
#pragma unroll
                        for (int i = 0; i < FMH; ++i)
                            accr[mh * FMH + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accr[mh * FMH + i], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            if (++stage == NST) stage = 0;
        }
        if ((MODE == 2) && !late) __builtin_amdgcn_s_barrier();      // arrival counts of the two groups equal again
    }

    if constexpr (RS) {
        if (rs_on && (lane >> 4) == 0) {                 // every accumulator row holds the sum: take row 0 of lanes 0..15
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * WM + i * 16 + lane;
                if (m < p.M) atomicAdd(p.a_rowsum + m, accr[i][0]);
            }
        }
    }
    CB_STAMP(2);
    tile_epilogue8w<BM, BN, WGM, WGN, SMEM_BYTES, 64, LA::TR, (LA::TR ? 0 : (LB::TR ? 2 : 1))>(p, acc, smem, m0, n0, tid, slab, (zbatch * (int)gridDim.y + bid.by) * (int)gridDim.x + bid.bx);
    CB_STAMP(3);
    CB_STAMP_FLUSH(p, stamp_lin, tid);
}


// ---------------------------------------------------------------------------------------------
// launchers (explicitly instantiated per tile in gemm8_inst_*.hip so that the tiles compile in parallel)
// ---------------------------------------------------------------------------------------------
#define CB_G8_LAUNCH(LA_, LB_, RS_)                                                                                        \
    do {                                                                                                                   \
        if (mode == 0) hipLaunchKernelGGL((gemm8_kernel<BM, BN, WGM, WGN, NST, 0, LA_, LB_, RS_>), grid, dim3(NT8), 0, st, p, ws);      \
        else if (mode == 1) hipLaunchKernelGGL((gemm8_kernel<BM, BN, WGM, WGN, NST, 1, LA_, LB_, RS_>), grid, dim3(NT8), 0, st, p, ws); \
        else hipLaunchKernelGGL((gemm8_kernel<BM, BN, WGM, WGN, NST, 2, LA_, LB_, RS_>), grid, dim3(NT8), 0, st, p, ws);         \
        return cb_launch_status("cb_gemm");                                                                                \
    } while (0)

template <int BM, int BN>
inline dim3 gemm8_grid(const GP& p) {
    return dim3((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.split_k * (p.batch > 1 ? p.batch : 1));
}

// forward forms: A ROWK | ROWK_GATHER, B ROWK
template <int BM, int BN, int WGM, int WGN, int NST>
int launch_gemm8_fwd(const GP& p, int mode, float* ws, hipStream_t st) {
    const dim3 grid = gemm8_grid<BM, BN>(p);
    using RA0 = RowkDma8<BM, false>; using RA1 = RowkDma8<BM, true>; using RB0 = RowkDma8<BN, false>;
    if (p.a_mode == CB_ROWK) CB_G8_LAUNCH(RA0, RB0, false);
    CB_G8_LAUNCH(RA1, RB0, false);
}
// data-gradient forms: A ROWK | ROWK_GATHER, B KROW | KROW_TAPS
template <int BM, int BN, int WGM, int WGN, int NST>
int launch_gemm8_dgrad(const GP& p, int mode, float* ws, hipStream_t st) {
    const dim3 grid = gemm8_grid<BM, BN>(p);
    using RA0 = RowkDma8<BM, false>; using RA1 = RowkDma8<BM, true>;
    using KB0 = KrowDma8<BN, KM_PLAIN>; using KB1 = KrowDma8<BN, KM_TAPS>;
    if (p.a_mode == CB_ROWK) CB_G8_LAUNCH(RA0, KB0, false);
    CB_G8_LAUNCH(RA1, KB1, false);
}
// weight-gradient forms: A KROW, B KROW | KROW_GATHER
template <int BM, int BN, int WGM, int WGN, int NST>
int launch_gemm8_wgrad(const GP& p, int mode, float* ws, hipStream_t st) {
    const dim3 grid = gemm8_grid<BM, BN>(p);
    using KA0 = KrowDma8<BM, KM_PLAIN>; using KB0 = KrowDma8<BN, KM_PLAIN>; using KB2 = KrowDma8<BN, KM_GATHER>;
    if (p.b_mode == CB_KROW_GATHER) CB_G8_LAUNCH(KA0, KB2, false);
    if (p.a_rowsum) CB_G8_LAUNCH(KA0, KB0, true);
    CB_G8_LAUNCH(KA0, KB0, false);
}

}  // namespace cbgemm
