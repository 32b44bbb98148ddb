/* clipbert_hip.h -- C ABI of libclipbert_hip.so, the MI355X (gfx950) implementation of the ClipBERT
 * forward/backward hot path.
 *
 * The reference (jayleicn/ClipBERT) has no FFI / plugin registry of its own: its seam is the Python
 * nn.Module API of src/modeling.  Each entry point below therefore names the
 * reference function(s) whose arithmetic it replaces; clipbert_amd/modeling.py keeps the reference's
 * module names / signatures / state-dict keys and reaches these entry points through ctypes
 * (INTEGRATION.md shows the binding a maintainer of the reference would add).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer borrowed from the caller (torch.Tensor.data_ptr()); nothing is
 *    allocated, retained or freed inside; no torch types appear in any signature;
 *  - kernels are enqueued on `stream` (a hipStream_t passed as void*) and the call returns at once;
 *  - return 0 on success, negative on error (cb_last_error() gives a thread-local message); nothing
 *    throws across the ABI;
 *  - `dtype` selects the storage / MFMA input type of activations and compute-weights:
 *    CB_BF16 (performance mode, bf16 in, fp32 accumulate on v_mfma_f32_16x16x32_bf16) or
 *    CB_F32 (parity mode, exact fp32 on v_mfma_f32_16x16x4_f32).  Statistics, softmax, losses,
 *    gradients of parameters and optimizer state are always fp32;
 *  - activations are NHWC / token-major; conv weights are KRSC ([Cout][R][S][Cin], i.e. the
 *    channels_last memory image of the reference's OIHW nn.Parameter); Linear weights are (out,in).
 */
#ifndef CLIPBERT_HIP_H
#define CLIPBERT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CB_F32 = 0, CB_BF16 = 1 };
enum { CB_SPLITK_WS_COUNTER_BYTES = 65536 };   /* tail of cb_gemm_desc.splitk_ws reserved for arrival counters (see there) */
enum { CB_ACT_NONE = 0, CB_ACT_RELU = 1, CB_ACT_GELU = 2, CB_ACT_TANH = 3,
       CB_ACT_GELU_SAVE_GRAD = 4,   /* cb_gemm: GELU; the second output C2, if given, receives gelu'(pre-activation) INSTEAD of the
                                       pre-activation: what the backward needs of it, computed where exp / erfc are at hand      */
       CB_ACT_SAVED_GRAD = 5 };      /* cb_gemm: no activation; gelu_grad_pre holds gelu'(pre) as stored by a CB_ACT_GELU_SAVE_GRAD
                                       launch: the result is multiplied by it as it is (no erf / exp in the backward epilogue)   */

/* operand addressing modes of cb_gemm (reduction index = k) */
enum {
    CB_ROWK = 0,        /* element (row,k) at base + row*ld + k            (k contiguous)            */
    CB_ROWK_GATHER = 1, /* row -> pixel via table; k = (tap, c): base + tab[row].off + r*sH + s*sW + c */
    CB_KROW = 2,        /* element (row,k) at base + k*ld + row            (row contiguous)           */
    CB_KROW_TAPS = 3,   /* weights KRSC read for dgrad: k = (tap, co): base + co*ld + tap'*Cin + row  */
    CB_KROW_GATHER = 4  /* k -> pixel via table; row = (tap, c): base + tab[k].off + r*sH + s*sW + c  */
};

/* one entry per output pixel of a convolution, built by cb_build_pixel_table */
typedef struct { int32_t off; int16_t ih0; int16_t iw0; } cb_pixel;

/* Describes C[M,N] (op)= epilogue( sum_k A(m,k) * B(n,k) ).  All strides in ELEMENTS. */
typedef struct {
    int32_t dtype;            /* CB_F32 | CB_BF16 : type of A, B, residual, mask, C (unless c_f32)     */
    int32_t M, N, K;          /* K = full reduction length (taps * Cin for convolutions)              */
    int32_t a_mode, b_mode;
    const void* A; const void* B;
    int64_t lda, ldb;
    const cb_pixel* a_tab;    /* CB_ROWK_GATHER: M entries */
    const cb_pixel* b_tab;    /* CB_KROW_GATHER: K entries */
    /* tap geometry shared by the gather / taps modes */
    int32_t R, S, Cin;        /* K (or N for KROW_GATHER) = R*S*Cin                                  */
    int32_t H, W;             /* bounds of the gathered image                                        */
    int64_t sH, sW;           /* element strides of one image row / one pixel                        */
    int32_t flip_taps;        /* CB_KROW_TAPS: read weight tap (R-1-r, S-1-s) (transposed conv)      */
    int32_t schedule;         /* tiles 5-7 only: 0 = default K-loop schedule, 1 / 2 / 3 = force schedule 0 / 1 / 2 of
                                 csrc/gemm8_impl.h (diagnostic: tools/gemm8_probe.py, tests) */
    /* output */
    void* C; int64_t ldc;
    const int32_t* c_rowmap;  /* optional: output row m is written at row c_rowmap[m]                */
    int32_t c_f32;            /* 1: C is fp32 regardless of dtype (parameter gradients, logits)      */
    int32_t accumulate;       /* 1: C += result (plain RMW; with split_k > 1: fp32 atomics, row-coalesced);
                                 2: FIRST WRITER -- C holds nothing of value and receives the result (no zero fill before the launch,
                                 no read-modify-write in it), and the K split stays the library's to choose as with 1, but only
                                 through the slab scratch (ordered in-launch reduce): without a scratch the problem runs unsplit */
    int32_t split_k;          /* >= 1; > 1 requires c_f32 && accumulate semantics (C pre-initialised) */
    int32_t act;              /* CB_ACT_* applied before the residual add                            */
    const float* scale;       /* optional per-n multiplier (FrozenBN scale)                          */
    const float* shift;       /* optional per-n addend (bias / FrozenBN shift)                       */
    const void* residual; int64_t ldr;    /* optional, added after act                               */
    int32_t relu_after;       /* ReLU after the residual add (ResNet block output)                    */
    int32_t zero_fill_pitch;  /* > 0 with c_rowmap: the rows orow+1, orow+pitch, orow+pitch+1 of C (and C2) are written
                               * with ZEROS -- data gradient of a stride-2 1x1 convolution: output pixel m owns a 2x2 patch
                               * of input pixels of which only the first receives a value (pitch = input row width W);
                               * needs 16-byte-aligned 8-column chunks (N, ldc % 8 == 0)                             */
    const void* mask; int64_t ldm;        /* optional: result zeroed where mask[m,n] <= 0 (ReLU bwd)  */
    void* C2; int64_t ldc2;   /* optional second output: value BEFORE act (GELU backward needs it)   */
    float alpha;              /* multiplies the accumulator first (1.0 if 0)                         */
    float dropout_p;          /* > 0: inverted dropout on the value before the residual add          */
    uint64_t dropout_seed;    /* mask = f(dropout_seed + *dropout_seed_ptr, m*N + n)                  */
    const uint64_t* dropout_seed_ptr;  /* optional DEVICE word added to the seed (varies per hipGraph replay) */
    int32_t tile;             /* 0 auto (tuned table, then heuristics), 1 = 128x128, 2 = 64x64, 3 = 128x64,
                                 4 = 128x128 with its registers capped so that two blocks share a CU  (bf16; fp32 parity mode
                                 always runs 64x64); 5 = 256x256, 6 = 128x256, 7 = 256x128: the 8-wave LDS-DMA kernels
                                 (bf16, 16-byte-aligned operands and 8-column output chunks; anything else falls back
                                 to auto); 8 = the streaming structure for HBM-bound products with K <= 256 and 10^4+
                                 rows (persistent workgroups, weights resident in LDS, next A tile and the epilogue's
                                 operands in flight while a tile is stored: the ResNet's 1x1 convolutions of res2 / res3
                                 and their data gradients; an error where it does not cover the problem); 9 = few rows
                                 (bf16, A k-contiguous, B k-contiguous or 16-byte-aligned reduction-major, one problem): 32 x 64
                                 output tiles whose four waves SPLIT the reduction and meet in LDS -- taken by itself for
                                 M <= 64 (the heads' products and their data gradients); an error where it does not cover the problem */
    int32_t xcd_order;        /* workgroup -> tile order: 0 auto, 1 = XCD-compact (each XCD, with its own L2, owns a
                                 contiguous run of tiles), 2 = dispatch order (consecutive tiles round-robin over XCDs) */
    int64_t a_bytes, b_bytes; /* sizes of the A / B buffers in bytes (0 = unknown).  When both are known, < 2 GiB
                                 and every row / tap start is 16-byte aligned the kernel uses range-checked
                                 buffer loads (fast path); otherwise element-wise guarded loads.           */
    const void* gelu_grad_pre; int64_t ld_gelu;  /* optional: result *= gelu'(pre[m,n]) (last step before the store):
                                 the GELU backward of BertIntermediate fused into the dgrad of BertOutput.dense;
                                 with act == CB_ACT_SAVED_GRAD the tensor already holds gelu'(pre): result *= it  */
    float* a_rowsum;          /* optional, weight-gradient form only (A and B CB_KROW): a_rowsum[m] += sum_k A(m,k),
                                 i.e. the bias gradient colsum(dY), computed on the matrix core next to dW (atomics) */
    int32_t batch;            /* > 1: `batch` independent problems of this shape in ONE launch (grid z); problem b uses
                                 A + b*batch_stride_a, B + b*batch_stride_b, C + b*batch_stride_c (elements of their
                                 types) and a_rowsum + b*batch_stride_rowsum.  Plain operands only.  Used for the weight
                                 gradients of all encoder layers at once (same shapes, layer-strided buffers).       */
    int32_t relu_bwd;         /* 1: ResNet-block backward epilogue (needs `mask` = the block's output y):
                                 t = (acc [+ C if accumulate] [+ residual]) where mask > 0, else 0;
                                 C2 = t * post_scale2[n] (t if null),  C = t * post_scale[n] (t if null).
                                 One dgrad launch thereby also does the ReLU x FrozenBN-scale backward of the block
                                 that consumes its result (F.relu_ + FrozenBatchNorm2d under autograd, grid_feat.py:95) */
    int64_t batch_stride_a, batch_stride_b, batch_stride_c, batch_stride_rowsum;
    const float* post_scale; const float* post_scale2;
    void* splitk_ws;          /* optional scratch for the K split of tiles 5-7: split_k * batch * M * N fp32 partial products
                                 (one slab per split), summed in index order by a second kernel that applies the whole
                                 epilogue -- deterministic, no atomics, every epilogue allowed.  Too small / null: no split.
                                 cb_gemm_group uses the same buffer for the split weight gradients of a group (see there).
                                 LAYOUT: the LAST CB_SPLITK_WS_COUNTER_BYTES bytes of the buffer are the arrival counters of the
                                 in-launch K-split reduce; the caller zeroes them ONCE when it allocates the buffer (cb_zero), every
                                 launch leaves them zero, and no launch puts partial products there (the payload is
                                 splitk_ws_bytes - CB_SPLITK_WS_COUNTER_BYTES).  The buffer -- partial products AND counters -- belongs
                                 to the launches of ONE stream (or of streams ordered by events): launches that may run concurrently
                                 must carry different buffers.  The library itself allocates nothing and keeps no such state. */
    int64_t splitk_ws_bytes;
    float* sq_slots;          /* optional, bf16 problems that STORE an fp32 C (accumulate 0 / 2; plain epilogue: weight gradients): every
                                 output tile also stores the sum of the squares of what it wrote -- its share of the squared gradient
                                 norm (torch.nn.utils.clip_grad_norm_, run_video_retrieval.py:477-482) -- to one slot of its own,
                                 sq_slots[t], t < sq_slots_n; the launch writes ceil(M / BM) * ceil(N / BN) * batch of them for the tile
                                 size it picked and leaves the rest untouched, so the caller zeroes the range once per step, reserves
                                 sq_slots_n >= ceil(M / 64) * ceil(N / 64) * batch, and adds the slots up in index order
                                 (cb_sq_sum_fold): a deterministic norm without a second pass over the gradients.  A call that cannot
                                 honour it (another epilogue, unaligned rows, fp32 operands) FAILS rather than skip it silently. */
    int64_t sq_slots_n;
} cb_gemm_desc;

/* GEMM / implicit-GEMM convolution, all forms.  Replaces torch.nn.Linear / F.conv2d (+ apex-amp
 * cuBLAS/cuDNN) at: BertSelfAttention q/k/v (src/modeling/transformers.py:238-249), BertSelfOutput
 * :297-301, BertIntermediate :363-366, BertOutput :377-381, BertPooler :470-476, heads
 * (src/modeling/modeling.py:534-539, transformers.py:504-515), detectron2 ResNet convs + FrozenBN
 * (src/modeling/grid_feat.py:95) and grid_encoder conv (:43-45,99), and their autograd backward. */
int cb_gemm(const cb_gemm_desc* d, void* stream);
/* What cb_gemm would launch for `d` (same validation, nothing launched; the operand pointers are only checked for alignment):
 * out4 = {tile, split_k, schedule, xcd_order} as the descriptor fields of those names.  d->tile / xcd_order / schedule == 0 (auto) are
 * resolved by the per-shape table measured on MI355X (csrc/gemm_tuned.h; use_table != 0) and, for shapes outside it, by the launch-cost
 * model fitted to the same sweeps (csrc/gemm_model.h, tools/fit_gemm_model.py).  tile 8 (the streaming structure, taken by itself only
 * with use_table != 0): out4 = {8, 1, instantiation, 2}.  For tools and tests. */
int cb_gemm_plan(const cb_gemm_desc* d, int32_t use_table, int32_t* out4);
/* Bytes of K-split scratch (cb_gemm_desc.splitk_ws) cb_gemm would use for `d` at most -- what it picks when the workspace is not the
 * constraint; 0 when the problem runs unsplit or through atomics.  A caller sizes ONE buffer by the maximum over the problems it will
 * launch on a stream (clipbert_amd keeps 128 MiB per device: the largest value over the three bench workloads is 115 MiB); a smaller or
 * absent buffer is never an error -- cb_gemm then chooses among the configurations that fit.  A non-zero answer includes the
 * CB_SPLITK_WS_COUNTER_BYTES tail. */
int cb_gemm_workspace_bytes(const cb_gemm_desc* d, int64_t* bytes);
/* `n` INDEPENDENT problems in as few launches as possible: the same results as n cb_gemm calls (no problem's output may overlap
 * another problem's output or operands), but problems that run the same kernel class -- weight gradients of Linear / 1x1
 * convolutions, weight gradients of gathered convolutions, plain forward products, gathered forward products; fast path (16-byte
 * aligned operands < 2 GiB); a strided batch and a_rowsum only on unsplit bf16 weight gradients of Linear layers (A and B CB_KROW,
 * tile 0 / 4: their own class on the 128x128 two-per-CU tile -- the encoder's four kinds of 12-layer weight gradients,
 * src/modeling/transformers.py:257-381 under autograd, are ONE launch whose last wave of tiles is shared) -- share ONE grid: every workgroup looks its (problem, tile) up in a
 * table that travels in the kernel arguments (capturable as is).  A launch of many small problems fills the 256 CUs where each
 * alone is a fraction of a round of workgroups: the weight gradients of all convolutions of a ResNet stage
 * (src/modeling/grid_feat.py:95 under autograd: one cuDNN call each) become two launches instead of ~13.  Everything else in the
 * list is launched problem by problem, in list order.  descs[i].tile: 0 = the library chooses tile and K splits for the group
 * (bf16: 64x64 or 128x128 with two workgroups per CU; a problem's K may be split only where cb_gemm's own rule allows atomics:
 * fp32 C, accumulate = 1, scale/alpha-only epilogue); 2 / 4 = that tile with descs[i].split_k as given (problems are grouped with
 * those that ask for the same tile).  CB_F32 problems run the 64x64 fp32 tile with the caller's split_k: bit-identical to n
 * cb_gemm calls wherever split_k == 1.
 * Split bf16 weight gradients of a group combine WITHOUT atomics when every split problem carries the same splitk_ws and it holds
 * all partial tiles (sum over the split problems of tiles x split_k x tile bytes): each K part writes its partial tile there, the
 * last part of a tile to arrive adds them in part order and applies the epilogue once -- an order-independent, bit-reproducible
 * sum (csrc/gemm_impl.h gemm_tile; the arrival counters are the zeroed tail of that same scratch, see cb_gemm_desc.splitk_ws: nothing
 * is allocated here, and two grouped launches in flight on two streams are independent exactly when their scratch buffers are).
 * No / too small a scratch, or CB_GROUP_SLAB=0: fp32 atomics as before (zero-initialised or accumulated-into C, order of addition
 * not fixed). */
int cb_gemm_group(const cb_gemm_desc* descs, int32_t n, void* stream);

/* Output-pixel table of a convolution: entry m=(n,oh,ow) -> offset of input pixel
 * (n, oh*stride-pad, ow*stride-pad) and its (ih0, iw0).  sN/sH/sW in elements. */
int cb_build_pixel_table(cb_pixel* tab, int32_t N, int32_t OH, int32_t OW, int32_t stride, int32_t pad,
                         int64_t sN, int64_t sH, int64_t sW, void* stream);

/* Stem input pack: (N,3,H,W) fp32 RGB mean-subtracted (or uint8 RGB with mean/std) -> zero-padded
 * NHWC4 image (N, H+2*pad(+), W+2*pad(+), 4) in BGR order, dtype T.  Fuses ImageNorm
 * (src/datasets/data_utils.py:266-276), .float() (dataloader.py:104) and the RGB->BGR gather
 * (src/modeling/grid_feat.py:92-94).  src_u8 = 1: src is uint8 and (v - mean[c]) / std[c] is applied.
 * mean3 / std3 are HOST arrays of 3 floats (RGB order) -- the only host pointers in this ABI. */
int cb_stem_pack(int32_t dtype, const void* src, int32_t src_u8, const float* mean3, const float* std3,
                 void* dst, int32_t N, int32_t H, int32_t W, int32_t Hp, int32_t Wp, int32_t pad,
                 void* stream);

/* ImageNorm alone (a1): uint8 (n) -> fp32 (x - mean[c]) / std[c], NCHW with plane size hw.
 * mean3 / std3: HOST arrays of 3 floats. */
int cb_image_norm(const uint8_t* src, float* dst, const float* mean3, const float* std3,
                  int64_t n_images, int64_t hw, void* stream);

/* Max pooling on NHWC (detectron2 BasicStem max_pool2d k3 s2 p1; grid_encoder MaxPool2d(2,2) +
 * ReLU, src/modeling/grid_feat.py:46-47).  relu = 1 applies ReLU after pooling. */
int cb_maxpool_fwd(int32_t dtype, const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C,
                   int32_t OH, int32_t OW, int32_t k, int32_t stride, int32_t pad, int32_t relu,
                   void* stream);
/* Backward of the above for k=2,s=2 (non-overlapping windows): dx gets dy at the arg-max, 0
 * elsewhere (including rows/cols dropped by floor mode); gated by y > 0 when relu = 1. */
int cb_maxpool2_bwd(int32_t dtype, const void* x, const void* y, const void* dy, void* dx, int32_t N,
                    int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t relu, void* stream);

/* The stem in ONE launch, forward only (round 5): 7x7 stride-2 convolution (3 -> 64) + FrozenBN + ReLU + 3x3 stride-2 max-pool (pad 1)
 * of detectron2's BasicStem, bf16.  packed: the (N, Hp, Wp, 4) image of cb_stem_pack (pad 3, Wp even); weight: [64][7][8 taps x 4 ch] bf16
 * (tap 7 and channel 3 zero); scale / shift: the folded FrozenBN; out: (N, PH, PW, 64) with OH x OW the convolution's map and
 * PH = (OH - 1) / 2 + 1.  Replaces cb_gemm (stem form) + cb_maxpool_fwd: the convolution output never leaves the CU. */
int cb_stem_pool(const void* packed, const void* weight, const float* scale, const float* shift, void* out, int32_t N, int32_t Hp,
                 int32_t Wp, int32_t OH, int32_t OW, int32_t PH, int32_t PW, void* stream);
/* cb_stem_pack folded into cb_stem_pool (round 6; SURVEY N4 "GPU input pipeline": the uint8 frames go straight into the first convolution):
 * frames: (N, 3, H, W) uint8 RGB planes as the loader hands them over; mean3 / std3: HOST arrays of ImageNorm (src/modeling/grid_feat.py /
 * e2e_model.py: the pixel statistics of the detectron2 config) in RGB order.  (x - mean) / std, the RGB -> BGR flip
 * (src/modeling/grid_feat.py:92-94) and the convolution's zero padding happen while an input tile moves into LDS -- the same arithmetic
 * as cb_stem_pack, bit-identical output to cb_stem_pack + cb_stem_pool; one launch and 27 MB of packed image fewer per 64 frames. */
int cb_stem_pool_u8(const uint8_t* frames, const float* mean3, const float* std3, const void* weight, const float* scale, const float* shift,
                    void* out, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t PH, int32_t PW, void* stream);

/* One bottleneck block of the res2 stage in ONE launch, forward only (round 5): detectron2's BottleneckBlock with 64 mid channels,
 * stride 1, FrozenBN (modeling/backbone/resnet.py as built by src/modeling/grid_feat.py:60-70; FREEZE_AT = 2 keeps the stage frozen):
 *     y1 = relu(bn1(conv1x1(x)));  y2 = relu(bn2(conv3x3(y1), pad 1));  out = relu(bn3(conv1x1(y2)) + shortcut)
 * bf16, NHWC.  x: (N, H, W, cin), cin = 64 (stage entry: shortcut = bn_sc(conv1x1(x)) through wsc) or 256 (identity shortcut, wsc
 * NULL); out: (N, H, W, 256).  Weights in their KRSC memory images: w1 [64][cin], w2 [64][3][3][64], w3 [256][64], wsc [256][64];
 * scale / shift: the folded fp32 FrozenBN vectors of each convolution.  Replaces three (four) cb_gemm launches whose 64-channel
 * intermediates made them HBM-bound; same arithmetic (bf16 y1 / y2, fp32 accumulation).  All pointers 16-byte aligned. */
typedef struct cb_res2_desc {
    const void* x; void* out;
    const void* w1; const void* w2; const void* w3; const void* wsc;
    const float* scale1; const float* shift1; const float* scale2; const float* shift2; const float* scale3; const float* shift3;
    const float* scale_sc; const float* shift_sc;
    int32_t N, H, W, cin;
} cb_res2_desc;
int cb_res2_block(const cb_res2_desc* desc, void* stream);

/* g = dy * (y > 0) * scale[c]  (ReLU + FrozenBN backward); optional second output dz = dy * (y > 0)
 * and third g2 = dz * scale2[c] (projection shortcut).  rows x C, contiguous. */
int cb_relu_scale_bwd(int32_t dtype, const void* dy, const void* y, const float* scale, void* g,
                      void* dz, const float* scale2, void* g2, int64_t rows, int32_t C, void* stream);

/* LayerNorm over the last dim (apex FusedLayerNorm, src/modeling/transformers.py:32,148).
 * y = (x - mean) * rstd * gamma + beta, fp32 statistics; mean/rstd (rows) are saved when non-null. */
/* Row segments: logical row r lives at physical row (r / seg_len) * seg_stride + seg_off + r % seg_len
 * (seg_len <= 0: rows are contiguous).  Lets one call address e.g. only the text rows of (B, L, D). */
int cb_layernorm_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, void* y,
                     float* mean, float* rstd, int64_t rows, int32_t D, float eps, int32_t seg_len,
                     int32_t seg_stride, int32_t seg_off, void* stream);
/* dx = LN backward; dgamma/dbeta (fp32, D) are ACCUMULATED with atomics.  dx2 (optional) receives
 * dx with inverted-dropout mask (seed, p) applied -- the gradient of the dropped GEMM output. */
int cb_layernorm_bwd(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                     const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int32_t D,
                     void* dx2, float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_ptr,
                     int32_t seg_len, int32_t seg_stride, int32_t seg_off, void* stream);
/* The same with DETERMINISTIC parameter gradients and no atomics: `nblocks` (1..1024) blocks are launched and block b stores
 * its partial sums to part[b][0|1][D] (dgamma | dbeta, fp32).  cb_ln_partials_reduce then adds, for njobs such calls at
 * once (part = [njobs][nblocks][2][D]), the partial rows in block order onto grad[off_gamma[j] ..], grad[off_beta[j] ..]
 * (element offsets into a flat fp32 gradient buffer; device arrays).  The encoder backward issues its 24 LayerNorm backwards
 * this way and one reduce (src/modeling/transformers.py:148,342-343 run 24 autograd nodes with their own reductions). */
int cb_layernorm_bwd_part(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                          const float* rstd, void* dx, float* part, int32_t nblocks, int64_t rows, int32_t D,
                          void* dx2, float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_ptr,
                          int32_t seg_len, int32_t seg_stride, int32_t seg_off, void* stream);
int cb_ln_partials_reduce(const float* part, float* grad, const int64_t* off_gamma, const int64_t* off_beta,
                          int32_t njobs, int32_t nblocks, int32_t D, void* stream);

/* Text embedding (BertEmbeddings.forward, src/modeling/transformers.py:172-199): out row
 * (b*L_total + t) = LN(word[ids[b,t]] + pos[t] + type[0]); pre-LN sum saved in `pre` when non-null.
 * rows_period = P > 0: the text batch is P rows repeated B/P times (the folded clip loop: every clip sees the same captions) --
 * ids / attn_mask hold P rows and row b reads row b % P (no repeated copy is materialised); 0: B distinct rows.
 * key_mask (optional, fp32 B x L_total): the attention key mask of ClipBertBaseModel.forward (src/modeling/modeling.py:217-220)
 * -- this call writes its text columns (attn_mask, or ones when attn_mask is null), cb_visual_embed_fwd the visual ones. */
int cb_text_embed_fwd(int32_t dtype, const int64_t* ids, const void* word, const void* pos, const void* type0,
                      const float* gamma, const float* beta, void* out, void* pre, float* mean, float* rstd,
                      int32_t B, int32_t Lt, int32_t L_total, int32_t D, float eps, const int64_t* attn_mask,
                      float* key_mask, int32_t rows_period, void* stream);
/* Visual embedding (VisualInputEmbedding.forward, src/modeling/modeling.py:62-101,124-153) fused
 * with repeat_tensor_rows (src/datasets/data_utils.py:344-357): out row (b*L_total + Lt + p) =
 * LN(mean_t grid[src_row[b], t, sel[p]] + row_emb[h] + col_emb[w] + type[0]).  sel (optional) is the
 * sorted pixel sub-sample of pre-training (modeling.py:80-88). */
int cb_visual_embed_fwd(int32_t dtype, const void* grid, const int32_t* src_row, const int32_t* sel,
                        const void* row_emb, const void* col_emb, const void* type0, const float* gamma,
                        const float* beta, void* out, void* pre, float* mean, float* rstd, int32_t B,
                        int32_t T, int32_t Hg, int32_t Wg, int32_t Lv, int32_t Lt, int32_t L_total, int32_t D,
                        float eps, float* key_mask, void* stream);
/* Backward of both embeddings given d(pre) rows in a (B, L_total, D) buffer: scatter-adds (fp32
 * atomics) into the embedding-table gradients and into dgrid (fp32, zero-initialised by caller). */
int cb_text_embed_bwd(int32_t dtype, const void* dpre, const int64_t* ids, float* dword, float* dpos,
                      float* dtype0, int32_t B, int32_t Lt, int32_t L_total, int32_t D, int64_t pad_id,
                      int32_t rows_period, void* stream);   /* rows with ids == pad_id get no word gradient (padding_idx) */
int cb_visual_embed_bwd(int32_t dtype, const void* dpre, const int32_t* src_row, const int32_t* sel,
                        float* dgrid, float* drow, float* dcol, float* dtype0, int32_t B, int32_t T,
                        int32_t Hg, int32_t Wg, int32_t Lv, int32_t Lt, int32_t L_total, int32_t D,
                        void* stream);

/* Self-attention core (BertSelfAttention.forward, src/modeling/transformers.py:257-282) on the fused
 * QKV buffer (B*L, 3*H*64): scores = QK^T / 8 + (1 - mask) * -10000, fp32 softmax, ctx = P V, heads
 * merged into ctx (B*L, H*64).  lse (B,H,L) fp32 is saved for the backward when non-null. */
int cb_attention_fwd(int32_t dtype, const void* qkv, const float* key_mask, void* ctx, float* lse,
                     int32_t B, int32_t L, int32_t H, float dropout_p, uint64_t dropout_seed,
                     const uint64_t* dropout_seed_ptr, void* stream);
/* dsum_ws: fp32 workspace of B*H*L elements (rowsum(dctx*ctx), produced and consumed inside). */
int cb_attention_bwd(int32_t dtype, const void* qkv, const float* key_mask, const void* ctx, const void* dctx,
                     const float* lse, float* dsum_ws, void* dqkv, int32_t B, int32_t L, int32_t H,
                     float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_ptr, void* stream);

/* Row-wise softmax cross-entropy with ignore_index (CrossEntropyLoss(reduction="none"),
 * src/modeling/modeling.py:287-298,562-566): loss[r] and (optional) dlogits = (softmax - onehot) *
 * dloss[r].  logits fp32 (rows, C). */
int cb_cross_entropy(const float* logits, int64_t ld, const int64_t* labels, float* loss, float* dlogits,
                     const float* dloss, int64_t rows, int32_t C, int64_t ignore_index, void* stream);

/* Column sums: out[n] (+)= sum_m g[m,n]  (bias gradients).  fp32 atomics into out. */
int cb_colsum(int32_t dtype, const void* g, int64_t ldg, float* out, int64_t M, int32_t N, void* stream);

/* Elementwise helpers on contiguous buffers. */
int cb_cast(int32_t src_dtype, const void* src, int32_t dst_dtype, void* dst, int64_t n, void* stream);
/* dx = dy * act'(.) ; `ref` is the PRE-activation for CB_ACT_GELU and the OUTPUT for RELU / TANH. */
int cb_act_bwd(int32_t dtype, int32_t act, const void* dy, const void* ref, void* dx, int64_t n, void* stream);

/* Fused AdamW over a flat fp32 parameter range (src/optimization/adamw.py:40-103) with global-norm
 * clipping (run_video_retrieval.py:477-482): p, g, m, v are fp32 arrays of n elements; `w16`
 * (optional) receives the bf16 compute copy.  grad_sq_sum: device scalar holding sum(g^2) over ALL
 * parameters (cb_sq_sum); clip coefficient = min(1, max_norm / (grad_scale*sqrt(sum) + 1e-6));
 * max_norm <= 0 disables clipping.  grad_scale multiplies g first (1/world_size when gradients were
 * summed across ranks). */
int cb_adamw(float* p, const float* g, float* m, float* v, void* w16, int64_t n, const float* hyper,
             const float* grad_sq_sum, void* stream);
/* hyper: DEVICE array of AT LEAST CB_HP_COUNT (10) floats (so a captured hipGraph sees new values each replay):
 * [lr, beta1, beta2, eps, weight_decay, 1-beta1^t, 1-beta2^t, max_norm, grad_scale, skip].  ABI version 2 (cb_version()) added
 * hyper[CB_HP_SKIP] as element 9: the kernel reads it unconditionally, so a caller written against version 1 (9 floats) must grow
 * its array to CB_HP_COUNT floats and set hyper[CB_HP_SKIP] = 0 -- a non-zero value there makes the launch a no-op. */
enum { CB_HP_LR = 0, CB_HP_BETA1, CB_HP_BETA2, CB_HP_EPS, CB_HP_WD, CB_HP_BC1, CB_HP_BC2, CB_HP_MAX_NORM,
       CB_HP_GRAD_SCALE, CB_HP_SKIP /* != 0: the launch does nothing (a captured, software-pipelined update with no step behind it) */,
       CB_HP_COUNT };
/* y = x * inverted-dropout mask(seed + *seed_ptr, index)  (forward and backward of nn.Dropout). */
int cb_dropout(int32_t dtype, const void* x, void* y, int64_t n, float p, uint64_t seed,
               const uint64_t* seed_ptr, void* stream);

/* Clip aggregation of the per-clip logits (the stack [n_clips][B*C], fp32) -- the reference does it with torch ops in its
 * task loops: training src/tasks/run_video_retrieval.py:402-411 (mean / max; "lse" is pooled inside the loss below),
 * inference :669-676 (mean / max / logsumexp over clips), src/tasks/run_video_qa.py:241-275 likewise.
 * `argmax` (int32 [B*C]) is written for CB_AGG_MAX and consumed by the backward. */
enum { CB_AGG_MEAN = 0, CB_AGG_MAX = 1, CB_AGG_LSE = 2 };
int cb_clip_aggregate_fwd(const float* logits, int32_t n_clips, int64_t bc, int32_t mode, float* out, int32_t* argmax,
                          void* stream);
int cb_clip_aggregate_bwd(const float* dout, const float* logits, const float* out, const int32_t* argmax, int32_t n_clips,
                          int64_t bc, int32_t mode, float* dlogits, void* stream);

/* LSE training loss (run_video_retrieval.py:415-418): loss[b] = logsumexp_{clip,class} - logsumexp_{clip}(class = labels[b]) on
 * clip-major logits [n_clips][B][C]; `dlogits` (optional) receives dloss[b] (1 if null) times its gradient. */
int cb_lse_loss(const float* logits, const int64_t* labels, int32_t n_clips, int32_t B, int32_t C, float* loss, const float* dloss,
                float* dlogits, void* stream);
/* Element-wise head losses on fp32 logits, forward and / or backward (reduction = "none" like the reference's):
 *   kind 0  MSELoss, num_labels == 1 (src/modeling/modeling.py:364-368, 436-440):  loss = (x - y)^2
 *   kind 1  instance_bce_with_logits(reduction="none") (:308-315, 370-372):          loss = max(x, 0) - x*y + log1p(exp(-|x|))
 *   kind 2  sigmoid margin ranking of the retrieval head (:567-575): x is (rows, group) with the positive first;
 *           loss (rows, group-1) = max(0, margin + sigmoid(x[:, 1:]) - sigmoid(x[:, :1]));  y unused
 * n = number of logits.  loss / dx may be null (forward only / backward only); dx = dloss (1 if null) times the gradient. */
int cb_head_loss(int32_t kind, const float* x, const float* y, float* loss, const float* dloss, float* dx, int64_t n, int32_t group,
                 float margin, void* stream);
/* Retrieval scores of the inference loop (src/tasks/run_video_retrieval.py:681-690): C == 2: softmax(logits)[:, 1]; C == 1: sigmoid. */
int cb_retrieval_scores(const float* logits, float* out, int64_t rows, int32_t C, void* stream);
/* Mean of n per-example losses (the runners' loss.mean(), run_video_retrieval.py:422) and its backward
 * dx[i] = *dmean / n; `*counter += inc` on a device word (the dropout seed word a captured step advances). */
int cb_mean_fwd(const float* x, int64_t n, float* out, void* stream);
int cb_mean_bwd(const float* dmean, int64_t n, float* dx, void* stream);
int cb_counter_add(int64_t* counter, int64_t inc, void* stream);
/* p[0, bytes) = 0 (any alignment): the zero fills of the step -- gradient ranges that are accumulated into (optimizer.zero_grad() of
 * run_video_retrieval.py:484), scatter targets of the embedding backwards, the squared-norm accumulator -- as a stream-ordered,
 * capturable kernel of this library. */
int cb_zero(void* p, int64_t bytes, void* stream);
/* the same for up to FOUR ranges in one launch (round 6): zero_grad() of a step whose weight gradients are stored by their first writers
 * leaves three gaps of the flat gradient buffer + the norm slots to fill -- one launch instead of four.  ptrs / bytes: HOST arrays. */
int cb_zero_ranges(void* const* ptrs, const int64_t* bytes, int32_t n, void* stream);
int cb_sq_sum(const float* g, int64_t n, float* out_accum, void* stream);
/* The same sum with a result that does not depend on the order in which workgroups retire (fixed grid of <= min(1024, ws_floats)
 * blocks -> `ws` partials -> one block adds them in index order): data-parallel ranks holding bit-identical all-reduced
 * gradients derive the bit-identical clip coefficient (torch.nn.utils.clip_grad_norm_ in run_video_retrieval.py:477-482 is
 * deterministic per rank too).  ws: caller-owned scratch of ws_floats floats. */
int cb_sq_sum_det(const float* g, int64_t n, float* out_accum, float* ws, int32_t ws_floats, void* stream);
/* The norm of a step whose weight-gradient launches already left their shares in slots (cb_gemm_desc.sq_slots): out_accum += the squares
 * of the nseg (<= 4) ranges g[seg_lo_hi[2 i], seg_lo_hi[2 i + 1]) that no such launch covers (seg_lo_hi: HOST array of element
 * offsets) + the sum of slots[0, nslots), everything added in a fixed order (same determinism contract as cb_sq_sum_det).  Two launches. */
int cb_sq_sum_fold(const float* g, const int64_t* seg_lo_hi, int32_t nseg, const float* slots, int64_t nslots, float* out_accum, float* ws,
                   int32_t ws_floats, void* stream);
/* The same two for bf16 GRADIENTS: in data-parallel runs the all-reduce travels in bf16 (as the reference's apex-O2 fp16 gradients
 * do through Horovod, run_video_retrieval.py:298-301); the optimizer then reads the reduced wire image directly instead of a
 * copy cast back to fp32 (same values: bf16 -> fp32 is exact). */
int cb_sq_sum_det_bf16(const void* g16, int64_t n, float* out_accum, float* ws, int32_t ws_floats, void* stream);
int cb_adamw_g16(float* p, const void* g16, float* m, float* v, void* w16, int64_t n, const float* hyper, const float* grad_sq_sum,
                 void* stream);

/* ELU followed by BatchNorm1d over the batch dimension -- regressor[1:3] of ClipBertForRegression
 * (src/modeling/modeling.py:461-466; torch.nn.ELU + torch.nn.BatchNorm1d semantics).  x, y: (B, D).  training = 1: batch
 * statistics (biased variance), running_mean / running_var updated in place with `momentum` (unbiased variance); training = 0:
 * the running statistics.  save_mean / save_invstd (D floats each, optional) are what the backward consumes. */
int cb_elu_bn1d_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    void* y, float* save_mean, float* save_invstd, int32_t B, int32_t D, int32_t training, float momentum, float eps,
                    void* stream);
/* dx = d(loss)/d(x) through BatchNorm1d and ELU; dgamma / dbeta (fp32, D) are ACCUMULATED (one writer per column). */
int cb_elu_bn1d_bwd(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* save_mean, const float* save_invstd,
                    void* dx, float* dgamma, float* dbeta, int32_t B, int32_t D, int32_t training, void* stream);

const char* cb_last_error(void);
/* ABI version: 1 = round 1-2; 2 = CB_HP_SKIP / CB_HP_COUNT 10 (cb_adamw*), cb_gemm_group, cb_gemm_workspace_bytes;
 * 3 = cb_gemm_desc.tile = 8 (the streaming structure; every earlier descriptor means what it meant), cb_head_loss, cb_retrieval_scores;
 * 4 = cb_res2_block, cb_stem_pool (round 5: the frozen front of the backbone as fused launches; nothing else changed);
 * 5 = the K-split arrival counters live in the tail of cb_gemm_desc.splitk_ws (CB_SPLITK_WS_COUNTER_BYTES, zeroed once by the caller)
 *     instead of a library-owned allocation: a caller of version 4 that passes a scratch must zero its last 64 KiB once;
 * 6 = cb_gemm_desc grew by sq_slots / sq_slots_n at its END (zero = off: older callers that memset the struct they allocate with the
 *     new size are unaffected), accumulate = 2 (first writer), cb_sq_sum_fold;
 * 7 = cb_gemm_desc.tile = 9 (few rows), chosen by itself for M <= 64: the same result up to the order of the fp32 additions;
 *     cb_gemm_group takes strided batches / a_rowsum on the unsplit bf16 weight-gradient form; cb_stem_pool_u8; cb_zero_ranges */
int cb_version(void);

/* ---- gradient exchange (one process per GPU, RCCL over xGMI) ----------------------------------------------------
 * Replaces Horovod's per-tensor NCCL all-reduce (src/tasks/run_video_retrieval.py:298-305, 432; src/utils/distributed.py).
 * cb_comm_unique_id: rank 0 creates the 128-byte id and the host distributes it (any channel).  cb_comm_init: every rank,
 * on its GPU (the calling thread's current HIP device); one communicator per process.  cb_allreduce_bucket: in-place SUM of
 * `count` elements (CB_F32 or CB_BF16: the flat gradient buffer or its bf16 wire image) over all ranks, enqueued on
 * `stream` -- asynchronous, ordered against other streams by HIP events, capturable into a hipGraph.  RCCL is loaded at
 * run time (librccl.so.1); without it these calls fail with a message and everything else works.  The communicator is process-
 * global state: call cb_comm_* from one host thread (the one that owns the GPU), as the rest of the ABI is used. */
int cb_comm_unique_id(void* id128);
int cb_comm_init(int32_t rank, int32_t world, const void* id128);
int cb_comm_info(int32_t* rank, int32_t* world);
int cb_allreduce_bucket(void* buf, int64_t count, int32_t dtype, void* stream);
/* The two halves of an all-reduce (direct reduce-scatter + all-gather over the xGMI mesh), so that the optimizer can
 * run on 1/world of the parameters in between.  cb_reduce_scatter_bucket: `send` holds world * recv_count elements; rank r receives
 * the sum over ranks of elements [r * recv_count, (r+1) * recv_count) in `recv` (recv == send + r * recv_count: in place).
 * cb_allgather_bucket: every rank contributes send_count elements, `recv` (world * send_count) gets rank r's at r * send_count
 * (send == recv + rank * send_count: in place).  cb_broadcast_bucket: root's buffer to all ranks (hvd.broadcast_parameters,
 * run_video_retrieval.py:304).  Same stream / capture rules as cb_allreduce_bucket. */
int cb_reduce_scatter_bucket(const void* send, void* recv, int64_t recv_count, int32_t dtype, void* stream);
int cb_allgather_bucket(const void* send, void* recv, int64_t send_count, int32_t dtype, void* stream);
int cb_broadcast_bucket(void* buf, int64_t count, int32_t dtype, int32_t root, void* stream);
int cb_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* CLIPBERT_HIP_H */
