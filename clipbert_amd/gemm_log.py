"""Measurement support: record every GEMM problem a step hands to the library (``ops.gemm`` and ``ops.gemm_group``) and sort the
problems into families BY THE ROOF THAT BOUNDS THEM.  Used by bench.py's roofline and tools/gemm_breakdown.py;
nothing on the product path imports it.

Families (reference functions: SURVEY.md a3 / a10-a14; shapes: Appendix B / C):
  encoder linear (fwd + dgrad)      nn.Linear of the 12 BERT layers and the heads, transformers.py:230-381      -> bf16 MFMA roof
  resnet 1x1 conv, K <= 256         res2 / res3 bottleneck 1x1 convolutions and their data gradients: 2*K flop per output element
                                    against >= 2+2 bytes moved -- below the machine balance (2.5 PF / 8 TB/s = 312 flop/B)   -> HBM roof
  resnet 1x1 conv, K > 256          res4 / res5 1x1 convolutions                                                 -> bf16 MFMA roof
  conv 3x3 / 7x7 (fwd + dgrad)      implicit-GEMM convolutions (pixel gather), grid_feat.py:43-48 + the stem     -> bf16 MFMA roof
  weight gradients                  every wgrad form (batched encoder layers, ResNet stages through cb_gemm_group) -> bf16 MFMA roof
  fused frozen front                round 5: cb_stem_pool (7x7 convolution + FrozenBN + ReLU + max-pool) and cb_res2_block (a res2 bottleneck
                                    block per launch): their 64-channel intermediates never leave the CU, what remains is input + output  -> HBM roof
                                    (algorithmic flops = the convolutions' 2*M*N*K, algorithmic bytes = the block's input once + its output once)
"""
from typing import Callable, Dict, List

import torch

from . import ops

MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E

FAMILY_BOUND = {
    "encoder linear (fwd + dgrad)": "mfma",
    "resnet 1x1 conv, K <= 256 (fwd + dgrad)": "hbm",
    "resnet 1x1 conv, K > 256 (fwd + dgrad)": "mfma",
    "conv 3x3 / 7x7 (fwd + dgrad)": "mfma",
    "weight gradients (linear + conv)": "mfma",
    "fused frozen front: stem + res2 blocks (fwd)": "hbm",
}
FUSED = "fused frozen front: stem + res2 blocks (fwd)"


def _problem(d) -> dict:
    """the figures of one problem from its cb_gemm_desc (ops.GemmDesc)"""
    batch = max(1, int(d.batch))
    taps = max(1, int(d.R) * int(d.S))
    esz = 4 if d.dtype == ops.CB_F32 else 2
    c_esz = 4 if d.c_f32 else esz
    wgrad = d.a_mode == ops.KROW
    gather = d.a_mode == ops.ROWK_GATHER or d.b_mode in (ops.KROW_GATHER, ops.KROW_TAPS)
    cnn = bool(d.scale) or bool(d.relu_bwd) or bool(d.a_tab) or bool(d.b_tab) or bool(d.c_rowmap) or d.zero_fill_pitch != 0 or bool(d.post_scale)
    M, N, K = int(d.M), int(d.N), int(d.K)
    # algorithmic reduction length (SURVEY 8d counts algorithmic flops): the 7x7x3 stem is LAUNCHED on the zero-padded NHWC4 image
    # with K = 7 rows x (8 taps x 4 channels) = 224, of which 7 x 7 x 3 = 147 products per output are the convolution's
    K_alg = 147 if (d.a_mode == ops.ROWK_GATHER and int(d.R) == 7 and int(d.Cin) == 32 and K == 224) else K
    # pixel counts of a 224-multiple input are multiples of 49 (7 x 7 at res5); token rows (41 per pair) and head rows are not
    cnn = cnn or (wgrad and K % 49 == 0 and K >= 49) or (not wgrad and M % 49 == 0 and M >= 49 * 16)
    if wgrad:
        fam = "weight gradients (linear + conv)"
        # A (K x M) + B (K x N; a gathered input counted once, not once per tap) + the fp32 gradient written (read too when accumulating)
        alg = batch * ((K * M + K * N / taps) * esz + (2 if d.accumulate else 1) * M * N * c_esz)
    else:
        if gather and taps > 1:
            fam = "conv 3x3 / 7x7 (fwd + dgrad)"
        elif cnn:
            fam = "resnet 1x1 conv, K <= 256 (fwd + dgrad)" if K <= 256 else "resnet 1x1 conv, K > 256 (fwd + dgrad)"
        else:
            fam = "encoder linear (fwd + dgrad)"
        extra = sum(1 for p in (d.residual, d.mask, d.C2, d.gelu_grad_pre) if p) + (1 if d.accumulate else 0) + (2 if d.relu_bwd else 0)
        alg = batch * ((M * K / taps + N * K) * esz + M * N * c_esz + extra * M * N * esz)
    return {"family": fam, "flop": 2.0 * M * N * K_alg * batch, "bytes": float(alg), "M": M, "N": N, "K": K, "batch": batch, "taps": taps,
            "form": "wgrad" if wgrad else ("dgrad" if d.b_mode in (ops.KROW, ops.KROW_TAPS) else "fwd")}


class GemmLog:
    """with GemmLog() as log: step()  -> log.launches: one entry per library call (cb_gemm or cb_gemm_group) with its problems and a
    ``replay()`` that issues the same call again on the same (still live) operands"""

    def __init__(self):
        self.launches: List[dict] = []

    def __enter__(self):
        self._gemm, self._group = ops.gemm, ops.gemm_group
        self._stem, self._res2 = ops.stem_pool, ops.res2_block

        def stem_pool(packed, weight, scale, shift, oh, ow):
            n, hp, wp, _ = packed.shape
            m = n * oh * ow
            pooled = n * ((oh - 1) // 2 + 1) * ((ow - 1) // 2 + 1)
            prob = {"family": FUSED, "flop": 2.0 * m * 64 * 147, "bytes": float(packed.numel() * 2 + pooled * 64 * 2), "M": m, "N": 64, "K": 147, "batch": 1,
                    "taps": 49, "form": "fwd"}
            self.launches.append({"problems": [prob], "replay": (lambda: self._stem(packed, weight, scale, shift, oh, ow))})
            return self._stem(packed, weight, scale, shift, oh, ow)

        self._stem8 = ops.stem_pool_u8

        def stem_pool_u8(frames, mean, std, weight, scale, shift):
            n, _c, h, w = frames.shape
            oh, ow = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
            m = n * oh * ow
            pooled = n * ((oh - 1) // 2 + 1) * ((ow - 1) // 2 + 1)
            prob = {"family": FUSED, "flop": 2.0 * m * 64 * 147, "bytes": float(frames.numel() + pooled * 64 * 2), "M": m, "N": 64, "K": 147, "batch": 1,
                    "taps": 49, "form": "fwd"}
            self.launches.append({"problems": [prob], "replay": (lambda: self._stem8(frames, mean, std, weight, scale, shift))})
            return self._stem8(frames, mean, std, weight, scale, shift)

        def res2_block(x, w1, w2, w3, ss1, ss2, ss3, wsc=None, sssc=None):
            n, h, w, cin = x.shape
            m = n * h * w
            kk = cin * 64 + 9 * 64 * 64 + 64 * 256 + (cin * 256 if wsc is not None else 0)
            prob = {"family": FUSED, "flop": 2.0 * m * kk, "bytes": float(m * (cin + 256) * 2), "M": m, "N": 256, "K": kk // 256, "batch": 1, "taps": 9, "form": "fwd"}
            self.launches.append({"problems": [prob], "replay": (lambda: self._res2(x, w1, w2, w3, ss1, ss2, ss3, wsc=wsc, sssc=sssc))})
            return self._res2(x, w1, w2, w3, ss1, ss2, ss3, wsc=wsc, sssc=sssc)

        def gemm(a, b, M, N, K, **kw):
            d = ops.gemm_desc(a, b, M, N, K, **kw)
            self.launches.append({"problems": [_problem(d)], "replay": (lambda a=a, b=b, M=M, N=N, K=K, kw=kw: self._gemm(a, b, M, N, K, **kw))})
            return self._gemm(a, b, M, N, K, **kw)

        def gemm_group(descs, like):
            descs = list(descs)
            if descs:
                self.launches.append({"problems": [_problem(d) for d in descs], "replay": (lambda descs=descs, like=like: self._group(descs, like))})
            return self._group(descs, like)

        ops.gemm, ops.gemm_group = gemm, gemm_group
        ops.stem_pool, ops.res2_block = stem_pool, res2_block
        ops.stem_pool_u8 = stem_pool_u8
        return self

    def __exit__(self, *exc):
        ops.gemm, ops.gemm_group = self._gemm, self._group
        ops.stem_pool, ops.res2_block = self._stem, self._res2
        ops.stem_pool_u8 = self._stem8
        return False

    def by_family(self) -> Dict[str, List[dict]]:
        """a grouped call is filed under the family of its problems (cb_gemm_group takes one kernel class: they share it)"""
        out: Dict[str, List[dict]] = {}
        for ln in self.launches:
            fams = {p["family"] for p in ln["problems"]}
            out.setdefault(sorted(fams)[0] if len(fams) == 1 else "mixed group", []).append(ln)
        return out


def time_family(launches: List[dict], reps: int = 3, outer: int = 5) -> float:
    """seconds for ONE pass over ``launches`` issued back to back: the calls are captured ``reps`` times into a hipGraph and the graph
    is replayed ``outer`` times between ONE pair of HIP events on the launch stream (events cost microseconds each on this stack, so
    bracketing ~20 us launches one by one would time the markers)"""
    def replay():
        for _ in range(reps):
            for ln in launches:
                ln["replay"]()
    replay()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        replay()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(outer):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (reps * outer)


def family_table(step_fn: Callable[[], None]) -> Dict[str, dict]:
    """run ``step_fn`` once with the log armed, then time every family: {family: {bound, launches, problems, gflop, mbytes, ms,
    tflops, gbs, frac (of the roof that bounds it)}}"""
    with GemmLog() as log:
        step_fn()
        torch.cuda.synchronize()
    out = {}
    for fam, launches in log.by_family().items():
        t = time_family(launches)
        probs = [p for ln in launches for p in ln["problems"]]
        fl, by = sum(p["flop"] for p in probs), sum(p["bytes"] for p in probs)
        bound = FAMILY_BOUND.get(fam, "mfma")
        tf, gbs = fl / t / 1e12, by / t / 1e9
        out[fam] = {"bound": bound, "launches": len(launches), "problems": len(probs), "gflop": round(fl / 1e9, 1), "algorithmic_mbytes": round(by / 1e6, 1),
                    "ms": round(t * 1e3, 3), "avg_launch_us": round(t / len(launches) * 1e6, 2), "tflops": round(tf, 1), "gbs": round(gbs, 1),
                    "frac": round(tf / MFMA_PEAK_TFLOPS if bound == "mfma" else gbs / HBM_PEAK_GBS, 4)}
    return out
