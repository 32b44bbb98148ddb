#!/usr/bin/env python
"""Vendor-library YARDSTICK for the step's GEMM / convolution shapes: hipBLASLt (through torch.matmul / bmm) on
the encoder products and MIOpen (through F.conv2d and torch.nn.grad) on the heaviest ResNet convolutions, timed next to cb_gemm on
the same box, the same way (N launches in a hipGraph between one pair of events; "hot" = back to back, "cold" = every launch behind
a 384 MB flush and a fresh write of its A operand, tools/tune_gemm.py's method).

MEASUREMENT ONLY: nothing under clipbert_amd/ imports torch.matmul / F.conv2d, and nothing here is on a timed or product path.

    python tools/gemm_yardstick.py [--out gpurun_out/yardstick] [--skip-conv]
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

FLUSH = {}


def graph_time(fn, inner, outer=3, best_of=3):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(best_of):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(outer):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (inner * outer))
    del g
    return best


def hot_cold(fn, a):
    """(hot us, cold us) of fn(); ``a`` = the operand the step's previous kernel would just have written"""
    hot = graph_time(fn, 16)
    if "buf" not in FLUSH:
        FLUSH["buf"] = torch.empty(384 << 20, dtype=torch.uint8, device="cuda")
    src = a.detach().clone()

    def pre():
        FLUSH["buf"].zero_()
        a.copy_(src)

    def full():
        pre()
        fn()
    base = graph_time(pre, 6, 2)
    cold = max(0.1, graph_time(full, 6, 2) - base)
    return round(hot, 2), round(cold, 2)


def encoder_rows(M=2624, nl=12):
    from clipbert_amd import ops
    dev, dt = "cuda", torch.bfloat16
    rows = []
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g, dtype=torch.float32) - 0.5).to(dt)
    for name, N, K in (("QKV", 2304, 768), ("attn.out", 768, 768), ("FFN1", 3072, 768), ("FFN2", 768, 3072)):
        x, w, bias = rnd(M, K), rnd(N, K), rnd(N).float()
        gy = rnd(M, N)
        y, dx = torch.empty(M, N, device=dev, dtype=dt), torch.empty(M, K, device=dev, dtype=dt)
        flop = 2.0 * M * N * K
        # forward  y = x W^T
        lib = hot_cold(lambda: torch.matmul(x, w.t(), out=y), x)
        libb = hot_cold(lambda: F.linear(x, w, bias.to(dt)), x)
        ours = hot_cold(lambda: ops.gemm(x, w, M, N, K, out=y), x)
        oursb = hot_cold(lambda: ops.gemm(x, w, M, N, K, out=y, shift=bias), x)
        rows.append(dict(kind="fwd", name=name, M=M, N=N, K=K, gflop=flop / 1e9, hipblaslt=lib, hipblaslt_bias=libb, cb_gemm=ours, cb_gemm_bias=oursb))
        # data gradient  dx = gy W   (M x K_in x N)
        lib = hot_cold(lambda: torch.matmul(gy, w, out=dx), gy)
        ours = hot_cold(lambda: ops.gemm(gy, w, M, K, N, out=dx, b_mode=ops.KROW), gy)
        rows.append(dict(kind="dgrad", name=name, M=M, N=K, K=N, gflop=flop / 1e9, hipblaslt=lib, cb_gemm=ours))
        # weight gradient of all layers: dW[l] = gy[l]^T x[l]  (fp32 out for ours, bf16 out for the library: it has no bf16 x bf16 -> fp32 path in torch)
        xs, gs = rnd(nl, M, K), rnd(nl, M, N)
        dw16 = torch.empty(nl, N, K, device=dev, dtype=dt)
        dw32 = torch.empty(nl, N, K, device=dev, dtype=torch.float32)
        db = torch.zeros(nl, N, device=dev, dtype=torch.float32)
        lib = hot_cold(lambda: torch.bmm(gs.transpose(1, 2), xs, out=dw16), gs)
        ours = hot_cold(lambda: ops.gemm(gs, xs, N, K, M, out=dw32[0], a_mode=ops.KROW, lda=N, b_mode=ops.KROW, ldb=K, ldc=K, accumulate=False, a_rowsum=db[0],
                                         batch=nl, batch_strides=(M * N, M * K, N * K, N)), gs)
        rows.append(dict(kind=f"wgrad x{nl}", name=name, M=N, N=K, K=M, gflop=flop * nl / 1e9, hipblaslt=lib, cb_gemm=ours))
        print(json.dumps(rows[-3]), flush=True)
        print(json.dumps(rows[-2]), flush=True)
        print(json.dumps(rows[-1]), flush=True)
    return rows


def wgrad_set(M=2624, nl=12):
    """round 6: ALL weight gradients of the encoder (four kinds x 12 layers) as the step issues them -- one cb_gemm_group launch (row-sum /
    strided-batch class) -- next to four cb_gemm launches and to the library's four bmm calls, each set timed as a whole (hot)."""
    from clipbert_amd import ops
    dev, dt = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g, dtype=torch.float32) - 0.5).to(dt)
    kinds, flop = [], 0.0
    for name, N, K in (("FFN2", 768, 3072), ("FFN1", 3072, 768), ("attn.out", 768, 768), ("QKV", 2304, 768)):
        xs, gs = rnd(nl, M, K), rnd(nl, M, N)
        kinds.append(dict(N=N, K=K, xs=xs, gs=gs, dw16=torch.empty(nl, N, K, device=dev, dtype=dt), dw32=torch.empty(nl, N, K, device=dev, dtype=torch.float32),
                          db=torch.zeros(nl, N, device=dev, dtype=torch.float32)))
        flop += 2.0 * M * N * K * nl

    def lib():
        for k in kinds:
            torch.bmm(k["gs"].transpose(1, 2), k["xs"], out=k["dw16"])

    def kw(k):
        return dict(out=k["dw32"][0], a_mode=ops.KROW, lda=k["N"], b_mode=ops.KROW, ldb=k["K"], ldc=k["K"], accumulate=False, a_rowsum=k["db"][0], batch=nl,
                    batch_strides=(M * k["N"], M * k["K"], k["N"] * k["K"], k["N"]))

    def four():
        for k in kinds:
            ops.gemm(k["gs"], k["xs"], k["N"], k["K"], M, **kw(k))

    def one():
        ops.gemm_group([ops.gemm_desc(k["gs"], k["xs"], k["N"], k["K"], M, tile=4, **kw(k)) for k in kinds], kinds[0]["gs"])

    res = dict(gflop=flop / 1e9, hipblaslt_four_bmm=round(graph_time(lib, 8), 1), cb_gemm_four_launches=round(graph_time(four, 8), 1),
               cb_gemm_group_one_launch=round(graph_time(one, 8), 1))
    res["tflops"] = {k: round(res["gflop"] / v * 1e3) for k, v in res.items() if k != "gflop"}
    return res


CONVS = [  # (name, cin, cout, k, stride, H_in) at 64 frames of 224 px
    ("stem 7x7 s2", 3, 64, 7, 2, 224),
    ("res2 conv2 3x3", 64, 64, 3, 1, 56), ("res2 conv3 1x1", 64, 256, 1, 1, 56), ("res2 conv1 1x1", 256, 64, 1, 1, 56),
    ("res3 conv2 3x3", 128, 128, 3, 1, 28), ("res3 conv3 1x1", 128, 512, 1, 1, 28), ("res3 conv1 1x1", 512, 128, 1, 1, 28),
    ("res4 conv2 3x3", 256, 256, 3, 1, 14), ("res4 conv3 1x1", 256, 1024, 1, 1, 14), ("res4 conv1 1x1", 1024, 256, 1, 1, 14),
    ("res5 conv2 3x3", 512, 512, 3, 1, 7), ("res5 conv3 1x1", 512, 2048, 1, 1, 7), ("res5 conv1 1x1", 2048, 512, 1, 1, 7),
    ("grid_encoder 3x3", 2048, 768, 3, 1, 7),
]


def conv_rows(nframes=64):
    """MIOpen (NHWC bf16, immediate mode) forward / input gradient / weight gradient next to the product's launches of the same
    convolutions (clipbert_amd.modeling._conv_fwd / _conv_dgrad / _conv_wgrad: FrozenBN scale + shift in the forward epilogue, which
    the library call does not do)"""
    from clipbert_amd.bench import step as bench_step
    from clipbert_amd import modeling as Mo
    st = bench_step.build(videos=2)
    model, rt = st.model, st.model.rt
    bb = model.cnn.feature.backbone
    mods = {"stem 7x7 s2": bb.stem.conv1,
            "res2 conv2 3x3": bb.res2[1].conv2, "res2 conv3 1x1": bb.res2[1].conv3, "res2 conv1 1x1": bb.res2[1].conv1,
            "res3 conv2 3x3": bb.res3[1].conv2, "res3 conv3 1x1": bb.res3[1].conv3, "res3 conv1 1x1": bb.res3[1].conv1,
            "res4 conv2 3x3": bb.res4[1].conv2, "res4 conv3 1x1": bb.res4[1].conv3, "res4 conv1 1x1": bb.res4[1].conv1,
            "res5 conv2 3x3": bb.res5[1].conv2, "res5 conv3 1x1": bb.res5[1].conv3, "res5 conv1 1x1": bb.res5[1].conv1,
            "grid_encoder 3x3": model.cnn.grid_encoder[0]}
    dev, dt = "cuda", torch.bfloat16
    rows = []
    for name, cin, cout, k, s, h in CONVS:
        pad = k // 2
        oh = (h + 2 * pad - k) // s + 1
        x = (torch.rand(nframes, cin, h, h, device=dev) - 0.5).to(dt).contiguous(memory_format=torch.channels_last)
        w = (torch.rand(cout, cin, k, k, device=dev) - 0.5).to(dt).contiguous(memory_format=torch.channels_last)
        gy = (torch.rand(nframes, cout, oh, oh, device=dev) - 0.5).to(dt).contiguous(memory_format=torch.channels_last)
        flop = 2.0 * nframes * oh * oh * cout * cin * k * k
        row = dict(name=name, cin=cin, cout=cout, k=k, stride=s, hw=h, gflop=flop / 1e9)
        try:
            row["miopen_fwd"] = hot_cold(lambda: F.conv2d(x, w, None, s, pad), x)
            row["miopen_dgrad"] = hot_cold(lambda: torch.nn.grad.conv2d_input(x.shape, w, gy, s, pad), gy)
            row["miopen_wgrad"] = hot_cold(lambda: torch.nn.grad.conv2d_weight(x, w.shape, gy, s, pad), gy)
        except Exception as e:                       # noqa: BLE001
            row["miopen_error"] = str(e)[:200]
        conv = mods.get(name)
        if conv is not None and name != "stem 7x7 s2":
            try:
                xn = x.permute(0, 2, 3, 1)            # NHWC view of the same memory
                gn = gy.permute(0, 2, 3, 1)
                assert xn.is_contiguous() and gn.is_contiguous()
                row["cb_fwd"] = hot_cold(lambda: Mo._conv_fwd(rt, xn, conv, act=Mo.ACT_RELU), x)
                row["cb_dgrad"] = hot_cold(lambda: Mo._conv_dgrad(rt, gn, conv, tuple(xn.shape)), gy)
                if rt.bank.grad_image(conv.weight) is not None:
                    row["cb_wgrad"] = hot_cold(lambda: Mo._conv_wgrad(rt, gn, xn, conv), gy)
            except Exception as e:                   # noqa: BLE001
                row["cb_error"] = str(e)[:200]
        rows.append(row)
        print(json.dumps(row), flush=True)
    return rows


def md(enc, conv):
    out = ["# GEMM yardstick: vendor libraries next to cb_gemm on the step's shapes (same box, us per launch: hot / cold)", "",
           "hot = 16 launches back to back in a hipGraph; cold = each launch behind a 384 MB flush + a fresh write of its A operand.", "",
           "## Encoder products (M = 2624 token rows; hipBLASLt through torch.matmul / bmm, bf16)", "",
           "| product | kind | M x N x K | GF | hipBLASLt hot / cold | TF hot | cb_gemm hot / cold | TF hot | cb / lib (hot, cold) |", "|---|---|---|---:|---:|---:|---:|---:|---:|"]
    for r in enc:
        l, c = r["hipblaslt"], r["cb_gemm"]
        out.append(f"| {r['name']} | {r['kind']} | {r['M']}x{r['N']}x{r['K']} | {r['gflop']:.1f} | {l[0]} / {l[1]} | {r['gflop'] / l[0] * 1e3:.0f} | {c[0]} / {c[1]} | "
                   f"{r['gflop'] / c[0] * 1e3:.0f} | {c[0] / l[0]:.2f}, {c[1] / l[1]:.2f} |")
        if "hipblaslt_bias" in r:
            l, c = r["hipblaslt_bias"], r["cb_gemm_bias"]
            out.append(f"| {r['name']} + bias | {r['kind']} | | | {l[0]} / {l[1]} | | {c[0]} / {c[1]} | | {c[0] / l[0]:.2f}, {c[1] / l[1]:.2f} |")
    if conv:
        out += ["", "## ResNet convolutions, 64 frames of 224 px (MIOpen through F.conv2d / torch.nn.grad, NHWC bf16, immediate mode)", "",
                "| convolution | GF | MIOpen fwd | cb fwd (+FrozenBN+ReLU) | MIOpen dgrad | cb dgrad | MIOpen wgrad | cb wgrad |", "|---|---:|---:|---:|---:|---:|---:|---:|"]
        f = lambda v: f"{v[0]} / {v[1]}" if v else "-"
        for r in conv:
            out.append(f"| {r['name']} ({r['cin']}->{r['cout']} @{r['hw']}) | {r['gflop']:.1f} | {f(r.get('miopen_fwd'))} | {f(r.get('cb_fwd'))} | {f(r.get('miopen_dgrad'))} | "
                       f"{f(r.get('cb_dgrad'))} | {f(r.get('miopen_wgrad'))} | {f(r.get('cb_wgrad'))} |")
    return "\n".join(out) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "yardstick"))
    ap.add_argument("--skip-conv", action="store_true")
    ap.add_argument("--wgrad-set", action="store_true", help="only the encoder's whole weight-gradient set: one grouped launch vs four vs the library")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.wgrad_set:
        res = wgrad_set()
        print(json.dumps(res))
        with open(os.path.join(a.out, "wgrad_set.json"), "w") as f:
            json.dump(res, f, indent=1)
        return
    enc = encoder_rows()
    with open(os.path.join(a.out, "yardstick.json"), "w") as f:
        json.dump({"encoder": enc}, f, indent=1)
    conv = []
    if not a.skip_conv:
        try:
            conv = conv_rows()
        except Exception as e:                       # noqa: BLE001
            print("conv yardstick failed:", e, flush=True)
    with open(os.path.join(a.out, "yardstick.json"), "w") as f:
        json.dump({"encoder": enc, "conv": conv}, f, indent=1)
    with open(os.path.join(a.out, "yardstick.md"), "w") as f:
        f.write(md(enc, conv))
    print(md(enc, conv))


if __name__ == "__main__":
    main()
