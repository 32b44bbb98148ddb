#!/usr/bin/env python
"""Does the captured training step give the same gradients every time it is replayed from the same state?

    python tools/replay_determinism.py [--lib VARIANT] [--replays 6] [--videos 16] [--dropout 0] [--eager 2]

Every run restores masters / moments / compute weights, replays (or runs eagerly) and snapshots the flat gradient buffer; tensors whose
gradient differs from run 0 by more than fp32-atomic-order noise (2e-6 of the tensor's max) are listed with the size of the difference.
--lib NAME loads clipbert_amd/lib/libclipbert_hip_NAME.so (python -m clipbert_amd.build --variant NAME --csrc DIR) instead of the product
library -- an A/B against an older source tree on the same box.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--replays", type=int, default=6)
    ap.add_argument("--eager", type=int, default=2)
    ap.add_argument("--videos", type=int, default=16)
    ap.add_argument("--dropout", type=int, default=0)
    a = ap.parse_args()
    from clipbert_amd import _lib
    if a.lib:
        from clipbert_amd.build import variant_path
        _lib._LIB = _lib.load(variant_path(a.lib))
    from clipbert_amd.bench import step as bench_step
    st = bench_step.build(videos=a.videos, dropout=bool(a.dropout))
    bank, opt = st.bank, st.opt
    init = dict(master=bank.master.clone(), m=bank.exp_avg.clone(), v=bank.exp_avg_sq.clone(), w16=bank.w16.clone())
    seed0 = st.model.rt.seed_dev.clone()

    def run(fn):
        bank.master.copy_(init["master"]); bank.exp_avg.copy_(init["m"]); bank.exp_avg_sq.copy_(init["v"]); bank.w16.copy_(init["w16"])
        st.model.rt.seed_dev.copy_(seed0)
        st.model.rt.forward_count = 0
        st.state["global_step"] = 0
        opt.step_count = 0
        st.host_prepare()
        out = fn()
        torch.cuda.synchronize()
        return bank.grad.clone(), float(opt.grad_norm()), out

    runs = []
    for i in range(a.eager):
        g, n, loss = run(st.device_step)
        runs.append((f"eager{i}", g, n))
        print(f"eager{i}: grad norm {n:.6e} loss {float(loss):.6f}", flush=True)
    graph, _ = st.capture()
    for i in range(a.replays):
        g, n, _ = run(graph.replay)
        runs.append((f"replay{i}", g, n))
        print(f"replay{i}: grad norm {n:.6e}", flush=True)
    ref_name, ref, _ = runs[0]
    bad_total = 0
    for name, g, n in runs[1:]:
        bad = []
        for pname, p in bank._trainable:
            off = bank.offset[id(p)]
            x, y = ref[off:off + p.numel()], g[off:off + p.numel()]
            scale = float(x.abs().max())
            d = float((x - y).abs().max())
            if not (d <= 2e-6 * scale + 1e-12):
                bad.append((pname, d, scale, int(((x - y).abs() > 2e-6 * scale + 1e-12).sum()), p.numel()))
        bad_total += len(bad)
        print(f"{name} vs {ref_name}: {len(bad)} tensors differ" + ("" if not bad else ": " + "; ".join(f"{b[0]} d={b[1]:.3e} max={b[2]:.3e} n={b[3]}/{b[4]}" for b in bad[:6])), flush=True)
    print("DETERMINISTIC" if bad_total == 0 else f"NOT DETERMINISTIC ({bad_total} tensor mismatches)")


if __name__ == "__main__":
    main()
