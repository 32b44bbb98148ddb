#!/usr/bin/env python
"""Round-5 debugging aid: why do the CNN weight gradients of the captured step turn to garbage from the second replay on?
   python tools/replay_debug.py [--videos 4] [--restore 0/1] [--prezero 0/1]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=4)
    ap.add_argument("--restore", type=int, default=0)
    ap.add_argument("--prezero", type=int, default=0)
    ap.add_argument("--replays", type=int, default=4)
    a = ap.parse_args()
    from clipbert_amd.bench import step as bench_step
    st = bench_step.build(videos=a.videos, dropout=False)
    bank, opt = st.bank, st.opt
    init = dict(master=bank.master.clone(), m=bank.exp_avg.clone(), v=bank.exp_avg_sq.clone(), w16=bank.w16.clone())
    st.host_prepare(); st.device_step(); torch.cuda.synchronize()
    ref = bank.grad.clone()
    graph, loss = st.capture()
    def report(tag):
        g = bank.grad
        bad = []
        for name, p in bank._trainable:
            off = bank.offset[id(p)]
            x = g[off:off + p.numel()]
            r = ref[off:off + p.numel()]
            fin = bool(torch.isfinite(x).all())
            d = float((x - r).abs().max()) if fin else float("nan")
            if not fin or d > 1e-3 * float(r.abs().max()) + 1e-12:
                bad.append((name, fin, d))
        gr = [(gi, a0, b0) for gi, (a0, b0) in enumerate(bank.group_range)]
        print(f"{tag}: loss {float(loss):.5f} norm {opt.grad_norm():.4e} bad tensors {len(bad)} e.g. {bad[:3]}  master finite {bool(torch.isfinite(bank.master).all())}", flush=True)
    for i in range(a.replays):
        if a.restore:
            bank.master.copy_(init["master"]); bank.exp_avg.copy_(init["m"]); bank.exp_avg_sq.copy_(init["v"]); bank.w16.copy_(init["w16"])
        if a.prezero:
            bank.grad.zero_()
        st.host_prepare()
        graph.replay()
        torch.cuda.synchronize()
        report(f"replay{i}")
    print("lazy span", bank.lazy_span, "groups", bank.group_range, "n_train", bank.n_train)

if __name__ == "__main__":
    main()
