#!/usr/bin/env python
"""In-situ decomposition of every cb_gemm launch of the captured metric step ("decompose the per-tile fixed cost").

Runs the step on the DIAGNOSTIC build of the library (python -m clipbert_amd.build --stamps -> lib/libclipbert_hip_stamps.so: thread 0
of every workgroup stamps the chip-wide 100 MHz counter at kernel entry / first K tile in LDS / K loop done / epilogue issued / stores
retired), replays the hipGraph a few times and writes, for the LAST replay, one row per launch:

    start spread, prologue, K loop, epilogue, store drain (medians and max over the workgroups), the launch's wall time
    (first entry -> last store retired) and the GAP to the next stamped launch (last retire -> first entry).

    python tools/stamps_run.py [--out gpurun_out/stamps] [--replays 6]

The stamps cost a few hundred cycles per workgroup (5 s_memrealtime + one extra s_waitcnt vmcnt(0) at the end); the step under
stamps is timed next to the same step on the product library so that the perturbation is visible.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stamps"))
    ap.add_argument("--replays", type=int, default=6)
    ap.add_argument("--videos", type=int, default=16)
    ap.add_argument("--lib", default="stamps", help="library variant (a CB_STAMPS build): lib/libclipbert_hip_<lib>.so")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    from clipbert_amd import _lib
    from clipbert_amd.build import variant_path
    path = variant_path(a.lib)
    lib = _lib.load(path)
    _lib._LIB = lib                                   # every ops.* call of this process goes to the diagnostic build
    lib.cb_debug_stamps_begin.argtypes = [C.c_void_p, C.c_int64]
    lib.cb_debug_stamps_area_words.restype = C.c_int64
    lib.cb_debug_stamps_count.restype = C.c_int64
    lib.cb_debug_stamps_desc.argtypes = [C.c_int64, C.c_char_p, C.c_int64]
    from clipbert_amd.bench import step as bench_step
    st = bench_step.build(videos=a.videos)
    dev = st.dev
    words = int(lib.cb_debug_stamps_area_words())
    n_areas = 400
    buf = torch.zeros(n_areas * words, dtype=torch.int64, device=dev)

    def reset():
        buf.zero_()
        buf.view(n_areas, words)[:, 0] = -1          # min start = ~0

    # warm up eagerly WITHOUT stamps (table builds, allocations), then capture WITH them
    lib.cb_debug_stamps_begin(None, 0)
    for _ in range(2):
        st.host_prepare()
        st.device_step()
    torch.cuda.synchronize()
    lib.cb_debug_stamps_begin(C.c_void_p(buf.data_ptr()), buf.numel() * 8)
    g, _loss = st.capture()
    n = int(lib.cb_debug_stamps_count())
    descs = []
    for i in range(n):
        b = C.create_string_buffer(1024)
        lib.cb_debug_stamps_desc(i, b, 1024)
        descs.append(json.loads(b.value.decode()))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        st.host_prepare()
        g.replay()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(a.replays):
        st.host_prepare()
        g.replay()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / a.replays
    reset()
    torch.cuda.synchronize()
    st.host_prepare()
    g.replay()
    torch.cuda.synchronize()
    raw = buf.view(n_areas, words)[:n].cpu().numpy().astype(np.uint64)
    np.savez_compressed(os.path.join(a.out, "stamps_raw.npz"), raw=raw, descs=json.dumps(descs))
    rows = summarize(raw, descs)
    with open(os.path.join(a.out, "stamps.json"), "w") as f:
        json.dump({"ms_per_step_stamped": ms, "launches": rows}, f, indent=0)
    print(f"[stamps] {n} stamped launches, step under stamps {ms:.3f} ms")
    print(table(rows))
    with open(os.path.join(a.out, "stamps.md"), "w") as f:
        f.write(f"step under stamps: {ms:.3f} ms/step, {n} stamped cb_gemm launches (ticks of 10 ns)\n\n" + table(rows) + "\n")


HDR, REC, WGS = 3, 6, 512
TILES = {1: "128x128", 2: "64x64", 3: "128x64", 4: "128x128o2", 5: "8w256x256", 6: "8w128x256", 7: "8w256x128", 8: "stream"}


def summarize(raw, descs):
    rows = []
    t_first = None
    for i, d in enumerate(descs):
        a = raw[i]
        t_min, t_max, nwg = int(a[0]), int(a[1]), int(a[2])
        if nwg == 0:
            continue
        rec = a[HDR:HDR + min(nwg, WGS) * REC].reshape(-1, REC).astype(np.int64)
        rec = rec[rec[:, 0] > 0]
        if t_first is None:
            t_first = t_min
        us = lambda x: float(x) / 100.0
        t0, t1, t2, t3, t4 = (rec[:, k] for k in range(5))
        row = dict(d)
        row.update(i=i, wgs=nwg, at_us=us(t_min - t_first), wall_us=us(t_max - t_min),
                   start_spread_us=us(np.percentile(t0 - t_min, 95)), start_max_us=us((t0 - t_min).max()),
                   prologue_med_us=us(np.median(t1 - t0)), prologue_max_us=us((t1 - t0).max()),
                   kloop_med_us=us(np.median(t2 - t1)), kloop_max_us=us((t2 - t1).max()),
                   epi_med_us=us(np.median(t3 - t2)), epi_max_us=us((t3 - t2).max()),
                   drain_med_us=us(np.median(t4 - t3)), drain_max_us=us((t4 - t3).max()),
                   wg_life_med_us=us(np.median(t4 - t0)), wg_life_max_us=us((t4 - t0).max()),
                   xcds=int(len(np.unique(rec[:, 5] >> 32))), t_min=t_min, t_max=t_max)
        rows.append(row)
    for j in range(len(rows) - 1):
        rows[j]["gap_next_us"] = (rows[j + 1]["t_min"] - rows[j]["t_max"]) / 100.0
    return rows


def table(rows):
    out = ["| # | at us | problem M x N x K (modes, batch) | tile s/sched | wgs | wall | start p95 | prologue med/max | K loop med/max | epilogue med/max | drain med/max | gap->next |",
           "|---|---:|---|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        flags = "".join(k[0] for k in ("c2", "residual", "dropout", "mask", "gelu_grad", "relu_bwd") if r.get(k)) + ("A" if r.get("act") else "")
        out.append(f"| {r['i']} | {r['at_us']:.0f} | {r['M']}x{r['N']}x{r['K']} ({r['a_mode']}/{r['b_mode']}, b{r['batch']}, t{r['taps']}{' ' + flags if flags else ''}"
                   f"{' g%d/%d' % (r['group_i'], r['group_n']) if r['group_n'] > 1 else ''}) | {TILES.get(r['tile'], r['tile'])} {r['split']}/{r['sched']} | {r['wgs']} | "
                   f"{r['wall_us']:.1f} | {r['start_spread_us']:.1f} | {r['prologue_med_us']:.1f}/{r['prologue_max_us']:.1f} | "
                   f"{r['kloop_med_us']:.1f}/{r['kloop_max_us']:.1f} | {r['epi_med_us']:.1f}/{r['epi_max_us']:.1f} | "
                   f"{r['drain_med_us']:.1f}/{r['drain_max_us']:.1f} | {r.get('gap_next_us', float('nan')):.1f} |")
    return "\n".join(out)


if __name__ == "__main__":
    main()
