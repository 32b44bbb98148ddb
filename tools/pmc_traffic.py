#!/usr/bin/env python
"""HBM traffic per GEMM launch, by family (clipbert_amd/gemm_log.py: the families of bench.py's roofline), from two rocprofv3 counter
passes over tools/gemm_breakdown.py (one eager training step of the bench workload with every cb_gemm / cb_gemm_group call logged):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -o f -- python tools/gemm_breakdown.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -o w -- python tools/gemm_breakdown.py
    python tools/pmc_traffic.py gpurun_out/gemm_calls.json out/f/f_counter_collection.csv out/w/w_counter_collection.csv profiles/rNN_pmc_traffic.json

Units / gfx950 correction as prescribed by MI355X_MICROARCH.md: both counters are KiB; FETCH_SIZE counts 64 B per 128 B
request on gfx950, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The calibration on adamw_kernel (exactly
16 B read + 14 B written per parameter) is re-checked and stored next to the result.

A library call may launch several kernels (a grouped call: one per chunk; an 8-wave K split: the GEMM + its slab reduce, folded
together here); gemm_calls.json says how many, so the LAST sum(kernels) GEMM dispatches of the trace are matched to the calls in order.
Families are also reported by FORM (fwd / dgrad / wgrad x linear / conv) for comparison with the round-3 table."""
import collections
import csv
import json
import sys


def per_dispatch(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    by = collections.OrderedDict()
    for r in rows:                                    # one row per (dispatch, counter[, dimension]): sum the dimensions
        key = int(r["Dispatch_Id"])
        by.setdefault(key, [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def gemm_dispatches(d):
    """one entry per GEMM kernel; the split-K reduce kernel that follows an 8-wave launch is part of that launch"""
    g = []
    for name, val in d:
        if "splitk_reduce_kernel" in name:
            if g:
                g[-1][1] += val
        elif "gemm_kernel" in name or "gemm8_kernel" in name or "gemm_dma_kernel" in name or "gemm_group_kernel" in name or "gemm8p_kernel" in name or "gemm_stream_kernel" in name or "gemm_streamk_kernel" in name or "res2_block_kernel" in name or "stem_pool_kernel" in name:       # (round 5: the fused front-end launches are logged calls too)
            g.append([name, val])
    return g


def xcd_floor(p):
    """Bytes a product must move when 8 non-coherent L2s each serve 1/8 of the tiles: every XCD fetches the operand panels its tiles touch.
    With the tiles of an XCD forming an (M / xm) x (N / xn) block (xm * xn = 8), A is fetched xn times and B xm times; outputs and epilogue
    operands once.  The best split is the floor; a launch cannot do better without cross-XCD sharing, whatever its tile order.  (K = the reduction length, A = M x K, B = N x K
    in bf16; round 6: convolutions too, with the pixel operand counted once per tap-free element.)"""
    a, b = 2.0 * p["M"] * p["K"] * p["batch"], 2.0 * p["N"] * p["K"] * p["batch"]
    if p["taps"] != 1:                        # implicit-GEMM convolution: the pixel operand holds M x K / taps unique elements (halo rows of a tile not
        if p["form"] == "wgrad":              # counted: a floor); the filter is the N x K operand
            return p["bytes"]
        a = a / p["taps"]
    rest = max(0.0, p["bytes"] - a - b)
    if p["form"] == "wgrad":                 # the reduction is the long axis: an XCD can own a K range instead (split-K), no operand re-read
        return p["bytes"]
    return rest + min(a * xn + b * xm for xm, xn in ((8, 1), (4, 2), (2, 4), (1, 8)))


def main():
    calls = json.load(open(sys.argv[1]))
    fetch = per_dispatch(sys.argv[2], "FETCH_SIZE")
    write = per_dispatch(sys.argv[3], "WRITE_SIZE")
    nk = sum(c["kernels"] for c in calls)
    gf, gw = gemm_dispatches(fetch)[-nk:], gemm_dispatches(write)[-nk:]
    assert len(gf) == nk == len(gw), (len(gf), len(gw), nk)

    def new():
        return dict(launches=0, kernels=0, problems=0, fetch_kib=0.0, write_kib=0.0, algorithmic_bytes=0.0, flop=0.0, floor=0.0, floor_ok=True)
    fam, form = collections.defaultdict(new), collections.defaultdict(new)
    pos = 0
    for c in calls:
        f = sum(x[1] for x in gf[pos:pos + c["kernels"]])
        w = sum(x[1] for x in gw[pos:pos + c["kernels"]])
        pos += c["kernels"]
        fams = sorted({p["family"] for p in c["problems"]})
        forms = sorted({f"cb_gemm<bf16> {p['form']}{' (implicit-GEMM conv)' if p['taps'] > 1 else ''}" for p in c["problems"]})
        for table, key in ((fam, fams[0] if len(fams) == 1 else "mixed group"), (form, forms[0] if len(forms) == 1 else "mixed group")):
            a = table[key]
            a["launches"] += 1; a["kernels"] += c["kernels"]; a["problems"] += len(c["problems"])
            a["fetch_kib"] += f; a["write_kib"] += w
            a["algorithmic_bytes"] += sum(p["bytes"] for p in c["problems"]); a["flop"] += sum(p["flop"] for p in c["problems"])
            fl = [xcd_floor(p) for p in c["problems"]]
            if any(x is None for x in fl):
                a["floor_ok"] = False
            else:
                a["floor"] += sum(fl)

    def table(t):
        out = {}
        for k, a in t.items():
            n = a["launches"]
            hbm = (2 * a["fetch_kib"] + a["write_kib"]) * 1024
            out[k] = {"launches_in_trace": n, "kernels": a["kernels"], "problems": a["problems"], "hbm_bytes_per_launch": round(hbm / n),
                      "algorithmic_bytes_per_launch": round(a["algorithmic_bytes"] / n), "hbm_over_algorithmic": round(hbm / a["algorithmic_bytes"], 2),
                      "hbm_mbytes_per_step": round(hbm / 1e6, 1), "algorithmic_mbytes_per_step": round(a["algorithmic_bytes"] / 1e6, 1),
                      "fetch_kib_raw_per_launch": round(a["fetch_kib"] / n, 1), "write_kib_per_launch": round(a["write_kib"] / n, 1),
                      "flop_per_launch": round(a["flop"] / n),
                      "xcd_floor_over_algorithmic": round(a["floor"] / a["algorithmic_bytes"], 2) if a["floor_ok"] else None,
                      "hbm_over_xcd_floor": round(hbm / a["floor"], 2) if a["floor_ok"] else None}
        return out
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python tools/gemm_breakdown.py "
                     "(one eager training step of the bench workload)",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (KiB counters; gfx950 FETCH_SIZE counts 64 B per 128 B request)",
           "xcd_floor": "bytes a product moves at best when 8 non-coherent L2s each serve 1/8 of its tiles (tools/pmc_traffic.py: xcd_floor); "
                        "1x1 / linear families only",
           "families": table(fam), "by_form": table(form)}
    ad_f = [x[1] for x in fetch if "adamw" in x[0]]
    ad_w = [x[1] for x in write if "adamw" in x[0]]
    if ad_f and ad_w:
        nsteps = max(1, len(ad_f) // 4)
        out["calibration_adamw"] = {"fetch_bytes_per_step": round(2 * sum(ad_f) * 1024 / nsteps), "write_bytes_per_step": round(sum(ad_w) * 1024 / nsteps),
                                    "expected": "16 B read + 14 B written per parameter (p, g, m, v fp32 in; p, m, v fp32 + bf16 copy out)"}
    # the step's fills (zero_kernel, runtime fillBuffer): bytes written per step
    zero = [x for x in write if "zero_kernel" in x[0] or "fillBuffer" in x[0]]
    if zero:
        nsteps = 2                                        # gemm_breakdown.py runs the eager step twice
        out["fills"] = {"launches_per_step": len(zero) // nsteps, "write_mbytes_per_step": round(sum(x[1] for x in zero) * 1024 / nsteps / 1e6, 1)}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    for k, v in out["families"].items():
        print(k, v)
    for k, v in out["by_form"].items():
        print(k, {kk: v[kk] for kk in ("launches_in_trace", "hbm_over_algorithmic", "xcd_floor_over_algorithmic", "hbm_over_xcd_floor", "hbm_mbytes_per_step")})
    print(out.get("calibration_adamw"), out.get("fills"))


if __name__ == "__main__":
    main()
