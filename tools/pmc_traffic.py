#!/usr/bin/env python
"""HBM traffic per cb_gemm launch, by kernel family, from two rocprofv3 counter passes over tools/gemm_breakdown.py
(one eager training step of the bench workload with every cb_gemm call logged):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -o f -- python tools/gemm_breakdown.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -o w -- python tools/gemm_breakdown.py
    python tools/pmc_traffic.py gpurun_out/gemm_calls.json out/f/f_counter_collection.csv out/w/w_counter_collection.csv profiles/rNN_pmc_traffic.json

Units / gfx950 correction as prescribed by MI355X_MICROARCH.md: both counters are KiB; FETCH_SIZE counts 64 B per 128 B
request on gfx950, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The calibration on adamw_kernel (exactly
16 B read + 14 B written per parameter) is re-checked and stored next to the result."""
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


def main():
    calls = json.load(open(sys.argv[1]))
    fetch = per_dispatch(sys.argv[2], "FETCH_SIZE")
    write = per_dispatch(sys.argv[3], "WRITE_SIZE")

    def gemms(d):
        # one entry per cb_gemm call; the split-K reduce kernel that follows an 8-wave launch is part of that call
        g = []
        for name, val in d:
            if "splitk_reduce_kernel" in name:
                if g:
                    g[-1][1] += val
            elif "gemm_kernel" in name or "gemm8_kernel" in name or "gemm_dma_kernel" in name:
                g.append([name, val])
        return g[-len(calls):]
    gf, gw = gemms(fetch), gemms(write)
    assert len(gf) == len(calls) == len(gw), (len(gf), len(gw), len(calls))
    fam = collections.defaultdict(lambda: dict(launches=0, fetch_kib=0.0, write_kib=0.0, algorithmic_bytes=0.0, flop=0.0))
    for c, f, w in zip(calls, gf, gw):
        key = f"cb_gemm<bf16> {c['form']}{' (implicit-GEMM conv)' if c['conv'] else ''}"
        a = fam[key]
        nb = c.get("batch", 1)
        taps = c.get("R", 1) * c.get("S", c.get("R", 1)) if c["conv"] else 1
        m, n, k = c["M"], c["N"], c["K"]
        if c["form"] == "wgrad":                      # A (k x m) + B (k x n, gathered input counted once) + fp32 C read+write
            alg = nb * ((k * m + k * n / taps) * c["esz"] + 2 * m * n * c["c_esz"])
        else:                                         # A (m x k, gathered input counted once) + B + C + the epilogue's M x N operands
            alg = nb * ((m * k / taps + n * k) * c["esz"] + m * n * c["c_esz"] + c.get("extra_mn", 0) * m * n * c["esz"])
        a["launches"] += 1; a["fetch_kib"] += f[1]; a["write_kib"] += w[1]; a["algorithmic_bytes"] += alg
        a["flop"] += 2.0 * m * n * k * nb
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python tools/gemm_breakdown.py "
                     "(one eager training step of the bench workload)",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (KiB counters; gfx950 FETCH_SIZE counts 64 B per 128 B request)",
           "families": {}}
    for k, a in fam.items():
        n = a["launches"]
        hbm = (2 * a["fetch_kib"] + a["write_kib"]) * 1024
        out["families"][k] = {"launches_in_trace": n, "hbm_bytes_per_launch": round(hbm / n),
                              "algorithmic_bytes_per_launch": round(a["algorithmic_bytes"] / n),
                              "hbm_over_algorithmic": round(hbm / a["algorithmic_bytes"], 2),
                              "fetch_kib_raw_per_launch": round(a["fetch_kib"] / n, 1), "write_kib_per_launch": round(a["write_kib"] / n, 1),
                              "flop_per_launch": round(a["flop"] / n)}
    ad_f = [x[1] for x in fetch if "adamw" in x[0]]
    ad_w = [x[1] for x in write if "adamw" in x[0]]
    if ad_f and ad_w:
        nsteps = max(1, len(ad_f) // 4)
        out["calibration_adamw"] = {"fetch_bytes_per_step": round(2 * sum(ad_f) * 1024 / nsteps), "write_bytes_per_step": round(sum(ad_w) * 1024 / nsteps),
                                    "expected": "16 B read + 14 B written per parameter (p, g, m, v fp32 in; p, m, v fp32 + bf16 copy out)"}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    for k, v in out["families"].items():
        print(k, v)
    print(out.get("calibration_adamw"))


if __name__ == "__main__":
    main()
