# schedule 4 (persistent 8-wave kernel, next tile's DMA requested before the epilogue) against schedules 0-2 and the 4-wave tiles
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O
timeout 500 python tools/gemm8_probe.py --quick --out $O/gemm8_probe.json 2>&1 | grep "\[probe\]" | tee $O/gemm8_probe.log
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03r", "gemm8_probe.json")))
for r in d["rows"]:
    fl = 2.0 * r["M"] * r["N"] * r["K"] * r["batch"]
    us = {k: v for k, v in r["us"].items() if v}
    best = {}
    for m in ("m0", "m1", "m2", "m3"):
        c = [k for k in us if k.startswith("8w") and f"/{m}" in k]
        if c:
            b = min(c, key=us.get)
            best[m] = f"{b} {fl / us[b] / 1e6:.0f} TF"
    print(f'{r["form"]:5s} {r["M"]}x{r["N"]}x{r["K"]} b{r["batch"]}: ' + " | ".join(f"{m}: {v}" for m, v in best.items()) + (f"  BAD {r['bad']}" if r["bad"] else ""))
PY
