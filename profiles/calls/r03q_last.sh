# Last call of round 3: the GPU suite + smoke + bench on the final HEAD (test-only and N > 1-only Python changes since the r03z set),
# and the auto-configured cb_gemm on large plain GEMMs (tile = 0: table / cost model) next to explicit tiles.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O
cd $R
(time timeout 1200 python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; grep smoke $O/smoke.log
(time timeout 600 python bench.py) > $O/bench.log 2>&1; grep -E "timed region" $O/bench.log
timeout 300 python tools/gemm8_probe.py --quick --out $O/gemm8_probe_quick.json 2>&1 | grep "\[probe\]" | tee $O/gemm8_probe_quick.log
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03q", "gemm8_probe_quick.json")))
for r in d["rows"]:
    fl = 2.0 * r["M"] * r["N"] * r["K"] * r["batch"]
    print(f'{r["form"]:5s} {r["M"]}x{r["N"]}x{r["K"]} b{r["batch"]}: auto {fl / r["us"]["auto"] / 1e6:.0f} TF, best explicit {r["best"]} {fl / r["us"][r["best"]] / 1e6:.0f} TF')
PY
