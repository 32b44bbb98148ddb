# Round 6, call X: launch table of the TGIF row (bench.py --mode tgif, BASELINE configs[3]) judged by its captured step (tools/tune_instep.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06x; mkdir -p $O; cd $R
(CB_BENCH_TUNE=$O/instep_tgif.json timeout 1500 python bench.py --mode tgif --no-cpu-baseline --no-roofline) > $O/tune.log 2>&1
grep -E "^\[instep\]" $O/tune.log | grep -E "KEEP|baseline|overrides" | cut -c1-330 | tail -30
