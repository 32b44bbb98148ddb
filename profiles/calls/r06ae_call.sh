# Round 6, call AE: a wider in-step search on the metric step: the 24 heaviest shapes x up to 12 configurations each
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ae; mkdir -p $O; cd $R
(CB_BENCH_TUNE=$O/instep_train_wide.json CB_BENCH_TUNE_SHAPES=24 CB_BENCH_TUNE_CANDS=12 timeout 2400 python bench.py --no-cpu-baseline --no-roofline) > $O/tune.log 2>&1
grep -E "^\[instep\]" $O/tune.log | grep -E "KEEP|baseline|overrides" | cut -c1-400 | tail -20
