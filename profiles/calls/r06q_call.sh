# Round 6, call Q: the 3x3 convolutions' weight gradients on the 8-wave LDS-DMA gather tiles (own launches, slab K split) instead of the grouped 4-wave tile, judged by the step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R
(CB_TUNE_WGRAD_8W=1 CB_BENCH_TUNE_WGRAD=$O/wgrad8_tuning.json timeout 1200 python bench.py --no-cpu-baseline --no-roofline) > $O/tune.log 2>&1
grep -E "^\[wgrad\]" $O/tune.log | cut -c1-600 | tail -8
