# Round 6, call L: tile / K split of the ResNet's grouped weight gradients per problem shape, judged by the captured step (tools/tune_instep.py run_wgrad_groups)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06l; mkdir -p $O; cd $R
(CB_BENCH_TUNE_WGRAD=$O/wgrad_tuning.json timeout 1500 python bench.py --no-cpu-baseline --no-roofline) > $O/tune.log 2>&1
grep -E "^\[wgrad\]" $O/tune.log | cut -c1-400 | tail -30
