# Round 6, call D: (1) LayerNorm backward geometries with the full-row specialisation; (2) the launch table judged again by the whole
# captured step now that the epilogues are cheaper (tools/tune_instep.py inside bench.py, candidates from the round-5 cold sweep)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
timeout 600 python tools/ln_bwd_probe.py 2>&1 | tee $O/ln_bwd_probe.txt | tail -40
(CB_BENCH_TUNE=$O/instep_tuning.json CB_BENCH_TUNE_CAND=$R/profiles/r05g_gemm_tuning_cold_train.json timeout 1500 python bench.py --no-cpu-baseline --no-roofline) > $O/instep.log 2>&1
tail -60 $O/instep.log | cut -c1-220
