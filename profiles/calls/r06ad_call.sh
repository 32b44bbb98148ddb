# Round 6, call AD: the metric step's launch table judged again by the captured step on the final kernels (tools/tune_instep.py), then the rows with the merged 448 table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ad; mkdir -p $O; cd $R
(CB_BENCH_TUNE=$O/instep_train.json timeout 1500 python bench.py --no-cpu-baseline --no-roofline) > $O/tune.log 2>&1
grep -E "^\[instep\]" $O/tune.log | grep -E "KEEP|baseline|overrides" | cut -c1-330 | tail -20
for m in "--size 448 --txt-len 20 --n-clips 4" "--mode tgif" "--mode infer16" ""; do (timeout 300 python bench.py $m --no-cpu-baseline --no-roofline) > $O/bench.log 2>&1; echo "bench $m: $(grep -E 'timed region' $O/bench.log | sed 's/.*done: //' | head -1)"; done
