R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ab; mkdir -p $O; cd $R
for m in "--mode infer16" "--mode infer16" ""; do (timeout 300 python bench.py $m --no-cpu-baseline --no-roofline) > $O/bench.log 2>&1; echo "bench $m: $(grep -E 'timed region' $O/bench.log | sed 's/.*done: //' | head -1)"; done
