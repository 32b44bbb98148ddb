# Round 5, call P: grouped weight-gradient launch model (rounds x K loop + atomics at ~1 TB/s) vs the round-4 model
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R
for i in 1 2; do
  (CB_GROUP_MODEL=r4 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_r4_$i.log 2>&1; echo "group model r4: $(grep -E 'timed region' $O/bench_r4_$i.log | sed 's/.*done: //')"
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_new_$i.log 2>&1; echo "group model r5: $(grep -E 'timed region' $O/bench_new_$i.log | sed 's/.*done: //')"
done
(CB_GEMM_TRACE=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1) 2>&1 | grep "cb_gemm_group\[" | sort | uniq -c | sort -rn | head -60 > $O/group_plan_new.txt
(CB_GROUP_MODEL=r4 CB_GEMM_TRACE=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1) 2>&1 | grep "cb_gemm_group\[" | sort | uniq -c | sort -rn | head -60 > $O/group_plan_r4.txt
