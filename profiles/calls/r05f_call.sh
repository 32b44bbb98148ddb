# Round 5, call F: write-through default (A/B vs plain vs +nt on the 2nd output), kernel trace of the step, the full GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O/trace; cd $R
for i in 1 2; do
  for w in 0 1 3; do
    (CB_GEMM_WT=$w timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_wt${w}_$i.log 2>&1; echo "WT=$w: $(grep -E 'timed region' $O/bench_wt${w}_$i.log | sed 's/.*done: //')"
  done
done
(timeout 600 python bench.py) > $O/bench.log 2>&1; grep '^{' $O/bench.log > $O/bench.json; grep -E "timed region" $O/bench.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; head -45 $O/train_step.md | cut -c1-160
python $R/tools/step_timeline.py $O/trace/bench_kernel_trace.csv > $O/step_phases.txt 2>&1; cat $O/step_phases.txt
gzip -f $O/trace/bench_kernel_trace.csv
cd $R
(time timeout 1500 python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
