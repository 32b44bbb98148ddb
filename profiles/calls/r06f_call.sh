# Round 6, call F: first-writer stores for the ResNet weight gradients + the squared norm from per-launch shares (no sq_sum pass, no zero fill
# of the CNN range); the pruned tree.  Full GPU suite (log kept), bench x3, determinism, trace.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
(timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
for i in 1 2 3; do (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$i.log 2>&1; echo "bench: $(grep -E 'timed region' $O/bench_$i.log | sed 's/.*done: //')"; done
(timeout 300 python tools/replay_determinism.py --replays 3 --eager 2) > $O/determinism.txt 2>&1; tail -2 $O/determinism.txt | cut -c1-200
cd /tmp; export TMPDIR=/tmp
mkdir -p $O/trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
(cd $R && timeout 300 python tools/gemm_breakdown.py > $O/trace/breakdown.log 2>&1; cp gpurun_out/gemm_calls.json $O/trace/gemm_calls.json)
python $R/tools/join_calls_trace.py $O/trace/gemm_calls.json $O/trace/bench_kernel_trace.csv > $O/gemm_by_shape_instep.txt 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; head -40 $O/train_step.md
python $R/tools/step_timeline.py $O/trace/bench_kernel_trace.csv > $O/step_phases.txt 2>&1; cat $O/step_phases.txt
rm -f $O/trace/bench_kernel_trace.csv
grep "weight gradients" $O/gemm_by_shape_instep.txt | head -12
