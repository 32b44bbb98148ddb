# Round 6, call AH: zero_grad's four fills in one launch (cb_zero_ranges): tests, bench, kernel counts
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ah; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_kernels_misc.py tests/test_bench_step.py tests/test_norm_fold.py tests/test_loops.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -2
for i in 1 2 3; do (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$i.log 2>&1; echo "bench: $(grep -E 'timed region' $O/bench_$i.log | sed 's/.*done: //')"; done
(timeout 600 python tools/replay_determinism.py) 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
mkdir -p $O/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; grep -E "zero|One steady" $O/train_step.md
rm -f $O/trace/bench_kernel_trace.csv
