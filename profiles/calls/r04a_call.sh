# Round-4 first GPU call: cb_gemm_group parity on the GPU, stage-level probe, bench A/B (grouped vs single ResNet weight gradients), kernel trace.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O/trace
cd $R
(time timeout 600 python -m pytest tests/test_gemm_group.py tests/test_model_small.py -x -q -m gpu) > $O/pytest_group.log 2>&1; tail -3 $O/pytest_group.log
(time timeout 600 python tools/group_probe.py --out $O/group_probe.json) > $O/group_probe.log 2>&1; tail -4 $O/group_probe.log | cut -c1-600
for i in 1 2; do
(timeout 300 env CB_NO_GROUP_WGRAD=1 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_single_$i.json 2> $O/bench_single_$i.err; grep -E "timed region" $O/bench_single_$i.err
(timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_group_$i.json 2> $O/bench_group_$i.err; grep -E "timed region" $O/bench_group_$i.err
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; head -12 $O/train_step.md
gzip -f $O/trace/bench_kernel_trace.csv
