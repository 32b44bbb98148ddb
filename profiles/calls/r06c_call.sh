# Round 6, call C (run twice: first with the three 8-wave kinds, then -- this record -- with the flags-templated fast epilogue on every tile):
# (CB_GEMM_FAST_EPI=0), alternating; kernel tests; per-shape in-step durations of both arms
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_kernels_gemm8.py tests/test_kernels_gemm.py tests/test_bench_step.py tests/test_parity_record.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do
  (CB_GEMM_FAST_EPI=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_generic_$i.log 2>&1; echo "generic epilogue: $(grep -E 'timed region' $O/bench_generic_$i.log | sed 's/.*done: //')"
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_fast_$i.log 2>&1; echo "fast epilogue:    $(grep -E 'timed region' $O/bench_fast_$i.log | sed 's/.*done: //')"
done
cd /tmp; export TMPDIR=/tmp
for arm in fast generic; do
  mkdir -p $O/trace_$arm
  if [ $arm = generic ]; then export CB_GEMM_FAST_EPI=0; else unset CB_GEMM_FAST_EPI; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$arm -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace_$arm/bench.log 2>&1
  (cd $R && timeout 300 python tools/gemm_breakdown.py > $O/trace_$arm/breakdown.log 2>&1; cp gpurun_out/gemm_calls.json $O/trace_$arm/gemm_calls.json)
  python $R/tools/join_calls_trace.py $O/trace_$arm/gemm_calls.json $O/trace_$arm/bench_kernel_trace.csv > $O/gemm_by_shape_instep_$arm.txt 2>&1
  python $R/tools/trace_summary.py $O/trace_$arm/bench_kernel_trace.csv > $O/train_step_$arm.md 2>&1
  rm -f $O/trace_$arm/bench_kernel_trace.csv
  grep "encoder linear" $O/gemm_by_shape_instep_$arm.txt | head -10
done
