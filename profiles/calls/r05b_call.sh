# Round 5, call B: rolled epilogue loops -- bench (plain / write-through, alternating), stamps, the GEMM GPU tests + the timed-step parity test
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
(timeout 600 python bench.py --no-cpu-baseline) > $O/bench_base.log 2>&1; grep '^{' $O/bench_base.log > $O/bench_base.json; grep -E "timed region" $O/bench_base.log
for i in 1 2; do
  (CB_GEMM_WT=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_wt$i.log 2>&1; grep -E "timed region" $O/bench_wt$i.log
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_plain$i.log 2>&1; grep -E "timed region" $O/bench_plain$i.log
done
(timeout 600 python tools/stamps_run.py --out $O/stamps) > $O/stamps.log 2>&1; tail -1 $O/stamps.log | cut -c1-200; grep "stamps\]" $O/stamps.log
(time timeout 900 python -m pytest tests/test_bench_step.py tests/test_kernels_gemm.py tests/test_kernels_gemm8.py tests/test_gemm_stream.py tests/test_gemm_group.py -x -q -m gpu) > $O/pytest_sub.log 2>&1; tail -5 $O/pytest_sub.log
