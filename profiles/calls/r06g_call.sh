# Round 6, call G: (1) GPU tests that call F did not reach (-x stopped at the streaming kernel's bit-equality: scale + shift is now one explicit
# fma in every path), (2) A/B: norm shares on / off (CB_BENCH_NO_FOLD=1: full sq_sum pass) x write-through / plain stores of the fp32 gradients
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
(timeout 2400 python -m pytest tests/test_gemm_stream.py tests/test_gemm_group.py tests/test_res2_block.py tests/test_norm_fold.py tests/test_bench_step.py tests/test_gemm_plan.py tests/test_kernels_gemm.py tests/test_kernels_gemm8.py tests/test_gpu_full.py tests/test_parity_record.py tests/test_zz_bench_cli.py -m gpu -q -p no:cacheprovider) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; grep FAILED $O/pytest_gpu.log | head
for i in 1 2 3; do
  (CB_BENCH_NO_FOLD=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_nofold_$i.log 2>&1; echo "full pass, plain stores:  $(grep -E 'timed region' $O/bench_nofold_$i.log | sed 's/.*done: //')"
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_fold_$i.log 2>&1; echo "shares, plain stores:     $(grep -E 'timed region' $O/bench_fold_$i.log | sed 's/.*done: //')"
  (CB_WG_STORE_WT=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_foldwt_$i.log 2>&1; echo "shares, write-through:    $(grep -E 'timed region' $O/bench_foldwt_$i.log | sed 's/.*done: //')"
done
