# Round 5, last call: FFN1 stores gelu'(pre) for the backward (CB_ACT_GELU_SAVE_GRAD / CB_ACT_SAVED_GRAD) vs the pre-activation + GELU' in the
# FFN2 data-gradient epilogue (CB_NO_GELU_SAVE_GRAD=1), alternating; parity tests of the kernels and of the timed step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05zz; mkdir -p $O; cd $R
for i in 1 2 3; do
  (CB_NO_GELU_SAVE_GRAD=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_pre_$i.log 2>&1; echo "pre-activation saved: $(grep -E 'timed region' $O/bench_pre_$i.log | sed 's/.*done: //')"
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_grad_$i.log 2>&1; echo "derivative saved:     $(grep -E 'timed region' $O/bench_grad_$i.log | sed 's/.*done: //')"
done
timeout 900 python -m pytest tests/test_kernels_gemm.py tests/test_kernels_gemm8.py tests/test_bench_step.py tests/test_model_small.py tests/test_parity_record.py -m gpu -x -q 2>&1 | tail -3
