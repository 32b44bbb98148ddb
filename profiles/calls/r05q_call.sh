# Round 5, call Q: per-pass prefetch of the epilogue operands (residual / mask / GELU pre-activation) in the 8-wave epilogue vs HEAD
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05q; mkdir -p $O; cd $R
L=clipbert_amd/lib; cp $L/libclipbert_hip.so $L/new.so.keep
for i in 1 2 3; do
  cp $L/libclipbert_hip_pre.so $L/libclipbert_hip.so
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_pre_$i.log 2>&1; echo "HEAD (no prefetch): $(grep -E 'timed region' $O/bench_pre_$i.log | sed 's/.*done: //')"
  cp $L/new.so.keep $L/libclipbert_hip.so
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_new_$i.log 2>&1; echo "operand prefetch:   $(grep -E 'timed region' $O/bench_new_$i.log | sed 's/.*done: //')"
done
cp $L/new.so.keep $L/libclipbert_hip.so
timeout 900 python -m pytest tests/test_kernels_gemm8.py tests/test_kernels_gemm.py tests/test_gpu_full.py tests/test_bench_step.py -m gpu -x -q 2>&1 | tail -3
