# Round 6, call J: the few-rows structure (cb_gemm tile 9, csrc/gemm_skinny.hip) on the heads' products: tests, A/B in the step, per-shape times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests/test_gemm_skinny.py tests/test_kernels_gemm.py tests/test_model_small.py tests/test_bench_step.py tests/test_gpu_full.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -3
b() { name=$1; shift; (env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$name.log 2>&1; echo "$name: $(grep -E 'timed region' $O/bench_$name.log | sed 's/.*done: //')"; }
for i in 1 2 3; do
  b tiled_$i CB_GEMM_NO_SKINNY=1
  b skinny_$i CB_X=0
done
cd /tmp; export TMPDIR=/tmp
mkdir -p $O/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; grep -E "skinny|few rows|One steady" $O/train_step.md
rm -f $O/trace/bench_kernel_trace.csv
