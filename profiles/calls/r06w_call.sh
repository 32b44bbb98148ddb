# Round 6, call W (experiment): every kernel instantiates only the specialised epilogue bodies its FORM meets (variant prune, -DCB_FE_PRUNE) vs all sixteen
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06w; mkdir -p $O; cd $R
(CB_LIB_VARIANT=prune timeout 900 python -m pytest tests/test_bench_step.py tests/test_kernels_gemm8.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -2
b() { name=$1; shift; (env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$name.log 2>&1; echo "$name: $(grep -E 'timed region' $O/bench_$name.log | sed 's/.*done: //')"; }
for i in 1 2 3; do
  b all_$i CB_X=0
  b prune_$i CB_LIB_VARIANT=prune
done
