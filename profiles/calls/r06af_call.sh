# Round 6, call AF (experiment): the 64x64 bf16 kernels compiled for 4 waves per SIMD (<= 128 registers: four workgroups per CU as the LDS allows) vs the compiler's choice (2-3)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06af; mkdir -p $O; cd $R
(CB_LIB_VARIANT=occ4 timeout 900 python -m pytest tests/test_bench_step.py tests/test_kernels_gemm.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -2
b() { name=$1; shift; (env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$name.log 2>&1; echo "$name: $(grep -E 'timed region' $O/bench_$name.log | sed 's/.*done: //')"; }
for i in 1 2 3; do
  b occ1_$i CB_X=0
  b occ4_$i CB_LIB_VARIANT=occ4
done
