# Round 6, call AG: uint8 frames straight into cb_stem_pool (cb_stem_pool_u8): tests, A/B against cb_stem_pack + cb_stem_pool, kernel times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ag; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_res2_block.py tests/test_bench_step.py tests/test_gpu_full.py tests/test_model_small.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -2
b() { name=$1; shift; (env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$name.log 2>&1; echo "$name: $(grep -E 'timed region' $O/bench_$name.log | sed 's/.*done: //')"; }
for i in 1 2 3; do
  b pack_$i CB_NO_STEM_U8=1
  b u8_$i CB_X=0
done
cd /tmp; export TMPDIR=/tmp
mkdir -p $O/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; grep -E "stem|One steady" $O/train_step.md
rm -f $O/trace/bench_kernel_trace.csv
