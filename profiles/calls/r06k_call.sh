# Round 6, call K: the encoder's four kinds of 12-layer weight gradients in ONE grouped launch (cb_gemm_group's row-sum / strided-batch class)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06k; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests/test_gemm_group.py tests/test_bench_step.py tests/test_norm_fold.py tests/test_model_small.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -3
b() { name=$1; shift; (env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$name.log 2>&1; echo "$name: $(grep -E 'timed region' $O/bench_$name.log | sed 's/.*done: //')"; }
for i in 1 2 3; do
  b four_$i CB_NO_GROUP_ENC_WGRAD=1
  b one_$i CB_X=0
done
(timeout 600 python tools/replay_determinism.py) 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
mkdir -p $O/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; grep -E "wgrad|One steady" $O/train_step.md
rm -f $O/trace/bench_kernel_trace.csv
