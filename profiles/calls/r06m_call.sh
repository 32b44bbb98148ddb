# Round 6, call M: embedding backward with sum / scatter block roles, the slot sum with independent loads; the encoder's whole weight-gradient set
# (one grouped launch / four launches / the library's four bmm calls) on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests/test_kernels_misc.py tests/test_model_small.py tests/test_bench_step.py tests/test_norm_fold.py tests/test_gpu_full.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -3
for i in 1 2 3; do (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$i.log 2>&1; echo "bench: $(grep -E 'timed region' $O/bench_$i.log | sed 's/.*done: //')"; done
(timeout 600 python tools/replay_determinism.py) 2>&1 | tail -1
(timeout 600 python tools/gemm_yardstick.py --wgrad-set --out $O) 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
mkdir -p $O/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; grep -E "embed|sum_two|One steady" $O/train_step.md
rm -f $O/trace/bench_kernel_trace.csv
