# Round 5, call I: fused stem (conv + FrozenBN + ReLU + max-pool) -- GPU tests, bench A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
(timeout 600 python -m pytest tests/test_res2_block.py tests/test_bench_step.py tests/test_gpu_full.py -x -q -m gpu) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
  (CB_NO_STEM_FUSE=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_unfused$i.log 2>&1; echo "two launches: $(grep -E 'timed region' $O/bench_unfused$i.log | sed 's/.*done: //')"
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_fused$i.log 2>&1; echo "fused stem:   $(grep -E 'timed region' $O/bench_fused$i.log | sed 's/.*done: //')"
done
