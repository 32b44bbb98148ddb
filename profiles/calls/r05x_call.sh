# Round 5, call X: slab K split of the grouped weight gradients (ordered in-kernel reduce by the last K part to arrive) vs the fp32 atomics
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05x; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gemm_group.py tests/test_bench_step.py tests/test_model_small.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  (CB_GROUP_SLAB=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_atomics_$i.log 2>&1; echo "atomics: $(grep -E 'timed region' $O/bench_atomics_$i.log | sed 's/.*done: //')"
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_slab_$i.log 2>&1; echo "slab:    $(grep -E 'timed region' $O/bench_slab_$i.log | sed 's/.*done: //')"
done
(timeout 300 python tools/replay_determinism.py --replays 4 --eager 2) > $O/determinism.txt 2>&1; tail -8 $O/determinism.txt | cut -c1-300
