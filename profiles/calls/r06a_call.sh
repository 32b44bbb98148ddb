# Round 6, call A: (1) GPU tests of what changed (K-split tickets in the caller's scratch, absolute pooled bound, timed step), (2) the
# concurrency A/B VERDICT r5 item 5 asks for, alternating: CB_OVERLAP_WGRAD = 0 (one branch) / 1 (the encoder's four batched weight-gradient
# launches on a second graph branch beside the ResNet backward) / 2 (each ResNet stage's grouped weight gradients + the grid encoder's beside
# the next stage's data gradients) / 3 (both), (3) a base trace for the round.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_abi.py tests/test_gemm_group.py tests/test_bench_step.py "tests/test_gpu_full.py::test_forward_matches_reference_golden" -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do
  for v in 0 1 2 3; do
    (CB_OVERLAP_WGRAD=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_ov${v}_$i.log 2>&1; echo "overlap $v: $(grep -E 'timed region' $O/bench_ov${v}_$i.log | sed 's/.*done: //')"
  done
done
(CB_OVERLAP_WGRAD=3 timeout 300 python tools/replay_determinism.py --replays 3 --eager 2) > $O/determinism_ov3.txt 2>&1; tail -5 $O/determinism_ov3.txt | cut -c1-300
cd /tmp; export TMPDIR=/tmp
mkdir -p $O/trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; head -30 $O/train_step.md
python $R/tools/step_timeline.py $O/trace/bench_kernel_trace.csv > $O/step_phases.txt 2>&1; cat $O/step_phases.txt
rm -f $O/trace/bench_kernel_trace.csv
tail -1 $O/trace/bench.log | cut -c1-600
