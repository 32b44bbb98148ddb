# Round 6, call I: LayerNorm forward full-row form (tests + bench), and the encoder's four batched weight-gradient launches on four
# concurrent graph branches (CB_OVERLAP_WGRAD=8), with the 4-wave tile of the table and with the 8-wave 256x256 tile pinned
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_kernels_misc.py tests/test_model_small.py tests/test_bench_step.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -2
W8="2,2,3072,768,2624,12,1,1=8w256x256/xcd/s1/m0;2,2,768,3072,2624,12,1,1=8w256x256/xcd/s1/m0;2,2,2304,768,2624,12,1,1=8w256x256/xcd/s1/m0"
b() { name=$1; shift; (env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$name.log 2>&1; echo "$name: $(grep -E 'timed region' $O/bench_$name.log | sed 's/.*done: //')"; }
for i in 1 2 3; do
  b base_$i CB_X=0
  b fan_$i CB_OVERLAP_WGRAD=8
  b w8_$i CB_LAUNCH_OVERRIDE="$W8"
  b fanw8_$i CB_OVERLAP_WGRAD=8 CB_LAUNCH_OVERRIDE="$W8"
done
