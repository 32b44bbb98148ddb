# Round 6, call O (experiment): BertOutput.dense forward (64x64 tile) through the grouped kernel instead of gemm_kernel: probe says 24.9 vs 28.0 us
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06o; mkdir -p $O; cd $R
b() { name=$1; shift; (env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$name.log 2>&1; echo "$name: $(grep -E 'timed region' $O/bench_$name.log | sed 's/.*done: //')"; }
for i in 1 2 3; do
  b base_$i CB_X=0
  b grp_$i CB_EXP_SLAB_ANY=1 CB_EXP_FWD_GROUP="2624,768,3072"
  b grp2_$i CB_EXP_SLAB_ANY=1 CB_EXP_FWD_GROUP="2624,768,3072;2624,768,768"
done
