# Round 5, call Y: with the slab K split a split costs less than the atomics the grouped launch model prices: scale its atomics term
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O; cd $R
for i in 1 2; do
  for sc in 1.0 0.5 0.25 0.0; do
    (CB_GROUP_ATOM_SCALE=$sc timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_${sc}_$i.log 2>&1; echo "atomics term x $sc: $(grep -E 'timed region' $O/bench_${sc}_$i.log | sed 's/.*done: //')"
  done
done
