# Round 5, call W: upper bound of what the split-K fp32 atomics of the grouped weight gradients cost in the step: a diagnostic build whose
# atomic epilogue uses plain stores (WRONG sums; timing only) against the product library, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05w; mkdir -p $O; cd $R
L=clipbert_amd/lib; cp $L/libclipbert_hip.so $L/new.so.keep
for i in 1 2; do
  cp $L/new.so.keep $L/libclipbert_hip.so
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_prod_$i.log 2>&1; echo "product:            $(grep -E 'timed region' $O/bench_prod_$i.log | sed 's/.*done: //')"
  cp $L/libclipbert_hip_noatomic.so $L/libclipbert_hip.so
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_noatomic_$i.log 2>&1; echo "stores for atomics: $(grep -E 'timed region' $O/bench_noatomic_$i.log | sed 's/.*done: //')"
done
cp $L/new.so.keep $L/libclipbert_hip.so
