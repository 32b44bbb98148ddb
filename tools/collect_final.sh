# Copy the round-end measurement set (tools/final_profiles.sh + tools/final_verify.sh -> gpurun_out/final/) into profiles/ under a round tag:
#   bash tools/collect_final.sh r05z
T=${1:?tag}; F=gpurun_out/final; P=profiles
cp $F/train_step.md $P/${T}_train_step.md
cp $F/step_phases.txt $P/${T}_step_phases.txt
cp $F/hbm_mfma_per_kernel.md $P/${T}_hbm_mfma_per_kernel.md
cp $F/pmc_traffic.json $P/${T}_pmc_traffic.json
cp $F/trace/bench_kernel_stats.csv $P/${T}_kernel_stats.csv
cp $F/trace/bench.log $P/${T}_bench_under_rocprof.log
for f in pytest_gpu.log smoke.log bf16_parity.json; do [ -f $F/$f ] && cp $F/$f $P/${T}_$f; done
[ -f $F/bench.json ] && cp $F/bench.json $P/${T}_bench.json
for m in tgif infer16 448c4; do [ -f $F/bench_$m.log ] && grep '^{' $F/bench_$m.log > $P/${T}_bench_$(echo $m | sed 's/448c4/448px_c4/').json; done
[ -f $F/bench_loopback_allreduce.log ] && { grep -hE "replay plan|timed region" $F/bench_loopback_allreduce.log $F/bench_loopback_owner_only.log | cut -c1-220 > $P/${T}_bench_loopback_dp_plans.log; }
[ -f $F/dp2_train.log ] && { for f in dp2_train_self_launched dp2_train dp2_infer dp2_train_owner_only; do echo "== $f"; grep -E "DP self-check|replay plan|timed region|supervisor|rows_gathered" $F/$f.log | cut -c1-220; done > $P/${T}_dp2_dryrun_shared_gpu_gloo.log; }
ls -la $P/${T}_*
