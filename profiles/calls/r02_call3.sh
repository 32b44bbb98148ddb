# Round 2, GPU call 3: LayerNorm-backward grid sweep, weight-gradient split sweep, bench.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02c; mkdir -p $O
cd $R
for c in 32 64 128 256; do CB_LN_BWD_BLOCKS=$c timeout 120 python tools/ln_probe.py 2>&1 | grep rows; done > $O/ln_probe.log 2>&1; cat $O/ln_probe.log
(time timeout 900 python tools/tune_gemm.py --modes train,tgif,infer16 --out $O/gemm_tuning.json) > $O/tune.log 2>&1; tail -2 $O/tune.log
(time timeout 300 python bench.py --no-cpu-baseline) > $O/bench_train.log 2>&1; grep -E "timed region|^real" $O/bench_train.log
