# Round-3 mid-round GPU call: full GPU suite on the tree with cb_zero / owner-only update / cost model, the metric bench, a cold sweep of
# a batch size that is in NO sweep (10 videos, 24 text tokens) to measure the cost model on unseen shapes, kernel names of the step.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O/trace
cd $R
(time timeout 1200 python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
(time timeout 600 python bench.py) > $O/bench.json 2> $O/bench.err; grep -E "timed region" $O/bench.err
(time timeout 700 python tools/tune_gemm.py --cold --modes "train:--videos 10 --txt-len 24" --out $O/holdout_sweep.json) > $O/holdout.log 2>&1; tail -1 $O/holdout.log
python tools/fit_gemm_model.py check $O/holdout_sweep.json > $O/holdout_check.txt 2>&1; cat $O/holdout_check.txt
export CB_BENCH_SHARE_GPU=1 CB_BENCH_BACKEND=gloo CB_BENCH_SHARD=1
(time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 2 --steps 3 --warmup 1) > $O/dp2_shard.log 2>&1; grep -E "DP self-check|replay plan|timed region" $O/dp2_shard.log
unset CB_BENCH_SHARE_GPU CB_BENCH_BACKEND CB_BENCH_SHARD
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; grep -c "at::native" $O/train_step.md
rm -f $O/trace/bench_kernel_trace.csv.keep; gzip -f $O/trace/bench_kernel_trace.csv
