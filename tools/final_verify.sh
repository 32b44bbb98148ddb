# Round-end verification set (run through gpurun from the repo root); outputs in gpurun_out/final/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O/trace
cd $R
# bf16 parity: the bounds are committed constants derived from the bf16 ORACLE (tests/golden/bf16_yardstick.json, tests/parity_bounds.py);
# nothing is re-recorded or regenerated here.  The suite writes what it measured to gpurun_out/r04_bf16_parity.json (a record, not a bound).
rm -f $R/gpurun_out/r04_bf16_parity.json
(time timeout 1200 python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; cp $R/gpurun_out/r04_bf16_parity.json $O/bf16_parity.json 2>/dev/null
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; grep smoke $O/smoke.log
(time timeout 600 python bench.py) > $O/bench.log 2>&1; grep -E "timed region|real" $O/bench.log; grep '^{' $O/bench.log > $O/bench.json
(time timeout 300 python bench.py --mode tgif --no-cpu-baseline) > $O/bench_tgif.log 2>&1; grep -E "timed region" $O/bench_tgif.log
(time timeout 300 python bench.py --mode infer16 --no-cpu-baseline) > $O/bench_infer16.log 2>&1; grep -E "timed region" $O/bench_infer16.log
(time timeout 300 python bench.py --size 448 --txt-len 20 --n-clips 4 --no-cpu-baseline) > $O/bench_448c4.log 2>&1; grep -E "timed region" $O/bench_448c4.log
# the data-parallel plans on one GPU with a world-size-1 communicator of the library (real cb_* collectives, captured into the step's graph)
(time timeout 300 env CB_BENCH_LOOPBACK=1 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_loopback_allreduce.log 2>&1; grep -E "replay plan|timed region" $O/bench_loopback_allreduce.log | cut -c1-200
(time timeout 300 env CB_BENCH_LOOPBACK=1 CB_BENCH_SHARD=1 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_loopback_owner_only.log 2>&1; grep -E "replay plan|timed region" $O/bench_loopback_owner_only.log | cut -c1-200
# N > 1 control flow on this 1-GPU box: two ranks share GPU 0, gloo collectives (captures, split replay plan, bucketed exchange,
# cross-rank parameter check, sharded inference + row gather).  A control-flow check, not a measurement.
export CB_BENCH_SHARE_GPU=1 CB_BENCH_BACKEND=gloo
(time timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1) > $O/dp2_train_self_launched.log 2>&1; grep -E "DP self-check|replay plan|timed region|supervisor" $O/dp2_train_self_launched.log
(time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1) > $O/dp2_train.log 2>&1; grep -E "DP self-check|replay plan|timed region" $O/dp2_train.log
(time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --mode infer16 --steps 3 --warmup 1) > $O/dp2_infer.log 2>&1; grep -E "timed region|rows_gathered" $O/dp2_infer.log | cut -c1-200
(time timeout 600 env CB_BENCH_SHARD=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --steps 3 --warmup 1) > $O/dp2_train_owner_only.log 2>&1; grep -E "DP self-check|replay plan|timed region" $O/dp2_train_owner_only.log
unset CB_BENCH_SHARE_GPU CB_BENCH_BACKEND
