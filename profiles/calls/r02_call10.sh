#!/bin/bash
# the four-graph N > 1 plan (ResNet backward cut after res5): dry run of the 8-rank plan on one GPU, then two gloo ranks sharing the GPU
cd /root/repo; O=gpurun_out/r02x; mkdir -p $O
CB_BENCH_DRY_DP=8 timeout 100 python bench.py --no-cpu-baseline --no-roofline > $O/dry4.json 2> $O/dry4.log; grep -E "replay plan|timed region|Error|error" $O/dry4.log | tail -4
CB_BENCH_DRY_DP=8 CB_BENCH_CNN_SPLIT=0 timeout 100 python bench.py --no-cpu-baseline --no-roofline > $O/dry3.json 2> $O/dry3.log; grep -E "replay plan|timed region" $O/dry3.log | tail -2
export CB_BENCH_SHARE_GPU=1 CB_BENCH_BACKEND=gloo
timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/dp2_4graph.log 2>&1; grep -E "DP self-check|replay plan|timed region|Error" $O/dp2_4graph.log | tail -5
