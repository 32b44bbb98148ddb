# Round-4 diagnostic: the 2-rank gloo dry run under torch.distributed.run that timed out once in r04h, run directly with full logs (twice), and the bare launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
cd $R
export CB_BENCH_SHARE_GPU=1 CB_BENCH_BACKEND=gloo
for i in 1 2; do
(time timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 2 --steps 2 --warmup 1 --mode train) > $O/torchrun_$i.log 2>&1; echo "torchrun $i rc=$?"; grep -E "timed region|supervisor|real|Traceback|Error|muted" $O/torchrun_$i.log | cut -c1-200
done
(time timeout 420 python bench.py --gpus 2 --steps 2 --warmup 1) > $O/bare.log 2>&1; echo "bare rc=$?"; grep -E "timed region|supervisor|real|Traceback|Error|muted" $O/bare.log | cut -c1-200
