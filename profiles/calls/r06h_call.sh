# Round 6, call H: attention backward with the key mask staged before the barrier (tools/attn_probe.py), bench x2, and the weight-gradient
# launches of one replayed step one by one (tools/group_launches.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
timeout 300 python tools/attn_probe.py 2>&1 | grep "L=41"
(timeout 600 python -m pytest tests/test_kernels_misc.py tests/test_model_small.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -2
for i in 1 2; do (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$i.log 2>&1; echo "bench: $(grep -E 'timed region' $O/bench_$i.log | sed 's/.*done: //')"; done
cd /tmp; export TMPDIR=/tmp
mkdir -p $O/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/group_launches.py $O/trace/bench_kernel_trace.csv | tee $O/group_launches.txt
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; grep -E "attn|layernorm|zero|sq_sum|sum_two" $O/train_step.md
rm -f $O/trace/bench_kernel_trace.csv
