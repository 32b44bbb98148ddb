# Round 2, GPU call 2: tuned table in place -- parity suite, bench lines, kernel trace, PMC traffic passes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O/trace $O/pmc_f $O/pmc_w
cd $R
(time timeout 300 python bench.py) > $O/bench_train.log 2>&1; grep -E "timed region|^real" $O/bench_train.log
(time timeout 240 python bench.py --mode tgif --no-cpu-baseline) > $O/bench_tgif.log 2>&1; grep -E "timed region|^real" $O/bench_tgif.log
(time timeout 240 python bench.py --mode infer16 --no-cpu-baseline) > $O/bench_infer16.log 2>&1; grep -E "timed region|^real" $O/bench_infer16.log
(time timeout 240 python bench.py --size 448 --txt-len 20 --n-clips 4 --no-cpu-baseline --steps 10) > $O/bench_448c4.log 2>&1; grep -E "timed region|^real" $O/bench_448c4.log
(time timeout 900 python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/trace/bench.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python $R/tools/gemm_breakdown.py > $O/pmc_f/log.txt 2>&1
cp $R/gpurun_out/gemm_calls.json $O/gemm_calls.json
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python $R/tools/gemm_breakdown.py > $O/pmc_w/log.txt 2>&1
ls $O/trace $O/pmc_f $O/pmc_w | head -30
tail -1 $O/trace/bench.log | cut -c1-200
# keep the merged output small: the raw traces of the counter passes are large
find $O -name "*_kernel_trace.csv" -size +20M -delete
