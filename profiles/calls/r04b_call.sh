# Round-4 second GPU call: the new parity bounds (bf16 oracle yardstick, decided-margin goldens) on the GPU, bench line with the per-roof
# families, the self-launching --gpus 2 path (two ranks on GPU 0, gloo), PMC traffic passes with the grouped launches.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O/pmc_f $O/pmc_w
cd $R
rm -f $R/gpurun_out/r04_bf16_parity.json
(time timeout 900 python -m pytest tests/test_gpu_full.py tests/test_parity_record.py -q -m gpu) > $O/pytest_parity.log 2>&1; tail -15 $O/pytest_parity.log | cut -c1-400
cp $R/gpurun_out/r04_bf16_parity.json $O/ 2>/dev/null
(time timeout 400 python bench.py) > $O/bench.json 2> $O/bench.err; grep -E "timed region" $O/bench.err; cut -c1-1500 $O/bench.json
(time timeout 900 python -m pytest tests/test_zz_bench_cli.py -q -m gpu -x) > $O/pytest_cli.log 2>&1; tail -5 $O/pytest_cli.log | cut -c1-600
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python $R/tools/gemm_breakdown.py > $O/pmc_f/log.txt 2>&1; tail -2 $O/pmc_f/log.txt | cut -c1-300
cp $R/gpurun_out/gemm_calls.json $O/gemm_calls.json
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python $R/tools/gemm_breakdown.py > $O/pmc_w/log.txt 2>&1
python $R/tools/pmc_traffic.py $O/gemm_calls.json $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; tail -12 $O/pmc_traffic.log | cut -c1-400
rm -f $O/pmc_f/*kernel_trace.csv $O/pmc_w/*kernel_trace.csv; gzip -f $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv
