# Round-end measurement set (run through gpurun from the repo root): kernel trace + stats of the bench command, the step's table and
# timeline, the two HBM-traffic PMC passes (separate runs, --kernel-trace only) and the MFMA-busy pass over one logged eager step of the
# same workload.  Outputs in gpurun_out/final/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O/trace $O/pmc_f $O/pmc_w $O/pmc_m
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; head -12 $O/train_step.md
python $R/tools/step_timeline.py $O/trace/bench_kernel_trace.csv > $O/step_phases.txt 2>&1; cat $O/step_phases.txt
gzip -f $O/trace/bench_kernel_trace.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python $R/tools/gemm_breakdown.py > $O/pmc_f/log.txt 2>&1
cp $R/gpurun_out/gemm_calls.json $O/gemm_calls.json
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python $R/tools/gemm_breakdown.py > $O/pmc_w/log.txt 2>&1
python $R/tools/pmc_traffic.py $O/gemm_calls.json $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; tail -14 $O/pmc_traffic.log | cut -c1-300
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_m -o m -- python $R/tools/gemm_breakdown.py > $O/pmc_m/log.txt 2>&1
gunzip -kf $O/trace/bench_kernel_trace.csv.gz
python $R/tools/hbm_table.py $O/trace/bench_kernel_trace.csv $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv $O/pmc_m/m_counter_collection.csv $O/hbm_mfma_per_kernel.md > /dev/null 2>&1; head -14 $O/hbm_mfma_per_kernel.md | cut -c1-200
rm -f $O/trace/bench_kernel_trace.csv $O/pmc_f/*kernel_trace.csv $O/pmc_w/*kernel_trace.csv $O/pmc_m/*kernel_trace.csv; gzip -f $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv $O/pmc_m/m_counter_collection.csv
tail -2 $O/trace/bench.log | cut -c1-300
