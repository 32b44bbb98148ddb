# Round-end measurement set (run through gpurun from the repo root): kernel trace + stats of the bench command, the two
# HBM-traffic PMC passes and the MFMA-busy pass over one logged eager step of the same workload.  Outputs in gpurun_out/final/.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O/trace $O/pmc_f $O/pmc_w $O/pmc_m
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/trace/bench.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python $R/tools/gemm_breakdown.py > $O/pmc_f/log.txt 2>&1
cp $R/gpurun_out/gemm_calls.json $O/gemm_calls.json
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python $R/tools/gemm_breakdown.py > $O/pmc_w/log.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_m -o m -- python $R/tools/gemm_breakdown.py > $O/pmc_m/log.txt 2>&1
ls -la $O/trace $O/pmc_f $O/pmc_w $O/pmc_m | head -40
tail -2 $O/trace/bench.log | cut -c1-300
