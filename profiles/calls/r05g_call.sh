# Round 5, call G: cold-cache sweep of every cb_gemm shape of the metric step under tile x order x split x schedule, on the kernels with
# the rolled epilogues + write-through stores (the round-3 table was measured with epilogues that cost 1.1 us per chunk)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R
(time timeout 1500 python tools/tune_gemm.py --cold --modes train --out $O/sweep_train_cold.json) > $O/sweep.log 2>&1; tail -4 $O/sweep.log | cut -c1-200
