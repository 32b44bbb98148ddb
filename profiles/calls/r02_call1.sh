# Round 2, GPU call 1: parity suite, first bench lines of the metric configuration, the tile / order sweep.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
(time timeout 300 python bench.py --steps 10 --warmup 3) > $O/bench_train.log 2>&1; tail -2 $O/bench_train.log | cut -c1-600
(time timeout 240 python bench.py --mode tgif --steps 10 --warmup 3 --no-cpu-baseline) > $O/bench_tgif.log 2>&1; tail -2 $O/bench_tgif.log | cut -c1-300
(time timeout 240 python bench.py --mode infer16 --steps 10 --warmup 3 --no-cpu-baseline) > $O/bench_infer16.log 2>&1; tail -2 $O/bench_infer16.log | cut -c1-300
(time timeout 600 python tools/tune_gemm.py --modes train,tgif,infer16 --out $O/gemm_tuning.json) > $O/tune.log 2>&1; tail -3 $O/tune.log
(time timeout 600 python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
