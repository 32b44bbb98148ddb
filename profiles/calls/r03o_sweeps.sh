# More cold sweeps for the launch-cost model: workloads whose shapes are in NO table (other batch sizes, text lengths, resolutions,
# clip counts), plus two that are held out of the fit entirely.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O
cd $R
(time timeout 1500 python tools/tune_gemm.py --cold --modes "train:--videos 6 --txt-len 40;train:--videos 24;train:--videos 8 --size 192;train:--videos 4 --n-clips 4 --frames 1;tgif:--videos 6;infer16:--repeat 32" --out $O/sweep_fit.json) > $O/sweep_fit.log 2>&1; tail -1 $O/sweep_fit.log
(time timeout 600 python tools/tune_gemm.py --cold --modes "train:--videos 12 --txt-len 20 --size 256;tgif:--videos 10" --out $O/sweep_holdout.json) > $O/sweep_holdout.log 2>&1; tail -1 $O/sweep_holdout.log
grep -c "^\[tune\]" $O/sweep_fit.log $O/sweep_holdout.log
