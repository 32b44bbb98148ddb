# Round 5, call A: same-box baseline, in-kernel stamps of the captured step, write-through / raster A/B, vendor-library yardstick
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
(time timeout 600 python bench.py --no-cpu-baseline) > $O/bench_base.log 2>&1; grep '^{' $O/bench_base.log > $O/bench_base.json; grep -E "timed region" $O/bench_base.log
(time timeout 600 python tools/stamps_run.py --out $O/stamps) > $O/stamps.log 2>&1; tail -3 $O/stamps.log | cut -c1-300
for i in 1 2; do
  (CB_GEMM_WT=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_wt$i.log 2>&1; grep -E "timed region" $O/bench_wt$i.log
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_plain$i.log 2>&1; grep -E "timed region" $O/bench_plain$i.log
  (CB_GEMM_RASTER_W=4 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_rw4_$i.log 2>&1; grep -E "timed region" $O/bench_rw4_$i.log
done
(CB_GEMM_RASTER_W=3 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_rw3.log 2>&1; grep -E "timed region" $O/bench_rw3.log
(CB_GEMM_RASTER_W=6 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_rw6.log 2>&1; grep -E "timed region" $O/bench_rw6.log
(time timeout 900 python tools/gemm_yardstick.py --out $O/yardstick) > $O/yardstick.log 2>&1; tail -40 $O/yardstick.log | cut -c1-250
