# In-step durations of every cb_gemm of the metric step (rocprofv3 kernel trace of one eager step, joined with the call log), with
# and without the 8-wave entries of the tuned table, plus the bench step time of both.  Run through gpurun from the repo root.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/instep; mkdir -p $O/a $O/b
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/a -o t -- python $R/tools/gemm_breakdown.py > $O/a/log.txt 2>&1
cp $R/gpurun_out/gemm_calls.json $O/a/gemm_calls.json
python $R/tools/join_gemm_trace.py $O/a/gemm_calls.json $(ls $O/a/*/*kernel_trace.csv $O/a/*kernel_trace.csv 2>/dev/null | head -1) 60 > $O/with8w.txt 2>&1
CB_GEMM_NO8W=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/b -o t -- python $R/tools/gemm_breakdown.py > $O/b/log.txt 2>&1
cp $R/gpurun_out/gemm_calls.json $O/b/gemm_calls.json
python $R/tools/join_gemm_trace.py $O/b/gemm_calls.json $(ls $O/b/*/*kernel_trace.csv $O/b/*kernel_trace.csv 2>/dev/null | head -1) 60 > $O/no8w.txt 2>&1
head -3 $O/with8w.txt; head -3 $O/no8w.txt
cd $R
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
CB_GEMM_NO8W=1 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
rm -rf $O/a/*/*.db $O/b/*/*.db 2>/dev/null; find $O -name "*kernel_trace.csv" -size +20M -delete
