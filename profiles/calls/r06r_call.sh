# Round 6, call R: kernel trace of the infer16 row (BASELINE configs[4]): where did 14.8 -> 18.0 ms come from
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06r; mkdir -p $O/trace; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o infer -- python $R/bench.py --mode infer16 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python - <<'P'
import csv, collections, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/tools")
import trace_summary as T
p = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06r/trace/infer_kernel_trace.csv"
rows = list(csv.DictReader(open(p)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    name = T.gemm_name(n) or n.split("(")[0][:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[name][0] += 1; agg[name][1] += d
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{v[1]/tot*100:5.1f}%  n={v[0]:5d}  avg {v[1]/v[0]:8.1f} us  {k}")
P
rm -f $O/trace/infer_kernel_trace.csv
