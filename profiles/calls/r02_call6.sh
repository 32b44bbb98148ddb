#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02x/trace; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench.log 2>&1
cd $R; python tools/trace_summary.py $(ls $O/*/*kernel_trace.csv $O/*kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/r02x/train_step.md 2>&1; head -30 gpurun_out/r02x/train_step.md
