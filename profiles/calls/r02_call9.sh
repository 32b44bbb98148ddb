#!/bin/bash
# does the 20-step timed region see steady-state clocks?  same box: default, long runs, long warm-up
cd /root/repo
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "steps 20  warmup 5   : $(b) $(b)"
echo "steps 200 warmup 5   : $(b --steps 200)"
echo "steps 1000 warmup 5  : $(b --steps 1000)"
echo "steps 20  warmup 300 : $(b --warmup 300) $(b --warmup 300)"
echo "steps 20  warmup 5   : $(b)"
