#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r02x
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline 2> gpurun_out/r02x/b$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'])"; done
timeout 300 python bench.py --mode tgif --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tgif', d['ms_per_step'], d['value'])"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
