#!/bin/bash
# build the product library + emulator here (the .so / .o files travel with the snapshot), then run a call script on an MI355X box:
#   tools/gpu.sh profiles/calls/r06b_call.sh [timeout_s]
set -e
cd "$(dirname "$0")/.."
python -m clipbert_amd.build > /tmp/gpu_build.log 2>&1 || { tail -30 /tmp/gpu_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "${2:-1800}" -- "bash $1"
