mkdir -p gpurun_out/pmc3; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in "0 4 8" "1 4 8" "2 4 8" "2 12 4" "2 2 16"; do
  set -- $cfg
  CB_GEMM_XCD_MODE=$1 CB_GEMM_ST_W=$2 CB_GEMM_ST_H=$3 timeout 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc3 -o m$1_$2_$3 -- python $R/tools/one_gemm.py ${SHAPE:-5248 768 3072} 2 > /dev/null 2>&1
done
