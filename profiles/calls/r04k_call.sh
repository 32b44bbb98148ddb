# Round-4: WHY do the ResNet forward kernels take what they take -- SQ wave-time decomposition and L2 counters per kernel (separate PMC passes, --kernel-trace only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O/sq $O/tcc
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq -o sq -- python $R/tools/resnet_fwd_probe.py > $O/sq/log.txt 2>&1; tail -1 $O/sq/log.txt
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/tcc -o tcc -- python $R/tools/resnet_fwd_probe.py > $O/tcc/log.txt 2>&1; tail -1 $O/tcc/log.txt
python $R/tools/pmc_kernel_table.py $O/resnet_fwd_pmc.md $O/sq/sq_counter_collection.csv $O/tcc/tcc_counter_collection.csv | cut -c1-330 | head -45
rm -f $O/sq/*kernel_trace.csv $O/tcc/*kernel_trace.csv; gzip -f $O/sq/sq_counter_collection.csv $O/tcc/tcc_counter_collection.csv
