cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r3_pmc_sq1 -- python3 $R/tools/bench_encode.py all-MiniLM-L6-v2 4096 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r3_pmc_sq2 -- python3 $R/tools/bench_encode.py all-MiniLM-L6-v2 4096 3 > /dev/null 2>&1
ls $R/gpurun_out/r3_pmc_sq1/*/ $R/gpurun_out/r3_pmc_sq2/*/
