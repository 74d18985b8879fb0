cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $R/gpurun_out/r3_pmc_sq3 -- python3 $R/tools/bench_encode.py all-MiniLM-L6-v2 4096 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL --kernel-trace --output-format csv -d $R/gpurun_out/r3_pmc_sq4 -- python3 $R/tools/bench_encode.py all-MiniLM-L6-v2 4096 3 > /dev/null 2>&1
