set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r3_prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r3_pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/r3_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r3_pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/r3_pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r3_pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/r3_pmc_mfma.log 2>&1
ls $R/gpurun_out/r3_pmc_mfma/*/ | head
