#!/bin/bash
# The round's judged profiles, on the GPU box: bash tools/profile_round.sh <tag>   (then tools/profile_summary.py here)
#   pass 1: rocprofv3 --kernel-trace --stats over the default bench (no CPU baseline)
#   passes 2-4: --pmc FETCH_SIZE / WRITE_SIZE / MFMA busy, each alone with --kernel-trace (MI355X_MICROARCH.md HBM section)
# The program follows `--` directly (no env / bash -c hop: the profiler has initialised the GPU by then).
set -e -o pipefail
tag=${1:-r03f}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/${tag}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/${tag}_mfma.log 2>&1
cd $R && python3 bench.py > gpurun_out/${tag}_bench.log 2>&1
tail -n 1 gpurun_out/${tag}_bench.log | cut -c1-600
