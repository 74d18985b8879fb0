python -m pytest tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -n 5
for i in 1 2; do
TSIM_LN_ROWS256=0 python tools/bench_encode.py all-MiniLM-L6-v2 4096 20
TSIM_LN_ROWS256=1 python tools/bench_encode.py all-MiniLM-L6-v2 4096 20
done
TSIM_LN_ROWS256=1 python tools/bench_encode.py all-MiniLM-L6-v2 16384 10
TSIM_LN_ROWS256=0 python tools/bench_encode.py all-MiniLM-L6-v2 16384 10
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_prof_lr -- python3 $GRAFT_REPO_ROOT/tools/bench_encode.py all-MiniLM-L6-v2 4096 10 > /dev/null 2>&1
