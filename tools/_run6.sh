for i in 1 2; do
TSIM_LN_TAIL_BM=32 python tools/bench_encode.py all-MiniLM-L6-v2 4096 20
TSIM_LN_TAIL_BM=64 python tools/bench_encode.py all-MiniLM-L6-v2 4096 20
done
python -m pytest tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -n 3
