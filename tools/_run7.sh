python -m pytest tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -n 3
for i in 1 2; do
python tools/bench_encode.py all-MiniLM-L6-v2 4096 20
TSIM_LN_ROWS256=0 python tools/bench_encode.py all-MiniLM-L6-v2 4096 20
done
python tools/bench_encode.py all-MiniLM-L6-v2 1024 20
TSIM_LN_ROWS256=0 python tools/bench_encode.py all-MiniLM-L6-v2 1024 20
