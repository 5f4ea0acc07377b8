cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -k "deterministic or independent" 2>&1 | tail -15
python tools/bench_b1.py 2>/dev/null
VS_GEMM_NO_KSPLIT=1 python tools/bench_b1.py 2>/dev/null
VS_DETERMINISTIC=1 python tools/bench_b1.py 2>/dev/null
python tools/bench_b1.py --scenes 2 2>/dev/null
VS_GEMM_NO_KSPLIT=1 python tools/bench_b1.py --scenes 2 2>/dev/null
