cd $GRAFT_REPO_ROOT
VS_DETERMINISTIC=1 python tools/batch_invariance3.py 2>/dev/null
VS_DETERMINISTIC=1 VS_GEMM_MI=4 python tools/batch_invariance3.py 2>/dev/null
VS_DETERMINISTIC=1 VS_GEMM_MI=4 VS_CONV_MI=4 VS_CONV_SPLIT_MI=4 python tools/batch_invariance3.py 2>/dev/null
VS_DETERMINISTIC=1 VS_GEMM_MI=4 VS_CONV_T256_MIN=1 python tools/batch_invariance3.py 2>/dev/null
VS_DETERMINISTIC=1 python tools/bench_b1.py --scenes 1 2>/dev/null
VS_DETERMINISTIC=1 VS_GEMM_MI=4 python tools/bench_b1.py --scenes 1 2>/dev/null
python tools/bench_b1.py --scenes 24 --iters 8 2>/dev/null
VS_DETERMINISTIC=1 python tools/bench_b1.py --scenes 24 --iters 8 2>/dev/null
VS_DETERMINISTIC=1 VS_GEMM_MI=4 python tools/bench_b1.py --scenes 24 --iters 8 2>/dev/null
