cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_split_path_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | tail -4
for b in 1 2 4; do
python tools/bench_b1.py --scenes $b --iters 15 2>/dev/null
VS_GEMM_MI2=0 python tools/bench_b1.py --scenes $b --iters 15 2>/dev/null
done
python tools/bench_b1.py --scenes 24 --iters 6 2>/dev/null
