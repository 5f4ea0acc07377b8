cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_e2e_gpu.py -x -q 2>&1 | tail -4
python tools/bench_raster_fb.py --scenes 8 --iters 5 --check 2>/dev/null
VS_RENDER_BLOCK=0 python tools/bench_raster_fb.py --scenes 8 --iters 5 --check 2>/dev/null
python tools/bench_raster_fb.py --scenes 8 --iters 5 --no-bwd 2>/dev/null
VS_RENDER_BLOCK=0 python tools/bench_raster_fb.py --scenes 8 --iters 5 --no-bwd 2>/dev/null
bash tools/raster_fb_prof.sh rblk 8 stats 2>&1 | grep -v "^{" | head -9
