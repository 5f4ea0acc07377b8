cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_raster_gpu.py -x -q -k "backward or twist or camera_lists or layout or memory" 2>&1 | tail -5
python tools/bench_raster_fb.py --scenes 8 --iters 5 --check 2>/dev/null
VS_RBWD_STAGED=0 python tools/bench_raster_fb.py --scenes 8 --iters 5 --check 2>/dev/null
bash tools/raster_fb_prof.sh stg 8 stats,sq 2>&1 | grep -v "^{" | head -30
