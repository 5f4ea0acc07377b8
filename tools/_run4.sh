cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_raster_gpu.py -x -q -k "backward or camera or twist or layout" 2>&1 | tail -5
python tools/bench_raster_fb.py --scenes 8 --iters 5 --check 2>/dev/null
bash tools/raster_fb_prof.sh k2b 8 stats 2>&1 | grep -v "^{" | head -8
