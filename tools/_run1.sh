cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_raster_gpu.py -x -q -k "backward" 2>&1 | tail -5
for v in main w4; do
  if [ $v = main ]; then unset VICASPLAT_HIP_LIB; else export VICASPLAT_HIP_LIB=$PWD/variants/libvicasplat_hip_$v.so; fi
  echo "== $v"; python tools/bench_raster_fb.py --scenes 8 --iters 5 --check
  echo "== $v old kernel"; VS_RBWD_WAVES=4 python tools/bench_raster_fb.py --scenes 8 --iters 5 --check
done
