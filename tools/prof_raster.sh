#!/bin/bash
# Per-kernel times of the rasterizer's forward kernels on the bench step (rocprofv3 kernel trace, 3 executed steps).  Run through gpurun:
#   gpurun -- 'bash tools/prof_raster.sh <tag>'  ->  gpurun_out/prof_<tag>_raster.txt
tag=${1:-x}
cd ${GRAFT_REPO_ROOT:-.}
out=$PWD/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python bench.py --mode infer --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --no-fast --no-targets70 > /tmp/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" > $out/prof_${tag}_raster.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("preprocess_kernel", "render_kernel", "tile_sort", "segment_sort", "scatter_kernel", "tile_scan")):
        ms = float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e6
        tot += ms
        print(f"{ms:8.3f} ms/launch  x{r['Calls']}  {n[:70]}")
print(f"{tot:8.3f} ms rasterizer forward per step")
print(f"{sum(float(r['TotalDurationNs']) for r in rows) / 3e6:8.2f} ms of kernel time per step (3 executed steps)")
PY
cp "$f" $out/prof_${tag}_kernel_stats.csv
cat $out/prof_${tag}_raster.txt
