#!/bin/bash
# Per-kernel times of the rasterizer kernels (forward + backward) inside a traced 16-bit training step: bash tools/prof_raster_bwd.sh <tag> [scenes]
tag=${1:-x}; sc=${2:-8}
cd ${GRAFT_REPO_ROOT:-.}
out=$PWD/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/profb_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb_$tag -- python tools/bench_train.py --scenes $sc --steps 2 --warmup 1 > /tmp/profb_$tag.log 2>&1
f=$(find /tmp/profb_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee $out/profb_${tag}_raster.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("preprocess", "render", "tile_sort", "segment_sort", "scatter_kernel", "tile_scan")):
        print(f"{float(r['TotalDurationNs']) / int(r['Calls']) / 1e6:8.3f} ms/launch  x{r['Calls']}  {n[:80]}")
print(f"{sum(float(r['TotalDurationNs']) for r in rows) / 3e6:8.2f} ms of kernel time per step (3 executed steps)")
PY
