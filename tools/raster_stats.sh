#!/bin/bash
# per-kernel time of the rasterizer kernels inside the default bench step: bash tools/raster_stats.sh <tag> [lib.so]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/rs_$1
mkdir -p $O
[ -n "$2" ] && export VICASPLAT_HIP_LIB=$R/$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python $R/bench.py --mode infer --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --no-fast --no-targets70 > $O/log.txt 2>&1
python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("render_kernel", "preprocess_kernel", "tile_sort", "scatter_kernel", "segment_sort", "tile_scan")):
        ms = float(r["TotalDurationNs"]) / 1e6 / int(r["Calls"]); tot += ms
        import re; print(f"  {re.search(r'([a-z_0-9]+_kernel)', n).group(1):24s} {ms:8.3f} ms/call x{r['Calls']}")
print(f"  rasterizer total {tot:.2f} ms")
PY
tail -n 1 $O/log.txt | cut -c1-300
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
