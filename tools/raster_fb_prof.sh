#!/bin/bash
# Rasterizer forward + backward on the bench scene: per-kernel time (stats pass), HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) and
# VALU / wave counters of the two backward kernels.  bash tools/raster_fb_prof.sh <tag> [scenes] [passes: stats,traffic,sq]
tag=${1:-x}; sc=${2:-8}; passes=${3:-stats,traffic,sq}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/rfb_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="python $R/tools/bench_raster_fb.py --scenes $sc --iters 3 --check"
$T > $O/untraced.json 2> $O/untraced.err; cat $O/untraced.json
RX="preprocess|render|tile_sort|segment_sort|scatter_kernel|tile_scan|grec|fill"
if [[ $passes == *stats* ]]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $T > $O/stats.log 2>&1
python - "$O" $sc <<'PY' | tee $O/kernels.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True)[0]
sc = 288.0 / (int(sys.argv[2]) * 12)
print("kernel, launches, us/launch, ms per 288 views (4 executed iterations)")
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("preprocess", "render", "tile_sort", "segment_sort", "scatter_kernel", "tile_scan", "grec", "fillBuffer", "FillFunctor")):
        per = float(r["TotalDurationNs"]) / int(r["Calls"])
        print(f"{n[:70]:70s} {r['Calls']:>5s} {per / 1e3:10.1f} {float(r['TotalDurationNs']) / 4 / 1e6 * sc:8.3f}")
PY
fi
if [[ $passes == *traffic* ]]; then
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- $T > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- $T > $O/write.log 2>&1
python - "$O" $sc <<'PY' | tee $O/traffic.txt
import csv, glob, collections, re, sys
O = sys.argv[1]; sc = 288.0 / (int(sys.argv[2]) * 12)
def load(pat, name):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(glob.glob(pat, recursive=True)[0])):
        if r["Counter_Name"] != name: continue
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:60]
        d[k][0] += 1; d[k][1] += float(r["Counter_Value"])
    return d
f, w = load(O + "/fetch/**/*counter_collection.csv", "FETCH_SIZE"), load(O + "/write/**/*counter_collection.csv", "WRITE_SIZE")
print("kernel, launches, fetch GB (raw KiB x2 -- x1 for the gather kernels render*), write GB: per 288 views (4 executed iterations)")
for k in sorted(set(f) | set(w)):
    mult = 1.0 if k.startswith("render") else 2.0
    print(f"{k:62s} {f[k][0]:4d} {mult * f[k][1] * 1024 / 4 / 1e9 * sc:8.2f} {w[k][1] * 1024 / 4 / 1e9 * sc:8.2f}")
PY
fi
if [[ $passes == *sq* ]]; then
RB="render_backward|preprocess_backward"
rocprofv3 --kernel-trace --kernel-include-regex "$RB" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/sqa -o r -- $T > $O/sqa.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RB" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sqb -o r -- $T > $O/sqb.log 2>&1
python - "$O" <<'PY' | tee $O/sq.txt
import csv, glob, collections, re, sys
for sub in ("sqa", "sqb"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.search(r"([a-z_0-9]+_kernel)", r["Kernel_Name"]).group(1)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        print(sub, k, len(n[k]), {c.replace("SQ_", ""): f"{x / len(n[k]):.4g}" for c, x in v.items()})
PY
fi
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
