#!/bin/bash
# SQ counters of the rasterizer's forward kernels on the bench step (two --pmc passes + durations from a stats pass): bash tools/pmc_raster.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pr_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --mode infer --no-cpu-baseline --no-f32 --no-fast --no-targets70 --no-roofline --steps 1 --warmup 1"
RX="preprocess_kernel|render_kernel|tile_sort_kernel|segment_sort_kernel|scatter_kernel"
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq1 -o g -- $B > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -o g -- $B > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $O/sq3 -o g -- $B > $O/sq3.log 2>&1
python - "$O" <<'PY'
import collections, csv, glob, sys
O = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ("sq1", "sq2", "sq3"):
    fs = glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = next((n for n in ("preprocess_kernel", "render_kernel", "tile_sort_kernel", "segment_sort_kernel", "scatter_kernel") if n in r["Kernel_Name"]), None)
        if k: cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in cnt.items():
    c = {n: sum(v) / len(v) for n, v in c.items()}
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    print(f"== {k}: waves {c.get('SQ_WAVES', 0):.0f}, busy cycles/32 {c.get('SQ_BUSY_CYCLES', 0) / 32:.3e}")
    for n in sorted(c):
        extra = f"  {100 * c[n] / wc:6.1f} % of wave cycles" if n.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES")) else (f"  {c[n] / max(c.get('SQ_WAVES', 1), 1):9.1f} per wave" if n.startswith("SQ_INSTS") else "")
        print(f"   {n:26s} {c[n]:.4e}{extra}")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
