#!/bin/bash
# SQ counters of render_backward_kernel inside one training step: bash tools/raster_bwd_pmc.sh
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/rbp
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RX="render_backward_kernel"
B="python $R/tools/bench_train.py --scenes 24 --steps 1 --warmup 1"
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/a -o r -- $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -o r -- $B > $O/b.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python - "$O" <<'PY'
import csv, glob, collections, re, sys
for sub in ("a", "b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.search(r"([a-z_0-9]+_kernel)", r["Kernel_Name"]).group(1)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        print(sub, k, len(n[k]), {c.replace("SQ_", ""): f"{x / len(n[k]):.3g}" for c, x in v.items()})
PY
