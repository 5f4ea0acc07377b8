R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/rp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RX="render_kernel|preprocess_kernel|tile_sort_kernel|scatter_kernel|segment_sort_kernel"
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/a -o r -- python $R/bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 > $O/a.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -o r -- python $R/bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 > $O/b.log 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out/rp"
for sub in ("a","b"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        print(sub,k,"dispatches",len(n[k]),{c:f"{x/len(n[k]):.4g}" for c,x in v.items()})
PY
