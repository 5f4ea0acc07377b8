#!/bin/bash
# SQ counters of the split attention kernel (attention_sp_kernel, or attention_split_kernel with VS_ATTN_PACKED=0) on one shape: bash tools/pmc_one_attn.sh <tag> encoder|video
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pa_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq1 -o g -- python $R/tools/one_attn.py $2 > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/sq2 -o g -- python $R/tools/one_attn.py $2 > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o g -- python $R/tools/one_attn.py $2 > $O/t.log 2>&1
python - "$O" <<'PY'
import collections, csv, glob, sys
O = sys.argv[1]
cnt = collections.defaultdict(list)
for tag in ("sq1", "sq2"):
    for r in csv.DictReader(open(glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True)[0])):
        if "attention_sp" in r["Kernel_Name"]:
            cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in cnt.items()}
dur = next(float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3 for r in csv.DictReader(open(glob.glob(f"{O}/t/**/*kernel_stats.csv", recursive=True)[0])) if "attention_sp" in r["Name"])
cyc = c["SQ_BUSY_CYCLES"] / 32
print(f"{dur:.1f} us/dispatch; clock {cyc / dur / 1e3:.2f} GHz; MFMA busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.1f} %")
for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
    print(f"  {k:22s} {100 * c[k] / c['SQ_WAVE_CYCLES']:.1f} % of wave cycles")
print(f"  VALU insts per wave {c['SQ_INSTS_VALU'] / max(c['SQ_WAVES'], 1):.0f}, LDS insts per wave {c['SQ_INSTS_LDS'] / max(c['SQ_WAVES'], 1):.0f}, waves {c['SQ_WAVES']:.0f}; LDS conflicts {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1f} %")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
