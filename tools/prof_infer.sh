#!/bin/bash
# Kernel stats of the headline inference step: bash tools/prof_infer.sh <tag> [ENV=.. ...] -> gpurun_out/prof_<tag>/stats.csv (+ top rows printed)
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o g -- python $R/bench.py --mode infer --no-cpu-baseline --no-f32 --no-fast --no-targets70 --no-roofline --steps 5 --warmup 2 > $O/run.log 2>&1
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
cp $f $O/stats.csv
find $O/t -name "*kernel_trace.csv" -delete; find $O/t -name "*.db" -delete
python - "$O/stats.csv" "${PROF_FILTER:-}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2]
for r in rows[:200]:
    n = r["Name"]
    if flt and not any(k in n for k in flt.split(",")): continue
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    print(f"{short:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:10.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
    if not flt and rows.index(r) > 28: break
PY
