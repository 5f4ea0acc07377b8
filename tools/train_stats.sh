#!/bin/bash
# per-kernel time of one training step (24 scenes x 12 targets): bash tools/train_stats.sh <tag> [topN]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/ts_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/tools/bench_train.py --scenes ${SCENES:-24} --steps 2 --warmup 1 $TRAIN_ARGS > $O/log.txt 2>&1
python - "$O" "${2:-28}" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f))); n = 3
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time {tot / 1e6 / n:.1f} ms/step")
for r in rows[:int(sys.argv[2])]:
    m = re.search(r"([A-Za-z_0-9]+_kernel[A-Za-z_0-9]*(<[^>]*>)?|Cijk\w+|at::native::\w+)", r["Name"])
    print(f"  {(m.group(1) if m else r['Name'][:50])[:56]:56s} x{int(r['Calls']) / n:7.1f} {float(r['TotalDurationNs']) / 1e6 / n:8.2f} ms/step")
PY
tail -n 1 $O/log.txt | cut -c1-200
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
