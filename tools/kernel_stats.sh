#!/bin/bash
# per-kernel time of the kernels whose name matches <pattern> inside the default bench step: bash tools/kernel_stats.sh <tag> <pattern> [lib.so]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/ks_$1
mkdir -p $O
[ -n "$3" ] && export VICASPLAT_HIP_LIB=$R/$3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python $R/bench.py --mode infer --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --no-fast --no-targets70 $BENCH_ARGS > $O/log.txt 2>&1
python - "$O" "$2" <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if re.search(sys.argv[2], n):
        ms = float(r["TotalDurationNs"]) / 1e6 / 3; tot += ms      # 3 executed steps
        print(f"  {n[:90]:90s} {ms:8.3f} ms/step  ({float(r['TotalDurationNs']) / 1e3 / int(r['Calls']):8.1f} us x {int(r['Calls']) // 3})")
print(f"  total {tot:.2f} ms/step")
PY
tail -n 1 $O/log.txt | cut -c1-200
