cd $GRAFT_REPO_ROOT
python bench.py --mode infer --no-cpu-baseline --no-f32 --no-fast --no-targets70 --no-roofline --steps 5 --warmup 2 > gpurun_out/b1_line.json 2> gpurun_out/b1_line.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b1_line.json").read().strip().splitlines()[-1])
print("headline ms", d["ms_per_step"], "latency", json.dumps(d["latency_b1"]))
PY
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_b1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o g -- python $GRAFT_REPO_ROOT/bench.py --scenes-per-gpu 1 --mode infer --no-cpu-baseline --no-f32 --no-fast --no-targets70 --no-roofline --no-latency --steps 20 --warmup 3 > $O/run.log 2>&1
tail -1 $O/run.log | cut -c1-300
cp $(find $O/t -name "*kernel_stats.csv" | head -1) $O/stats.csv
find $O/t -name "*kernel_trace.csv" -delete; find $O/t -name "*.db" -delete
python - "$O/stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (23 executed steps):", tot / 23 / 1e6)
for r in rows[:40]:
    short = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
    print(f"{short:80s} calls/step {int(r['Calls'])/23:7.1f}  avg {float(r['AverageNs'])/1e3:9.1f} us  ms/step {float(r['TotalDurationNs'])/23/1e6:8.3f}")
PY
