#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel of the 8-scene split-class training step (separate passes): where the training step's HBM traffic goes.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/ptrain
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="python $R/tools/bench_train.py --scenes 8 --steps 1 --warmup 1 --dtype ${1:-split}"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- $T > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- $T > $O/write.log 2>&1
python - <<PY
import csv, glob, collections, re
def load(pat, name):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(glob.glob(pat, recursive=True)[0])):
        if r["Counter_Name"] != name: continue
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:70]
        d[k][0] += 1; d[k][1] += float(r["Counter_Value"])
    return d
f, w = load("$O/fetch/**/*counter_collection.csv", "FETCH_SIZE"), load("$O/write/**/*counter_collection.csv", "WRITE_SIZE")
rows = sorted(((k, f[k][0], 2 * f[k][1] * 1024 / 2 / 1e9, w[k][1] * 1024 / 2 / 1e9) for k in set(f) | set(w)), key=lambda t: -(t[2] + t[3]))
print("per step (2 executed steps averaged; fetch x2 gfx950 correction): kernel, launches/step, fetch GB, write GB")
for k, n, fb, wb in rows[:40]: print(f"{k:72s} {n // 2:5d} {fb:8.2f} {wb:8.2f}")
print("total", sum(r[2] for r in rows), sum(r[3] for r in rows))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
