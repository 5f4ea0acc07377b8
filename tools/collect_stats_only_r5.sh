#!/bin/bash
# Re-collect only the per-kernel time passes of tools/collect_profiles_r5.sh (after a kernel change late in the round) and the committed bench line.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --mode infer --no-cpu-baseline --no-f32 --no-fast --no-targets70"
$B --steps 2 --warmup 1 > /dev/null 2>&1
rm -rf $O/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --no-roofline --steps 4 --warmup 1 > $O/stats.log 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
cd $R && python bench.py > $O/bench_full.log 2>&1; tail -1 $O/bench_full.log > $O/bench_line.json
