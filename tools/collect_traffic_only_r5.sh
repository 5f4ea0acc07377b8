#!/bin/bash
# Re-collect only the HBM-traffic passes of tools/collect_profiles_r5.sh (FETCH_SIZE / WRITE_SIZE in separate passes) after a late kernel change.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --mode infer --no-cpu-baseline --no-f32 --no-fast --no-targets70"
rm -rf $O/fetch $O/write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o b -- $B --no-roofline --steps 1 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o b -- $B --no-roofline --steps 1 --warmup 1 > $O/write.log 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
