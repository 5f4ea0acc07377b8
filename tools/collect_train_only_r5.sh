#!/bin/bash
# Re-collect only the training-step passes of tools/collect_profiles_r5.sh (after a late change on the training path).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/train $O/train_split
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python $R/tools/bench_train.py --scenes 24 --steps 3 --warmup 1 > $O/train.log 2>&1
python $R/tools/bench_train.py --scenes 24 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/train_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_split -o t -- python $R/tools/bench_train.py --scenes 8 --steps 2 --warmup 1 --dtype split > $O/train_split.log 2>&1
python $R/tools/bench_train.py --scenes 8 --steps 3 --warmup 1 --dtype split 2>/dev/null | tail -1 > $O/train_split_line.json
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
