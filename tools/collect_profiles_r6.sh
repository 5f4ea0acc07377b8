#!/bin/bash
# Round-6 profile collection on the GPU box (run from the repo root through gpurun).  Raw rocprofv3 output goes to gpurun_out/p6/; the
# summaries committed under profiles/ are derived from it by tools/refresh_profiles_r6.py.  Parts: bash tools/collect_profiles_r6.sh [line,infer,traffic,rbwd,b1,train]
parts=${1:-line,infer,traffic,rbwd,b1,train}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --mode infer --no-cpu-baseline --no-f32 --no-fast --no-targets70 --no-latency"
if [[ $parts == *line* ]]; then      # the full default bench line of this build (what the driver runs)
  python $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
fi
if [[ $parts == *infer* ]]; then     # per-kernel time of the default (split) inference step, 5 executed steps
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --no-roofline --steps 4 --warmup 1 > $O/stats.log 2>&1
fi
if [[ $parts == *traffic* ]]; then   # HBM traffic of the split step: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (2 executed steps each) + the rasterizer's VALU counts
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o b -- $B --no-roofline --steps 1 --warmup 1 > $O/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o b -- $B --no-roofline --steps 1 --warmup 1 > $O/write.log 2>&1
  rocprofv3 --kernel-trace --kernel-include-regex "preprocess_kernel|render_kernel" --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/raster_valu -o b -- $B --no-roofline --steps 1 --warmup 1 > $O/raster_valu.log 2>&1
fi
if [[ $parts == *rbwd* ]]; then      # rasterizer forward + backward of the training batch (24 scenes): time, traffic, SQ counters
  bash $R/tools/collect_raster_bwd_r6.sh 24 > $O/rbwd.log 2>&1
  cd /tmp
fi
if [[ $parts == *b1* ]]; then        # one-scene batch (the latency leg's workload), 23 executed steps
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -o g -- python $R/bench.py --scenes-per-gpu 1 --mode infer --no-cpu-baseline --no-f32 --no-fast --no-targets70 --no-roofline --no-latency --steps 20 --warmup 3 > $O/b1.log 2>&1
  python $R/tools/bench_b1.py --iters 30 > $O/b1_latency.json 2>/dev/null
fi
if [[ $parts == *train* ]]; then     # the training step: f16 class 24 scenes (4 executed steps), split class 8 scenes (3 executed steps), and their untraced lines
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python $R/tools/bench_train.py --scenes 24 --steps 3 --warmup 1 > $O/train.log 2>&1
  python $R/tools/bench_train.py --scenes 24 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/train_line.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_split -o t -- python $R/tools/bench_train.py --scenes 8 --steps 2 --warmup 1 --dtype split > $O/train_split.log 2>&1
  python $R/tools/bench_train.py --scenes 8 --steps 3 --warmup 1 --dtype split 2>/dev/null | tail -1 > $O/train_split_line.json
fi
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O; find $O -type f | head -60
