#!/bin/bash
# Round-5 profile collection on the GPU box (run from the repo root through gpurun).  Raw rocprofv3 output goes to gpurun_out/p5/; the
# summaries committed under profiles/ are derived from it by tools/refresh_profiles_r5.py.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --mode infer --no-cpu-baseline --no-f32 --no-fast --no-targets70"
$B --steps 2 --warmup 1 > /dev/null 2>&1   # untraced warm-up run
# 1. per-kernel time of the default (split) bench step, of the f16 fast path and of the exact-f32 path (5 / 5 / 3 executed steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --no-roofline --steps 4 --warmup 1 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats16 -o bench -- $B --no-roofline --dtype f16 --steps 4 --warmup 1 > $O/stats16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats32 -o bench -- $B --no-roofline --dtype f32 --steps 2 --warmup 1 > $O/stats32.log 2>&1
# 2. HBM traffic of the split step: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (2 executed steps each)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o b -- $B --no-roofline --steps 1 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o b -- $B --no-roofline --steps 1 --warmup 1 > $O/write.log 2>&1
# 2b. executed VALU wave-instructions of the rasterizer's two issue-bound kernels (own pass; 2 executed steps)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/raster_valu -o b -- $B --no-roofline --steps 1 --warmup 1 > $O/raster_valu.log 2>&1
# 3. SQ / MFMA counters of gemm256_kernel<split> on the ViT-L fc1 shape at the bench's row count (two passes) + its duration
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/gemm_sq1 -o g -- python $R/tools/one_gemm.py 49152 4096 1024 > $O/gemm_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_LDS --output-format csv -d $O/gemm_sq2 -o g -- python $R/tools/one_gemm.py 49152 4096 1024 > $O/gemm_sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gemm_t -o g -- python $R/tools/one_gemm.py 49152 4096 1024 > $O/gemm_t.log 2>&1
# 3b. SQ / LDS / MFMA counters of attention_sp_kernel on the video shape and the frame-encoder shape (tools/pmc_one_attn.sh: two --pmc passes + one stats pass each)
bash $R/tools/pmc_one_attn.sh r5_video video > $O/attn_video.txt 2>&1
bash $R/tools/pmc_one_attn.sh r5_enc encoder > $O/attn_encoder.txt 2>&1
cd /tmp
# 4. the training step (24 scenes, 12 targets: BASELINE configs 4 / 5), 4 executed steps
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python $R/tools/bench_train.py --scenes 24 --steps 3 --warmup 1 > $O/train.log 2>&1
python $R/tools/bench_train.py --scenes 24 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/train_line.json
# 5. the same step in the split class (reference precision forward AND backward), 8 scenes, 3 executed steps
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_split -o t -- python $R/tools/bench_train.py --scenes 8 --steps 2 --warmup 1 --dtype split > $O/train_split.log 2>&1
python $R/tools/bench_train.py --scenes 8 --steps 3 --warmup 1 --dtype split 2>/dev/null | tail -1 > $O/train_split_line.json
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O; find $O -type f | head -60
