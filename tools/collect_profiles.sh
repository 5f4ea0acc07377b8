#!/bin/bash
# Round-2 profile collection on the GPU box (run from the repo root through gpurun).  Raw rocprofv3 output goes to gpurun_out/p2/, the
# summaries that are committed under profiles/ are derived from it by tools/pmc_traffic.py and by hand (profiles/README.md).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --mode infer --no-cpu-baseline --no-f32 --steps 2 --warmup 1 > /dev/null 2>&1   # untraced warm-up run
# 1. per-kernel time of the default bench step (5 executed steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --mode infer --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 > $O/stats.log 2>&1
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (2 executed steps each)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o b -- python $R/bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o b -- python $R/bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 > $O/write.log 2>&1
# 3. FETCH_SIZE calibration: wide streaming read vs 16-byte gathers of 48-byte records
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/probe/fetch_calib.hip -o /tmp/fetch_calib > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -o c -- /tmp/fetch_calib > $O/calib.log 2>&1
# 4. SQ / MFMA counters of gemm256_kernel on the ViT-L fc1 shape (two passes)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/gemm_sq1 -o g -- python $R/tools/one_gemm.py 49152 4096 1024 > $O/gemm_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_LDS --output-format csv -d $O/gemm_sq2 -o g -- python $R/tools/one_gemm.py 49152 4096 1024 > $O/gemm_sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gemm_t -o g -- python $R/tools/one_gemm.py 49152 4096 1024 > $O/gemm_t.log 2>&1
# 5. the training step (24 scenes, 12 targets: BASELINE configs 4 / 5), 4 executed steps
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python $R/tools/bench_train.py --scenes 24 --steps 3 --warmup 1 > $O/train.log 2>&1
python $R/tools/bench_train.py --scenes 24 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/train_line.json
python $R/tools/bench_train.py --scenes 24 --steps 3 --warmup 1 --checkpoint 2>/dev/null | tail -1 > $O/train_line_ckpt.json
# keep the merge small: the traces themselves are not needed, only stats + counter tables
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O; find $O -type f | head -40
