"""Copy the summaries tools/collect_profiles_r5.sh left under gpurun_out/p4 into profiles/ (round-4 names), derive the PMC tables and print
the per-family tables of profiles/README.md.  python tools/refresh_profiles_r5.py [bench_line.json]"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P3, OUT = os.path.join(R, "gpurun_out", "p5"), os.path.join(R, "profiles")
def one(pat):
    f = glob.glob(os.path.join(P3, pat), recursive=True); assert f, pat; return f[0]
shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(OUT, "round5_bench_kernel_stats.csv"))
shutil.copy(one("stats16/**/*kernel_stats.csv"), os.path.join(OUT, "round5_f16_kernel_stats.csv"))
shutil.copy(one("stats32/**/*kernel_stats.csv"), os.path.join(OUT, "round5_f32_kernel_stats.csv"))
tr = "train2" if os.path.isdir(os.path.join(P3, "train2")) else "train"
shutil.copy(one(tr + "/**/*kernel_stats.csv"), os.path.join(OUT, "round5_train_kernel_stats.csv"))
if os.path.isdir(os.path.join(P3, "train_split")):
    shutil.copy(one("train_split/**/*kernel_stats.csv"), os.path.join(OUT, "round5_train_split_kernel_stats.csv"))
for src, dst in (("train_line.json", "round5_train_step.json"), ("train_split_line.json", "round5_train_split_step.json")):
    if os.path.exists(os.path.join(P3, src)) and open(os.path.join(P3, src)).read().strip():
        json.dump(json.loads(open(os.path.join(P3, src)).read().strip().splitlines()[-1]), open(os.path.join(OUT, dst), "w"), indent=1)
subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_traffic.py"), one("fetch/**/*counter_collection.csv"), one("write/**/*counter_collection.csv"),
                       "2", os.path.join(OUT, "round5_pmc_traffic.json")])
pm = json.load(open(os.path.join(OUT, "round5_pmc_traffic.json")))
pm["workload"]["dtype"] = "split"
# ---- executed VALU wave-instructions of preprocess_kernel / render_kernel (bench.py: roofline_rasterizer.valu_issue) ----
try:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one("raster_valu/**/*counter_collection.csv"))):
        for k in ("preprocess_kernel", "render_kernel"):
            if k in r["Kernel_Name"] and "backward" not in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for r in csv.DictReader(open(one("stats/**/*kernel_stats.csv"))):
        for k in ("preprocess_kernel", "render_kernel"):
            if k in r["Name"] and "backward" not in r["Name"]:
                dur[k] = float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3
    P, S, V = 524288, pm["workload"]["scenes_per_gpu"], 12
    R = None
    try:
        R = json.load(open(sys.argv[1]))["roofline_rasterizer"]["num_rendered"] if len(sys.argv) > 1 else None
    except Exception:
        pass
    units = {"preprocess_kernel": (float(P) * S * V, "(Gaussian, camera) pair"), "render_kernel": (float(S * V) * 65536, "pixel")}
    pm["raster_valu"] = {k: dict(valu_insts_per_launch=sum(v["SQ_INSTS_VALU"]) / len(v["SQ_INSTS_VALU"]), waves=sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"]),
                                 traced_us=dur[k], units_per_launch=units[k][0], unit="VALU lane-instructions per " + units[k][1])
                         for k, v in acc.items() if k in dur}
    pm["raster_valu_command"] = "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES (own pass) + the kernel-stats pass for the durations"
except Exception as e:
    print("raster_valu: not derived:", repr(e))
json.dump(pm, open(os.path.join(OUT, "round5_pmc_traffic.json"), "w"), indent=1)
if len(sys.argv) > 1:
    line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    json.dump(line, open(os.path.join(OUT, "round5_bench_line.json"), "w"), indent=1)
# ---- SQ counters of gemm256_kernel<split> ----
cnt = collections.defaultdict(list)
for tag in ("gemm_sq1", "gemm_sq2"):
    for r in csv.DictReader(open(one(tag + "/**/*counter_collection.csv"))):
        if "gemm256_kernel" in r["Kernel_Name"]:
            cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in cnt.items()}
dur = next(float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3 for r in csv.DictReader(open(one("gemm_t/**/*kernel_stats.csv"))) if "gemm256_kernel" in r["Name"])
cyc = c["SQ_BUSY_CYCLES"] / 32
M, N, K = 49152, 4096, 1024
fl = 2.0 * M * N * K
md = f"""# gemm256_kernel<split> PMC, round 5 (tools/one_gemm.py {M} {N} {K}: the ViT-L fc1 shape at the bench's row count, split operands, store epilogue; 5 dispatches averaged)

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -- python tools/one_gemm.py {M} {N} {K}
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_LDS -- python tools/one_gemm.py {M} {N} {K}
    rocprofv3 --kernel-trace --stats -- python tools/one_gemm.py {M} {N} {K}        # {dur:.1f} us per dispatch = {fl / dur / 1e6:.0f} TFLOP/s algorithmic, {3 * fl / dur / 1e6:.0f} executed

| counter (per dispatch) | value | reading |
|---|---|---|
| SQ_WAVE_CYCLES | {c['SQ_WAVE_CYCLES']:.3e} | |
| SQ_ACTIVE_INST_ANY | {c['SQ_ACTIVE_INST_ANY']:.3e} | {100 * c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % of wave cycles issuing |
| SQ_WAIT_INST_ANY | {c['SQ_WAIT_INST_ANY']:.3e} | {100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % issue-stalled (a wave streaming MFMAs sits here by construction) |
| SQ_WAIT_ANY | {c['SQ_WAIT_ANY']:.3e} | {100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % parked at `s_barrier` / `s_waitcnt` (f16 kernel, round 2: 30.3 %) |
| SQ_WAIT_INST_LDS | {c['SQ_WAIT_INST_LDS']:.3e} | {100 * c['SQ_WAIT_INST_LDS'] / c['SQ_WAVE_CYCLES']:.1f} % |
| SQ_BUSY_CYCLES | {c['SQ_BUSY_CYCLES']:.3e} | / 32 SEs = {cyc / 1e3:.0f} k cycles for a {dur:.0f} us dispatch => {cyc / dur / 1e3:.2f} GHz sustained under this load (f16 kernel: 1.80 GHz) |
| SQ_VALU_MFMA_BUSY_CYCLES | {c['SQ_VALU_MFMA_BUSY_CYCLES']:.3e} | = 3 x {fl / 1e9:.1f} GFLOP / 1024 FLOP/cycle/SIMD; / ({cyc / 1e3:.0f} k x 1024 SIMDs) = **{100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.1f} % MFMA-busy at the clock the chip ran** ({100 * 3 * fl / dur / 1e6 / 2500:.1f} % against the 2.5 PF headline) |
| SQ_ACTIVE_INST_VALU | {c['SQ_ACTIVE_INST_VALU']:.3e} | the in-LDS conversion + epilogue ({100 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.1f} % of wave cycles) |
| SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE | {c['SQ_LDS_BANK_CONFLICT']:.3e} / {c['SQ_LDS_IDX_ACTIVE']:.3e} | {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1f} % conflict cycles: the in-place conversion's ds_write_b128 in fragment row order (rows 2k / 2k + 1 share a chunk slot under the stores' 32-bank modulus); the build after this profile takes the rows even-first (gemm256.h, `crow`) |
"""
open(os.path.join(OUT, "round5_pmc_gemm256.md"), "w").write(md)
att = "# attention_sp_kernel<3> PMC, round 5 (tools/pmc_one_attn.sh: packed q | k | v, split class; video = 24 scenes x 12 heads x 2064 x 2064, encoder = 192 frames x 16 heads x 257 x 257)\n\n"
for tag in ("video", "encoder"):
    f = os.path.join(P3, f"attn_{tag}.txt")
    if os.path.exists(f):
        att += f"## {tag}\n\n```\n" + "".join(l for l in open(f) if "amdgpu.ids" not in l) + "```\n\n"
open(os.path.join(OUT, "round5_pmc_attention.md"), "w").write(att)
print(md)
FAM = [("gemm", r"gemm256_kernel|gemm_kernel|gemm_smallm|gemm_skinny|conv7x7_256|stem_up_stream|split_pack"), ("wgrad", r"tn_splitk|wgrad|splitk_reduce|head1x1_bwd"), ("conv3x3", r"conv3x3"), ("attention bwd", r"attn_bwd|attn_delta"),
       ("attention", r"attention"), ("raster bwd", r"render_backward|preprocess_backward"), ("raster fwd", r"render_kernel|preprocess_kernel|tile_sort|scatter_kernel|segment_sort|tile_scan"),
       ("layernorm", r"layernorm"), ("upsample", r"upsample"), ("adapter", r"adapter"), ("adamw", r"multi_tensor_apply"),
       ("torch glue", r"at::native|rocclr|Cijk"), ("other hip", r".")]
def table(path, steps):
    fam = {}; tot = 0.0
    for r in csv.DictReader(open(path)):
        ms = float(r["TotalDurationNs"]) / 1e6 / steps; tot += ms
        k = next(n for n, rx in FAM if re.search(rx, r["Name"])); fam[k] = fam.get(k, 0.0) + ms
    print(f"{os.path.basename(path)}: {tot:.1f} ms of kernel time per step")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]): print(f"  {k:16s} {v:8.2f} ms  {100 * v / tot:5.1f} %")
    return fam
table(os.path.join(OUT, "round5_bench_kernel_stats.csv"), 5)
table(os.path.join(OUT, "round5_f16_kernel_stats.csv"), 5)
table(os.path.join(OUT, "round5_f32_kernel_stats.csv"), 3)
table(os.path.join(OUT, "round5_train_kernel_stats.csv"), 4)
print("Cijk rows in the training trace:", sum("Cijk" in r["Name"] for r in csv.DictReader(open(os.path.join(OUT, "round5_train_kernel_stats.csv")))))
for k, v in pm["kernels"].items(): print(f"  traffic {k:36s} {v['hbm_bytes_per_step'] / 1e9:8.2f} GB/step  per launch {v.get('hbm_bytes_per_launch', 0) / 1e6:10.1f} MB")
