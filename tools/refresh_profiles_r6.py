"""Copy the summaries tools/collect_profiles_r6.sh left under gpurun_out/p6 (+ gpurun_out/rbw6) into profiles/ (round-6 names), derive the PMC
table and print the per-family tables of profiles/README.md.  python tools/refresh_profiles_r6.py"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, OUT = os.path.join(R, "gpurun_out", "p6"), os.path.join(R, "profiles")
def one(pat, base=P):
    f = glob.glob(os.path.join(base, pat), recursive=True)
    return f[0] if f else None
def cp(src, dst):
    if src and os.path.exists(src):
        shutil.copy(src, os.path.join(OUT, dst)); return True
    print("missing:", dst); return False
cp(one("stats/**/*kernel_stats.csv"), "round6_bench_kernel_stats.csv")
cp(one("b1/**/*kernel_stats.csv"), "round6_b1_kernel_stats.csv")
cp(one("train/**/*kernel_stats.csv"), "round6_train_kernel_stats.csv")
cp(one("train_split/**/*kernel_stats.csv"), "round6_train_split_kernel_stats.csv")
RB = os.path.join(R, "gpurun_out", "rbw6")
cp(os.path.join(RB, "round6_pmc_raster_bwd.json"), "round6_pmc_raster_bwd.json")
cp(os.path.join(RB, "round6_raster_fb_kernel_stats.csv"), "round6_raster_fb_kernel_stats.csv")
for src, dst in (("train_line.json", "round6_train_step.json"), ("train_split_line.json", "round6_train_split_step.json"), ("b1_latency.json", "round6_b1_latency.json"),
                 ("bench_line.json", "round6_bench_line.json")):
    f = os.path.join(P, src)
    if os.path.exists(f) and open(f).read().strip():
        json.dump(json.loads(open(f).read().strip().splitlines()[-1]), open(os.path.join(OUT, dst), "w"), indent=1)
    else:
        print("missing:", dst)
fe, wr = one("fetch/**/*counter_collection.csv"), one("write/**/*counter_collection.csv")
if fe and wr:
    subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_traffic.py"), fe, wr, "2", os.path.join(OUT, "round6_pmc_traffic.json")])
    pm = json.load(open(os.path.join(OUT, "round6_pmc_traffic.json")))
    pm["workload"]["dtype"] = "split"
    try:    # executed VALU wave-instructions of preprocess_kernel / render_kernel (bench.py: roofline_rasterizer.valu_issue)
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(one("raster_valu/**/*counter_collection.csv"))):
            for k in ("preprocess_kernel", "render_kernel"):
                if k in r["Kernel_Name"] and "backward" not in r["Kernel_Name"]:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur = {}
        for r in csv.DictReader(open(os.path.join(OUT, "round6_bench_kernel_stats.csv"))):
            for k in ("preprocess_kernel", "render_kernel"):
                if k in r["Name"] and "backward" not in r["Name"]:
                    dur[k] = float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3
        Pn, S, V = 524288, pm["workload"]["scenes_per_gpu"], 12
        units = {"preprocess_kernel": (float(Pn) * S * V, "(Gaussian, camera) pair"), "render_kernel": (float(S * V) * 65536, "pixel")}
        pm["raster_valu"] = {k: dict(valu_insts_per_launch=sum(v["SQ_INSTS_VALU"]) / len(v["SQ_INSTS_VALU"]), waves=sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"]),
                                     traced_us=dur[k], units_per_launch=units[k][0], unit="VALU lane-instructions per " + units[k][1])
                             for k, v in acc.items() if k in dur}
        pm["raster_valu_command"] = "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES (own pass) + the kernel-stats pass for the durations"
    except Exception as e:
        print("raster_valu: not derived:", repr(e))
    json.dump(pm, open(os.path.join(OUT, "round6_pmc_traffic.json"), "w"), indent=1)
    for k, v in pm["kernels"].items(): print(f"  traffic {k:36s} {v['hbm_bytes_per_step'] / 1e9:8.2f} GB/step  per launch {v.get('hbm_bytes_per_launch', 0) / 1e6:10.1f} MB")
FAM = [("gemm", r"gemm256_kernel|gemm_kernel|gemm_smallm|gemm_skinny|conv7x7_256|stem_up_stream|split_pack"), ("wgrad", r"tn_splitk|wgrad|splitk_reduce|head1x1_bwd"), ("conv3x3", r"conv3x3"), ("attention bwd", r"attn_bwd|attn_delta"),
       ("attention", r"attention"), ("raster bwd", r"render_backward|preprocess_backward"), ("raster fwd", r"render_kernel|preprocess_kernel|tile_sort|scatter_kernel|segment_sort|tile_scan"),
       ("layernorm", r"layernorm"), ("upsample", r"upsample"), ("adapter", r"adapter"), ("transposes / packs", r"transpose|split16|im2col"), ("adamw", r"multi_tensor_apply"),
       ("torch glue", r"at::native|rocclr|Cijk"), ("other hip", r".")]
def table(name, steps, top=0):
    path = os.path.join(OUT, name)
    if not os.path.exists(path): return
    fam = {}; tot = 0.0; rows = list(csv.DictReader(open(path)))
    for r in rows:
        ms = float(r["TotalDurationNs"]) / 1e6 / steps; tot += ms
        k = next(n for n, rx in FAM if re.search(rx, r["Name"])); fam[k] = fam.get(k, 0.0) + ms
    print(f"{name}: {tot:.2f} ms of kernel time per step")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]): print(f"  {k:20s} {v:8.2f} ms  {100 * v / tot:5.1f} %")
    for r in rows[:top]:
        short = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        print(f"    {short:70s} x{int(r['Calls']) / steps:7.1f}  {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step")
table("round6_bench_kernel_stats.csv", 5, 12)
table("round6_b1_kernel_stats.csv", 23, 14)
table("round6_train_kernel_stats.csv", 4, 30)
table("round6_train_split_kernel_stats.csv", 3, 30)
table("round6_raster_fb_kernel_stats.csv", 4, 12)
