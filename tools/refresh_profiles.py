"""Copy the summaries tools/collect_profiles.sh left under gpurun_out/p2 into profiles/ (round-2 names) and print the per-family tables
of profiles/README.md from them.  python tools/refresh_profiles.py [bench_line.json]"""
import csv, glob, json, os, re, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P2, OUT = os.path.join(R, "gpurun_out", "p2"), os.path.join(R, "profiles")
def one(pat):
    f = glob.glob(os.path.join(P2, pat), recursive=True); assert f, pat; return f[0]
shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(OUT, "round2_bench_kernel_stats.csv"))
shutil.copy(one("train/**/*kernel_stats.csv"), os.path.join(OUT, "round2_train_kernel_stats.csv"))
subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_traffic.py"), one("fetch/**/*counter_collection.csv"), one("write/**/*counter_collection.csv"),
                       "2", os.path.join(OUT, "round2_pmc_traffic.json")])
tl = [json.loads(open(os.path.join(P2, n)).read().strip().splitlines()[-1]) for n in ("train_line.json", "train_line_ckpt.json")]
json.dump(dict(plain=tl[0], checkpointed=tl[1]), open(os.path.join(OUT, "round2_train_step.json"), "w"), indent=1)
if len(sys.argv) > 1:
    line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    json.dump(line, open(os.path.join(OUT, "round2_bench_line.json"), "w"), indent=1)
FAM = [("gemm", r"gemm256_kernel|gemm_kernel|gemm_smallm"), ("wgrad", r"tn_splitk|wgrad|splitk_reduce"), ("conv3x3", r"conv3x3"), ("attention bwd", r"attn_bwd"),
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
table(os.path.join(OUT, "round2_bench_kernel_stats.csv"), 5)
for r in csv.DictReader(open(os.path.join(OUT, "round2_bench_kernel_stats.csv"))):
    if re.search(r"render_kernel|preprocess_kernel|tile_sort|scatter_kernel|segment_sort", r["Name"]):
        print("    ", re.search(r"([a-z_]+_kernel)", r["Name"]).group(1), round(float(r["TotalDurationNs"]) / 5e6, 2))
table(os.path.join(OUT, "round2_train_kernel_stats.csv"), 4)
pm = json.load(open(os.path.join(OUT, "round2_pmc_traffic.json")))
for k, v in pm["kernels"].items(): print(f"  traffic {k:36s} {v['hbm_bytes_per_step'] / 1e9:8.2f} GB/step")
