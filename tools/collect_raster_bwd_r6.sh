#!/bin/bash
# profiles/round6_pmc_raster_bwd.json + round6_raster_fb_kernel_stats.csv: the rasterizer forward + backward on the bench scene family
# (24 eight-view scenes, 12 target views each = the per-GPU batch of the training legs), per-kernel time (stats pass), HBM traffic
# (separate FETCH_SIZE / WRITE_SIZE passes) and VALU / LDS counters of the two backward kernels (SQ pass).  bash tools/collect_raster_bwd_r6.sh [scenes]
sc=${1:-24}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/rbw6
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="python $R/tools/bench_raster_fb.py --scenes $sc --iters 3"
$T > $O/untraced.json 2> $O/untraced.err; cat $O/untraced.json
RX="preprocess|render|tile_sort|segment_sort|scatter_kernel|tile_scan|fillBuffer"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $T > $O/stats.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- $T > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- $T > $O/write.log 2>&1
RB="render_backward|preprocess_backward"
rocprofv3 --kernel-trace --kernel-include-regex "$RB" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/sqa -o r -- $T > $O/sqa.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RB" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sqb -o r -- $T > $O/sqb.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/round6_raster_fb_kernel_stats.csv
python - "$O" $sc <<'PY'
import csv, glob, collections, json, re, sys
O, sc = sys.argv[1], int(sys.argv[2]); views = sc * 12; iters = 4
def load(pat, name):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(glob.glob(pat, recursive=True)[0])):
        if r["Counter_Name"] != name: continue
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
        d[k][0] += 1; d[k][1] += float(r["Counter_Value"])
    return d
f, w = load(O + "/fetch/**/*counter_collection.csv", "FETCH_SIZE"), load(O + "/write/**/*counter_collection.csv", "WRITE_SIZE")
ker = {}
for k in sorted(set(f) | set(w)):
    mult = 1.0 if k.startswith("render") else 2.0      # FETCH_SIZE x2 on gfx950 except the 16-byte gather kernels (profiles/round2_fetch_calibration.md)
    ker[k] = dict(launches_per_iteration=max(f[k][0], w[k][0]) / iters, fetch_bytes_per_view=mult * f[k][1] * 1024 / iters / views,
                  write_bytes_per_view=w[k][1] * 1024 / iters / views)
stats = {re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"]).split("(")[0].split("<")[0]: float(r["TotalDurationNs"]) / iters
         for r in csv.DictReader(open(O + "/round6_raster_fb_kernel_stats.csv"))}
sq = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for sub in ("sqa", "sqb"):
    for fn in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = re.search(r"([a-z_0-9]+_kernel)", r["Kernel_Name"]).group(1)
            sq[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, sub)].add(r["Dispatch_Id"])
bwd = ["__amd_rocclr_fillBufferAligned", "render_backward_seg_kernel", "preprocess_backward_kernel"]
fwdk = ["preprocess_kernel", "tile_scan_kernel", "scatter_kernel", "tile_sort_kernel", "segment_sort_kernel", "render_kernel"]
tot = lambda names: sum(ker[k]["fetch_bytes_per_view"] + ker[k]["write_bytes_per_view"] for k in names if k in ker)
valu = {}
for k in ("render_backward_seg_kernel", "preprocess_backward_kernel"):
    if k in sq and k in stats:
        n = len(nd[(k, "sqb")]) or 1
        insts = sq[k]["SQ_INSTS_VALU"] / n; us = stats[k] / 1e3
        valu[k] = dict(valu_wave_insts_per_launch=int(insts), insts_lds_per_launch=int(sq[k]["SQ_INSTS_LDS"] / n), traced_us=round(us, 1),
                       valu_issue_frac=round(insts / (us * 1e-6) / (1024 * 2.4e9 / 2.0), 4),
                       wait_inst_lds_frac_of_wave_cycles=round(sq[k]["SQ_WAIT_INST_LDS"] / max(sq[k]["SQ_WAVE_CYCLES"], 1), 4),
                       wait_any_frac_of_wave_cycles=round(sq[k]["SQ_WAIT_ANY"] / max(sq[k]["SQ_WAVE_CYCLES"], 1), 4))
out = dict(command="tools/collect_raster_bwd_r6.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_* (separate passes) -- python tools/bench_raster_fb.py --scenes %d --iters 3" % sc,
           workload=dict(scenes=sc, target_views=12, views=views, note="encoder-predicted Gaussians of the bench's synthetic scenes (golden weights), MSE against a random target"),
           units="KiB as reported; FETCH_SIZE x2 (gfx950) except render* (16-byte gathers, calibrated x1); per rendered view",
           kernels=ker, traced_ns_per_iteration=stats,
           backward=dict(hbm_bytes_per_view=tot(bwd), kernels=bwd, basis="PMC counters (FETCH_SIZE x2 gfx950 correction except the gather kernel, + WRITE_SIZE), per rendered view x views of the step"),
           forward=dict(hbm_bytes_per_view=tot(fwdk), kernels=fwdk), valu_issue=valu)
json.dump(out, open(O + "/round6_pmc_raster_bwd.json", "w"), indent=1)
print(json.dumps(dict(backward_GB_per_288=tot(bwd) * 288 / 1e9, forward_GB_per_288=tot(fwdk) * 288 / 1e9, valu=valu)))
for k in bwd + fwdk:
    if k in stats: print(f"{k:40s} {stats[k] / 1e6 * 288 / views:8.3f} ms per 288 views")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
rm -rf $O/stats $O/fetch $O/write $O/sqa $O/sqb
du -sh $O
