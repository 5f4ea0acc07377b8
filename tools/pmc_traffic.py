"""Aggregate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs) of bench.py into profiles/round2_pmc_traffic.json.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps_executed> <out.json> [scenes_per_gpu=24]
(bench.py only uses the file when its "workload" block matches the workload being benchmarked: the default 24 scenes x 8 views x 12 targets)
HBM bytes per kernel family, gfx950 corrections as MI355X_MICROARCH.md prescribes: rocprofv3 reports FETCH_SIZE / WRITE_SIZE
in KiB; FETCH_SIZE counts a wide coalesced read at half its size on gfx950 (x2); WRITE_SIZE is taken as reported.
Round 2: the x2 is NOT applied to render_kernel, whose reads are 16-byte gathers of 48-byte records: calibrated with
tools/probe/fetch_calib.hip (profiles/round2_fetch_calibration.md) -- a streaming read of 1 GiB reports 512 MiB (x2 confirmed), a random
gather of 16 M 48-byte records reports 1.66x the requested bytes, which is already the line-granular traffic (doubling it would exceed the
HBM rate the kernel's duration allows).
"""
import csv, json, re, sys, collections

FAMILIES = [("gemm", ("gemm256_kernel", "gemm_kernel", "gemm_smallm_kernel", "gemm_skinny_kernel", "conv7x7_256_kernel", "stem_up_stream_kernel")), ("conv3x3", ("conv3x3_256_kernel", "conv3x3_kernel")),
            ("attention", ("attention_kernel", "attention_res_kernel", "attention_split_kernel", "attention_sp_kernel", "attention_f32_kernel")), ("rasterizer", ("preprocess_kernel", "scatter_kernel", "tile_scan_kernel", "tile_sort_kernel",
                                                                  "segment_sort_kernel", "render_kernel")),
            ("layernorm", ("layernorm_mod_kernel", "layernorm_rows_kernel")), ("upsample", ("upsample2x_kernel", "upsample2x_f32")), ("adapter", ("adapter_",))]


def family(name):
    for fam, pats in FAMILIES:
        if any(p in name for p in pats):
            return fam
    return "other"


GATHER_KERNELS = ("render_kernel",)   # FETCH_SIZE taken as reported (calibrated), everything else x2


def load(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])  # family -> [dispatches, KiB (fetch: already corrected)]
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        fam = family(r["Kernel_Name"])
        key = (r["Dispatch_Id"], fam)
        if key not in seen:
            seen.add(key); per[fam][0] += 1
        v = float(r["Counter_Value"])
        if counter == "FETCH_SIZE" and not any(k in r["Kernel_Name"] for k in GATHER_KERNELS):
            v *= 2.0
        per[fam][1] += v
        if fam == "conv3x3":        # per-instantiation detail of the convolutions (where the family's fetch bytes go)
            short = re.search(r"conv3x3\w*(<[^>]*>)?", r["Kernel_Name"]).group(0).replace(" ", "")
            per["conv3x3/" + short][1] += v
            if (r["Dispatch_Id"], short) not in seen:
                seen.add((r["Dispatch_Id"], short)); per["conv3x3/" + short][0] += 1
        if fam == "rasterizer":     # per-kernel detail of the rasterizer
            short = next(k for k in FAMILIES[3][1] if k in r["Kernel_Name"])
            per["rasterizer/" + short][1] += v
            if (r["Dispatch_Id"], short) not in seen:
                seen.add((r["Dispatch_Id"], short)); per["rasterizer/" + short][0] += 1
    return per


def main():
    fetch_csv, write_csv, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    scenes = int(sys.argv[5]) if len(sys.argv) > 5 else 24
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    kernels = {}
    for fam in sorted(set(f) | set(w)):
        n = max(f[fam][0], w[fam][0], 1)
        fb, wb = f[fam][1] * 1024.0, w[fam][1] * 1024.0
        kernels[fam] = dict(launches=n, fetch_kib_raw=f[fam][1], write_kib_raw=w[fam][1], fetch_bytes_corrected_per_launch=int(fb / n),
                            write_bytes_per_launch=int(wb / n), hbm_bytes_per_launch=int((fb + wb) / n),
                            hbm_bytes_per_step=int((fb + wb) / steps))
    json.dump(dict(command="rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python bench.py "
                           "--mode infer --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32", steps=steps,
                   units="KiB as reported; fetch corrected x2 for gfx950 (MI355X_MICROARCH.md, HBM section) except render_kernel (16-byte gathers: x1, calibrated); write as reported",
                   kernels=kernels, workload=dict(scenes_per_gpu=scenes, context_views=8, target_views=12)), open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:12s} launches {v['launches']:6d}  HBM/launch {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB   HBM/step {v['hbm_bytes_per_step'] / 1e9:8.2f} GB")


if __name__ == "__main__":
    main()
