"""Aggregate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs) of bench.py into profiles/round1_pmc_traffic.json.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps_executed> <out.json> [scenes_per_gpu=24]
(bench.py only uses the file when its "workload" block matches the workload being benchmarked: the default 24 scenes x 8 views x 12 targets)
HBM bytes per kernel family, gfx950 corrections as MI355X_MICROARCH.md prescribes: rocprofv3 reports FETCH_SIZE / WRITE_SIZE
in KiB; FETCH_SIZE counts a wide coalesced read at half its size on gfx950 (x2); WRITE_SIZE is taken as reported.
"""
import csv, json, sys, collections

FAMILIES = [("gemm", ("gemm256_kernel", "gemm_kernel", "gemm_smallm_kernel")), ("conv3x3", ("conv3x3_256_kernel", "conv3x3_kernel")),
            ("attention", ("attention_kernel", "attention_res_kernel")), ("rasterizer", ("preprocess_kernel", "scatter_kernel", "tile_scan_kernel", "tile_sort_kernel",
                                                                  "segment_sort_kernel", "render_kernel")),
            ("layernorm", ("layernorm_mod_kernel",)), ("upsample", ("upsample2x_kernel",)), ("adapter", ("adapter_",))]


def family(name):
    for fam, pats in FAMILIES:
        if any(p in name for p in pats):
            return fam
    return "other"


def load(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])  # family -> [dispatches, KiB]
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        fam = family(r["Kernel_Name"])
        key = (r["Dispatch_Id"], fam)
        if key not in seen:
            seen.add(key); per[fam][0] += 1
        per[fam][1] += float(r["Counter_Value"])
    return per


def main():
    fetch_csv, write_csv, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    scenes = int(sys.argv[5]) if len(sys.argv) > 5 else 24
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    kernels = {}
    for fam in sorted(set(f) | set(w)):
        n = max(f[fam][0], w[fam][0], 1)
        fb, wb = 2.0 * f[fam][1] * 1024.0, w[fam][1] * 1024.0
        kernels[fam] = dict(launches=n, fetch_kib_raw=f[fam][1], write_kib_raw=w[fam][1], fetch_bytes_corrected_per_launch=int(fb / n),
                            write_bytes_per_launch=int(wb / n), hbm_bytes_per_launch=int((fb + wb) / n),
                            hbm_bytes_per_step=int((fb + wb) / steps))
    json.dump(dict(command="rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python bench.py "
                           "--steps 1 --warmup 1 --no-cpu-baseline --no-roofline", steps=steps,
                   units="KiB as reported; fetch corrected x2 for gfx950 (MI355X_MICROARCH.md, HBM section); write as reported",
                   kernels=kernels, workload=dict(scenes_per_gpu=scenes, context_views=8, target_views=12)), open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:12s} launches {v['launches']:6d}  HBM/launch {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB   HBM/step {v['hbm_bytes_per_step'] / 1e9:8.2f} GB")


if __name__ == "__main__":
    main()
