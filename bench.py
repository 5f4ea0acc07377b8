"""bench.py -- headline benchmark of the VicaSplat hot path on MI355X (BASELINE.json metric):

    scenes/sec, 8-view 256x256 scenes: ViT-L encoder + video/camera decoder + DPT heads + Gaussian adapter
    (-> 524 288 Gaussians + 7 poses per scene) followed by rasterization of 12 target views per scene
    (re10k_8view: num_context_views 8, num_target_views 12, config/experiment/re10k_8view.yaml:19-20).

A "step" = one pass of that path over one batch of `--scenes-per-gpu` synthetic scenes already resident in HBM.
`python bench.py --gpus N --steps K --warmup W`.  N ranks, one per GPU, over RCCL: under a launcher (`torch.distributed.run ... bench.py
--gpus N`, WORLD_SIZE set) this process IS one of the ranks; started plainly (`python bench.py --gpus N`, no WORLD_SIZE) it launches the N
ranks itself -- like a plain `python -m src.main` under Lightning (src/main.py:104-116) -- and every rank asserts world size == --gpus.
Scenes are independent: they are sharded over ranks with NO data-path collective ("scaling": "weak").
Prints ONE JSON line (rank 0) with `roofline` (dominant hand-written kernel, measured with HIP events on the launch
stream in an extra instrumented step) and `cpu_baseline` (the CPU oracle timed on the host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_16BIT_TFLOPS = 2500.0  # dense bf16/f16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
DTYPE_NOTE = {"split": "f32 (f32 activations; products as 3 x f16 MFMA on hi+lo operand pairs, f32 accumulate)", "f16": "f16", "bf16": "bf16",
              "f32": "f32 (exact-f32 MFMA)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenes-per-gpu", type=int, default=24)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--targets", type=int, default=12)
    ap.add_argument("--dtype", default="split", choices=["split", "f16", "bf16", "f32"],
                    help="operand class of the headline: split (default) = f32 activations, f16 hi+lo operands, three MFMAs per product -- the "
                         "precision that meets the render tolerance; f16 / bf16 = the fast TF32-class path; f32 = exact-f32 MFMA")
    ap.add_argument("--mode", default="both", choices=["infer", "train", "both"],
                    help="infer: the headline forward metric only; both (default): also time the full training step (BASELINE config 4/5) "
                         "and report it in the `train` object of the same JSON line; train: `value` IS the training throughput")
    ap.add_argument("--train-scenes-per-gpu", type=int, default=24)    # README.md:104 / distill.yaml: 24 scenes per GPU
    ap.add_argument("--train-steps", type=int, default=3)
    ap.add_argument("--train-split-scenes", type=int, default=8, help="scenes per GPU of the reference-precision (split class) training leg: f32 "
                    "activations double the footprint of the 16-bit step (24 scenes: 158 GB), so the secondary leg runs a third of the batch")
    ap.add_argument("--no-train-split", action="store_true", help="skip the split-class training leg")
    ap.add_argument("--no-train-split24", action="store_true", help="skip the 24-scene checkpointed split-class training leg")
    ap.add_argument("--train-timeout", type=float, default=240.0, help="seconds after which a stalled training leg is abandoned and the headline line is printed without it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the exact-f32 MFMA leg")
    ap.add_argument("--no-fast", action="store_true", help="skip the 16-bit (f16) fast-path leg")
    ap.add_argument("--no-targets70", action="store_true", help="skip the 70-view demo workload leg")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-scene (B = 1) latency leg")
    ap.add_argument("--latency-iters", type=int, default=25)
    ap.add_argument("--dry-run-dist", action="store_true",
                    help="no GPU work: run the N > 1 control flow of this file (process-group init from the launcher's environment, barriers, MAX-reduce of "
                         "the elapsed time, the training leg's watchdog and one GradReducer exchange) on CPU with the gloo backend and print the JSON line")
    ap.add_argument("--leg-steps", type=int, default=10, help="timed steps of the secondary precision legs")
    return ap.parse_args()


def _auto_blocks(B, dt, dev):
    from vicasplat_amd.model.encoder.train_forward import auto_checkpoint_blocks
    return auto_checkpoint_blocks(B, 1.0, torch.cuda.get_device_properties(dev).total_memory / 2 ** 30, 24, 12, half=dt != "split")


def target_cameras(B, Vt, dev):
    """Vt target cameras per scene in frame-0 coordinates: identity rotation, x translation j*0.05."""
    E = torch.eye(4, device=dev).repeat(B, Vt, 1, 1)
    E[:, :, 0, 3] = (torch.arange(Vt, device=dev) * 0.05)[None]
    K = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]], device=dev).repeat(B, Vt, 1, 1)
    near = torch.full((B, Vt), 0.01, device=dev)
    far = torch.full((B, Vt), 100.0, device=dev)
    return E, K, near, far


class KernelTimer:
    """Wraps the HIP op front-ends with event pairs recorded on the launch stream (torch's current stream)."""

    def __init__(self):
        self.rec = []

    def wrap(self, mod, name, meta_fn):
        orig = getattr(mod, name)

        def f(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*a, **k)
            e.record()
            self.rec.append((name, s, e, meta_fn(r, *a, **k)))
            return r

        setattr(mod, name, f)
        return orig

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, s, e, meta in self.rec:
            d = out.setdefault(name, dict(ms=0.0, calls=0, flops=0.0, bytes=0.0))
            d["ms"] += s.elapsed_time(e); d["calls"] += 1
            d["flops"] += meta.get("flops", 0.0); d["bytes"] += meta.get("bytes", 0.0)
        return out


def raster_roofline(rs, traffic, design_min, R, gaussians, views, pmc_file):
    """The rasterizer against the rooflines that bind it.  HBM: `achieved` = the bytes the PMC counters saw per forward
    (FETCH_SIZE + WRITE_SIZE of its six kernels, committed pass of the same workload) / the live event-timed duration; `frac` = that / 8 TB/s.
    Without a counter file for this workload: the bytes this DESIGN must move (`design_min_*`: attributes once per scene).  SURVEY 8(d)'s
    per-view formula (P*280 + R*68 + 1.8 MB per view: counts the P*232 input bytes once per VIEW, 12x per scene -- bytes this design does
    not move) is kept as a footnote only.  VALU: `preprocess_kernel` and `render_kernel` are issue-bound, not HBM-bound: their executed
    VALU wave-instructions (SQ_INSTS_VALU, committed PMC pass) x 2 cycles (a wave64 instruction occupies its SIMD-32 for two cycles,
    MI355X_MICROARCH.md) / (1024 SIMDs x 2.4 GHz x the kernel's traced duration)."""
    sec = rs["ms"] * 1e-3
    byt = traffic if traffic else design_min
    d = dict(bound="hbm", achieved=round(byt / sec / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(byt / sec / 1e9 / PEAK_HBM_GBS, 4),
             traffic=traffic, bytes_basis=("PMC counters (FETCH_SIZE x2 gfx950 correction except the gather kernel, + WRITE_SIZE), per forward"
                                           if traffic else "design minimum (no counter file for this workload)"),
             design_min_bytes=int(design_min), design_min_achieved=round(design_min / sec / 1e9, 1),
             design_min_frac=round(design_min / sec / 1e9 / PEAK_HBM_GBS, 4),
             traffic_over_design_min=(round(traffic / design_min, 3) if traffic else None),
             num_rendered=int(R), gaussians=gaussians, views=views, ms=round(rs["ms"], 3),
             footnote_survey_8d_formula=dict(algorithmic_bytes=int(rs["bytes"]), achieved=round(rs["bytes"] / sec / 1e9, 1),
                                             frac=round(rs["bytes"] / sec / 1e9 / PEAK_HBM_GBS, 4),
                                             note="P*280 + R*68 + 1.8 MB per rendered VIEW; not the bytes this design moves"))
    try:
        pm = json.load(open(pmc_file))
        valu = {}
        for k, v in pm.get("raster_valu", {}).items():
            peak_rate = 1024 * 2.4e9 / 2.0                         # wave-instructions per second, all SIMDs
            valu[k] = dict(valu_insts_per_launch=int(v["valu_insts_per_launch"]), traced_us=round(v["traced_us"], 1),
                           frac=round(v["valu_insts_per_launch"] / (v["traced_us"] * 1e-6) / peak_rate, 4),
                           insts_per_unit=round(v["valu_insts_per_launch"] * 64.0 / v["units_per_launch"], 1), unit=v["unit"])
        d["valu_issue"] = dict(bound="valu-issue", peak="1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction", kernels=valu) if valu else None
    except Exception:
        d["valu_issue"] = None
    return d


def raster_backward_roofline(step, B, V, Vt):
    """the rasterizer BACKWARD (cuda_splatting.py:226-235 under autograd) against the HBM roofline, measured inside one
    extra (untimed-for-the-headline) training step: vs_raster_backward bracketed with events on the launch stream (its three parts: the zero
    fill of the per-(camera, Gaussian) gradient records, the render replay, the per-Gaussian reduction), the forward beside it.  `achieved`
    = counter bytes (FETCH_SIZE x2 gfx950 correction except the gather kernel + WRITE_SIZE of those kernels, separate PMC passes of
    tools/raster_fb_prof.sh on the same scene family, committed as profiles/round6_pmc_raster_bwd.json) / the live duration; the bytes this
    design must move at minimum are stated beside it: list + record gathers R * 52, the pixel data, the gradient records written once and
    read once (views * P * 40 each way), radii / clamp masks views * P * 5, the scene's attributes in and gradients out S * P * (232 + 352)."""
    from vicasplat_amd import raster
    rec = {}

    def wrap(name):
        orig = getattr(raster, name)

        def f(*a, **k):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(); r = orig(*a, **k); e_.record()
            rec.setdefault(name, []).append((s_, e_, r if name == "_forward_impl" else a[1]))
            return r

        setattr(raster, name, f)
        return orig

    o1, o2 = wrap("_forward_impl"), wrap("_backward_impl")
    try:
        step()
        torch.cuda.synchronize()
    finally:
        raster._forward_impl, raster._backward_impl = o1, o2
    fwd_ms = sum(s_.elapsed_time(e_) for s_, e_, _ in rec["_forward_impl"])
    bwd_ms = sum(s_.elapsed_time(e_) for s_, e_, _ in rec["_backward_impl"])
    st = rec["_forward_impl"][0][2][1]
    S, Pn, Cn, M, H, W, _ = st["dims"]
    R = sum(x[2][1]["num_rendered"] for x in rec["_forward_impl"])
    design_min = R * 52.0 + Cn * H * W * 32.0 + 2.0 * Cn * Pn * 40.0 + Cn * Pn * 5.0 + S * Pn * (232.0 + 352.0)
    d = dict(bound="hbm", peak=PEAK_HBM_GBS, unit="GB/s", ms=round(bwd_ms, 3), forward_ms=round(fwd_ms, 3), views=Cn, gaussians=S * Pn, num_rendered=int(R),
             ms_per_288_views=round(bwd_ms * 288.0 / Cn, 2), design_min_bytes=int(design_min),
             design_min_achieved=round(design_min / (bwd_ms * 1e-3) / 1e9, 1), design_min_frac=round(design_min / (bwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
             kernels="hipMemset of the gradient records + render_backward_seg_kernel (segment-parallel replay from the forward's checkpoints) + preprocess_backward_kernel")
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "round6_pmc_raster_bwd.json")))
        per_view = pm["backward"]["hbm_bytes_per_view"]
        traffic = per_view * Cn
        d["traffic_basis"] = pm["backward"]["basis"]
        d["valu_issue"] = pm.get("valu_issue")
    except Exception:
        pass
    byt = traffic if traffic else design_min
    d.update(achieved=round(byt / (bwd_ms * 1e-3) / 1e9, 1), frac=round(byt / (bwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), traffic=(int(traffic) if traffic else None),
             traffic_over_design_min=(round(traffic / design_min, 3) if traffic else None))
    return d


def train_leg(args, enc, dec, dev, rank, world, dist, cdt=None, scenes=None, steps=None, checkpoint=False):
    """BASELINE configs 4 / 5: the full training step -- encoder + decoder + rasterizer forward, MSE, backward on the HIP kernels,
    gradient exchange (GradReducer: bucketed all-reduce over RCCL overlapped with backward; N > 1 only), clip 0.5, AdamW -- on
    `--train-scenes-per-gpu` 8-view scenes with `--targets` target views each (re10k_8view.yaml:19-20).  Timed like the headline:
    barrier + synchronize on both sides, max over ranks.  lr is 1e-12: random-init weights + a real learning rate throw the scene off
    screen within a few steps, which would change the rasterizer's work between timed steps; every kernel of the step still runs."""
    from vicasplat_amd import callers, synthetic
    from vicasplat_amd import dist as vdist
    B, V, Vt = (scenes or args.train_scenes_per_gpu), args.views, args.targets
    # operand class of the step: 16-bit (the fast class) or "split" -- the reference's precision in BOTH directions (f32 activations and
    # gradients, three f16 MFMAs per product; round 3)
    dt = cdt if cdt is not None else (torch.bfloat16 if args.dtype == "bf16" else torch.float16)
    nsteps = steps or args.train_steps
    peak_tf = PEAK_MFMA_16BIT_TFLOPS / 3.0 if dt == "split" else PEAK_MFMA_16BIT_TFLOPS
    img, K = synthetic.synthetic_input(B, V, 256, seed=100 + rank)
    tE, tK, tn, tf = target_cameras(B, Vt, dev)
    gen = torch.Generator().manual_seed(7 + rank)
    target = torch.rand(B, Vt, 3, 256, 256, generator=gen).to(dev)
    batch = dict(context=dict(image=img.to(dev), intrinsics=K.to(dev)), target=dict(image=target, extrinsics=tE, intrinsics=tK, near=tn, far=tf))
    enc.train().requires_grad_(True)
    enc.backbone.gradient_checkpointing = bool(checkpoint)      # enable_gradient_checkpointing(): per-block recomputation (backbone_vica.py:464-516)
    enc.backbone.checkpoint_blocks = "auto" if checkpoint == "auto" else None      # "auto": only as many blocks as 288 GB require (train_forward.auto_checkpoint_blocks)
    opt, _ = callers.configure_optimizer(enc, lr=1e-12)
    reducer = vdist.GradReducer(enc.parameters()) if world > 1 else None
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2):      # warm-up: optimizer state, then the caching allocator's pool in its steady shape (one step is not enough: the second step of
        # a leg can still grow the pool by tens of GB of fresh hipMallocs -- seen once as a 2.2 s "step" in the two-step split leg)
        r = callers.training_step(enc, dec, batch, opt, compute_dtype=dt, reducer=reducer)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        r = callers.training_step(enc, dec, batch, opt, compute_dtype=dt, reducer=reducer)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms = el / nsteps * 1e3
    # (every rank runs the instrumented step: under N > 1 it contains the gradient exchange's collectives; rank 0's reading is reported)
    try:
        rb = raster_backward_roofline(lambda: callers.training_step(enc, dec, batch, opt, compute_dtype=dt, reducer=reducer), B, V, Vt)
    except Exception as ex:      # an instrumentation failure must not take the leg with it
        rb = dict(error=repr(ex)[:200])
    # algorithmic FLOPs of the step: 3 407 GFLOP forward per 8-view scene, backward = 2x forward
    flops = 3.0 * 3407e9 * (V / 8.0) * B
    tf_s = flops / (ms * 1e-3) / 1e12
    enc.eval().requires_grad_(False)
    enc.backbone.gradient_checkpointing = False
    enc.backbone.checkpoint_blocks = None
    if reducer is not None:
        reducer.remove()
    for p in enc.parameters():
        p.grad = None
    return dict(metric="scenes/sec training step (fwd+bwd+clip+AdamW)", value=round(world * B / (ms * 1e-3), 3), unit="scenes/s",
                ms_per_step=round(ms, 2), steps=nsteps, scenes_per_gpu=B, context_views=V, target_views=Vt,
                dtype=("split (f32 activations and gradients, 3 x f16 MFMA per product, forward and backward)" if dt == "split" else "bf16" if dt == torch.bfloat16 else "f16"),
                loss_scale=float(r["loss_scale"]), gradient_checkpointing=("auto: the first %d encoder / %d decoder blocks" % _auto_blocks(B, dt, dev) if checkpoint == "auto" else bool(checkpoint)),
                loss=float(r["loss"]), grad_norm=float(r["grad_norm"]), skipped=bool(r["skipped"]),
                peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 1),
                gradient_exchange=("none (1 GPU)" if world == 1 else f"GradReducer: 64 MiB buckets all-reduced during backward, RCCL x{world}"),
                roofline_rasterizer_backward=rb,
                roofline=dict(bound="mfma", achieved=round(tf_s, 1), peak=round(peak_tf, 1), unit="TFLOP/s",
                              frac=round(tf_s / peak_tf, 4),
                              what="whole step: 3 x 3407 GFLOP per 8-view scene (fwd + 2x bwd, SURVEY 8d) / step time"))


def dry_run_dist(args):
    """The multi-rank control flow of main() / train_leg() with the GPU work replaced by a toy computation: what the driver's
    `torch.distributed.run ... bench.py --gpus N` exercises before any kernel matters.  CPU, gloo, rendezvous on 127.0.0.1."""
    import threading

    import torch.distributed as dist
    from vicasplat_amd import dist as vdist
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
    nseen = 1
    if world > 1:
        ones = torch.ones(1)
        dist.all_reduce(ones)
        nseen = int(ones.item())
        assert nseen == args.gpus
    B = args.scenes_per_gpu
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(B, 32, 32, generator=g)                         # this rank's scene shard: scenes are sharded, no data-path collective

    def step():
        return (x @ x.transpose(1, 2)).sum()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # training leg: watchdog + overlapped gradient exchange on a toy replica
    fired = []
    dog = threading.Timer(args.train_timeout, lambda: fired.append(1))
    dog.daemon = True
    dog.start()
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.GELU(), torch.nn.Linear(64, 8))
    reducer = vdist.GradReducer(model.parameters(), bucket_bytes=4096) if world > 1 else None
    ncoll = 0
    for _ in range(2):
        if reducer is not None:
            reducer.zero_grad()
        else:
            model.zero_grad()
        model(x.mean(1)).square().mean().backward()
        if reducer is not None:
            ncoll = reducer.finish()
    chk = torch.stack([p.grad.double().sum() for p in model.parameters()])
    same = True
    if world > 1:
        all_chk = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(all_chk, chk)
        same = all(torch.equal(all_chk[0], c) for c in all_chk)
    dog.cancel()
    if rank == 0:
        print(json.dumps(dict(metric="scenes/sec (8-view 256x256) encode+rasterize", value=round(world * B * args.steps / max(elapsed, 1e-9), 3), unit="scenes/s",
                              n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True,
                              scaling="weak", vs_baseline=None, dtype="dry run (CPU toy computation)", data="synthetic", dry_run=True,
                              distributed=dict(backend="gloo" if world > 1 else "none (1 rank)", ranks_seen=nseen),
                              config=dict(workload="control-flow dry run: no kernel of the hot path is executed", scenes_per_gpu=B,
                                          parallelism=f"scene-sharded x{world} (no collective)"),
                              train=dict(gradient_exchange=("none (1 rank)" if world == 1 else f"GradReducer: {ncoll} bucket all-reduces launched during backward, gloo x{world}"),
                                         replicas_identical=bool(same), watchdog_fired=bool(fired)))), flush=True)
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here (src/main.py:104-116: Lightning spawns one process per
    visible GPU from a plain `python -m src.main`).  Re-executes this file under torch.distributed.run on 127.0.0.1 with a free port; the
    children see WORLD_SIZE and take the rank path of main().  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what this pool's host driver supports (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def check_world(args, world):
    """--gpus is a contract, not a hint: a run whose rank count differs from it would print a line for a different machine."""
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; pass --gpus {world} or fix the launcher")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    check_world(args, int(os.environ.get("WORLD_SIZE", 1)))
    if args.dry_run_dist:
        return dry_run_dist(args)
    if args.mode == "train":      # only the training leg is of interest: keep the forward part to one untimed-quality pass
        args.steps, args.warmup, args.no_roofline, args.no_cpu_baseline = 1, 0, True, True
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    # what the collective library itself saw: one all-reduce of ones over RCCL = the rank count, reported in the line beside n_gpus
    dist_info = dict(backend="none (1 rank)", ranks_seen=1)
    if dist is not None:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ver = torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else None
        dist_info = dict(backend="nccl (RCCL)", rccl_version=".".join(str(v) for v in ver) if ver else None, ranks_seen=int(ones.item()),
                         launcher=("torch.distributed.run (external)" if os.environ.get("TORCHELASTIC_RUN_ID") else "environment"))
        assert dist_info["ranks_seen"] == args.gpus, f"RCCL counted {dist_info['ranks_seen']} ranks, --gpus says {args.gpus}"

    import json as _json
    from vicasplat_amd import ops, raster, synthetic
    from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    from vicasplat_amd.model.types import Gaussians

    DT = {"split": "split", "f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
    dt = DT[args.dtype]
    # executed MFMA FLOPs per algorithmic FLOP and the matrix peak they run against: split operands issue three 16-bit MFMAs per product
    mfma_mult = 3.0 if args.dtype == "split" else 1.0
    peak_tf = 157.3 if args.dtype == "f32" else PEAK_MFMA_16BIT_TFLOPS
    B, V, Vt = args.scenes_per_gpu, args.views, args.targets
    shapes = _json.load(open(os.path.join(ROOT, "tests", "golden", "shapes_full.json")))
    W = synthetic.golden_weights(shapes, seed=0)
    enc, _ = get_encoder(default_cfg())
    enc.load_state_dict(W, strict=True)
    enc = enc.to(dev).eval().requires_grad_(False)     # inference legs: frozen weights -> the fused no-grad path (train_leg re-enables)
    enc.set_compute_dtype(dt)
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(dev)
    img, K = synthetic.synthetic_input(B, V, 256, seed=rank)
    ctx = dict(image=img.to(dev), intrinsics=K.to(dev))
    tE, tK, tnear, tfar = target_cameras(B, Vt, dev)

    # The rasterizer's exact mode copies its instance count back to size the sort buffers (one host synchronisation per batched call;
    # upstream does it per view).  The bench step runs the CAPACITY mode instead: one untimed exact pass measures the
    # instance count, the timed steps use buffers 1.25x that size and never touch the host; the device-side overflow flags of every timed
    # call are checked after the timed region (an overflowed call would have rendered background only: the run is then invalid).
    cap = {"n": None, "flags": []}

    def step():
        out = enc(ctx, compute_viewspace_depth=False)
        g = out["gaussians"]
        gs = Gaussians(g.means, g.covariances, g.harmonics, g.opacities)
        with raster.instance_capacity(cap["n"]) as scope:
            r = dec(gs, tE, tK, tnear, tfar, (256, 256))
        if cap["n"] is None:
            cap["n"] = int(max(n for n, _ in scope.calls) * 1.25) + 65536
        else:
            cap["flags"].append(scope.overflow_flag())
        return out, r

    step()                                   # untimed: exact mode, sizes the capacity
    for _ in range(max(0, args.warmup - 1)):
        step()
    cap["flags"].clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    if cap["flags"] and bool(torch.stack(cap["flags"]).max().item() != 0):
        raise RuntimeError("a timed rasterizer call outgrew its instance capacity (it rendered background only): the measurement is invalid")

    roofline, extra = None, {}
    if rank == 0 and not args.no_roofline:
        kt = KernelTimer()
        P = V * 256 * 256

        def gemm_meta(r, a, w, *rest, **k):
            M = k.get("M") or a.shape[0]
            return dict(flops=2.0 * M * w.shape[0] * a.shape[1])

        def attn_meta(r, q, k_, v, out, **k):
            nb, H, Lq = k["nbatch"], k["H"], k["Lq"]
            Lk = k.get("Lk") or 0
            if k.get("kv_seg") is not None:
                Lk = 2 * Lq if V > 2 else Lq
            return dict(flops=4.0 * nb * H * Lq * Lk * 64)

        def raster_meta(r, *a, **k):
            st = r[1]
            S, Pn, Cn, M, H, Wd, _ = st["dims"]
            # P*280 + R*68 + 1.8 MB per rendered VIEW -- counts the P*232 input bytes once per view.  This design reads them
            # once per SCENE (preprocess_kernel loops over the scene's cameras with the attributes in registers), so the bytes it
            # must move at minimum are S*P*232 + views*(P*48 + 1.8 MB) + R*68: reported beside the 8(d) figure.
            R_ = st["num_rendered"]
            return dict(bytes=Cn * (Pn * 280.0 + 1.8e6) + R_ * 68.0, R=R_, design_min=S * Pn * 232.0 + Cn * (Pn * 48.0 + 1.8e6) + R_ * 68.0)

        def conv_meta(r, x, w, *a, **k):
            return dict(flops=2.0 * r.numel() * w.shape[1] * w.shape[2] * w.shape[3])

        wrapped = [(ops, "gemm", gemm_meta), (ops, "gemm_qkv_rope", gemm_meta), (ops, "attention", attn_meta),
                   (ops, "layernorm_mod", lambda r, x, *a, **k: dict(bytes=x.numel() * 6.0)),
                   (ops, "conv3x3_nhwc", conv_meta),
                   (ops, "conv3x3_head1x1_nhwc", lambda r, x, w, b, w2, b2, n_out, *a, **k: dict(
                       flops=2.0 * x.shape[0] * x.shape[1] * x.shape[2] * (9.0 * w.shape[3] * w.shape[0] + w.shape[0] * n_out))),
                   (ops, "conv7x7_rgb_nhwc", lambda r, x, w, *a, **k: dict(flops=2.0 * r.numel() * 147)),
                   (ops, "stem7x7_up_split_stream", lambda r, x, w, *a, **k: dict(flops=2.0 * r.data.numel() * 147)),      # (round 5: the streaming stem)
                   (ops, "upsample2x_nhwc", lambda r, x, *a, **k: dict(bytes=r.numel() * 2.0 * (2.25 if (len(a) or k.get("add") is not None) else 1.25))),
                   (ops, "gaussian_adapter", lambda r, *a, **k: dict()),
                   (raster, "_forward_impl", raster_meta)]
        saved = [(m, n, kt.wrap(m, n, f)) for m, n, f in wrapped]
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cap["n"] = None      # the instrumented step runs the rasterizer's exact mode: its state carries the true instance count R
        s.record(); out, r = step(); e.record()
        summ = kt.summary()
        for m, n, o in saved:
            setattr(m, n, o)
        zero = dict(ms=0.0, calls=0, flops=0.0, bytes=0.0)
        for k_ in ("gemm", "gemm_qkv_rope", "conv3x3_nhwc", "conv3x3_head1x1_nhwc", "conv7x7_rgb_nhwc", "stem7x7_up_split_stream", "upsample2x_nhwc",
                   "gaussian_adapter"):
            summ.setdefault(k_, dict(zero))
        for k_ in ("ms", "calls", "flops"):      # the 7x7 stem: tile route or streaming kernel, one family
            summ["conv7x7_rgb_nhwc"][k_] = summ["conv7x7_rgb_nhwc"].get(k_, 0.0) + summ["stem7x7_up_split_stream"].get(k_, 0.0)
        tot_ms = s.elapsed_time(e)
        # every vs_gemm_* launch: plain/fused-epilogue GEMMs and the qkv projections with RoPE in the epilogue
        gm = {k_: summ["gemm"][k_] + summ["gemm_qkv_rope"][k_] for k_ in ("ms", "calls", "flops")}
        at, rs, ln = summ["attention"], summ["_forward_impl"], summ["layernorm_mod"]
        # 3x3 convolutions: the plain implicit-GEMM launches + the two fused conv3 -> ReLU -> conv1 head kernels
        cv = {k_: summ["conv3x3_nhwc"][k_] + summ["conv3x3_head1x1_nhwc"][k_] for k_ in ("ms", "calls", "flops", "bytes")}
        up, ad, stem = summ["upsample2x_nhwc"], summ["gaussian_adapter"], summ["conv7x7_rgb_nhwc"]
        R = [m for n, _, _, m in kt.rec if n == "_forward_impl"][0]["R"]
        raster_min = sum(m["design_min"] for n, _, _, m in kt.rec if n == "_forward_impl")
        gemm_tf = gm["flops"] / (gm["ms"] * 1e-3) / 1e12
        # dominant hand-written kernel family of the step: the MFMA GEMM (ViT encoder/decoder linears, 1x1 convolutions)
        traffic = {}
        pmc_file = None
        try:  # HBM bytes per launch from the committed PMC passes (profiles/README.md); null unless they cover this workload and operand class
            fn = {"split": "round6_pmc_traffic.json", "f16": "round2_pmc_traffic.json"}.get(args.dtype)
            for older in ("round5_pmc_traffic.json", "round4_pmc_traffic.json", "round3_pmc_traffic.json"):
                if fn and fn.startswith("round") and args.dtype == "split" and not os.path.exists(os.path.join(ROOT, "profiles", fn)):
                    fn = older
            pmc_file = os.path.join(ROOT, "profiles", fn)
            pm = json.load(open(pmc_file))
            if pm.get("workload", {}).get("scenes_per_gpu") == B and V == 8 and Vt == 12 and pm["workload"].get("dtype", "f16") == args.dtype:
                traffic = {k: v["hbm_bytes_per_launch"] for k, v in pm["kernels"].items()}
                traffic["rasterizer"] = pm["kernels"]["rasterizer"]["hbm_bytes_per_step"]  # one forward = 6 kernels
        except Exception:
            pass
        # `achieved` = ALGORITHMIC FLOPs (2 M N K per product) / time; split operands execute three 16-bit MFMAs per product, so the
        # matrix pipes see mfma_mult x that: `frac` is the MFMA utilisation (executed MFMA FLOPs / dense 16-bit peak), `frac_algorithmic`
        # prices the algorithmic rate against peak / mfma_mult -- the same number by construction, reported so that neither is hidden.
        kname = {"split": "gemm256_kernel/gemm_kernel<split: f16 hi+lo, 3 MFMA per product> (vs_gemm_split)", "f32": "gemm256_kernel/gemm_kernel<f32> (exact-f32 MFMA)"}.get(
            args.dtype, f"gemm256_kernel/gemm_kernel<{args.dtype}> (vs_gemm_bias_act, vs_gemm_qkv_rope)")
        roofline = dict(kernel=kname, bound="mfma",
                        achieved=round(gemm_tf, 1), peak=round(peak_tf / mfma_mult, 1), unit="TFLOP/s",
                        frac=round(gemm_tf * mfma_mult / peak_tf, 4), traffic=traffic.get("gemm"),
                        executed_mfma_tflops=round(gemm_tf * mfma_mult, 1), mfma_peak=peak_tf, mfma_per_product=mfma_mult,
                        # the other reading (VERDICT r3 "weak 7"): the ALGORITHMIC rate against the hardware's dense 16-bit peak itself --
                        # `peak` above (833 TF/s for the split class) is that figure divided by the three MFMAs a product costs, not a
                        # hardware number
                        frac_algorithmic_vs_f16_peak=round(gemm_tf / PEAK_MFMA_16BIT_TFLOPS, 4),
                        launches=gm["calls"], avg_launch_us=round(gm["ms"] * 1e3 / gm["calls"], 2))
        if args.dtype != "f32":
            # what the matrix pipe SUSTAINS on this box: back-to-back 16-bit MFMAs on register operands with random data, no memory traffic
            # (vs_probe_mfma_rate, ~40 ms).  The chip is power-limited under matrix load (DVFS), so this -- not the 2.4 GHz headline -- is the
            # ceiling an MFMA kernel can approach here; `frac` above stays priced against the headline peak.
            try:
                sus = ops.sustained_mfma_tflops(dev)
                roofline["sustained_mfma_tflops"] = round(sus, 1)
                roofline["frac_of_sustained"] = round(gemm_tf * mfma_mult / sus, 4)
            except Exception as ex:
                roofline["sustained_mfma_tflops"] = None
                roofline["sustained_error"] = repr(ex)[:200]
        known = gm["ms"] + at["ms"] + ln["ms"] + rs["ms"] + cv["ms"] + up["ms"] + ad["ms"] + stem["ms"]
        mfma_flops = gm["flops"] + at["flops"] + cv["flops"] + stem["flops"]
        mfma_ms = gm["ms"] + at["ms"] + cv["ms"] + stem["ms"]
        extra = dict(
            step_breakdown_ms=dict(total=round(tot_ms, 3), gemm=round(gm["ms"], 3), conv3x3=round(cv["ms"], 3),
                                   conv7x7_stem=round(stem["ms"], 3), attention=round(at["ms"], 3), layernorm=round(ln["ms"], 3),
                                   upsample=round(up["ms"], 3), adapter=round(ad["ms"], 3), rasterizer=round(rs["ms"], 3),
                                   other_glue=round(tot_ms - known, 3)),
            roofline_conv=dict(bound="mfma", achieved=round(cv["flops"] / (cv["ms"] * 1e-3) / 1e12, 1), peak=round(peak_tf / mfma_mult, 1),
                               unit="TFLOP/s", frac=round(cv["flops"] / (cv["ms"] * 1e-3) / 1e12 * mfma_mult / peak_tf, 4)),
            roofline_attention=dict(bound="mfma", achieved=round(at["flops"] / (at["ms"] * 1e-3) / 1e12, 1), peak=round(peak_tf / mfma_mult, 1), unit="TFLOP/s",
                                    frac=round(at["flops"] / (at["ms"] * 1e-3) / 1e12 * mfma_mult / peak_tf, 4)),
            roofline_rasterizer=raster_roofline(rs, traffic.get("rasterizer"), raster_min, R, P * B, B * Vt, pmc_file),
            mfma_util_step=round(mfma_flops / (mfma_ms * 1e-3) / 1e12 * mfma_mult / peak_tf, 4))

    cpu_baseline = None
    psnr_vs_oracle = None
    psnr_ctx = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline leg: the oracle (a restatement of the reference, "port"), timed on the host cores ----
        from oracle import encoder_ref as er
        from oracle import raster_ref as rr
        import numpy as np
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        ncores = max(1, min(avail, 32))  # more threads than that only add sync overhead on this problem size
        torch.set_num_threads(ncores)
        cfg = er.default_cfg()
        # bounded sample (target: 10-30 s of CPU work): ONE 2-view scene through the full ViT-L encoder first; when that is
        # quick enough, the full 8-view scene as well, so the encoder leg needs no extrapolation
        Vs = 2
        t1 = time.perf_counter()
        o = er.forward(W, cfg, img[:1, :Vs], K[:1, :Vs])
        t_enc = time.perf_counter() - t1
        enc_scale = 3407.0 / 817.6  # encoder+decoder+heads GFLOP at 8 views / at 2 views
        if t_enc * enc_scale < 15.0:
            Vs = V
            t1 = time.perf_counter()
            o = er.forward(W, cfg, img[:1, :Vs], K[:1, :Vs])
            t_enc = time.perf_counter() - t1
            enc_scale = 1.0
        g = o["gaussians"]
        nv = 2 if Vs == 2 else 3  # views compared with the product chain below
        nvt = nv if Vs == 2 else Vt  # rasterized sample views: every target view of the scene, one host thread each
        sc = dict(means=g["means"].reshape(-1, 3).numpy(), covariances=g["covariances"].reshape(-1, 3, 3).numpy(),
                  harmonics=g["harmonics"].reshape(-1, 3, 25).numpy(), opacities=g["opacities"].reshape(-1).numpy(),
                  extrinsics=tE[0, :nvt].cpu().numpy(), intrinsics=tK[0, :nvt].cpu().numpy(), near=tnear[0, :nvt].cpu().numpy(),
                  far=tfar[0, :nvt].cpu().numpy())
        ras_threads = max(1, min(ncores, nvt))
        t1 = time.perf_counter()
        o_views = rr.render_views(sc, res=256, threads=ras_threads)
        t_ras_all = time.perf_counter() - t1          # wall time of the nvt views on ras_threads threads
        o_views = o_views[:nv]
        # ---- render PSNR vs the oracle chain (BASELINE.json metric, second half): the SAME scene and target cameras through the
        # product chain (HIP encoder -> HIP rasterizer, compute dtype of this run) against oracle encoder (f32) -> C rasterizer ----
        from oracle import chain
        h_out = enc(dict(image=img[:1, :Vs].to(dev), intrinsics=K[:1, :Vs].to(dev)), compute_viewspace_depth=False)
        hg = h_out["gaussians"]
        h_r = dec(Gaussians(hg.means, hg.covariances, hg.harmonics, hg.opacities), tE[:1, :nv], tK[:1, :nv], tnear[:1, :nv], tfar[:1, :nv],
                  (256, 256))
        torch.cuda.synchronize()
        cm = chain.compare_renders(h_r.color[0].cpu().numpy(), o_views)
        pose_err = float((h_out["gaussian_camera_extrins"][0].cpu() - o["gaussian_camera_extrins"][0]).abs().max())
        if Vs == V:
            psnr_ctx = (nv, o_views, o)
        psnr_vs_oracle = dict(value=round(min(cm["psnr_between"]), 2), unit="dB", per_view=[round(p, 2) for p in cm["psnr_between"]],
                              dpsnr_common_target=[float(f"{p:.2e}") for p in cm["dpsnr_common_target"]], pose_max_abs_err=float(f"{pose_err:.2e}"),
                              what=f"PSNR(HIP encoder[{args.dtype}] -> HIP rasterizer, oracle encoder[f32] -> C oracle rasterizer), scene 0, "
                                   f"{Vs} context views, first {nv} of {Vt} target views; min over views.  Yardstick: the reference's own CUDA "
                                   "precision (TF32 operands) emulated on the oracle scores 19-20 dB against the same f32 chain on this "
                                   "synthetic scene (tests/test_chain_cpu.py); the f32 path of this build: see f32_path.psnr_vs_oracle")
        # extrapolation to one bench scene: encoder by FLOPs when only 2 views were run, rasterizer ~ Gaussians x views
        est = t_enc * enc_scale + t_ras_all * (V / Vs) * (Vt / nvt)
        cpu_baseline = dict(value=round(1.0 / est, 5), unit="scenes/s", cores=ncores, kind="port",
                            sample=f"oracle (restated reference) on ONE {Vs}-view scene: encoder {t_enc:.1f}s f32 on {ncores} threads"
                                   f"{'' if enc_scale == 1.0 else f' (x{enc_scale:.2f} by FLOPs to 8 views)'}, rasterizer {t_ras_all:.2f}s for "
                                   f"{nvt} of {Vt} target views ({Vs * 65536 // 1000}k Gaussians) on {ras_threads} threads, one view per thread; "
                                   f"scene time = encoder + rasterizer")

    # ---- secondary precision legs on the SAME workload (N = 1 only, like the CPU leg): the 16-bit fast path (TF32-class: what the
    # reference's CUDA run computes with, but outside the 1e-4 dB render tolerance against an fp32 evaluation) and the exact-f32 MFMA
    # path, each timed for --leg-steps steps after one warm-up, with its render PSNR against the oracle chain computed above ----
    def precision_leg(name, peak, mult):
        try:
            enc.set_compute_dtype(DT[name])
            torch.cuda.empty_cache()
            cap["n"] = None
            step()                           # exact mode: sizes this precision's capacity
            step()
            cap["flags"].clear()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.leg_steps):
                o_, r_ = step()
            torch.cuda.synchronize()
            ms_ = (time.perf_counter() - t1) / args.leg_steps * 1e3
            if cap["flags"] and bool(torch.stack(cap["flags"]).max().item() != 0):
                raise RuntimeError("rasterizer capacity overflow in a timed step")
            tf_ = 3407e9 * (V / 8.0) * B / (ms_ * 1e-3) / 1e12
            leg = dict(metric="scenes/sec (8-view 256x256) encode+rasterize", value=round(B / (ms_ * 1e-3), 2), unit="scenes/s", dtype=name,
                       ms_per_step=round(ms_, 2), scenes_per_gpu=B, steps=args.leg_steps,
                       roofline=dict(bound="mfma", achieved=round(tf_, 1), peak=round(peak / mult, 1), unit="TFLOP/s", frac=round(tf_ * mult / peak, 4),
                                     what="whole step: 3407 GFLOP per 8-view scene (SURVEY 8d) / step time, against the MFMA peak of the operand class"))
            if psnr_ctx is not None:
                from oracle import chain
                nv_, o_views_, oo_ = psnr_ctx
                cm_ = chain.compare_renders(r_.color[0, :nv_].cpu().numpy(), o_views_)
                leg["psnr_vs_oracle"] = dict(value=round(min(cm_["psnr_between"]), 2), unit="dB", per_view=[round(p_, 2) for p_ in cm_["psnr_between"]],
                                             dpsnr_common_target=[float(f"{p_:.2e}") for p_ in cm_["dpsnr_common_target"]],
                                             pose_max_abs_err=float(f"{float((o_['gaussian_camera_extrins'][0].cpu() - oo_['gaussian_camera_extrins'][0]).abs().max()):.2e}"))
            return leg
        except Exception as e:      # a failing optional leg must not take the headline line with it
            return dict(error=repr(e)[:300])
        finally:
            enc.set_compute_dtype(dt)
            torch.cuda.empty_cache()

    # ---- the demo workload (demo.py:204-243, SURVEY 8d "report both"): 70 novel views per scene -- 10 cameras interpolated on each of the
    # 7 intervals between the 8 PREDICTED context cameras -- all sharing the scene's one Gaussian set (one batched rasterizer call, no
    # replication).  Same encoder pass, same precision as the headline; N = 1 only. ----
    def targets70_leg():
        try:
            from vicasplat_amd import callers
            t_ = torch.linspace(0, 1, 10, dtype=torch.float32, device=dev)
            n70 = (V - 1) * 10
            near70, far70 = torch.full((B, n70), 0.01, device=dev), torch.full((B, n70), 100.0, device=dev)
            cap70 = {"n": None, "flags": []}
            K_host = ctx["intrinsics"].float().cpu()

            def step70():
                out = enc(ctx, compute_viewspace_depth=False)
                # the camera path is host code in the demo too (a few hundred 4x4 matrices; on the device its float64 least-squares solves
                # cost more than the whole render): predicted poses -> host -> 70 cameras per scene -> device
                P_ = out["gaussian_camera_extrins"].cpu()
                tc = t_.cpu()
                E70 = callers.interpolate_extrinsics(P_[:, :-1].reshape(-1, 4, 4), P_[:, 1:].reshape(-1, 4, 4), tc).reshape(B, n70, 4, 4).float().to(dev)
                Kc = K_host
                K70 = callers.interpolate_intrinsics(Kc[:, :-1].reshape(-1, 3, 3), Kc[:, 1:].reshape(-1, 3, 3), tc).reshape(B, n70, 3, 3).float().to(dev)
                g = out["gaussians"]
                with raster.instance_capacity(cap70["n"]) as scope:
                    r = dec(Gaussians(g.means, g.covariances, g.harmonics, g.opacities), E70, K70, near70, far70, (256, 256))
                if cap70["n"] is None:
                    cap70["n"] = int(max(n for n, _ in scope.calls) * 1.25) + 65536
                    cap70["R"] = max(n for n, _ in scope.calls)
                else:
                    cap70["flags"].append(scope.overflow_flag())
                return r

            step70(); step70()
            cap70["flags"].clear()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_ = max(2, args.leg_steps // 2)
            for _ in range(n_):
                step70()
            torch.cuda.synchronize()
            ms_ = (time.perf_counter() - t1) / n_ * 1e3
            if cap70["flags"] and bool(torch.stack(cap70["flags"]).max().item() != 0):
                raise RuntimeError("rasterizer capacity overflow in a timed step")
            return dict(metric="scenes/sec (8-view 256x256) encode + 70 interpolated novel views", value=round(B / (ms_ * 1e-3), 2), unit="scenes/s",
                        dtype=DTYPE_NOTE[args.dtype], ms_per_step=round(ms_, 2), steps=n_, scenes_per_gpu=B, views_per_scene=n70,
                        rendered_views=B * n70, num_rendered=int(cap70["R"]),
                        workload="demo.py:204-243: per scene 70 cameras interpolated between the predicted context poses, one Gaussian set")
        except Exception as e:
            return dict(error=repr(e)[:300])
        finally:
            torch.cuda.empty_cache()

    targets70 = None
    if rank == 0 and world == 1 and args.mode != "train" and not args.no_targets70:
        targets70 = targets70_leg()

    # ---- single-scene latency: the reference's ONLY published number for this path is "~0.1 s" per scene end to end
    # (README.md:16), and its demo is a B = 1 caller (demo.py:180-243).  One 8-view scene: encoder + 12 target views, and encoder + the
    # demo's 70 interpolated views; every iteration is synchronised (a latency, not a throughput), median and p90 of `--latency-iters`. ----
    def latency_leg():
        try:
            from vicasplat_amd import callers
            import statistics
            ctx1 = dict(image=ctx["image"][:1].contiguous(), intrinsics=ctx["intrinsics"][:1].contiguous())
            E12, K12, n12, f12 = tE[:1], tK[:1], tnear[:1], tfar[:1]
            t_ = torch.linspace(0, 1, 10, dtype=torch.float32)
            n70 = (V - 1) * 10
            near70, far70 = torch.full((1, n70), 0.01, device=dev), torch.full((1, n70), 100.0, device=dev)
            K_host = ctx1["intrinsics"].float().cpu()
            caps = {"12": None, "70": None}

            def run12():
                out = enc(ctx1, compute_viewspace_depth=False)
                g = out["gaussians"]
                with raster.instance_capacity(caps["12"]) as scope:
                    r = dec(Gaussians(g.means, g.covariances, g.harmonics, g.opacities), E12, K12, n12, f12, (256, 256))
                if caps["12"] is None:
                    caps["12"] = int(max(n for n, _ in scope.calls) * 1.25) + 65536
                return r

            def run70():
                out = enc(ctx1, compute_viewspace_depth=False)
                P_ = out["gaussian_camera_extrins"].cpu()        # the demo's camera path is host code (demo.py:204-243): includes its device -> host round trip
                E70 = callers.interpolate_extrinsics(P_[:, :-1].reshape(-1, 4, 4), P_[:, 1:].reshape(-1, 4, 4), t_).reshape(1, n70, 4, 4).float().to(dev)
                K70 = callers.interpolate_intrinsics(K_host[:, :-1].reshape(-1, 3, 3), K_host[:, 1:].reshape(-1, 3, 3), t_).reshape(1, n70, 3, 3).float().to(dev)
                g = out["gaussians"]
                with raster.instance_capacity(caps["70"]) as scope:
                    r = dec(Gaussians(g.means, g.covariances, g.harmonics, g.opacities), E70, K70, near70, far70, (256, 256))
                if caps["70"] is None:
                    caps["70"] = int(max(n for n, _ in scope.calls) * 1.25) + 65536
                return r

            def enc_only():
                return enc(ctx1, compute_viewspace_depth=False)

            res = {}
            for name, fn in (("encoder_plus_12_views", run12), ("encoder_plus_70_views", run70), ("encoder_only", enc_only)):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(args.latency_iters):
                    t1 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1) * 1e3)
                ts.sort()
                res[name] = dict(median_ms=round(statistics.median(ts), 3), p90_ms=round(ts[int(0.9 * (len(ts) - 1))], 3), min_ms=round(ts[0], 3))
            return dict(metric="single-scene latency (B = 1, 8 context views 256x256), synchronised per iteration", unit="ms", iters=args.latency_iters,
                        dtype=DTYPE_NOTE[args.dtype], reference_published="~0.1 s per scene (README.md:16, other hardware, 2-view demo class)", **res)
        except Exception as e:
            return dict(error=repr(e)[:300])
        finally:
            torch.cuda.empty_cache()

    latency_b1 = None
    if rank == 0 and world == 1 and args.mode != "train" and not args.no_latency:
        latency_b1 = latency_leg()

    f32_path = fast_path = None
    if rank == 0 and world == 1 and args.mode != "train":
        if not args.no_fast and args.dtype not in ("f16", "bf16"):
            fast_path = precision_leg("f16", PEAK_MFMA_16BIT_TFLOPS, 1.0)
        if not args.no_f32 and args.dtype != "f32":
            f32_path = precision_leg("f32", 157.3, 1.0)

    def headline(train):
        return dict(metric="scenes/sec (8-view 256x256) encode+rasterize", value=round(value, 3), unit="scenes/s", n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype=DTYPE_NOTE[args.dtype], data="synthetic", distributed=dist_info,
                    config=dict(workload="re10k_8view full pipeline fwd: ViT-L encoder+decoder+DPT heads -> 524288 Gaussians/scene, "
                                         f"{Vt} target views/scene rasterized at 256x256", scenes_per_gpu=B, context_views=V, target_views=Vt,
                                parallelism=f"scene-sharded x{world} (no collective)"),
                    roofline=roofline, cpu_baseline=cpu_baseline, psnr_vs_oracle=psnr_vs_oracle, targets70=targets70, latency_b1=latency_b1, fast_path=fast_path, f32_path=f32_path, train=train, **extra)

    train = None
    if args.mode in ("train", "both"):
        del ctx
        # The training leg is the only part of this file with collectives in its data path.  If it stalls (a rank lost inside an
        # all-reduce), the already-measured headline must still be reported: a watchdog prints it from rank 0 and ends every rank.
        import threading

        def give_up():
            if rank == 0 and args.mode != "train":
                print(json.dumps(headline(dict(error=f"training leg did not finish within {args.train_timeout} s"))), flush=True)
            os._exit(0 if args.mode != "train" else 3)

        dog = threading.Timer(args.train_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            if args.mode == "train" and args.dtype == "split":
                train = train_leg(args, enc, dec, dev, rank, world, dist, cdt="split", scenes=args.train_split_scenes if args.train_scenes_per_gpu == 24 else None)
            else:
                train = train_leg(args, enc, dec, dev, rank, world, dist)
                if world == 1 and not args.no_train_split:      # the same step at the reference's precision (secondary leg, reduced batch)
                    try:
                        train["split_class"] = train_leg(args, enc, dec, dev, rank, world, None, cdt="split", scenes=args.train_split_scenes, steps=2)
                    except Exception as e:
                        train["split_class"] = dict(error=repr(e)[:300])
                    # BASELINE config 5's per-GPU batch at the reference's precision (VERDICT r3 "missing 5"): 24 scenes only fit with
                    # enable_gradient_checkpointing() (re10k_8view.yaml:19-20,61; the reference trains with it on)
                    if args.train_scenes_per_gpu == 24 and not args.no_train_split24:
                        try:
                            train["split_class_batch24_checkpointed"] = train_leg(args, enc, dec, dev, rank, world, None, cdt="split", scenes=24, steps=2,
                                                                                   checkpoint=True)
                        except Exception as e:
                            train["split_class_batch24_checkpointed"] = dict(error=repr(e)[:300])
                        # the same batch with only as many blocks checkpointed as the 288 GB require (late round 5): recomputation is a memory
                        # measure, and the reference's all-blocks policy (re10k_8view.yaml:61) is sized for 80 GB devices
                        try:
                            train["split_class_batch24_checkpoint_auto"] = train_leg(args, enc, dec, dev, rank, world, None, cdt="split", scenes=24, steps=2,
                                                                                     checkpoint="auto")
                        except Exception as e:
                            train["split_class_batch24_checkpoint_auto"] = dict(error=repr(e)[:300])
        except Exception as e:      # the headline line must survive a failure of the optional training leg
            if args.mode == "train":
                raise
            train = dict(error=repr(e)[:300])
            if dist is not None:    # the ranks may have diverged inside a collective: nothing after this point may need them in step
                dist = None
        finally:
            dog.cancel()
    if rank == 0 and args.mode == "train":
        print(json.dumps(dict(metric=train["metric"], value=train["value"], unit="scenes/s", n_gpus=world, steps=args.train_steps, warmup=1,
                              ms_per_step=train["ms_per_step"], higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.dtype,
                              data="synthetic", config=dict(workload="re10k_8view full pipeline fwd+bwd: one training step",
                                                            scenes_per_gpu=train["scenes_per_gpu"], context_views=V, target_views=Vt,
                                                            parallelism=f"data-parallel x{world}: " + train["gradient_exchange"]),
                              roofline=train["roofline"], cpu_baseline=None, train=train)))
    elif rank == 0:
        line = headline(train)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
