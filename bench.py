#!/usr/bin/env python
"""bench.py -- train-step images/sec (fwd+bwd raster) @ 1M Gaussians 1080p on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 via torch.distributed.run, one
rank per GPU over RCCL).  One "step" = one pass of the hot path on one synthetic view per rank:
GaussianRasterizer forward (preprocess, scan, duplicateWithKeys, radix sort, ranges, per-tile
alpha blend + exposure/CRF tone-map) and backward (per-pixel backward, preprocess backward),
followed at N>1 by one RCCL all-reduce of the per-Gaussian gradients.  Inputs are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[2] ("c3": 1M Gaussians, 1920x1080, SH degree 3, HDR linear
radiance + learned CRF tone-map), the configuration the metric is quoted on; at N>1 each rank
renders its own view of the same cloud (configs[4]).  Synthetic scene: SURVEY.md 8(d).

Extra objects on the line:
  roofline      -- the dominant kernel (render_bwd_kernel): algorithmic bytes per launch
                   (76*R' + 20*W*H) / its average duration measured here with HIP events
                   on the launch stream; peak = 8 TB/s HBM3E.  `fwd_bwd` repeats the figure for the
                   per-tile alpha-blend forward+backward pair (116*R' + 40*W*H + 8*tiles), the
                   quantity BASELINE.md's 40 % target is stated on.
  cpu_baseline  -- the pure-PyTorch CPU autograd rasterizer (oracle/torch_rasterizer.py) timed on
                   this box's host cores on a bounded sample of the same frame.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)

CONFIGS = {
    # name: (P, W, H, sh_degree, hdr, n_poses)
    "c2": (100_000, 800, 800, 0, False, 1),
    "c3": (1_000_000, 1920, 1080, 3, True, 1),
    "c4": (1_000_000, 1920, 1080, 3, True, 8),
}


def build_step(cfg, rank, world, dev):
    from casualhdrsplat_amd import GaussianRasterizationSettings, GaussianRasterizer, synthetic as S
    from casualhdrsplat_amd.distributed import all_reduce_gradients, exchange_view_gradients
    P, W, H, deg, hdr, n_poses = cfg
    sc = S.make_scene(P, W, H, deg, seed=0, hdr=hdr)
    # one view per rank: yaw in [-5, +5] degrees about the cloud centre; rank 0 of a
    # single-GPU run uses the frontal camera the cloud was laid out for.
    if world > 1:
        yaw = -5.0 + 10.0 * rank / (world - 1)
        cam = S.yaw_camera(W, H, yaw)
    else:
        cam = sc.camera
    kw = {}
    exposure = crf = None
    if hdr:
        exposure = sc.exposure.clone().to(dev).requires_grad_(True)
        crf = sc.crf_table.clone().to(dev).requires_grad_(True)
        kw.update(exposure=exposure, crf_table=crf, crf_range=sc.crf_range)
    if n_poses > 1:
        cams = S.blur_poses(W, H, n_poses)
        kw.update(viewmatrices=torch.stack([c.viewmatrix for c in cams]).to(dev),
                  projmatrices=torch.stack([c.projmatrix for c in cams]).to(dev),
                  camposes=torch.stack([c.campos for c in cams]).to(dev))
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=sc.bg.to(dev),
        scale_modifier=1.0, viewmatrix=cam.viewmatrix.to(dev), projmatrix=cam.projmatrix.to(dev),
        sh_degree=deg, campos=cam.campos.to(dev), prefiltered=False, debug=False, **kw)
    params = dict(
        means3D=sc.means3D.to(dev).requires_grad_(True),
        means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
        opacities=sc.opacities.to(dev).requires_grad_(True),
        shs=sc.shs.to(dev).requires_grad_(True),
        scales=sc.scales.to(dev).requires_grad_(True),
        rotations=sc.rotations.to(dev).requires_grad_(True),
    )
    dL = sc.dL_dimage.to(dev)
    plist = list(params.values()) + ([exposure, crf] if hdr else [])

    non_sh = [p for k, p in params.items() if k != "shs"] + ([exposure, crf] if hdr else [])

    def make_rasterizer(capacity):
        # two front ends over the same kernels: plain, and with the SH gradient deferred to the view exchange
        return {"allreduce": GaussianRasterizer(rs, capacity=capacity),
                "views": GaussianRasterizer(rs, capacity=capacity, defer_sh_grad=True),
                "views_overlap": GaussianRasterizer(rs, capacity=capacity, defer_sh_grad=True, gather_group=True)}

    # exchange = (what goes on the wire, all-reduce algorithm); chosen by measurement in main() when world > 1
    state = {"rast": make_rasterizer(None), "out": None, "exchange": ("allreduce", "rccl")}

    def step():
        mode, algo = state["exchange"]
        rast = state["rast"][mode if world > 1 else "allreduce"]
        for p in plist:
            p.grad = None
        out = rast(params["means3D"], params["means2D"], params["opacities"], shs=params["shs"],
                   scales=params["scales"], rotations=params["rotations"])
        torch.autograd.backward(out[0], grad_tensors=dL)
        if world > 1:
            if mode != "allreduce":
                exchange_view_gradients(non_sh, params["shs"], rast.deferred, algo=algo)
            else:
                all_reduce_gradients(plist, algo=algo)
        state["out"] = out
        return out

    return step, state, make_rasterizer, sc, dL, plist


def derived_counts(out, W, H, n_poses):
    """R, R' = sum over tiles of max n_contrib (entries a tile must fetch), E = sum n_contrib."""
    from casualhdrsplat_amd import inspect_state
    st = inspect_state(out[0])
    nc = st["n_contrib"].to(torch.int64)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pad = torch.zeros(n_poses, gy * 16, gx * 16, dtype=torch.int64, device=nc.device)
    pad[:, :H, :W] = nc
    tmax = pad.reshape(n_poses, gy, 16, gx, 16).amax(dim=(2, 4))
    return int(st["num_rendered"]), int(tmax.sum()), int(nc.sum()), gx * gy * n_poses


def time_stage(fn, iters):
    """Average duration (ms) of fn() measured with HIP events on the current (= launch) stream."""
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ts) / len(ts), ts[len(ts) // 2]


def cpu_baseline(sc, cfg, n_tiles_sample=96):
    """Pure-PyTorch CPU autograd rasterizer on a bounded sample: full preprocess + binning of the frame,
    then forward+backward of every k-th tile; extrapolated to images/s of the whole frame."""
    from oracle import torch_rasterizer as TR
    P, W, H, deg, hdr, n_poses = cfg
    cam = sc.camera
    view = TR.View(W, H, cam.tanfovx, cam.tanfovy, cam.viewmatrix, cam.projmatrix, cam.campos)
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ntiles = gx * gy
    stride = max(1, ntiles // n_tiles_sample)
    tiles = list(range(stride // 2, ntiles, stride))
    t0 = time.time()
    pre = TR.preprocess(view, leaves[0], leaves[1], deg, shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    point_list, ranges, _ = TR.bin_tiles(view, pre)
    t1 = time.time()
    color, _, _ = TR.render(view, pre, point_list, ranges, sc.bg, tiles=tiles)
    if hdr:
        color = TR.tonemap(color, sc.exposure, sc.crf_table, sc.crf_range)
    (color * sc.dL_dimage).sum().backward()
    t2 = time.time()
    t_pre, t_tiles = t1 - t0, t2 - t1
    t_full = t_pre + t_tiles * (ntiles / len(tiles))
    return {
        "value": 1.0 / t_full, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
        "host_cpus": os.cpu_count(),
        "sample": (f"pure-PyTorch fp32 autograd rasterizer (oracle/torch_rasterizer.py): full preprocess+binning of "
                   f"the {P}-Gaussian {W}x{H} frame ({t_pre:.1f} s) + fwd+bwd of {len(tiles)} of {ntiles} tiles "
                   f"({t_tiles:.1f} s, backward also covers preprocess); value extrapolates the tile part to all tiles"),
        "measured_seconds": t2 - t0,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-iters", type=int, default=10)
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback exists)")
    import torch.distributed as dist
    from casualhdrsplat_amd.distributed import init_from_env
    # HS_BENCH_BACKEND=gloo is a plumbing check for boxes with fewer GPUs than ranks (ranks then share devices);
    # measured runs use "nccl" (= RCCL over xGMI), one rank per GPU.
    rank, world, local = init_from_env(os.environ.get("HS_BENCH_BACKEND", "nccl"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = CONFIGS[args.config]
    P, W, H, deg, hdr, n_poses = cfg

    step, state, make_rasterizer, sc, dL, plist = build_step(cfg, rank, world, dev)
    # first step in the upstream-compatible synchronous mode learns num_rendered; the timed steps use the
    # sync-free mode with a fixed binning capacity (25 % headroom), overflow checked lazily every step.
    out = step()
    torch.cuda.synchronize()
    R, Rp, E, vtiles = derived_counts(out, W, H, n_poses)
    state["rast"] = make_rasterizer(int(R * 1.25) + 4096)
    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    allreduce_info = None
    if world > 1:
        # measure, don't guess: time a few whole steps with each gradient exchange on this node's links and keep the
        # fastest (max over ranks, so every rank takes the same decision).  "allreduce" = all-reduce of the flat
        # per-Gaussian gradient buffer; "views" = all-reduce of the non-SH part + all-gather of per-view colour
        # gradients with the SH gradient rebuilt locally; "views_overlap" = the same with the all-gather started
        # inside the backward, under its per-Gaussian half; "rccl" / "direct" = library ring vs 1-hop all-to-all form
        times = {}
        for mode in ("allreduce", "views", "views_overlap"):
            for algo in ("rccl", "direct"):
                state["exchange"] = (mode, algo)
                try:  # a backend that lacks a collective raises on every rank alike: skip that strategy
                    step()
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        step()
                    barrier()
                    dt_ = time.perf_counter() - t0
                except RuntimeError as e:
                    if rank == 0:
                        print(f"[bench] exchange {mode}/{algo} unavailable: {e}", file=sys.stderr)
                    continue
                t = torch.tensor([dt_], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                times[f"{mode}/{algo}"] = float(t.item()) / 3 * 1e3
        best = min(times, key=times.get) if times else "allreduce/rccl"
        state["exchange"] = tuple(best.split("/"))
        allreduce_info = {"choice": best, "step_ms": times}

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = world * args.steps / elapsed

    line = {
        "metric": "train-step images/sec (fwd+bwd raster) @ 1M Gaussians 1080p",
        "value": images_per_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {P} Gaussians, {W}x{H}, SH degree {deg}, "
                               f"{'HDR radiance + CRF tone-map' if hdr else 'LDR'}, {n_poses} pose(s)/view, "
                               f"{world} view(s)/step (one per GPU)" + (", gradients summed over views (config.gradient_exchange)" if world > 1 else ""),
                   "num_rendered_R": R, "R_prime": Rp, "pixel_pair_evals_E": E,
                   "binning": "sync-free fixed capacity 1.25*R"},
        "mpix_per_s": world * args.steps * W * H * n_poses / elapsed / 1e6,
    }
    if allreduce_info is not None:
        line["config"]["gradient_exchange"] = allreduce_info

    if rank == 0:
        from casualhdrsplat_amd import _lib as L
        from casualhdrsplat_amd.rasterizer import replay_backward, replay_forward
        # a fresh forward whose autograd graph is kept (never .backward()-ed) so its stages can be replayed
        for p_ in plist:
            p_.grad = None
        out = state["rast"]["allreduce"](*[plist[i] for i in (0, 1, 2)], shs=plist[3], scales=plist[4], rotations=plist[5])
        R, Rp, E, vtiles = derived_counts(out, W, H, n_poses)
        WH = W * H * n_poses
        bwd_ms, bwd_med = time_stage(lambda: replay_backward(out[0], dL, L.HS_BWD_RENDER), args.kernel_iters)
        fwd_ms, fwd_med = time_stage(lambda: replay_forward(out[0], L.HS_STAGE_RENDER), args.kernel_iters)
        bytes_bwd = 76 * Rp + 20 * WH
        bytes_fwd = 40 * Rp + 20 * WH + 8 * vtiles
        traffic = valu_busy = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                pmc = json.load(open(tpath)).get(args.config, {})
                traffic = pmc.get("render_bwd_kernel_hbm_bytes")
                valu_busy = pmc.get("render_bwd_kernel_valu_busy_frac")
            except Exception:
                traffic = valu_busy = None
        ach = bytes_bwd / (bwd_ms * 1e-3) / 1e9
        line["roofline"] = {
            "kernel": "render_bwd_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes": bytes_bwd, "avg_ms": bwd_ms, "median_ms": bwd_med,
            "valu_busy_frac": valu_busy,  # PMC (profiles/pmc_traffic.json): what actually bounds this kernel
            "note": "the kernel is VALU-issue bound (DESIGN.md 4, profiles/README.md), not HBM bound; traffic = L2-fabric bytes from profiles/pmc_traffic.json",
            "fwd_bwd": {"kernels": "render_fwd_kernel + render_bwd_kernel", "algorithmic_bytes": bytes_fwd + bytes_bwd,
                        "avg_ms": fwd_ms + bwd_ms,
                        "achieved": (bytes_fwd + bytes_bwd) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9,
                        "frac": (bytes_fwd + bytes_bwd) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "pair_evals_per_s": 2 * E / ((fwd_ms + bwd_ms) * 1e-3)},
        }
        # live per-stage times of the same frame (HIP events on the launch stream), for the reader of the line
        stages = {"render_bwd": bwd_ms, "render_fwd": fwd_ms}
        for name, fn in (("binning", lambda: replay_forward(out[0], L.HS_STAGE_BIN)),
                         ("segsum_and_preprocess_bwd", lambda: replay_backward(out[0], dL, L.HS_BWD_PREPROCESS)),
                         ("crf_gradient", (lambda: replay_backward(out[0], dL, L.HS_BWD_CRF)) if hdr else None)):
            if fn is not None:
                stages[name] = time_stage(fn, args.kernel_iters)[0]
        line["stages_ms"] = {k: round(v, 4) for k, v in stages.items()}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sc, cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
