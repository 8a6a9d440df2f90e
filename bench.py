#!/usr/bin/env python
"""bench.py -- train-step images/sec (fwd+bwd raster) @ 1M Gaussians 1080p on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  One rank per GPU over RCCL; at N > 1 the ranks come
either from the caller (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / WORLD_SIZE
are in the environment) or, when `python bench.py --gpus N` is run bare, from bench.py itself: before anything touches
a GPU it starts `python -m torch.distributed.run` with N fresh child processes on a free port, forwards rank 0's JSON
line and exits with the children's code.

One "step" = one pass of the hot path on one synthetic view per rank: GaussianRasterizer forward (preprocess, scan,
duplicateWithKeys, radix sort, ranges, per-tile alpha blend + exposure/CRF tone-map) and backward (per-pixel backward,
preprocess backward), followed at N>1 by the exchange that sums the per-Gaussian gradients over the views.  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[2] ("c3": 1M Gaussians, 1920x1080, SH degree 3, HDR linear radiance + learned CRF
tone-map), the configuration the metric is quoted on; at N>1 each rank renders its own view of the same cloud
(configs[4]).  Synthetic scene: SURVEY.md 8(d); seeds {0,1,2} are each timed the same way (W warm-up + K steps, fresh scene
and rasterizers) and the headline `value` / `ms_per_step` is their MEDIAN (`seeds_ms_per_step` has all three).

Extra objects on the line (all measured in this run unless marked `from_profiles`):
  roofline      -- the dominant kernel (render_bwd_kernel): algorithmic bytes per launch (76*R' + 20*W*H) /
                   its average duration measured here with HIP events on the launch stream; peak = 8 TB/s HBM3E.
                   `fwd_bwd` repeats the figure for the per-tile alpha-blend forward+backward pair
                   (116*R' + 40*W*H + 8*tiles), the quantity BASELINE.md's 40 % target is stated on.  `valu` is the
                   roofline that actually binds these kernels: wave-level vector instructions per launch =
                   (instructions per compositing-loop trip, read from the ISA by scripts/isa_loop_counts.py) x (trips,
                   counted live by the diagnostic instantiation of the kernels), against the issue rate of the 1024
                   SIMDs (one wave64 VALU instruction per 4 cycles per SIMD at 2.4 GHz); `lane_utilisation` = active
                   pixels / (128 x trips).
  cpu_baseline  -- the pure-PyTorch CPU autograd rasterizer (oracle/torch_rasterizer.py) timed on this box's host
                   cores: BASELINE config c1 in full (median of 5), c2 in full when it fits the time budget, and a
                   bounded sample of the c3 frame extrapolated to the whole frame (`value`).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
HBM_MEASURED_GBS = 6290.0  # the guide's measured float4 copy rate: the practical ceiling
# VALU issue peak: 256 CUs x 4 SIMDs, one wave64 vector instruction per 4 cycles per SIMD (16 lanes/clk), 2.4 GHz
N_SIMD, CLK_HZ, CYCLES_PER_VALU = 1024, 2.4e9, 4.0
VALU_PEAK_WAVE_INSTR_PER_S = N_SIMD * CLK_HZ / CYCLES_PER_VALU

CONFIGS = {
    # name: (P, W, H, sh_degree, hdr, n_poses)
    "c1": (1_000, 128, 128, 0, False, 1),
    "c2": (100_000, 800, 800, 0, False, 1),
    "c3": (1_000_000, 1920, 1080, 3, True, 1),
    "c4": (1_000_000, 1920, 1080, 3, True, 8),
}


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(argv: list[str], gpus: int, port: int) -> list[str]:
    """The command a bare `python bench.py --gpus N` turns into: N fresh processes, one per GPU."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(argv: list[str], gpus: int) -> int:
    """Parent of a bare multi-GPU run.  Nothing here imports torch or touches HIP: the ranks are started as ordinary
    child processes (never an exec of a process that initialised the GPU) and this process only relays their output."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL peer access)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(gpus, 1))))
    cmd = launch_command(argv, gpus, free_port())
    print(f"[bench] --gpus {gpus} without a torchrun environment: starting {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in proc.stdout:  # rank 0's JSON line (and anything else the ranks print) goes straight through
        sys.stdout.write(ln)
        sys.stdout.flush()
    return proc.wait()


def build_step(cfg, rank, world, dev, seed=0):
    import torch
    from casualhdrsplat_amd import GaussianRasterizationSettings, GaussianRasterizer, SortChainStalled, synthetic as S
    from casualhdrsplat_amd.distributed import all_reduce_gradients, exchange_view_gradients
    P, W, H, deg, hdr, n_poses = cfg
    sc = S.make_scene(P, W, H, deg, seed=seed, hdr=hdr)
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
    # what a view-parallel step sums over the ranks: the gradients of the PARAMETERS.  means2D is not one -- its
    # "gradient" is this view's screen-space gradient, the densification statistic a trainer accumulates per view
    # (norm per view, summed at densification time), so it stays on the rank that rendered the view
    reduced = [p for k, p in params.items() if k != "means2D"] + ([exposure, crf] if hdr else [])
    non_sh = [p for k, p in params.items() if k not in ("shs", "means2D")] + ([exposure, crf] if hdr else [])

    def make_rasterizer(capacity):
        # front ends over the same kernels: plain, and with the SH gradient deferred to the view exchange
        return {"allreduce": GaussianRasterizer(rs, capacity=capacity),
                "allreduce_overlap": GaussianRasterizer(rs, capacity=capacity, reduce_group=True, reduce_chunks=4),
                "views": GaussianRasterizer(rs, capacity=capacity, defer_sh_grad=True),
                "views_overlap": GaussianRasterizer(rs, capacity=capacity, defer_sh_grad=True, gather_group=True)}

    # exchange = (what goes on the wire, all-reduce algorithm); chosen by measurement in main() when world > 1
    state = {"rast": make_rasterizer(None), "out": None, "exchange": ("allreduce", "rccl")}

    def step():
        mode, algo = state["exchange"]
        rast = state["rast"][mode if (world > 1 and mode != "none") else "allreduce"]
        rast.gather_direct = algo == "direct"
        for p in plist:
            p.grad = None
        for attempt in (0, 1):
            out = rast(params["means3D"], params["means2D"], params["opacities"], shs=params["shs"],
                       scales=params["scales"], rotations=params["rotations"])
            try:
                torch.autograd.backward(out[0], grad_tensors=dL)
                break
            except SortChainStalled:
                # ranks sharing one GPU (the gloo plumbing check) can stall each other's blockIdx-ordered radix passes; the
                # library has switched to ticket order: repeat the step -- unless this backward already started a
                # collective of its own (the other ranks run it once)
                if attempt or (world > 1 and mode in ("views_overlap", "allreduce_overlap")):
                    raise
                for p in plist:
                    p.grad = None
        if world > 1 and mode != "none":   # ("none": the probe's yardstick, a step without the exchange)
            if mode == "allreduce_overlap":
                rast.finish_reduce()      # the backward itself issued the chunked all-reduce, under its per-Gaussian half
            elif mode != "allreduce":
                exchange_view_gradients(non_sh, params["shs"], rast.deferred, algo=algo)
            else:
                all_reduce_gradients(reduced, algo=algo)
        state["out"] = out
        return out

    return step, state, make_rasterizer, sc, dL, plist


def whole_step_bytes(cfg, R, Rp, vtiles, n_visible):
    """Algorithmic bytes of ONE whole step (every row of the table; I = P x poses instances): what a
    perfect implementation must move, whatever this one moves.  Sort = one ideal pass (24 R).  The Gaussian ROW (means,
    scales, rotation, opacity, SH: 236 B at degree 3) is charged once per GAUSSIAN -- N poses project the same row, and a
    perfect implementation reads it once (this one does: 1.06 GB counted for c4's preprocess forward, profiles/
    r05_pmc_traffic_c4.csv) -- and only the per-instance outputs / gradient inputs once per instance (VERDICT r5 weak #3:
    until round 6 the row was charged per instance, which put c4's preprocess stages ABOVE the measured copy rate)."""
    P, W, H, deg, hdr, n_poses = cfg
    M = (deg + 1) ** 2
    I, WH = P * n_poses, W * H * n_poses
    in_row = 12 + 12 + 16 + 4 + 12 * M                     # means, scales, rotation, opacity, SH coefficients
    parts = {
        "preprocess_fwd": in_row * P + 75 * n_visible,
        "scan": 8 * I,
        "duplicate_with_keys": 20 * I + 12 * R,
        "sort_one_ideal_pass": 24 * R,
        "tile_ranges": 8 * R + 8 * vtiles,
        "render_fwd": 40 * Rp + 20 * WH + 8 * vtiles,
        "render_bwd": 76 * Rp + 20 * WH,
        "preprocess_bwd": in_row * P + 60 * I + in_row * P,   # read the row + 60 B per instance, write the row
        "hdr_images": (2 * 12 * WH) if hdr else 0,
    }
    parts["total"] = sum(parts.values())
    return parts


def derived_counts(out, W, H, n_poses):
    """R, R' = sum over tiles of max n_contrib (entries a tile must fetch), E = sum n_contrib."""
    import torch
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
    import torch
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


def _cpu_full_frame(cfg, seed=0):
    """One full fwd+bwd of a BASELINE config with the pure-PyTorch CPU rasterizer; returns seconds."""
    from casualhdrsplat_amd import synthetic as S
    from oracle import torch_rasterizer as TR
    P, W, H, deg, hdr, n_poses = cfg
    sc = S.make_scene(P, W, H, deg, seed=seed, hdr=hdr)
    cam = sc.camera
    view = TR.View(W, H, cam.tanfovx, cam.tanfovy, cam.viewmatrix, cam.projmatrix, cam.campos)
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
    t0 = time.time()
    color = TR.rasterize(view, leaves[0], leaves[1], deg, sc.bg, shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    (color * sc.dL_dimage).sum().backward()
    return time.time() - t0


def cpu_baseline(sc, cfg, n_tiles_sample=256, with_c2=False):
    """CPU baseline of the bench configuration.  `value` = the pure-PyTorch CPU autograd rasterizer of BASELINE.md
    section 3 (oracle/torch_rasterizer.py) on the box's host cores, on a bounded sample of the same frame (preprocess
    + binning forward and the preprocess backward of the WHOLE frame, timed exactly; render forward+backward of every k-th
    tile in four interleaved subsets, extrapolated to all tiles -- `extrapolation` holds the four estimates and their spread);
    `c_oracle_single_thread` = the C oracle (one thread) on the WHOLE frame (_c_oracle_frame), no sampling; c1 in full
    with the PyTorch rasterizer (median of 5); c2 in full as well ("config 2 if it completes in < 10 min" --
    2.5 minutes on the 128 host cores of the MI355X box) unless --no-cpu-c2."""
    import torch
    from oracle import torch_rasterizer as TR
    P, W, H, deg, hdr, n_poses = cfg
    t_start = time.time()
    c1 = sorted(_cpu_full_frame(CONFIGS["c1"]) for _ in range(5))
    c1_med = c1[2]
    out = {"c1_full": {"images_per_s": 1.0 / c1_med, "seconds_median_of_5": c1_med,
                       "workload": "1k Gaussians, 128x128, SH 0, fwd+bwd, whole frame"}}
    cam = sc.camera
    view = TR.View(W, H, cam.tanfovx, cam.tanfovy, cam.viewmatrix, cam.projmatrix, cam.campos)
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ntiles = gx * gy
    stride = max(1, ntiles // n_tiles_sample)
    tiles = list(range(stride // 2, ntiles, stride))
    # the frame's cost = preprocess + binning forward (whole frame, timed exactly) + the tile part (sampled) + the backward
    # of preprocess (whole frame, timed exactly).  The tile sample is split into four interleaved subsets, each rendered and
    # differentiated down to the preprocess outputs on its own: four independent estimates of the per-tile cost, whose
    # spread is the extrapolation's error bar
    t0 = time.time()
    pre = TR.preprocess(view, leaves[0], leaves[1], deg, shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    point_list, ranges, _ = TR.bin_tiles(view, pre)
    t_pre = time.time() - t0
    mids = [pre[k] for k in ("xy", "conic", "opacity", "rgb")]
    acc = [torch.zeros_like(m) for m in mids]
    n_sub = 4
    per_tile_est, t_tiles = [], 0.0
    for k in range(n_sub):
        sub = tiles[k::n_sub]
        t1 = time.time()
        color, _, _ = TR.render(view, pre, point_list, ranges, sc.bg, tiles=sub)
        if hdr:
            color = TR.tonemap(color, sc.exposure, sc.crf_table, sc.crf_range)
        g = torch.autograd.grad((color * sc.dL_dimage).sum(), mids, allow_unused=True)
        dt = time.time() - t1
        for a_, g_ in zip(acc, g):
            if g_ is not None:
                a_ += g_
        per_tile_est.append(dt / len(sub))
        t_tiles += dt
    t2 = time.time()
    torch.autograd.backward(mids, acc)
    t_pre_bwd = time.time() - t2
    per_tile = t_tiles / len(tiles)
    t_full = t_pre + t_pre_bwd + per_tile * ntiles
    est = [1.0 / (t_pre + t_pre_bwd + e * ntiles) for e in per_tile_est]
    mean_e = sum(per_tile_est) / n_sub
    std_err = (sum((e - mean_e) ** 2 for e in per_tile_est) / (n_sub - 1)) ** 0.5 / n_sub ** 0.5
    # c2 in full costs ~170 x the c1 frame just timed (151-165 s on the MI355X box's 128 host threads): run it unless that
    # prediction says the box's CPU would need more than six minutes for it (the default run must finish within minutes)
    c2_predicted = 170.0 * c1_med
    if with_c2 and c2_predicted > 360.0:
        out["c2_full"] = {"skipped": f"predicted {c2_predicted:.0f} s on this host (170 x the c1 frame's {c1_med:.2f} s) > 360 s budget; "
                                     "last measured: profiles/r04_bench_c3.json"}
    elif with_c2:
        t = _cpu_full_frame(CONFIGS["c2"])
        out["c2_full"] = {"images_per_s": 1.0 / t, "seconds": t, "workload": "100k Gaussians, 800x800, SH 0, LDR, fwd+bwd, whole frame"}
    else:
        out["c2_full"] = {"skipped": "--no-cpu-c2 given (about 150 s of CPU time); last measured: profiles/r02_cpu_baseline_c2.json"}
    torch_port = {
        "value": 1.0 / t_full, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": (f"pure-PyTorch fp32 autograd rasterizer (oracle/torch_rasterizer.py) on the {P}-Gaussian {W}x{H} frame: "
                   f"preprocess+binning forward of the whole frame ({t_pre:.1f} s) + preprocess backward of the whole frame "
                   f"({t_pre_bwd:.1f} s), both exact, + render fwd+bwd of {len(tiles)} of {ntiles} tiles in {n_sub} interleaved "
                   f"subsets ({t_tiles:.1f} s); value extrapolates the tile part only"),
        "extrapolation": {"tiles_sampled": len(tiles), "tiles": ntiles, "subsets": n_sub,
                          "images_per_s_by_subset": est, "per_tile_seconds_by_subset": per_tile_est,
                          "per_tile_rel_std_err": std_err / mean_e,
                          "exact_seconds": {"preprocess_and_binning_fwd": t_pre, "preprocess_bwd": t_pre_bwd}},
    }
    # the headline baseline is the one BASELINE.json's north_star words: the pure-PyTorch CPU rasterizer on the box's host
    # cores (bounded sample, extrapolated); the C oracle on ONE core -- the whole frame, no sampling, and ~50x faster than
    # the PyTorch rasterizer on all cores -- is reported next to it
    out.update(torch_port)
    if n_poses == 1:
        out["c_oracle_single_thread"] = _c_oracle_frame(sc, cfg)
    out["host_cpus"] = os.cpu_count()
    out["measured_seconds"] = time.time() - t_start
    return out


def _c_oracle_frame(sc, cfg):
    """The C restatement (oracle/hs_oracle.c, single thread) on the WHOLE frame of the bench configuration: preprocess,
    binning, render forward, tone-map forward/backward (HDR configs) and the full backward -- the same work as one GPU
    step.  About half a minute at c3."""
    import numpy as np
    from oracle import c_oracle as O
    P, W, H, deg, hdr, n_poses = cfg
    cam = sc.camera
    ocam = O.Camera(W, H, cam.tanfovx, cam.tanfovy, cam.viewmatrix.numpy(), cam.projmatrix.numpy(), cam.campos.numpy(),
                    sc.bg.numpy(), 1.0, deg)
    kw = dict(shs=sc.shs.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy())
    means, opac, dL = sc.means3D.numpy(), sc.opacities.numpy(), sc.dL_dimage.numpy()
    t0 = time.time()
    f = O.forward(ocam, means, opac, **kw)
    if hdr:
        dt, tab, (umin, umax) = float(sc.exposure), sc.crf_table.numpy(), sc.crf_range
        O.tonemap_fwd(f["color"], dt, tab, umin, umax)
        dL = O.tonemap_bwd(f["color"], dt, tab, umin, umax, dL)[0]
    O.backward(ocam, f, np.ascontiguousarray(dL, dtype=np.float32), means, **kw)
    t = time.time() - t0
    return {"value": 1.0 / t, "unit": "images/s", "cores": 1, "kind": "port",
            "sample": (f"C oracle (oracle/hs_oracle.c, one thread), the whole {P}-Gaussian {W}x{H} frame: preprocess + "
                       f"binning + render forward + {'tone-map + ' if hdr else ''}full backward in {t:.1f} s "
                       f"(R = {int(f['R'])} pairs) -- no sampling, no extrapolation")}


def kernel_source_hash(path):
    """SHA-256 of a kernel source with blank and comment-only lines dropped: the committed ISA counts and PMC counters
    (profiles/) stay attached to the CODE they were taken on when only a comment changes."""
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for ln in f:
            t = ln.strip()
            if t and not t.startswith(b"//"):
                h.update(t + b"\n")
    return h.hexdigest()


def isa_counts():
    """Per-trip vector-instruction counts of the two compositing loops (scripts/isa_loop_counts.py reads them from
    the compiler's ISA for gfx950 and commits them as profiles/isa_loop_counts.json).  They are a property of the
    source, so the file records the hash of render.hip it was made from; `stale` says whether that still matches."""
    path = os.path.join(ROOT, "profiles", "isa_loop_counts.json")
    if not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
        src = os.path.join(ROOT, "casualhdrsplat_amd", "csrc", "render.hip")
        d["stale"] = kernel_source_hash(src) != d.get("render_hip_sha256")
        return d
    except Exception:
        return None


def valu_roofline(stats, isa, fwd_ms, bwd_ms):
    """The roofline that binds the render kernels (SURVEY.md 8d "which roofline"): wave-level vector instructions
    issued per launch / launch time, against one wave64 VALU instruction per 4 cycles on each of the 1024 SIMDs."""
    out = {"bound": "valu_issue", "peak": VALU_PEAK_WAVE_INSTR_PER_S, "unit": "wave64 VALU instr/s",
           "peak_note": "1024 SIMDs x 2.4 GHz / 4 cycles per wave64 vector instruction (transcendentals and "
                        "v_permlane*_swap take 8: `issue_cycles_frac` weights them)"}
    for side, ms in (("bwd", bwd_ms), ("fwd", fwd_ms)):
        trips, empty, pix = stats[f"{side}_trips"], stats[f"{side}_empty_trips"], stats[f"{side}_active_pixels"]
        o = {"trips": trips, "empty_trips": empty,
             "lane_utilisation": pix / (128.0 * trips) if trips else None,
             "lane_utilisation_note": "active pixels / (128 pixels x trips), counted live (hs_render_stats)"}
        if side == "bwd":
            o["trips_by_active_lanes"] = {k[len("bwd_hist_"):]: stats[k] for k in stats if k.startswith("bwd_hist_")}
        k = (isa or {}).get(f"render_{side}_kernel")
        if k:
            full, short = k["valu_per_trip"], k.get("valu_per_empty_trip", k["valu_per_trip"])
            cyc_full, cyc_short = k["valu_cycles_per_trip"], k.get("valu_cycles_per_empty_trip", k["valu_cycles_per_trip"])
            instr = full * (trips - empty) + short * empty
            cycles = cyc_full * (trips - empty) + cyc_short * empty
            o.update({"valu_instr_per_trip": full, "valu_instr_per_empty_trip": short,
                      "valu_instr_per_launch_loop_only": instr, "achieved": instr / (ms * 1e-3),
                      "frac": instr / (ms * 1e-3) / VALU_PEAK_WAVE_INSTR_PER_S,
                      "issue_cycles_frac": cycles / (ms * 1e-3 * N_SIMD * CLK_HZ),
                      "isa_from_profiles": True, "isa_source": (isa or {}).get("source"),
                      "isa_stale": (isa or {}).get("stale")})
        out[side] = o
    return out


def offline_profile(cfg_name):
    """Counter-derived figures of an EARLIER rocprofv3 --pmc run (they cannot be collected inside a timed run): kept
    apart from the live numbers and labelled with where they came from."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return None
    try:
        pmc = json.load(open(tpath)).get(cfg_name)
        if not pmc:
            return None
        src = os.path.join(ROOT, "casualhdrsplat_amd", "csrc", "render.hip")
        same = bool(pmc.get("render_hip_sha256")) and \
            kernel_source_hash(src) == pmc.get("render_hip_sha256")
        return {"from_profiles": True, "file": "profiles/pmc_traffic.json", "source": pmc.get("source"),
                "source_commit": pmc.get("source_commit"), "same_kernel_source": same,
                "render_bwd_kernel_hbm_bytes": pmc.get("render_bwd_kernel_hbm_bytes"),
                "render_fwd_kernel_hbm_bytes": pmc.get("render_fwd_kernel_hbm_bytes"),
                "render_bwd_kernel_tile_replay_hbm_bytes": pmc.get("render_bwd_kernel_tile_replay_hbm_bytes"),
                "render_bwd_kernel_valu_busy_frac": pmc.get("render_bwd_kernel_valu_busy_frac"),
                "render_fwd_kernel_valu_busy_frac": pmc.get("render_fwd_kernel_valu_busy_frac"),
                "method": pmc.get("method")}
    except Exception:
        return None


FALLBACK = ("allreduce", "rccl")   # the plain library all-reduce of the flat gradient buffer: never dropped unless it raises
EXCHANGES = [("allreduce", "rccl"), ("allreduce_overlap", "rccl"), ("views", "rccl"), ("views_overlap", "rccl"),  # library collectives only ...
             ("allreduce", "direct"), ("views", "direct"), ("views_overlap", "direct")]  # ... then the 1-hop all-to-all forms
# xGMI of an 8 x MI355X node (MI355X_MICROARCH.md / SURVEY.md 5): a full mesh, 7 links of ~153 GB/s per GPU
XGMI_LINKS, XGMI_LINK_GBS = 7, 153.0


def probe_order(sh_degree: int, everything: bool = False) -> list:
    """Order in which choose_exchange tries the strategies.  With SH colours (degree >= 1) the view forms go first -- they
    put 5x fewer bytes on the wire (exchange_bytes) and are the only ones whose model reaches 6x at 8 GPUs
    (exchange_model) -- then the plain all-reduce with its chunked overlap, the plain all-reduce last: it is the fallback
    whatever it costs.  Degree 0: the SH row IS the colour gradient, nothing to gain from the view form."""
    lib = [e for e in EXCHANGES if e[1] == "rccl"]
    if sh_degree >= 1:
        lib = [("views_overlap", "rccl"), ("views", "rccl"), ("allreduce_overlap", "rccl"), FALLBACK]
    else:
        lib = [("allreduce_overlap", "rccl"), FALLBACK, ("views", "rccl"), ("views_overlap", "rccl")]
    return lib + ([e for e in EXCHANGES if e[1] == "direct"] if everything else [])


def exchange_bytes(mode, world, cfg, crf_K=256):
    """Bytes one rank SENDS per step with an exchange strategy (ring / 1-hop all-reduce of n bytes: 2 (w-1)/w n; all-gather
    of n bytes per rank: (w-1) n) and the payload it is made of -- from the tensor sizes, not from a measurement."""
    P, W, H, deg, hdr, n_poses = cfg
    M = (deg + 1) ** 2
    other = P * (3 + 1 + 3 + 4) + (1 + 3 * crf_K if hdr else 0)          # means3D, opacity, scales, rotations, exposure, CRF table
    sh = P * M * 3
    if mode == "allreduce":
        payload = {"all_reduced_bytes": 4 * (other + sh), "all_gathered_bytes_per_rank": 0}
    else:
        payload = {"all_reduced_bytes": 4 * other, "all_gathered_bytes_per_rank": 4 * (n_poses * P * 3 + n_poses * 3)}
    sent = 2 * (world - 1) / world * payload["all_reduced_bytes"] + (world - 1) * payload["all_gathered_bytes_per_rank"]
    return {**payload, "sent_per_rank_bytes": int(sent)}


def exchange_model(mode, world, cfg, base_ms=None, crf_K=256):
    """Analytic cost of one gradient exchange on the xGMI mesh, so the measured probe can be checked against it (and so the
    scaling a strategy can reach is known before any 8-GPU run).  Per rank: an all-reduce of B bytes moves 2 (w-1)/w B, an
    all-gather of b bytes per rank (w-1) b.  Two bounds per collective: `ring` = one ring over ONE link per hop
    (2 (w-1)/w B / link for the all-reduce, (w-1) b / link for the all-gather) and `one_hop` = every peer over its own
    link at once (2 (B / w) / link; b / link) -- the library lands between them.  `hidden_under_ms`: kernel time of the same
    step the collective travels beside (the per-Gaussian backward, 0.13 ms per 1M Gaussians and pose at SH degree 3, for
    what starts after the record sums; the SH rebuild, 0.06 ms per 1M Gaussians and view, for the all-reduce of the view
    forms) -- r03 stage timings scaled by P.  exposed = what is left; `extra_kernel_ms` = the SH rebuild of the view forms
    minus the SH rows their backward no longer writes; expected scaling = w x base / (base + exposed + extra)."""
    P, W, H, deg, hdr, n_poses = cfg
    b = exchange_bytes("views" if mode.startswith("views") else "allreduce", world, cfg, crf_K)
    link = XGMI_LINK_GBS * 1e9
    w = max(world, 2)
    ar, ag = b["all_reduced_bytes"], b["all_gathered_bytes_per_rank"]
    t = {"all_reduce_ring_ms": 2 * (w - 1) / w * ar / link * 1e3, "all_reduce_one_hop_ms": 2 * (ar / w) / link * 1e3,
         "all_gather_ring_ms": (w - 1) * ag / link * 1e3, "all_gather_one_hop_ms": ag / link * 1e3}
    project_ms = 0.13 * P / 1e6 * n_poses * (0.4 + 0.6 * (3 * (deg + 1) ** 2) / 48.0)
    # hs_sh_backward_views over V = world x poses views: (12 V + 12 M) bytes per Gaussian at ~4.6 TB/s plus V basis evaluations
    rebuild_ms = (0.045 + 0.008 * world * n_poses) * P / 1e6 * ((deg + 1) ** 2 / 16.0) if deg >= 1 else 0.0
    saved_ms = 0.04 * P / 1e6 * ((deg + 1) ** 2 / 16.0) if deg >= 1 else 0.0   # the backward no longer writes the SH rows
    hide = {"allreduce": (0.0, 0.0), "allreduce_overlap": (project_ms * 0.75, 0.0),   # (under it: all-reduce, all-gather)
            "views": (rebuild_ms, 0.0), "views_overlap": (rebuild_ms, project_ms)}[mode]
    out = {**t, "hidden_under_ms": {"all_reduce": hide[0], "all_gather": hide[1]},
           "extra_kernel_ms": (rebuild_ms - saved_ms) if mode.startswith("views") else 0.0,
           "note": "bounds for 7 x 153 GB/s xGMI links per GPU: `ring` = one link per hop, `one_hop` = all peers at once; "
                   "UNMEASURED on hardware until an 8-GPU node runs this"}
    for algo in ("ring", "one_hop"):
        exposed = max(0.0, t[f"all_reduce_{algo}_ms"] - hide[0]) + max(0.0, t[f"all_gather_{algo}_ms"] - hide[1])
        # the view forms run the SH rebuild kernel instead of writing the SH rows in the backward (about the same bytes)
        out[f"exposed_{algo}_ms"] = exposed
        if base_ms:
            out[f"expected_scaling_{algo}"] = world * base_ms / (base_ms + exposed + out["extra_kernel_ms"])
    return out


def choose_exchange(step, state, barrier, dev, rank, world, backend, cfg):
    """Which gradient exchange the timed steps use (config c5).  HS_BENCH_EXCHANGE=<mode>/<algo> pins one and skips the
    probe.  Otherwise: measure, don't guess -- a few whole steps with each strategy on this node's links, safest first
    (plain library all-reduce of the flat gradient buffer, which also is the fallback), the fastest wins; MAX over ranks,
    and a strategy counts only if EVERY rank finished it, so all ranks decide alike.  Guards for an unattended run:
      * every probe step is timed on its own; a strategy whose first step takes more than 20 x the step without any
        exchange (HS_BENCH_PROBE_CAP_X) gets one second step -- a collective's first use pays one-time set-up -- and is
        dropped on all ranks if that is over the cap too, or at once beyond 50 x the cap (or if it raises, or -- the 1-hop
        forms -- does not reproduce dist.all_reduce on a test vector), before it can cost more.  (Over gloo on a shared GPU -- the plumbing check -- a
        236 MB exchange takes 50-330 ms and the cap drops everything but the fallback; over xGMI it is 1-3 ms);
      * the probe covers the library-collective strategies only (all-reduce, all-reduce + all-gather); the 1-hop
        all-to-all forms, which no multi-GPU node has run yet, join it with HS_BENCH_PROBE=all.
    "allreduce" = all-reduce of the flat per-Gaussian gradient buffer; "allreduce_overlap" = the same bytes, issued chunk by
    chunk from inside the backward while its per-Gaussian half computes the next chunk; "views" = all-reduce of the non-SH part +
    all-gather of per-view colour gradients, SH gradient rebuilt locally; "views_overlap" = the same with the all-gather
    started inside the backward, under its per-Gaussian half; "rccl" / "direct" = library ring vs 1-hop all-to-all."""
    import torch
    import torch.distributed as dist
    from casualhdrsplat_amd.distributed import validate_direct

    def agree(vals):   # MAX over ranks of a few floats
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def timed_steps(n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        barrier()
        return (time.perf_counter() - t0) / n * 1e3

    info = {"backend": backend,
            # collectives one step puts on the wire per rank: the flat buffer is ONE all-reduce; its chunked
            # overlap one coalesced all-reduce per chunk (4; distributed.LAST_EXCHANGE counts what was actually issued, below);
            # the view forms one all-reduce of the non-SH gradients + two all-gathers (colour gradients, camera centres)
            "collectives_per_step": {"allreduce/rccl": 1, "allreduce_overlap/rccl": 4, "views/rccl": 3, "views_overlap/rccl": 3,
                                     "allreduce/direct": 3, "views/direct": 5, "views_overlap/direct": 5},
            "bytes": {f"{m}/{a}": exchange_bytes("views" if m.startswith("views") else "allreduce", world, cfg)
                      for m, a in EXCHANGES if a == "rccl"},
            "model_ms": {m: exchange_model(m, world, cfg) for m, a in EXCHANGES if a == "rccl"}}
    pinned = os.environ.get("HS_BENCH_EXCHANGE", "").strip()
    if pinned:
        mode, _, algo = pinned.partition("/")
        if (mode, algo) not in EXCHANGES:
            raise SystemExit(f"HS_BENCH_EXCHANGE={pinned!r}: expected one of {['/'.join(e) for e in EXCHANGES]}")
        state["exchange"] = (mode, algo)
        info.update(choice=pinned, pinned=True, step_ms={})
        return info
    # the step without any exchange: the yardstick of the per-strategy cap
    state["exchange"] = ("none", "rccl")
    step()
    base_ms = agree([timed_steps(2)])[0]
    cap_ms = float(os.environ.get("HS_BENCH_PROBE_CAP_X", "20")) * base_ms
    info["model_ms"] = {m: exchange_model(m, world, cfg, base_ms) for m, a in EXCHANGES if a == "rccl"}
    candidates = probe_order(cfg[3], os.environ.get("HS_BENCH_PROBE") == "all")
    direct_ok = None
    times, dropped, first_steps = {}, {}, {}
    for mode, algo in candidates:
        name = f"{mode}/{algo}"
        state["exchange"] = (mode, algo)
        ok, first_ms, avg_ms, err = 1.0, 0.0, 0.0, ""
        try:  # a strategy the backend cannot run must not take the run down with it
            if algo == "direct":
                if direct_ok is None:
                    direct_ok = validate_direct(dev)
                if not direct_ok:
                    raise RuntimeError("all_reduce_direct does not reproduce dist.all_reduce on this backend")
            first_ms = timed_steps(1)
        except RuntimeError as e:
            ok, err = 0.0, str(e)[:200]
        first_ms, bad = agree([first_ms, 1.0 - ok])
        first_steps[name] = [round(first_ms, 3)]      # (what a strategy is kept or dropped on: MAX over ranks, ms)
        # (the plain library all-reduce is the fallback whatever it costs: only a failure removes it)
        over = first_ms > cap_ms and (mode, algo) != FALLBACK
        if over and not bad and first_ms <= 50.0 * cap_ms:
            # the first use of a collective pays one-time costs (RCCL sets its channels up lazily): one second chance,
            # decided on the agreed MAX like everything else, so every rank takes the same branch
            second_ms = 0.0
            try:
                second_ms = timed_steps(1)
            except RuntimeError as e:
                ok, err = 0.0, str(e)[:200]
            second_ms, bad = agree([second_ms, 1.0 - ok])
            first_steps[name].append(round(second_ms, 3))
            over = second_ms > cap_ms
            if over:
                err = err or f"first two steps {first_ms:.1f}, {second_ms:.1f} ms > cap {cap_ms:.1f} ms"
        if bad or over:
            dropped[name] = err or ("failed on another rank" if bad else f"first step {first_ms:.1f} ms > cap {cap_ms:.1f} ms")
            continue
        try:
            avg_ms = timed_steps(3)
        except RuntimeError as e:
            ok, err = 0.0, str(e)[:200]
        avg_ms, bad = agree([avg_ms, 1.0 - ok])
        if bad:
            dropped[name] = err or "failed on another rank"
            continue
        times[name] = avg_ms
        if mode == "allreduce_overlap":
            from casualhdrsplat_amd import distributed as D_
            info["collectives_per_step"][name] = int(D_.LAST_EXCHANGE["collectives"])   # counted, not assumed
    best = min(times, key=times.get) if times else "allreduce/rccl"
    state["exchange"] = tuple(best.split("/"))
    if rank == 0:
        print(f"[bench] exchange probe: step without exchange {base_ms:.3f} ms, first-step cap {cap_ms:.1f} ms; first step(s) per "
              f"strategy (ms, MAX over ranks) {first_steps}; kept {times}; dropped {dropped}; chosen {best}", file=sys.stderr)
    info.update(choice=best, step_ms=times, dropped=dropped, no_exchange_step_ms=base_ms, first_step_cap_ms=cap_ms,
                first_step_ms=first_steps)
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(c for c in CONFIGS if c != "c1"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-c2", action="store_true",
                    help="skip BASELINE config c2 in full on the CPU (BASELINE.md 3: 'config 2 if it completes in < 10 min'; "
                         "it takes ~2.5 min on the 128 host cores of the MI355X box and is part of the default run)")
    ap.add_argument("--cpu-c2", action="store_true", help="(kept for older command lines: c2 in full is the default now)")
    ap.add_argument("--no-extras", action="store_true", help="profiling runs: one seed only, no CPU baseline")
    ap.add_argument("--one-seed", action="store_true", help="time seed 0 only (the headline is then NOT the median of seeds)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default c3 runs also time BASELINE configs c2 and c4 once each (other_configs, ~10 s): skip that")
    ap.add_argument("--kernel-iters", type=int, default=10)
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the whole step as one HIP graph (single GPU).  auto = only where the host can pace the "
                         "step -- eager steps shorter than 0.6 ms (c2) -- and only if the replay reproduces the eager step "
                         "bit for bit; on = also for the long steps of c3 / c4 (measured: 0.2 %% there)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback exists)")
    import torch.distributed as dist
    from casualhdrsplat_amd.distributed import init_from_env
    # HS_BENCH_BACKEND=gloo is a plumbing check for boxes with fewer GPUs than ranks (ranks then share devices);
    # measured runs use "nccl" (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("HS_BENCH_BACKEND", "nccl")
    if int(os.environ.get("WORLD_SIZE", "1")) > torch.cuda.device_count() and backend == "nccl":
        raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible (RCCL needs one per rank; "
                         "HS_BENCH_BACKEND=gloo shares devices for a plumbing check)")
    rank, world, local = init_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    exchange = {"info": None, "choice": None}   # the gradient exchange is chosen once (first seed) and reused

    def bench_config(cfg_name, seeds, steps, warmup):
        """Time one BASELINE config (median over `seeds`, each by the same procedure) and, on rank 0, its per-stage /
        roofline legs on the median seed's scene.  Returns (line, scene, cfg)."""
        cfg = CONFIGS[cfg_name]
        P, W, H, deg, hdr, n_poses = cfg

        def prepare_seed(seed):
            """Everything ahead of the timed region, identical for every seed: scene, one synchronous step (learns num_rendered,
            as the published API does), rasterizers in the sync-free mode with a fixed binning capacity (25 % headroom; overflow
            checked lazily every step), the exchange strategy (N > 1) and the launch form (eager or one HIP graph)."""
            step, state, make_rasterizer, sc, dL, plist = build_step(cfg, rank, world, dev, seed=seed)
            out = step()
            torch.cuda.synchronize()
            counts = derived_counts(out, W, H, n_poses)
            del out
            state["out"] = None
            state["rast"] = make_rasterizer(int(counts[0] * 1.25) + 4096)
            if world > 1:
                if exchange["choice"] is None:
                    exchange["info"] = choose_exchange(step, state, barrier, dev, rank, world, backend, cfg)
                    exchange["choice"] = state["exchange"]
                state["exchange"] = exchange["choice"]
            # One launch per step: the sync-free step (fixed binning capacity: no host read in forward or backward) is captured
            # in a HIP graph and replayed -- the same kernels on the same buffers, minus ~40 launches of host work per step (at
            # c2 the host, not the GPU, paces the eager step).  Used only if the replay reproduces the eager step bit for bit.
            launch, run_step, gstep = "eager (one enqueue per kernel)", step, None
            want_graph = args.graph == "on"
            probe = {"graph_flag": args.graph}
            if world == 1 and args.graph == "auto":
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                probe_ms = (time.perf_counter() - t0) / 10 * 1e3
                want_graph = probe_ms < 0.6   # a step this short is at the mercy of the box's CPU
                # what the probe saw and why it chose (two boxes chose differently at c2 -- the eager
                # step there is paced by the host, 0.3 ms of enqueue work, so either form times the same GPU work)
                probe.update({"eager_probe_ms_per_step": round(probe_ms, 4), "threshold_ms": 0.6,
                              "rule": "hip_graph iff 10 eager steps average below the threshold (host-paced step) and the "
                                      "replay reproduces the eager step bit for bit",
                              "wanted": "hip_graph" if want_graph else "eager"})
            if world == 1 and want_graph:
                try:
                    from casualhdrsplat_amd.graphs import GraphedStep
                    step()
                    torch.cuda.synchronize()
                    want = [state["out"][0].detach().clone()] + [p_.grad.detach().clone() for p_ in plist]
                    # no autograd graph of an earlier (default-stream) step may be alive when the capture starts: its
                    # AccumulateGrad nodes would run on the default stream and break the capture
                    state["out"] = None
                    for p_ in plist:
                        p_.grad = None
                    gstep = GraphedStep(step, [state["rast"]["allreduce"]])
                    gstep.step()
                    gstep.check_overflow()
                    got = [state["out"][0].detach()] + [p_.grad.detach() for p_ in plist]
                    if not all(torch.equal(a_, b_) for a_, b_ in zip(want, got)):
                        raise RuntimeError("graph replay differs from the eager step")
                    launch, run_step = "hip_graph (whole step captured once, replayed)", gstep.step
                    probe["graph_bit_identical"] = True
                except Exception as e:  # noqa: BLE001 -- any capture problem: the eager step is always there
                    probe["graph_rejected"] = f"{type(e).__name__}: {str(e)[:120]}"
                    if args.graph == "on":
                        raise
                    print(f"[bench] HIP-graph step unavailable ({type(e).__name__}: {str(e)[:200]}); timing the eager step",
                          file=sys.stderr)
                    gstep, run_step = None, step
                    for p_ in plist:
                        p_.grad = None
            return dict(step=step, state=state, sc=sc, dL=dL, plist=plist, counts=counts, launch=launch, run_step=run_step,
                        gstep=gstep, seed=seed, probe=probe)

        def time_seed(seed):
            """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize, MAX over ranks: the same
            procedure for every seed of SURVEY.md 8(d); returns (context, ms per step)."""
            ctx = prepare_seed(seed)
            for _ in range(warmup):
                ctx["run_step"]()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                ctx["run_step"]()
            barrier()
            elapsed = time.perf_counter() - t0
            if ctx["gstep"] is not None:
                ctx["gstep"].check_overflow()   # the timed frames all fitted their binning capacity
            if world > 1:
                t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            return ctx, elapsed / steps * 1e3

        # "seeds {0,1,2}, report median".  Every seed is timed by the same function -- same warm-up, same K
        # steps, a fresh scene, fresh rasterizers, an emptied allocator cache -- and the headline is the MEDIAN seed; the
        # per-stage / roofline legs below then run on that seed's scene.
        per_seed, ctx = {}, None
        for seed in seeds:
            ctx = None                       # (the previous seed's scene and state buffers go before the next is built)
            torch.cuda.empty_cache()
            ctx, ms = time_seed(seed)
            per_seed[seed] = {"ms_per_step": ms, "R": ctx["counts"][0], "R_prime": ctx["counts"][1]}
        order = sorted(seeds, key=lambda s_: per_seed[s_]["ms_per_step"])
        med_seed = order[len(order) // 2]
        ms_per_step = per_seed[med_seed]["ms_per_step"]
        if ctx["seed"] != med_seed and rank == 0 and world == 1:
            ctx = None
            torch.cuda.empty_cache()
            ctx = prepare_seed(med_seed)     # the median seed's scene for the per-stage legs (not timed again)
        step, state, sc, dL, plist, launch = ctx["step"], ctx["state"], ctx["sc"], ctx["dL"], ctx["plist"], ctx["launch"]
        R, Rp, E, vtiles = ctx["counts"]
        allreduce_info = exchange["info"]
        images_per_s = world * 1e3 / ms_per_step

        line = {
            "metric": "train-step images/sec (fwd+bwd raster) @ 1M Gaussians 1080p",
            "value": images_per_s, "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_name}: {P} Gaussians, {W}x{H}, SH degree {deg}, "
                                   f"{'HDR radiance + CRF tone-map' if hdr else 'LDR'}, {n_poses} pose(s)/view, "
                                   f"{world} view(s)/step (one per GPU)" + (", gradients summed over views (config.gradient_exchange)" if world > 1 else ""),
                       "num_rendered_R": R, "R_prime": Rp, "pixel_pair_evals_E": E,
                       "seed": ctx["seed"],
                       "seeds": f"value / ms_per_step = the median of seeds {seeds}, each timed alike ({warmup} warm-up + "
                                f"{steps} steps); roofline / stages / counts belong to seed {ctx['seed']}",
                       "binning": "sync-free fixed capacity 1.25*R", "launch": launch, "launch_probe": ctx["probe"]},
            "mpix_per_s": world * W * H * n_poses / ms_per_step / 1e3,
            "seeds_ms_per_step": {**{str(k): v["ms_per_step"] for k, v in per_seed.items()},
                                  **{f"R_{k}": v["R"] for k, v in per_seed.items()},
                                  **{f"R_prime_{k}": v["R_prime"] for k, v in per_seed.items()},
                                  "median": ms_per_step, "median_seed": med_seed, "median_images_per_s": images_per_s,
                                  "spread": (max(v["ms_per_step"] for v in per_seed.values()) /
                                             min(v["ms_per_step"] for v in per_seed.values()) - 1.0)},
        }
        if allreduce_info is not None:
            line["config"]["gradient_exchange"] = allreduce_info

        if rank == 0:
            from casualhdrsplat_amd import _lib as L
            from casualhdrsplat_amd.rasterizer import render_stats, replay_backward, replay_forward
            # a fresh forward whose autograd graph is kept (never .backward()-ed) so its stages can be replayed
            for p_ in plist:
                p_.grad = None
            out = state["rast"]["allreduce"](*[plist[i] for i in (0, 1, 2)], shs=plist[3], scales=plist[4], rotations=plist[5])
            R, Rp, E, vtiles = derived_counts(out, W, H, n_poses)
            WH = W * H * n_poses
            it = args.kernel_iters
            # order matters: the binning replay clears the pair flags the backward sets, so every backward stage is timed
            # first, on the state a real step leaves behind (forward -> render bwd -> segmented sum + preprocess bwd)
            bwd_ms, bwd_med = time_stage(lambda: replay_backward(out[0], dL, L.HS_BWD_RENDER), it)
            stages = {"render_bwd": bwd_ms}
            stages["segsum_and_preprocess_bwd"] = time_stage(lambda: replay_backward(out[0], dL, L.HS_BWD_PREPROCESS), it)[0]
            # (the segmented sum on its own as well, HERE: it reads the records whose flags the render backward set -- after
            # a binning replay has cleared them it finds nothing to add and takes 30 us instead of 80)
            stages["pair_segsum"] = time_stage(lambda: replay_backward(out[0], dL, L.HS_BWD_SEGSUM), it)[0]
            if hdr:
                stages["crf_gradient"] = time_stage(lambda: replay_backward(out[0], dL, L.HS_BWD_CRF), it)[0]
            fwd_ms, fwd_med = time_stage(lambda: replay_forward(out[0], L.HS_STAGE_RENDER), it)
            stages["render_fwd"] = fwd_ms
            stats = render_stats(out[0], dL)
            stages["binning"] = time_stage(lambda: replay_forward(out[0], L.HS_STAGE_BIN), it)[0]
            stages["preprocess_fwd_and_binning"] = time_stage(
                lambda: replay_forward(out[0], L.HS_STAGE_PREPROCESS | L.HS_STAGE_BIN), it)[0]
            # the preprocess kernel exactly as a single-enqueue forward runs it (it carries the binning stage's prologue), and
            # the two kernels of the backward's last stage on their own
            stages["preprocess_fwd"] = time_stage(
                lambda: replay_forward(out[0], L.HS_STAGE_PREPROCESS | L.HS_STAGE_BIN | L.HS_STAGE_PREPROCESS_ONLY), it)[0]
            stages["binning_in_step"] = stages["preprocess_fwd_and_binning"] - stages["preprocess_fwd"]
            stages["preprocess_bwd"] = time_stage(lambda: replay_backward(out[0], dL, L.HS_BWD_PROJECT), it)[0]
            # (leave the state as a step leaves it for whatever follows: the binning replays cleared the pair flags)
            replay_forward(out[0], L.HS_STAGE_RENDER)
            bytes_bwd = 76 * Rp + 20 * WH
            bytes_fwd = 40 * Rp + 20 * WH + 8 * vtiles
            ach = bytes_bwd / (bwd_ms * 1e-3) / 1e9
            from casualhdrsplat_amd import inspect_state
            ist = inspect_state(out[0])
            n_visible = int((ist["radii"] > 0).sum())
            line["config"]["tile_sort"] = {0: "radix passes over (tile, instance) pairs", 1: "counting (small frame)",
                                           2: "hierarchical: one pass over (super-tile, instance) elements + expansion"}[int(ist["tile_sort"])]
            # which depth sort the frame got: by counting below 2^21 instances unless the process or the
            # environment asked for the look-back passes
            ds_env = os.environ.get("HS_DEPTH_SORT", "")[:1]
            by_counting = cfg[0] * cfg[5] < (2 << 20) and (ds_env == "m" or (ds_env != "l" and L.load().hs_depth_sort(-1) == 1))
            line["config"]["depth_sort"] = ("counting pass over the top 12 varying key bits + range sorts in LDS" if by_counting
                                            else "look-back radix passes over the varying key bits")
            line["config"]["depth_sort_ranges_off_chip"] = int(ist["depth_slow_ranges"])
            step_bytes = whole_step_bytes(cfg, R, Rp, vtiles, n_visible)
            step_gbs = step_bytes["total"] / (ms_per_step * 1e-3) / 1e9
            line["roofline"] = {
                "kernel": "render_bwd_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "frac_hbm_measured": ach / HBM_MEASURED_GBS, "peak_hbm_measured": HBM_MEASURED_GBS,
                "whole_step": {"algorithmic_bytes": step_bytes, "ms_per_step": ms_per_step, "achieved": step_gbs,
                               "frac": step_gbs / HBM_PEAK_GBS, "frac_hbm_measured": step_gbs / HBM_MEASURED_GBS,
                               "note": "sum of SURVEY.md 8(d)'s per-stage algorithmic bytes / the timed step (all kernels, "
                                       "host gaps included)"},
                "traffic": None,  # PMC counters cannot be read inside a timed run: see `offline_profile`
                "algorithmic_bytes": bytes_bwd, "avg_ms": bwd_ms, "median_ms": bwd_med,
                "note": "the kernel is VALU-issue bound (roofline.valu; DESIGN.md 4), not HBM bound",
                "fwd_bwd": {"kernels": "render_fwd_kernel + render_bwd_kernel", "algorithmic_bytes": bytes_fwd + bytes_bwd,
                            "avg_ms": fwd_ms + bwd_ms,
                            "achieved": (bytes_fwd + bytes_bwd) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9,
                            "frac": (bytes_fwd + bytes_bwd) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "frac_hbm_measured": (bytes_fwd + bytes_bwd) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9 / HBM_MEASURED_GBS,
                            "pair_evals_per_s": 2 * E / ((fwd_ms + bwd_ms) * 1e-3)},
                "valu": valu_roofline(stats, isa_counts(), fwd_ms, bwd_ms),
            }
            # The HBM-bound stages one by one: HIP-event time of the stage on the launch stream against the
            # algorithmic bytes of SURVEY.md 8(d) (the segmented sum is this implementation's own stage: 36 B per record the
            # render backward wrote + 40 B per instance sum), as fractions of the 8 TB/s peak and of the measured copy rate
            M_ = (deg + 1) ** 2
            I_ = P * n_poses
            in_row = 12 + 12 + 16 + 4 + 12 * M_
            pk_bytes = {
                "preprocess_fwd": step_bytes["preprocess_fwd"],
                "binning_in_step": step_bytes["scan"] + step_bytes["duplicate_with_keys"] + step_bytes["sort_one_ideal_pass"] + step_bytes["tile_ranges"],
                "render_fwd": bytes_fwd, "render_bwd": bytes_bwd,
                "pair_segsum": 36 * Rp + 40 * I_,
                "preprocess_bwd": in_row * P + 40 * I_ + in_row * P,   # row read once per Gaussian + 40 B per instance sum + row written
            }
            if hdr:
                pk_bytes["crf_gradient"] = 2 * 12 * WH
            line["roofline"]["per_kernel"] = {
                k: {"ms": stages[k], "algorithmic_bytes": int(bts), "achieved_gbs": bts / (stages[k] * 1e-3) / 1e9,
                    "frac": bts / (stages[k] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "frac_hbm_measured": bts / (stages[k] * 1e-3) / 1e9 / HBM_MEASURED_GBS}
                for k, bts in pk_bytes.items() if stages.get(k, 0) > 0}
            # a stage "faster" than the measured copy rate means its byte model is wrong, not that the kernel is fast
            over = {k: round(v["frac_hbm_measured"], 3) for k, v in line["roofline"]["per_kernel"].items() if v["frac_hbm_measured"] > 1.0}
            if over:
                raise SystemExit(f"[bench] per_kernel.frac_hbm_measured > 1.0 for {over} at {cfg_name}: the algorithmic byte "
                                 "model charges bytes the kernel cannot have moved -- fix whole_step_bytes / pk_bytes")
            non_render = [k for k in pk_bytes if not k.startswith("render")]
            nr_ms, nr_b = sum(stages[k] for k in non_render), sum(pk_bytes[k] for k in non_render)
            line["roofline"]["non_render"] = {"stages": non_render, "ms": nr_ms, "algorithmic_bytes": int(nr_b),
                                              "frac": nr_b / (nr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "frac_hbm_measured": nr_b / (nr_ms * 1e-3) / 1e9 / HBM_MEASURED_GBS}
            # ... and as the step sees them: the whole step minus the two render launches timed alone.  Smaller than
            # the sum of the stages: inside the step the CRF gradient's first stage runs as the last workgroups of
            # render_bwd_kernel's launch (`crf_gradient` above is the stand-alone launch), and a stage replayed alone pays a
            # launch latency the captured step does not
            in_step_ms = ms_per_step - fwd_ms - bwd_ms
            if in_step_ms > 0:
                line["roofline"]["non_render"].update({
                    "ms_in_step": in_step_ms, "frac_in_step": nr_b / (in_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "in_step_note": "ms_per_step (the seed the stages were timed on) - render_fwd - render_bwd stage times"})
            off = offline_profile(cfg_name)
            if off is not None:
                line["roofline"]["offline_profile"] = off
                if off.get("same_kernel_source") and off.get("render_bwd_kernel_hbm_bytes"):
                    # counters of an earlier rocprofv3 --pmc pass over this same workload, taken from kernels compiled
                    # from the very render.hip that is loaded now (hash checked); otherwise `traffic` stays null
                    # The launch of a whole step carries the CRF gradient's first stage as its last workgroups (HDR): those
                    # read dL/dLDR and the radiance image(s), 24 B per pixel -- on the algorithmic side of THIS comparison
                    # (VERDICT r5 weak #4).  `traffic_tile_replay`: the same kernel launched by HS_BWD_RENDER alone (no tail),
                    # to be set against `algorithmic_bytes` = 76 R' + 20 W H directly.
                    tail_b = 24 * WH if hdr else 0
                    line["roofline"]["traffic"] = off["render_bwd_kernel_hbm_bytes"]
                    line["roofline"]["traffic_algorithmic_bytes"] = bytes_bwd + tail_b
                    line["roofline"]["traffic_over_algorithmic"] = off["render_bwd_kernel_hbm_bytes"] / (bytes_bwd + tail_b)
                    if off.get("render_bwd_kernel_tile_replay_hbm_bytes"):
                        line["roofline"]["traffic_tile_replay"] = off["render_bwd_kernel_tile_replay_hbm_bytes"]
                        line["roofline"]["traffic_tile_replay_over_algorithmic"] = off["render_bwd_kernel_tile_replay_hbm_bytes"] / bytes_bwd
                    line["roofline"]["traffic_note"] = (
                        "bytes per launch at the L2-fabric interface from profiles/pmc_traffic.json (separate --pmc passes, gfx950 "
                        "FETCH_SIZE correction), same render.hip.  `traffic` = the launch of a whole step, which in HDR mode ends with "
                        "the CRF gradient's first-stage workgroups (+ 24 W H algorithmic bytes: traffic_algorithmic_bytes); "
                        "`traffic_tile_replay` = the kernel launched without that tail, against algorithmic_bytes = 76 R' + 20 W H")
            line["stages_ms"] = {k: round(v, 4) for k, v in stages.items()}
            line["render_stats"] = stats
        return line, sc, cfg

    seeds_main = [0] if (args.one_seed or args.no_extras) else [0, 1, 2]
    line, sc, cfg = bench_config(args.config, seeds_main, args.steps, args.warmup)
    if rank == 0:
        # BASELINE configs[1] and configs[3] on the driver's record too: after the c3 legs the default
        # run times c2 and c4 once each -- seed 0, the same time_seed procedure -- and prints their step, stages and roofline
        if world == 1 and args.config == "c3" and not (args.no_extras or args.one_seed or args.no_other_configs):
            other = {}
            for name, k_steps in (("c2", max(args.steps, 50)), ("c4", max(5, min(args.steps, 12)))):
                torch.cuda.empty_cache()
                o, _, _ = bench_config(name, [0], k_steps, args.warmup)
                other[name] = {"ms_per_step": o["ms_per_step"], "images_per_s": o["value"], "mpix_per_s": o["mpix_per_s"],
                               "steps": k_steps, "warmup": args.warmup, "seed": 0, "workload": o["config"]["workload"],
                               "launch": o["config"]["launch"], "launch_probe": o["config"]["launch_probe"], "num_rendered_R": o["config"]["num_rendered_R"],
                               "R_prime": o["config"]["R_prime"], "stages_ms": o["stages_ms"],
                               "roofline": {k: o["roofline"][k] for k in ("kernel", "achieved", "frac", "frac_hbm_measured",
                                                                          "algorithmic_bytes", "avg_ms")},
                               "roofline_fwd_bwd_frac": o["roofline"]["fwd_bwd"]["frac"],
                               "roofline_whole_step_frac": o["roofline"]["whole_step"]["frac"],
                               "per_kernel": o["roofline"]["per_kernel"]}
            line["other_configs"] = other
            torch.cuda.empty_cache()
        if world == 1 and not (args.no_cpu_baseline or args.no_extras):
            line["cpu_baseline"] = cpu_baseline(sc, cfg, with_c2=not args.no_cpu_c2)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
