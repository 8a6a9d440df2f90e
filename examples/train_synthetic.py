#!/usr/bin/env python3
"""The loop a CasualHDRSplat-style trainer runs around the rasterizer, on synthetic captures (no dataset needed).

Ground truth: a synthetic Gaussian cloud, a camera moving along a cubic SE(3) B-spline, one exposure time per captured frame
(which scales the radiance AND sets the length of the exposure window, i.e. the blur), a camera response curve.  The
observations are the blurred LDR frames `image_formation.HDRBlurFormation` renders from that.  The run then starts from
perturbed radiance and opacities, wrong exposure times and a wrong trajectory, and optimises all of them jointly with
Adam against the observations -- what /root/reference/Readme.md:54 describes ("jointly estimating exposure time with
camera motion") -- through the HIP kernels: every step is frames x (N-pose forward + backward).

    python examples/train_synthetic.py --steps 300

Gauge: exposure x radiance x response is determined only up to a common factor, so the response curve and the first frame's
exposure are held at their true values (a real capture pins them with EXIF exposure ratios or a calibrated response).
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from casualhdrsplat_amd import synthetic as S
from casualhdrsplat_amd.graphs import GraphedStep
from casualhdrsplat_amd.image_formation import (FrameRasterizers, HDRBlurFormation, ImplicitCRF, TrajectorySpline,
                                                  knots_from_lookat)

CLOUD = ("means3D", "opacities", "shs", "scales", "rotations")


def mean_by_rows(x: torch.Tensor) -> torch.Tensor:
    """x.mean() as two small reductions (rows of 256, then the row sums).  Inside THIS captured step a plain .mean() / .sum()
    over more than a few ten thousand elements -- PyTorch's two-pass kernel -- reads a wrong value from the second replay on
    (23.6 instead of 0.0406: scripts/repro/graph_step_mean.py).  Round 5 blamed a memset node; round 6's reproducers show
    that neither HIP's memset nodes nor this reduction fail in isolation and that the step fails with a torch-only stand-in
    for the rasterizer too: the capture of the long torch step, not the reduction, not this library.  The
    gradient of a mean does not depend on its value, so training was right all along; the printed numbers were not."""
    if os.environ.get("HS_EXAMPLE_PLAIN_MEAN"):      # (scripts/repro/graph_step_mean.py: the reduction this function avoids)
        return x.mean()
    n = x.numel()
    pad = (-n) % 256
    flat = x.reshape(-1)
    if pad:
        flat = torch.cat([flat, flat.new_zeros(pad)])
    return flat.reshape(-1, 256).sum(dim=1).sum() / n


def run(P=20000, W=320, H=208, frames=4, virtual=5, steps=200, seed=0, deg=1, log_every=25, device="cuda", quiet=False,
        graph=False, capacity=None):
    """Returns a dict of the run's first / last loss, PSNR and parameter errors (also what the GPU test checks)."""
    dev = torch.device(device)
    sc = S.make_scene(P, W, H, deg, seed=seed, hdr=True)
    cam = sc.camera
    torch.manual_seed(seed)
    knots = knots_from_lookat(frames + 3, radius=0.25)
    dt_true = torch.tensor([1.0, 0.5, 1.6, 0.8, 1.3, 0.6, 1.1, 0.9])[:frames]

    def formation(crf, **kw):
        traj = TrajectorySpline(knots, kind="cubic")
        return HDRBlurFormation(traj, frames, W, H, cam.tanfovx, cam.tanfovy, n_virtual=virtual, crf=crf, sh_degree=deg,
                                window_from_exposure=True, window_scale=0.6, **kw).to(dev)

    truth = formation(ImplicitCRF(K=128))
    with torch.no_grad():
        truth.log_exposure.copy_(dt_true.log())
        gen = torch.Generator().manual_seed(seed + 1)
        truth.trajectory.delta.copy_(0.004 * torch.randn(truth.trajectory.delta.shape, generator=gen))   # the true motion is not the prior
        cloud_true = {k: getattr(sc, k).to(dev) for k in CLOUD}
        targets = [truth(i, *[cloud_true[k] for k in CLOUD])[0] for i in range(frames)]

    # the learner: the response curve is given, everything else starts off
    # --graph: one persistent sync-free rasterizer per captured frame, so that the step's gradient computation can be
    # recorded once and replayed (graphs.GraphedStep): no host time for the few thousand tiny kernels of the pose arithmetic
    per_frame = FrameRasterizers(capacity=capacity or 40 * P * virtual) if graph else None
    model = formation(ImplicitCRF(K=128), **({"rasterizer_factory": per_frame} if graph else {}))
    model.crf.load_state_dict(truth.crf.state_dict())
    for p_ in model.crf.parameters():
        p_.requires_grad_(False)
    gen = torch.Generator().manual_seed(seed + 2)
    shs0 = sc.shs.clone()
    shs0[:, 0] += 0.25 * torch.randn(shs0[:, 0].shape, generator=gen)
    raw_opac0 = torch.logit(sc.opacities.clamp(1e-3, 1 - 1e-3)) + 0.5 * torch.randn(sc.opacities.shape, generator=gen)
    shs = shs0.to(dev).requires_grad_(True)
    raw_opac = raw_opac0.to(dev).requires_grad_(True)
    fixed = {k: cloud_true[k] for k in ("means3D", "scales", "rotations")}
    with torch.no_grad():
        model.log_exposure[0] = truth.log_exposure[0]       # the gauge (see the module docstring)
    opt = torch.optim.Adam([
        {"params": [shs], "lr": 1e-2}, {"params": [raw_opac], "lr": 2e-2},
        {"params": [model.log_exposure], "lr": 1e-2}, {"params": [model.trajectory.delta], "lr": 5e-4}])

    def errors():
        with torch.no_grad():
            e_dt = float((model.log_exposure[1:] - truth.log_exposure[1:]).abs().mean()) if frames > 1 else 0.0
            e_pose = float((model.trajectory.knots()[:, :3, 3] - truth.trajectory.knots()[:, :3, 3]).norm(dim=1).mean())
            return e_dt, e_pose

    learn = [shs, raw_opac, model.log_exposure, model.trajectory.delta]

    def gradients():
        """Forward of every frame + backward of the summed loss; returns (per-frame losses, per-frame MSE) as tensors."""
        for p_ in learn:
            p_.grad = None
        # one pass over the spline for all frames (its few hundred tiny tensor operations are the step's host cost), one
        # rasterizer call per frame, one backward of the summed loss
        cams = model.cameras_all()
        opac = torch.sigmoid(raw_opac)
        losses, mses = [], []
        for i in range(frames):
            ldr, _, _, _ = model(i, fixed["means3D"], opac, shs, fixed["scales"], fixed["rotations"], cameras=cams)
            losses.append(mean_by_rows((ldr - targets[i]).abs()))
            mses.append(mean_by_rows((ldr.detach() - targets[i]) ** 2))
        torch.stack(losses).sum().backward()
        return torch.stack([l_.detach() for l_ in losses]), torch.stack(mses)

    step_fn = gradients
    if graph:
        gradients()                                   # (creates the per-frame rasterizers)
        captured = GraphedStep(gradients, per_frame.rasterizers(frames), params=learn)

        def step_fn():
            out = captured.step()
            for p_, g_ in zip(learn, captured.grads):   # (the optimizer reads .grad: the graph's static gradient tensors)
                p_.grad = g_
            return out

    hist = []
    t0 = time.time()
    for it in range(steps + 1):
        losses, mses = step_fn()
        if graph and (it % 50 == 0 or it == steps):
            captured.check_overflow()
        total = float(losses.sum())
        ps = float((-10.0 * torch.log10(mses.clamp_min(1e-12))).sum())
        if it < steps:
            g0 = model.log_exposure.grad
            if g0 is not None:
                g0[0] = 0.0                                    # frame 0's exposure is the gauge
            opt.step()
        e_dt, e_pose = errors()
        hist.append(dict(step=it, loss=total / frames, psnr=ps / frames, exposure_log_err=e_dt, knot_pos_err=e_pose))
        if not quiet and (it % log_every == 0 or it == steps):
            print(f"step {it:4d}  L1 {total / frames:.5f}  PSNR {ps / frames:6.2f} dB  |log dt - truth| {e_dt:.4f}  "
                  f"knot position error {e_pose:.5f}  ({(time.time() - t0) / max(it, 1) * 1e3:.1f} ms/step)")
    return dict(first=hist[0], last=hist[-1], history=hist)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--P", type=int, default=20000)
    ap.add_argument("--W", type=int, default=320)
    ap.add_argument("--H", type=int, default=208)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--virtual", type=int, default=5, help="virtual poses per captured frame")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--deg", type=int, default=1)
    ap.add_argument("--graph", action="store_true", help="record the gradient computation once as a HIP graph and replay it")
    a = ap.parse_args(argv)
    r = run(a.P, a.W, a.H, a.frames, a.virtual, a.steps, a.seed, a.deg, graph=a.graph)
    f, l = r["first"], r["last"]
    print(f"loss {f['loss']:.5f} -> {l['loss']:.5f}; PSNR {f['psnr']:.2f} -> {l['psnr']:.2f} dB; exposure error "
          f"{f['exposure_log_err']:.4f} -> {l['exposure_log_err']:.4f}; knot error {f['knot_pos_err']:.5f} -> {l['knot_pos_err']:.5f}")


if __name__ == "__main__":
    main()
