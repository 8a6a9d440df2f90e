"""Bisecting the in-situ hazard (scripts/repro/graph_step_mean.py reproduces it): which ingredient of a captured step makes
a plain .mean() read wrong from the second replay on?  Every variant computes loss = (image - target).abs().mean() (+
backward) inside graphs.GraphedStep with NO parameter update, so every replay must print the eager value.
usage: python scripts/repro/graph_bisect.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import helpers as Hh
from casualhdrsplat_amd import synthetic as S, GaussianRasterizer
from casualhdrsplat_amd.graphs import GraphedStep
dev = "cuda"


def variant(name, hdr=False, poses=1, pose_grad=False, use_rast=True, big=200_000):
    torch.manual_seed(0)
    W, H, P = 320, 208, 20000
    sc = S.make_scene(P, W, H, 1, seed=0, hdr=hdr)
    cams = S.blur_poses(W, H, poses, step=0.02) if poses > 1 else None
    rs, expo, crf = Hh.settings_from_scene(sc, dev, cams, hdr, requires_grad=hdr)
    if pose_grad:
        V = rs.viewmatrices.clone().requires_grad_(True); PV = rs.projmatrices.clone().requires_grad_(True); C = rs.camposes.clone().requires_grad_(True)
        rs = rs._replace(viewmatrices=V, projmatrices=PV, camposes=C)
    leaves = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
    target = torch.rand(3, H, W, device=dev)
    extra = torch.randn(big, device=dev, requires_grad=True)
    rast = GaussianRasterizer(rs, capacity=40 * P * poses) if use_rast else None
    params = leaves + [extra] + ([expo, crf] if hdr else []) + ([V, PV, C] if pose_grad else [])

    def fn():
        for p in params:
            p.grad = None
        if rast is not None:
            out = rast(leaves[0], torch.zeros_like(leaves[0]), leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
            loss = (out[0] - target).abs().mean() + (extra * extra).mean()
        else:
            loss = (extra * extra).mean() + (target * extra[0]).abs().mean()
        loss.backward()
        return loss.detach()

    want = float(fn()); torch.cuda.synchronize()
    g = GraphedStep(fn, [rast] if rast is not None else [], params=params)
    vals = []
    for _ in range(6):
        vals.append(float(g.step())); torch.cuda.synchronize()
    bad = [v for v in vals if abs(v - want) > 1e-4 * abs(want)]
    print(f"{name:46s} eager {want:.6f} replays {' '.join('%.6f' % v for v in vals)}  {'WRONG' if bad else 'ok'}", flush=True)
    return bool(bad)


r = {}
r["torch only"] = variant("torch only (no rasterizer)", use_rast=False)
r["ldr"] = variant("rasterizer LDR, 1 pose")
r["hdr"] = variant("rasterizer HDR + CRF gradients", hdr=True)
r["poses"] = variant("rasterizer HDR, 5 poses", hdr=True, poses=5)
r["pose_grad"] = variant("rasterizer HDR, 5 poses, pose gradients", hdr=True, poses=5, pose_grad=True)
print("RESULT graph_bisect:", {k: ("WRONG" if v else "ok") for k, v in r.items()})


def formation_variant(name, frames=4, virtual=5, learn_traj=True, learn_exp=True, per_frame_rast=True, eager_first=True,
                      stack_out=True, crf_in_step=True, cams_in_step=True, exp_in_step=True, zeros_in_step=True, collect=False, fake_rast=False, probe=False):
    """The example's step, ingredient by ingredient (casualhdrsplat_amd.image_formation)."""
    from casualhdrsplat_amd.image_formation import (FrameRasterizers, HDRBlurFormation, ImplicitCRF, TrajectorySpline,
                                                    knots_from_lookat)
    torch.manual_seed(0)
    W, H, P, deg = 320, 208, 20000, 1
    sc = S.make_scene(P, W, H, deg, seed=0, hdr=True)
    cam = sc.camera
    per_frame = FrameRasterizers(capacity=40 * P * virtual)
    if fake_rast:
        class _Fake:      # torch-only stand-in for the rasterizer: an image that depends on the same inputs
            def __init__(self, settings): self.s = settings
            def __call__(self, means3D, means2D, opac, shs=None, scales=None, rotations=None):
                st = self.s
                v = (st.viewmatrices.sum() + st.projmatrices.sum() * 1e-3 + st.camposes.sum()) * st.exposure + st.crf_table.mean()
                noise = torch.sin(torch.arange(3 * H * W, device=dev, dtype=torch.float32)).reshape(3, H, W)
                img = noise * v + shs[:, 0, :].mean(0).reshape(3, 1, 1)
                return img, torch.zeros(1, device=dev), img * 2
        per_frame = _Fake
    model = HDRBlurFormation(TrajectorySpline(knots_from_lookat(frames + 3, radius=0.25), kind="cubic"), frames, W, H, cam.tanfovx,
                             cam.tanfovy, n_virtual=virtual, crf=ImplicitCRF(K=128), sh_degree=deg, window_scale=0.6,
                             rasterizer_factory=per_frame).to(dev)
    for p in model.crf.parameters():
        p.requires_grad_(False)
    model.trajectory.delta.requires_grad_(learn_traj)
    model.log_exposure.requires_grad_(learn_exp)
    if not crf_in_step:
        tab = model.crf.table().detach()
        model.crf.table = lambda: tab
    if not cams_in_step:
        cams0 = tuple(t.detach() for t in model.cameras_all())
        model.cameras_all = lambda: cams0
    if not exp_in_step:
        import casualhdrsplat_amd.image_formation as IFm
        le = model.log_exposure.detach().clone()
        expo = [torch.exp(le[i]) for i in range(frames)]
        model._parameters["log_exposure"] = None
        model.log_exposure = type("E", (), {"__getitem__": lambda self, i: torch.log(expo[i]), "dtype": le.dtype, "device": le.device})()
    cloud = [getattr(sc, k).to(dev) for k in ("means3D", "opacities", "shs", "scales", "rotations")]
    shs = cloud[2].clone().requires_grad_(True)
    targets = [torch.rand(3, H, W, device=dev) for _ in range(frames)]
    learn = [shs] + ([model.trajectory.delta] if learn_traj else []) + ([model.log_exposure] if learn_exp else [])

    def fn():
        for p in learn:
            p.grad = None
        cams = model.cameras_all()
        losses = []
        for i in range(frames):
            ldr, _, _, _ = model(i, cloud[0], cloud[1], shs, cloud[3], cloud[4], cameras=cams)
            losses.append((ldr - targets[i]).abs().mean())
        torch.stack(losses).sum().backward()
        return torch.stack([l_.detach() for l_ in losses]) if stack_out else losses[0].detach()

    def acc_id(t):
        return id(t.view_as(t).grad_fn.next_functions[0][0])
    before = [acc_id(p) for p in learn] if probe else []
    if eager_first:
        want = fn().sum().item()
    if collect:
        import gc
        gc.collect()
    if probe:
        print("   accumulators kept since before the eager step:", [a == acc_id(p) for a, p in zip(before, learn)], flush=True)
    g = GraphedStep(fn, [] if fake_rast else (per_frame.rasterizers(frames) if eager_first else []), params=learn)
    if not eager_first:
        want = None
    vals = []
    for _ in range(5):
        vals.append(float(g.step().sum())); torch.cuda.synchronize()
    want = vals[0] if want is None else want
    bad = [v for v in vals if abs(v - want) > 1e-4 * abs(want)]
    print(f"{name:46s} eager {want:.6f} replays {' '.join('%.6f' % v for v in vals)}  {'WRONG' if bad else 'ok'}", flush=True)
    return bool(bad)


f = {}
f["full"] = formation_variant("formation step as in the example")
f["const cams+crf"] = formation_variant("... cameras and CRF table constants", learn_traj=False, learn_exp=False, crf_in_step=False, cams_in_step=False, frames=1)
f["const cams"] = formation_variant("... cameras constant, CRF in step", learn_traj=False, learn_exp=False, cams_in_step=False, frames=1)
f["const crf"] = formation_variant("... CRF constant, cameras in step", learn_traj=False, learn_exp=False, crf_in_step=False, frames=1)
f["no eager first"] = formation_variant("... no eager step before the capture", eager_first=False, frames=1)
f["gc"] = formation_variant("... as in the example + gc.collect() before the capture", collect=True)
f["fake rasterizer"] = formation_variant("... torch-only stand-in for the rasterizer", fake_rast=True)
f["fake rasterizer, 1 frame"] = formation_variant("... torch-only stand-in, one frame", fake_rast=True, frames=1)
print("RESULT graph_bisect formation:", {k: ("WRONG" if v else "ok") for k, v in f.items()})
