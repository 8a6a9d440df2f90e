"""Shared test plumbing: scene -> (HIP path through the C ABI) and scene -> (CPU oracle), plus comparators.

The HIP path is casualhdrsplat_amd (product).  The oracle (oracle/) is only ever the checker.
"""
from __future__ import annotations

import numpy as np
import torch

from casualhdrsplat_amd import synthetic as S


def settings_from_scene(sc: S.Scene, device, cameras=None, hdr=False, blur_domain="ldr", requires_grad=False):
    from casualhdrsplat_amd import GaussianRasterizationSettings
    cam = sc.camera
    kw = {}
    exposure = crf = None
    if hdr:
        exposure = sc.exposure.clone().to(device).requires_grad_(requires_grad)
        crf = sc.crf_table.clone().to(device).requires_grad_(requires_grad)
        kw.update(exposure=exposure, crf_table=crf, crf_range=sc.crf_range, blur_domain=blur_domain)
    if cameras is not None:
        kw.update(viewmatrices=torch.stack([c.viewmatrix for c in cameras]).to(device),
                  projmatrices=torch.stack([c.projmatrix for c in cameras]).to(device),
                  camposes=torch.stack([c.campos for c in cameras]).to(device))
    rs = GaussianRasterizationSettings(
        image_height=cam.H, image_width=cam.W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=sc.bg.to(device),
        scale_modifier=1.0, viewmatrix=cam.viewmatrix.to(device), projmatrix=cam.projmatrix.to(device),
        sh_degree=sc.sh_degree, campos=cam.campos.to(device), prefiltered=False, debug=False, **kw)
    return rs, exposure, crf


def run_hip(sc: S.Scene, device="cuda", cameras=None, hdr=False, blur_domain="ldr", backward=True, capacity=None,
            use_cov_precomp=None, use_colors_precomp=None, grad_hdr=None):
    """Forward (+ backward with sc.dL_dimage) through GaussianRasterizer on the GPU."""
    from casualhdrsplat_amd import GaussianRasterizer, inspect_state
    rs, exposure, crf = settings_from_scene(sc, device, cameras, hdr, blur_domain, requires_grad=backward)
    leaf = {}

    def mk(name, t):
        leaf[name] = t.clone().to(device).requires_grad_(backward)
        return leaf[name]

    means3D = mk("means3D", sc.means3D)
    means2D = mk("means2D", torch.zeros_like(sc.means3D))
    opac = mk("opacities", sc.opacities)
    kwargs = {}
    if use_colors_precomp is not None:
        kwargs["colors_precomp"] = mk("colors_precomp", use_colors_precomp)
    else:
        kwargs["shs"] = mk("shs", sc.shs)
    if use_cov_precomp is not None:
        kwargs["cov3D_precomp"] = mk("cov3D_precomp", use_cov_precomp)
    else:
        kwargs["scales"] = mk("scales", sc.scales)
        kwargs["rotations"] = mk("rotations", sc.rotations)
    rast = GaussianRasterizer(rs, capacity=capacity)
    out = rast(means3D, means2D, opac, **kwargs)
    res = {"color": out[0].detach().cpu().numpy(), "radii": out[1].cpu().numpy()}
    if hdr:
        res["hdr"] = out[2].detach().cpu().numpy()
    if backward:
        st = inspect_state(out[0])
        res["state"] = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
        loss = (out[0] * sc.dL_dimage.to(device)).sum()
        if grad_hdr is not None:
            loss = loss + (out[2] * grad_hdr.to(device)).sum()
        loss.backward()
        for k, v in leaf.items():
            res["d_" + k] = v.grad.detach().cpu().numpy() if v.grad is not None else None
        if hdr:
            res["d_exposure"] = exposure.grad.detach().cpu().numpy()
            res["d_crf_table"] = crf.grad.detach().cpu().numpy()
    torch.cuda.synchronize()
    return res


def oracle_camera(O, sc: S.Scene, cam=None):
    cam = cam or sc.camera
    return O.Camera(cam.W, cam.H, cam.tanfovx, cam.tanfovy, cam.viewmatrix.numpy(), cam.projmatrix.numpy(),
                    cam.campos.numpy(), sc.bg.numpy(), 1.0, sc.sh_degree)


def run_oracle(O, sc: S.Scene, cam=None, dL=None, backward=True, use_cov_precomp=None, use_colors_precomp=None):
    """Single-pose LDR/linear render through the C oracle (a4..a12)."""
    ocam = oracle_camera(O, sc, cam)
    kw = {}
    if use_colors_precomp is not None:
        kw["colors_precomp"] = use_colors_precomp.numpy()
    else:
        kw["shs"] = sc.shs.numpy()
    if use_cov_precomp is not None:
        kw["cov3D_precomp"] = use_cov_precomp.numpy()
    else:
        kw["scales"] = sc.scales.numpy()
        kw["rotations"] = sc.rotations.numpy()
    f = O.forward(ocam, sc.means3D.numpy(), sc.opacities.numpy(), **kw)
    b = None
    if backward:
        dL = sc.dL_dimage.numpy() if dL is None else dL
        b = O.backward(ocam, f, dL, sc.means3D.numpy(), **kw)
    return f, b


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rel_err(got, ref, floor):
    """max |got-ref| / max(|ref|, floor) and the fraction of elements above 1e-4 of that measure."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    e = np.abs(got - ref) / np.maximum(np.abs(ref), floor)
    return float(e.max()) if e.size else 0.0, float((e > 1e-4).mean()) if e.size else 0.0


def grad_floor(ref):
    """Absolute floor for relative comparison of a gradient tensor: 1e-3 of its RMS.  An fp32 sum of
    many signed per-pixel terms carries an absolute error proportional to the magnitude of the terms,
    not of the (possibly cancelling) total, so elements far below the tensor's scale are compared
    against the scale."""
    ref = np.asarray(ref, np.float64)
    rms = float(np.sqrt((ref ** 2).mean())) if ref.size else 0.0
    return max(1e-3 * rms, 1e-30)
