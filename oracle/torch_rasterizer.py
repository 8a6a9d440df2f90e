"""Pure-PyTorch CPU autograd rasterizer (BASELINE.json config 1; the "pure-PyTorch CPU rasterizer"
that north_star wants timed beside the MI355X numbers).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by casualhdrsplat_amd.  PARITY UNPINNED: /root/reference has no code, tests or golden
vectors; the rules are those of SURVEY.md 8(a) (published
diff_gaussian_rasterization semantics) and the HDR epilogue order of
/root/reference/assets/pipeline.png (H -> exposure, CRF -> I -> average over poses -> B).

Independent of oracle/hs_oracle.c by construction: the forward is written with tensor ops and ALL
gradients come from torch.autograd (the C oracle and the HIP kernels use hand-derived backward
formulas), so agreement of the three pins the derivatives.  Runs in float32 or float64.

Gradient conventions reproduced from the published rasterizer (they differ from naive autograd):
  * alpha = min(0.99, o*G) passes gradient straight through the clamp;
  * the 1.3*tanfov clamp of t.x/t.z, t.y/t.z treats the clamped value as a constant;
  * skip / termination decisions (power > 0, alpha < 1/255, T < 1e-4) are constants;
  * the means2D gradient is the pixel-space gradient times (0.5 W, 0.5 H).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

TILE = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


@dataclass
class View:
    W: int
    H: int
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor  # [4,4] transposed convention
    projmatrix: torch.Tensor
    campos: torch.Tensor


def _xform(m, i, p):
    # p_i' = m[0,i] x + m[1,i] y + m[2,i] z + m[3,i]   (flat index 4*j+i == m[j,i])
    return m[0, i] * p[:, 0] + m[1, i] * p[:, 1] + m[2, i] * p[:, 2] + m[3, i]


def sh_basis(deg, d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, SH_C0)]
    if deg >= 1:
        b += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
        if deg >= 3:
            b += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
                  SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
                  SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=1)  # [P, (deg+1)^2]


def sh_backward_views(means3D, camposes, view_colors, M, sh_degree):
    """Checker for hs_sh_backward_views: dL/dsh[g,k,c] = sum_v Y_k(normalize(mean_g - campos_v)) * view_colors[v,g,c]
    (rows k >= (sh_degree+1)^2 zero).  Plain tensor algebra; tests pin it against autograd of `preprocess`."""
    P = means3D.shape[0]
    out = torch.zeros(P, M, 3, dtype=view_colors.dtype, device=view_colors.device)
    nc = (sh_degree + 1) ** 2
    for v in range(camposes.shape[0]):
        d = means3D.detach().to(view_colors.dtype) - camposes[v].to(view_colors.dtype)[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        out[:, :nc, :] += sh_basis(sh_degree, d)[:, :, None] * view_colors[v][:, None, :]
    return out


def quat_to_R(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def preprocess(view: View, means3D, opacities, sh_degree, shs=None, colors_precomp=None, scales=None,
               rotations=None, cov3D_precomp=None, scale_modifier=1.0, means2D=None, antialiasing=False,
               radiance_activation="relu_shift"):
    """a4.  Returns a dict; differentiable w.r.t. every float input.  Culled Gaussians have radii == 0.
    antialiasing (newer published rasterizer): opacity is scaled by sqrt(max(0.000025, det(cov2D) / det(cov2D +
    0.3 I))), compensating the energy the 0.3-pixel dilation adds to small splats."""
    dt = means3D.dtype
    V = view.viewmatrix.to(dt)
    PM = view.projmatrix.to(dt)
    W, H = view.W, view.H
    P = means3D.shape[0]
    pv = torch.stack([_xform(V, 0, means3D), _xform(V, 1, means3D), _xform(V, 2, means3D)], dim=1)
    vis = pv[:, 2] > 0.2
    ph_x, ph_y, ph_w = _xform(PM, 0, means3D), _xform(PM, 1, means3D), _xform(PM, 3, means3D)
    pw = 1.0 / (ph_w + 1e-7)
    ndc = torch.stack([ph_x * pw, ph_y * pw], dim=1)
    if cov3D_precomp is not None:
        c6 = cov3D_precomp
        Sigma = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]],
                            dim=1).reshape(P, 3, 3)
    else:
        R = quat_to_R(rotations)
        S2 = (scale_modifier * scales) ** 2
        Sigma = (R * S2[:, None, :]) @ R.transpose(1, 2)
    fx, fy = W / (2 * view.tanfovx), H / (2 * view.tanfovy)
    limx, limy = 1.3 * view.tanfovx, 1.3 * view.tanfovy
    tz = torch.where(vis, pv[:, 2], torch.ones_like(pv[:, 2]))
    txtz, tytz = pv[:, 0] / tz, pv[:, 1] / tz
    cx, cy = (txtz < -limx) | (txtz > limx), (tytz < -limy) | (tytz > limy)
    # clamped coordinate is a constant for the backward (published rule)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tz).detach(), pv[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tz).detach(), pv[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], dim=1).reshape(P, 2, 3)
    Wv = V[:3, :3].t()  # standard rotation: Wv[i,j] = V[j,i]
    A = J @ Wv
    cov = A @ Sigma @ A.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = vis & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_s, -b / det_s, a / det_s], dim=1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)
    if means2D is not None:
        # gradient tap: d(loss)/d(means2D) := d(loss)/d(ndc), the NDC-scaled screen-space gradient
        pix = pix + means2D[:, :2] * torch.tensor([0.5 * W, 0.5 * H], dtype=dt)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    pd = pix.detach()

    def tl(v, g):
        return torch.clamp(torch.trunc(v / TILE), 0, g).to(torch.int64)

    rminx, rminy = tl(pd[:, 0] - radius, gx), tl(pd[:, 1] - radius, gy)
    rmaxx, rmaxy = tl(pd[:, 0] + radius + (TILE - 1), gx), tl(pd[:, 1] + radius + (TILE - 1), gy)
    area = (rmaxx - rminx) * (rmaxy - rminy)
    ok = ok & (area > 0)
    if colors_precomp is not None:
        rgb = colors_precomp
        clamped = torch.zeros(P, 3, dtype=torch.bool)
    else:
        d = means3D - view.campos.to(dt)[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        B = sh_basis(sh_degree, d)
        raw = (B[:, :, None] * shs[:, :B.shape[1], :]).sum(dim=1)
        if radiance_activation == "exp":           # linear radiance through a positive activation
            rgb, clamped = torch.exp(raw), torch.zeros_like(raw, dtype=torch.bool)
        elif radiance_activation == "softplus":
            rgb, clamped = torch.nn.functional.softplus(raw), torch.zeros_like(raw, dtype=torch.bool)
        elif radiance_activation == "relu_shift":  # the published rule: + 0.5, clamped below only
            raw = raw + 0.5
            clamped = raw < 0
            rgb = torch.clamp_min(raw, 0.0)
        else:
            raise ValueError(radiance_activation)
    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)
    opac = opacities.reshape(-1)
    if antialiasing:
        det0 = (a - 0.3) * (c - 0.3) - b * b
        opac = opac * torch.sqrt(torch.clamp(det0 / det_s, min=0.000025))
    return dict(xy=pix, conic=conic, opacity=opac, rgb=rgb, depth=pv[:, 2], radii=radii,
                rect=torch.stack([rminx, rminy, rmaxx, rmaxy], dim=1), tiles_touched=torch.where(ok, area, torch.zeros_like(area)),
                clamped=clamped, visible=ok)


def bin_tiles(view: View, pre: dict):
    """a5..a8 with torch integer ops: returns (point_list, ranges[ntiles,2], keys_sorted)."""
    gx, gy = (view.W + TILE - 1) // TILE, (view.H + TILE - 1) // TILE
    ok = pre["visible"]
    idx = torch.nonzero(ok).reshape(-1)
    rect = pre["rect"][idx]
    w = rect[:, 2] - rect[:, 0]
    cnt = pre["tiles_touched"][idx]
    rep = torch.repeat_interleave(torch.arange(idx.numel()), cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(rep.numel()) - start[rep]
    ty = rect[rep, 1] + local // w[rep]
    tx = rect[rep, 0] + local % w[rep]
    tile = ty * gx + tx
    dbits = pre["depth"].detach().to(torch.float32)[idx].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    keys = (tile << 32) | dbits[rep]
    order = torch.sort(keys, stable=True).indices
    keys_sorted = keys[order]
    point_list = idx[rep][order]
    tiles_sorted = keys_sorted >> 32
    ntiles = gx * gy
    counts = torch.bincount(tiles_sorted, minlength=ntiles)
    ends = torch.cumsum(counts, 0)
    ranges = torch.stack([ends - counts, ends], dim=1)
    ranges[counts == 0] = 0
    return point_list, ranges, keys_sorted


def render(view: View, pre: dict, point_list, ranges, bg, tiles=None, want_invdepth=False):
    """a9.  Returns (color[3,H,W], final_T[H,W], n_contrib[H,W]); tiles = optional iterable of tile ids to
    render (others stay at bg / T=1) -- used to time a bounded sample of a large frame.  want_invdepth appends the
    expected inverse depth image sum_i alpha_i T_i / z_i (no background term), as newer published rasterizers do."""
    W, H = view.W, view.H
    dt = pre["xy"].dtype
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    bg = bg.to(dt)
    color = bg[:, None, None].expand(3, H, W).clone()
    final_T = torch.ones(H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    invdepth = torch.zeros(H, W, dtype=dt)
    oy, ox = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
    tile_ids = range(gx * gy) if tiles is None else tiles
    xy, conic, opac, rgb = pre["xy"], pre["conic"], pre["opacity"], pre["rgb"]
    for t in tile_ids:
        r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
        ty, tx = divmod(int(t), gx)
        y0, x0 = ty * TILE, tx * TILE
        h, w = min(TILE, H - y0), min(TILE, W - x0)
        if r1 <= r0:
            continue
        ids = point_list[r0:r1]
        pxs = (x0 + ox[:h, :w]).reshape(-1).to(dt)
        pys = (y0 + oy[:h, :w]).reshape(-1).to(dt)
        gxy = xy[ids]
        dx = gxy[None, :, 0] - pxs[:, None]
        dy = gxy[None, :, 1] - pys[:, None]
        con = conic[ids]
        power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
        raw = opac[ids][None, :] * torch.exp(power)
        alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()  # straight-through clamp
        valid = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
        one_m = torch.where(valid, 1.0 - alpha, torch.ones_like(alpha))
        T_incl = torch.cumprod(one_m, dim=1)
        alive = T_incl.detach() >= 0.0001  # prefix mask: False from the terminating entry on
        contrib = valid & alive
        T_before = T_incl / one_m
        wgt = torch.where(contrib, alpha * T_before, torch.zeros_like(alpha))
        C = wgt @ rgb[ids]
        if want_invdepth:
            invdepth[y0:y0 + h, x0:x0 + w] = (wgt @ (1.0 / pre["depth"][ids])).reshape(h, w)
        Tf = torch.where(alive, one_m, torch.ones_like(one_m)).prod(dim=1)
        pos = torch.arange(1, ids.numel() + 1)
        last = torch.where(contrib, pos[None, :], torch.zeros_like(pos)[None, :]).amax(dim=1)
        out = C + Tf[:, None] * bg[None, :]
        color[:, y0:y0 + h, x0:x0 + w] = out.t().reshape(3, h, w)
        final_T[y0:y0 + h, x0:x0 + w] = Tf.reshape(h, w)  # differentiable: alpha image = 1 - final_T
        n_contrib[y0:y0 + h, x0:x0 + w] = last.reshape(h, w)
    if want_invdepth:
        return color, final_T, n_contrib, invdepth
    return color, final_T, n_contrib


def tonemap(hdr, exposure, table, u_range, eps=1e-8):
    """a15: LDR = PWL(table_c, ln(max(H*dt, eps))) with flat extrapolation outside [umin, umax]."""
    K = table.shape[1]
    umin, umax = u_range
    x = hdr * exposure
    u = torch.log(torch.clamp(x, min=eps))
    s = torch.clamp((u - umin) / (umax - umin) * (K - 1), 0.0, float(K - 1))
    i = torch.clamp(torch.floor(s.detach()), max=K - 2).to(torch.int64)
    f = s - i.to(s.dtype)
    flat = i.reshape(3, -1)
    t0 = torch.gather(table, 1, flat).reshape(hdr.shape)
    t1 = torch.gather(table, 1, flat + 1).reshape(hdr.shape)
    return t0 * (1 - f) + t1 * f


def rasterize(view: View, means3D, opacities, sh_degree, bg, shs=None, colors_precomp=None, scales=None,
              rotations=None, cov3D_precomp=None, scale_modifier=1.0, means2D=None, tiles=None, return_state=False,
              antialiasing=False, radiance_activation="relu_shift"):
    pre = preprocess(view, means3D, opacities, sh_degree, shs, colors_precomp, scales, rotations, cov3D_precomp,
                     scale_modifier, means2D, antialiasing, radiance_activation)
    point_list, ranges, keys_sorted = bin_tiles(view, pre)
    if return_state:
        color, final_T, n_contrib, invdepth = render(view, pre, point_list, ranges, bg, tiles, want_invdepth=True)
        return color, dict(pre=pre, point_list=point_list, ranges=ranges, keys_sorted=keys_sorted, final_T=final_T,
                           n_contrib=n_contrib, invdepth=invdepth)
    return render(view, pre, point_list, ranges, bg, tiles)[0]


def rasterize_hdr(views, means3D, opacities, sh_degree, bg, exposure, crf_table, crf_range, blur_domain="ldr", **kw):
    """N-pose HDR image formation: returns (B_ldr, H_mean)."""
    hs = [rasterize(v, means3D, opacities, sh_degree, bg, **kw) for v in views]
    Hm = torch.stack(hs).mean(dim=0)
    if blur_domain == "ldr":
        ldr = torch.stack([tonemap(h, exposure, crf_table, crf_range) for h in hs]).mean(dim=0)
    else:
        ldr = tonemap(Hm, exposure, crf_table, crf_range)
    return ldr, Hm
