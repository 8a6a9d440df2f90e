"""Worker of tests/test_gpu_parity.py::test_view_parallel_step_two_ranks_on_one_gpu (launched by torch.distributed.run).

Each rank renders its own view of the same Gaussians on cuda:0 (HIP kernels), then sums the gradients over the ranks
with each exchange strategy (flat all-reduce: library / 1-hop; view exchange: library / 1-hop) over gloo, and compares
with the sum of both views' gradients computed locally with the plain backward."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import helpers as H
    from casualhdrsplat_amd import GaussianRasterizer
    from casualhdrsplat_amd import synthetic as S
    from casualhdrsplat_amd.distributed import all_reduce_gradients, exchange_view_gradients, init_from_env
    # HS_DIST_BACKEND=nccl: one GPU per rank over RCCL (boxes with >= 2 GPUs); default gloo: both ranks share cuda:0
    backend = os.environ.get("HS_DIST_BACKEND", "gloo")
    rank, world, local = init_from_env(backend)
    idx = local % torch.cuda.device_count() if backend == "nccl" else 0
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    W, Hh = 160, 120
    sc = S.make_scene(3000, W, Hh, 3, seed=7)
    names = ("means3D", "means2D", "opacities", "shs", "scales", "rotations")

    def backward(view_rank, defer, overlap=False, chunked=False):
        sc.camera = S.yaw_camera(W, Hh, -5.0 + 10.0 * view_rank / max(world - 1, 1))
        rs, _, _ = H.settings_from_scene(sc, dev)
        leaf = {k: t.clone().to(dev).requires_grad_(True) for k, t in
                dict(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities, shs=sc.shs,
                     scales=sc.scales, rotations=sc.rotations).items()}
        rast = GaussianRasterizer(rs, defer_sh_grad=defer, gather_group=True if overlap else None,
                                  reduce_group=True if chunked else None, reduce_chunks=3)
        out = rast(leaf["means3D"], leaf["means2D"], leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"],
                   rotations=leaf["rotations"])
        (out[0] * sc.dL_dimage.to(dev)).sum().backward()
        return leaf, rast

    want = None
    for r in range(world):
        leaf, _ = backward(r, False)
        g = {k: leaf[k].grad.clone() for k in names}
        want = g if want is None else {k: want[k] + g[k] for k in names}

    for mode in ("allreduce", "allreduce_overlap", "views", "views_overlap"):
        for algo in ("rccl", "direct"):
            if mode == "allreduce_overlap" and algo == "direct":
                continue
            leaf, rast = backward(rank, mode.startswith("views"), overlap=mode == "views_overlap",
                                  chunked=mode == "allreduce_overlap")
            if mode == "views_overlap":
                assert rast.deferred["gather"] is not None
            if mode == "allreduce_overlap":
                # the backward ran its per-Gaussian half in 3 chunks and started each chunk's all-reduce itself; means2D (this
                # view's screen-space gradient) is not part of it
                assert rast.finish_reduce() > 0
                all_reduce_gradients([leaf["means2D"]], algo=algo)
            elif mode != "allreduce":
                n = exchange_view_gradients([leaf[k] for k in names if k != "shs"], leaf["shs"], rast.deferred, algo=algo)
                assert n["all_gathered"] == world * (3000 * 3 + 3), n
            else:
                all_reduce_gradients([leaf[k] for k in names], algo=algo)
            for k in names:
                a, b = leaf[k].grad, want[k]
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max())), (mode, algo, k)
    # the usual 3DGS wiring -- activations between the parameters and the rasterizer, so the rasterizer's
    # inputs are NOT leaves: autograd runs the activations' backward right after the rasterizer's, on rows the chunked
    # all-reduce may still be summing.  The backward must have waited for its collectives by then (finish_reduce() finds
    # nothing left), and the raw parameters receive the gradients summed over the views.
    def activated(view_rank, chunked):
        sc.camera = S.yaw_camera(W, Hh, -5.0 + 10.0 * view_rank / max(world - 1, 1))
        rs, _, _ = H.settings_from_scene(sc, dev)
        raw = dict(means3D=sc.means3D.clone(), raw_opac=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)),
                   raw_scales=torch.log(sc.scales), raw_rot=sc.rotations * 1.7, dc=sc.shs[:, :1].clone(),
                   rest=sc.shs[:, 1:].clone())
        raw = {k: t.to(dev).requires_grad_(True) for k, t in raw.items()}
        m2d = torch.zeros_like(raw["means3D"], requires_grad=True)
        rast = GaussianRasterizer(rs, reduce_group=True if chunked else None, reduce_chunks=3)
        out = rast(raw["means3D"], m2d, torch.sigmoid(raw["raw_opac"]), shs=torch.cat([raw["dc"], raw["rest"]], dim=1),
                   scales=torch.exp(raw["raw_scales"]), rotations=torch.nn.functional.normalize(raw["raw_rot"], dim=1))
        (out[0] * sc.dL_dimage.to(dev)).sum().backward()
        return raw, rast

    want_raw = None
    for r in range(world):
        raw, _ = activated(r, False)
        g = {k: v.grad.clone() for k, v in raw.items()}
        want_raw = g if want_raw is None else {k: want_raw[k] + g[k] for k in g}
    raw, rast = activated(rank, True)
    assert rast._cell.get("reduce_waited_in_backward", 0) > 0 and rast.finish_reduce() == 0
    for k, v in raw.items():
        b = want_raw[k]
        assert torch.allclose(v.grad, b, rtol=1e-5, atol=1e-6 * float(b.abs().max())), ("activated", k)
    # a leaf that feeds TWO rasterizer calls before one backward (a multi-frame step, a second view per
    # rank).  The autograd engine then sums the two calls' gradients on the compute stream: neither call may leave its
    # chunked all-reduce in flight -- both backwards must have waited (finish_reduce() finds nothing) -- and every leaf
    # ends with the sum over ranks AND calls.  A single call on fresh leaves afterwards may again stay in flight.
    def two_calls(view_ranks, chunked):
        leaf = {k: t.clone().to(dev).requires_grad_(True) for k, t in
                dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations).items()}
        rasts, loss = [], 0.0
        for vr in view_ranks:
            sc.camera = S.yaw_camera(W, Hh, -5.0 + 10.0 * vr / max(2 * world - 1, 1))
            rs, _, _ = H.settings_from_scene(sc, dev)
            rast = GaussianRasterizer(rs, reduce_group=True if chunked else None, reduce_chunks=3)
            out = rast(leaf["means3D"], torch.zeros_like(leaf["means3D"]), leaf["opacities"], shs=leaf["shs"],
                       scales=leaf["scales"], rotations=leaf["rotations"])
            loss = loss + (out[0] * sc.dL_dimage.to(dev)).sum()
            rasts.append(rast)
        loss.backward()
        return leaf, rasts

    want2 = None
    for r in range(world):
        leaf, _ = two_calls((2 * r, 2 * r + 1), False)
        g = {k: v.grad.clone() for k, v in leaf.items()}
        want2 = g if want2 is None else {k: want2[k] + g[k] for k in g}
    leaf, rasts = two_calls((2 * rank, 2 * rank + 1), True)
    assert all(r_._cell.get("reduce_waited_in_backward", 0) > 0 for r_ in rasts), [r_._cell for r_ in rasts]
    assert all(r_.finish_reduce() == 0 for r_ in rasts)
    for k, v in leaf.items():
        b = want2[k]
        assert torch.allclose(v.grad, b, rtol=1e-5, atol=1e-6 * float(b.abs().max())), ("two calls per backward", k)
    from casualhdrsplat_amd.rasterizer import _OPEN_CONSUMERS
    assert not any(e[0] for e in _OPEN_CONSUMERS.values()), "every view-parallel call's backward has run: no consumer left open"
    leaf, rast = backward(rank, False, chunked=True)     # a sole consumer again: the collectives outlive backward()
    assert rast.finish_reduce() > 0
    torch.cuda.synchronize()
    print("VIEW-EXCHANGE-OK", rank, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
