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
    torch.cuda.synchronize()
    print("VIEW-EXCHANGE-OK", rank, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
