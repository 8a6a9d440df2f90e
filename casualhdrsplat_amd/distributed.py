"""View-parallel training step support (BASELINE.json config 5).

The hot path shards by camera view: every rank (one process per GPU) holds the full Gaussian
set, renders its own view forward+backward, and the only exchange is ONE all-reduce(sum) of the
per-Gaussian gradients (P x 59 fp32 at SH degree 3, plus the CRF-table and exposure gradients).
`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI between the 8 MI355X of a node);
"gloo" runs the same code on CPU for tests.  The reference shows no multi-GPU code at all
(/root/reference/Readme.md:1-58), so there is nothing to mirror here beyond the collective.

The rasterizer's backward writes every gradient into one flat fp32 buffer
(rasterizer._launch_backward), so when the `.grad`s of the Gaussian parameters are still views of
that buffer the reduction is a single in-place collective with no packing copy.

`exchange_view_gradients` is the cheaper exchange for SH-coloured Gaussians: the SH gradient row of a
Gaussian is the outer product of its view's RGB gradient (3 floats) with the SH basis of its view
direction (M floats), so ranks all-gather the 3 floats and the camera centres and rebuild the summed
[P, M, 3] gradient locally -- 12 B instead of 12*M B per Gaussian and view on the wire (at SH degree 3
and 8 views: 56 MB all-reduced + 12 MB all-gathered per rank instead of 236 MB all-reduced).
"""
from __future__ import annotations

from typing import Iterable, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the process group
    when WORLD_SIZE > 1."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # no silent default: a fixed port collides with any other job on the node, and ranks started by hand cannot
            # agree on a free one by themselves (torch.distributed.run sets it; bench.py's own launcher picks a free port)
            raise RuntimeError("init_from_env: WORLD_SIZE > 1 but MASTER_PORT is not set -- launch the ranks with "
                               "`python -m torch.distributed.run --master-addr 127.0.0.1 --master-port <free port> ...`")
        kw = {}
        if backend == "nccl":
            dev_index = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(dev_index)
            if os.environ.get("HS_DIST_BIND_DEVICE") == "1":   # eager communicator bound to this rank's GPU (opt-in)
                kw["device_id"] = torch.device("cuda", dev_index)
        # a collective that one rank never enters must end the job, not hang it (HS_DIST_TIMEOUT_S, default 15 minutes: ranks of a fresh box may start minutes apart)
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(os.environ.get("HS_DIST_TIMEOUT_S", "900")))
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except TypeError:  # a torch without `device_id`
            kw.pop("device_id", None)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def _shared_flat(grads: Sequence[torch.Tensor]) -> torch.Tensor | None:
    """If all gradients are contiguous slices of one storage, return a 1-D tensor spanning exactly the range from
    the first to the last of them (never a byte more: whatever lies behind the last listed gradient -- another
    slice of the rasterizer's flat buffer, or its uninitialised pad -- is not the caller's to reduce)."""
    if not grads:
        return None
    st = grads[0].untyped_storage()
    base = st.data_ptr()
    lo, hi = None, None
    for g in grads:
        if g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != base:
            return None
        o = g.storage_offset()
        lo = o if lo is None else min(lo, o)
        hi = o + g.numel() if hi is None else max(hi, o + g.numel())
    if sum(g.numel() for g in grads) < 0.9 * (hi - lo):
        return None  # sparse cover: packing is cheaper than reducing the gaps
    return torch.empty(0, dtype=torch.float32, device=grads[0].device).set_(st, lo, (hi - lo,))


def all_reduce_direct(flat: torch.Tensor, group=None) -> None:
    """In-place sum of `flat` over the ranks as a 1-hop reduce-scatter + all-gather built from two all-to-alls.

    xGMI on an 8 x MI355X node is a full mesh of point-to-point links (7 x ~153 GB/s per GPU), so a ring
    all-reduce is bound by ONE link (2*(n-1)/n * bytes / link) while sending shard j straight to rank j uses
    all seven links at once (~0.4 ms vs ~2.7 ms for 236 MB).  Every shard is summed by
    exactly one rank in rank order and then sent to all, so all ranks end with bitwise identical results.
    Any length works: the first world * (n // world) elements take the two all-to-alls, the (< world) elements
    left over take one tiny library all-reduce.  One temporary of the buffer's size (the received shards); the
    second exchange sends every peer the SAME reduced shard, so it needs no staging copy."""
    world = dist.get_world_size(group)
    n = flat.numel()
    chunk = n // world
    main = chunk * world
    if chunk > 0:
        body = flat[:main]
        recv = torch.empty_like(body)
        dist.all_to_all_single(recv, body, group=group)          # recv[j*chunk:(j+1)*chunk] = rank j's copy of MY shard
        mine = recv.view(world, chunk).sum(dim=0)                # fixed summation order
        # flat[j*chunk:(j+1)*chunk] = rank j's reduced shard.  RCCL: grouped point-to-point sends (1 hop, all links);
        # backends without a list all-to-all (gloo, CPU tests) use the equivalent all-gather
        if dist.get_backend(group) == "nccl":
            dist.all_to_all(list(body.view(world, chunk).unbind(0)), [mine] * world, group=group)
        else:
            dist.all_gather_into_tensor(body, mine, group=group)
    if main < n:
        dist.all_reduce(flat[main:], op=dist.ReduceOp.SUM, group=group)


def validate_direct(device, group=None, n: int = 100003) -> bool:
    """all_reduce_direct against the library all-reduce on a test vector (odd length: the leftover path runs too), on every
    rank; True only if ALL ranks reproduce it (sums of `world` integers-valued floats: exact in any order).  Called once
    before the 1-hop form is trusted with gradients (autotune_all_reduce, bench.py's probe)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randint(-1000, 1000, (n,), generator=g).to(torch.float32).to(device)
    want = x.clone()
    dist.all_reduce(want, op=dist.ReduceOp.SUM, group=group)
    ok = 1.0
    try:
        all_reduce_direct(x, group)
        ok = 1.0 if torch.equal(x, want) else 0.0
    except RuntimeError:
        ok = 0.0
    t = torch.tensor([ok], dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item() == 1.0) and world >= 1


_ALGO = {"choice": "rccl"}


def autotune_all_reduce(flat: torch.Tensor, group=None, iters: int = 3) -> str:
    """Time the library all-reduce against all_reduce_direct on a scratch copy of `flat` and keep the faster one
    for later all_reduce_gradients(..., algo="auto") calls.  The decision is taken on rank 0's maximum-over-ranks
    timings, so every rank chooses the same algorithm."""
    import time
    scratch = torch.zeros_like(flat)
    sync = torch.cuda.synchronize if flat.is_cuda else (lambda: None)
    times = {}
    candidates = [("rccl", lambda: dist.all_reduce(scratch, group=group))]
    if validate_direct(flat.device, group):   # never trust the 1-hop form with gradients before it reproduced the library's sum
        candidates.append(("direct", lambda: all_reduce_direct(scratch, group)))
    for name, fn in candidates:
        fn()
        dist.barrier(group)
        sync()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        times[name] = float(t.item()) / iters
    _ALGO["choice"] = "direct" if times.get("direct", float("inf")) < times["rccl"] else "rccl"
    _ALGO["times_ms"] = {k: v * 1e3 for k, v in times.items()}
    return _ALGO["choice"]


def all_reduce_gradients(params: Iterable[torch.Tensor], group=None, average: bool = False, algo: str = "rccl",
                         pending: list | None = None) -> int:
    """Sum (or average) `.grad` of `params` over all ranks with one exchange.  Returns the number of fp32 elements
    exchanged (0 when not distributed).  algo: "rccl" = torch.distributed.all_reduce, "direct" = all_reduce_direct,
    "auto" = whichever autotune_all_reduce measured faster.  `pending`: if a list is given and the exchange is a
    single library all-reduce, it is issued asynchronously and a completion callable is appended instead of waiting
    (the caller overlaps independent work and then calls it)."""
    grads = [p.grad for p in params if p is not None and p.grad is not None]
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    if algo == "auto":
        algo = _ALGO["choice"]
    flat = _shared_flat(grads)
    if flat is not None:
        if algo == "direct":
            all_reduce_direct(flat, group)
        elif pending is not None:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)

            def finish(work=work, flat=flat):
                work.wait()
                if average:
                    flat.div_(world)
            pending.append(finish)
            return flat.numel()
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        return flat.numel()
    packed = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    if average:
        packed.div_(dist.get_world_size(group))
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(packed[o:o + n].view_as(g))
        o += n
    return packed.numel()


def chunk_bounds(P: int, chunks: int, align: int = 128) -> list:
    """[(g0, g1)] covering [0, P) in at most `chunks` ascending pieces whose starts are multiples of `align` (the
    workgroup size of the per-Gaussian backward: hs_bwd_args.g_begin)."""
    if P <= 0:
        return []
    chunks = max(1, int(chunks))
    per = max(align, ((P + chunks - 1) // chunks + align - 1) // align * align)
    return [(g0, min(P, g0 + per)) for g0 in range(0, P, per)]


def _all_reduce_pieces(pieces: Sequence[torch.Tensor], group=None):
    """ONE asynchronous collective summing several tensors over the ranks; returns (completion callable, collectives issued).
    RCCL (and gloo on host tensors): `allreduce_coalesced` -- RCCL brackets the per-tensor all-reduces in one
    ncclGroupStart / End = one launch on the communication stream, no copies.  Where the backend cannot coalesce what it is
    given (gloo with device tensors: the one-GPU plumbing check), the pieces are packed into one buffer, reduced by one
    all-reduce and copied back when it is waited for -- still one collective."""
    pieces = [t for t in pieces if t.numel() > 0]
    for t in pieces:
        if not t.is_contiguous():
            raise ValueError("chunked_all_reduce: gradient rows must be contiguous")
    if not pieces:
        return None, 0
    if len(pieces) == 1:
        return dist.all_reduce(pieces[0], op=dist.ReduceOp.SUM, group=group, async_op=True).wait, 1
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    if dist.get_backend(group) == "nccl" or not pieces[0].is_cuda:
        try:
            opts = dist.AllreduceCoalescedOptions()
            opts.reduceOp = dist.ReduceOp.SUM
            work = pg.allreduce_coalesced(list(pieces), opts)
            return work.wait, 1
        except (AttributeError, RuntimeError, NotImplementedError):
            pass
    flat = torch.cat([t.reshape(-1) for t in pieces])
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def finish(work=work, flat=flat, pieces=pieces):
        work.wait()
        o = 0
        for t in pieces:
            n = t.numel()
            t.copy_(flat[o:o + n].view_as(t))
            o += n
    return finish, 1


LAST_EXCHANGE = {"collectives": 0}   # collectives the latest chunked_all_reduce of this process issued (bench.py prints it)


def chunked_all_reduce(rows: Sequence[torch.Tensor], P: int, chunks: int, compute_chunk, tail: Sequence[torch.Tensor] = (),
                       group=None) -> list:
    """The plain gradient exchange, overlapped with the kernels that produce the gradients (BASELINE.json configs[4]).

    rows           per-Gaussian gradient tensors [P, ...] (contiguous; in the rasterizer: views of its flat buffer);
    compute_chunk  callable (g0, g1): enqueue the computation of rows [g0, g1) of every tensor in `rows` on the current
                   stream (the rasterizer: hs_backward(HS_BWD_PROJECT, g_begin, g_end));
    tail           gradients that are complete before the first chunk (exposure, CRF table): reduced with the first one.
    For each ascending chunk: compute it, then issue ONE asynchronous collective for exactly those rows of every tensor
    (round 5: `allreduce_coalesced` -- a step puts `chunks` collectives on the wire instead of chunks x tensors small ones,
    each of which pays RCCL's launch) -- on RCCL the collective waits (on its own stream) for what the current stream holds
    at that moment, i.e. for the chunk just enqueued, and travels while the next chunk computes.  Returns the list of
    completion callables (finish_pending waits for them); [] when not distributed (the chunks are still computed).
    Element for element the result is that of one all-reduce of the whole buffer: every element is summed over the ranks
    exactly once."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    pending: list = []
    first = True
    n_coll = 0
    for g0, g1 in chunk_bounds(P, chunks):
        compute_chunk(g0, g1)
        if not distributed:
            continue
        pieces = [t[g0:g1] for t in rows] + (list(tail) if first else [])
        first = False
        fin, n = _all_reduce_pieces(pieces, group)
        if fin is not None:
            pending.append(fin)
        n_coll += n
    LAST_EXCHANGE["collectives"] = n_coll
    return pending


def finish_pending(pending) -> int:
    """Wait for the collectives chunked_all_reduce / all_reduce_gradients(pending=...) left in flight."""
    n = 0
    for fin in pending or ():
        fin()
        n += 1
    return n


def _gather_rows(out: torch.Tensor, inp: torch.Tensor, group, direct: bool, async_op: bool = False):
    """out[r * n : (r + 1) * n] = rank r's `inp` (n = inp.shape[0]).  direct: every rank sends its block straight to
    every peer (a list all-to-all = grouped point-to-point sends over all seven xGMI links at once) instead of the
    library's ring all-gather; backends without a list all-to-all (gloo) always use the all-gather."""
    if direct and dist.get_backend(group) == "nccl":
        world = dist.get_world_size(group)
        return dist.all_to_all(list(out.chunk(world, dim=0)), [inp] * world, group=group, async_op=async_op)
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)


def start_view_gather(view_colors: torch.Tensor, camposes: torch.Tensor, group=None, direct: bool = False):
    """Issue the all-gathers of one view-parallel step asynchronously (called from the rasterizer's backward between
    its two halves when GaussianRasterizer(..., gather_group=...) is used).  Returns None when not distributed, else
    (works, colours of all ranks [world*N, P, 3], camera centres of all ranks [world*N, 3])."""
    if group is True:
        group = None
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    world = dist.get_world_size(group)
    cams = camposes.reshape(-1, 3).contiguous()
    vc = view_colors.contiguous()
    vc_all = torch.empty((world * vc.shape[0],) + tuple(vc.shape[1:]), dtype=vc.dtype, device=vc.device)
    cams_all = torch.empty((world * cams.shape[0], 3), dtype=cams.dtype, device=cams.device)
    works = [_gather_rows(vc_all, vc, group, direct, async_op=True),
             dist.all_gather_into_tensor(cams_all, cams, group=group, async_op=True)]
    return works, vc_all, cams_all


def exchange_view_gradients(params: Iterable[torch.Tensor], shs: torch.Tensor, deferred: dict, group=None,
                            average: bool = False, algo: str = "rccl", sh_backward=None) -> dict:
    """Gradient exchange of one view-parallel step when the rasterizer ran with `defer_sh_grad=True`.

    params   -- the non-SH parameters (their `.grad`s are summed over ranks by all_reduce_gradients);
    shs      -- the SH parameter [P, M, 3]; its `.grad` is set to the gradient summed over ALL ranks' views;
    deferred -- `GaussianRasterizer.deferred` after backward (this rank's view colour gradients [N, P, 3], camera
                centres [N, 3], means3D, M, sh_degree).
    Works at world size 1 too (no collective, same kernel).  `sh_backward` lets CPU tests inject a reference for
    the HIP kernel; the product default is rasterizer.sh_backward_views, which raises without the extension.
    Returns the element counts put on the wire."""
    if not deferred or "view_colors" not in deferred:
        raise RuntimeError("exchange_view_gradients: run backward through GaussianRasterizer(..., defer_sh_grad=True) first")
    vc, cams = deferred["view_colors"], deferred["camposes"].reshape(-1, 3)
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    gathered = 0
    pending: list = []
    started = deferred.get("gather")
    if world > 1 and started is not None:  # the backward already issued the all-gathers: just wait for them
        works, vc_all, cams_all = started
        for w in works:
            w.wait()
        gathered = vc_all.numel() + cams_all.numel()
        vc, cams = vc_all, cams_all
    elif world > 1:
        # rank-major concatenation along dim 0 (the layout every backend accepts)
        vc_all = torch.empty((world * vc.shape[0],) + tuple(vc.shape[1:]), dtype=vc.dtype, device=vc.device)
        cams_all = torch.empty((world * cams.shape[0], 3), dtype=cams.dtype, device=cams.device)
        _gather_rows(vc_all, vc.contiguous(), group, direct=(algo == "direct"))
        dist.all_gather_into_tensor(cams_all, cams.contiguous(), group=group)
        gathered = vc_all.numel() + cams_all.numel()
        vc, cams = vc_all, cams_all
    # the all-reduce of the non-SH gradients is issued behind the all-gathers and left in flight while the SH rows
    # are rebuilt (the kernel only needs the gathered colours)
    reduced = all_reduce_gradients(params, group=group, average=average, algo=algo, pending=pending)
    if sh_backward is None:
        from .rasterizer import sh_backward_views as sh_backward
    g = sh_backward(deferred["means3D"], cams, vc, deferred["M"], deferred["sh_degree"])
    for finish in pending:
        finish()
    if average:
        g = g / world
    shs.grad = g if shs.grad is None else shs.grad + g
    return {"all_reduced": reduced, "all_gathered": gathered}
