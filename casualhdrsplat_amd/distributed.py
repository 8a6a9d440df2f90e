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
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _shared_flat(grads: Sequence[torch.Tensor]) -> torch.Tensor | None:
    """If all gradients are contiguous slices of one storage, return a 1-D tensor spanning them."""
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


def all_reduce_gradients(params: Iterable[torch.Tensor], group=None, average: bool = False) -> int:
    """Sum (or average) `.grad` of `params` over all ranks with one collective.  Returns the number
    of fp32 elements exchanged (0 when not distributed)."""
    grads = [p.grad for p in params if p is not None and p.grad is not None]
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    flat = _shared_flat(grads)
    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(dist.get_world_size(group))
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
