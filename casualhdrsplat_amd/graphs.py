"""Whole training steps as HIP graphs (one launch per step instead of ~40).

A step of the rasterizer at BASELINE config c2 (100k Gaussians, 800x800) is 0.4 ms of kernels behind ~35 launches and
their Python glue: the host, not the GPU, sets its pace.  The sync-free mode of GaussianRasterizer (capacity=N: no
host read anywhere in forward or backward) makes the step capturable: `GraphedStep` records forward + the caller's
loss + backward once (torch.cuda.graph -- hipGraph on ROCm) and replays it.  Inputs are the tensors `fn` closes over:
update them IN PLACE between replays (optimizer steps do), never rebind them.

The reference is a trainer around this hot path (Readme.md:54); nothing in it prescribes how launches reach the GPU.

What `fn` may contain (ROCm 7.2 / PyTorch 2.10 on MI355X): everything this package enqueues is kernels only -- no
hipMemsetAsync / hipMemcpyAsync on a captured path.  A LONG captured autograd step (the multi-frame image-formation step)
returns wrong VALUES from PyTorch's two-pass reduction -- a `.mean()` / `.sum()` over more than a few ten thousand elements
-- from the second replay on, while its gradients stay right (examples/train_synthetic.py: mean_by_rows avoids that
kernel).  Round 5 blamed memset nodes of the runtime; round 6's reproducers (scripts/repro/) show that
neither HIP's memset nodes nor torch's reduction fail in isolation, that rasterizer-only captured steps are right, and that
the formation step fails with a torch-only stand-in for the rasterizer as well: it is the capture of that long torch step,
not this library.  Check a captured step's printed scalars against an eager step once.  No torch.linalg solver calls, no
element-wise fills of device tensors from host scalars (host-to-device copies).
"""
from __future__ import annotations

import os
from typing import Callable, Sequence

import torch

from .rasterizer import BinningOverflow, GaussianRasterizer, grown_capacity


class GraphedStep:
    """Capture `fn()` -- forward through GaussianRasterizer(..., capacity=N), loss, backward; every tensor it reads must
    already live on the GPU -- and replay it with `step()`.

    fn            returns a tensor or tuple of tensors (outputs of the step; they are overwritten by every replay).
                  Gradients land in the `.grad` of the leaves as usual: set them to None before constructing this object,
                  so they are allocated inside the graph's memory pool and keep their addresses.
    rasterizers   the GaussianRasterizer objects `fn` calls (all with a fixed `capacity`): their device overflow counters
                  are copied to pinned memory inside the graph; `check_overflow()` waits for the step and raises
                  BinningOverflow if a frame did not fit (rebuild the step with the capacity it names).
    warmup        eager steps on a side stream before the capture (allocator warm-up; also fills the pinned-buffer pool
                  the capture may not allocate from).
    Drop every reference to outputs of EARLIER eager steps on the same leaves before constructing this object (an autograd
    graph kept alive from the default stream keeps its AccumulateGrad nodes there, which breaks the capture), and read
    gradients through `grads` -- `leaf.grad` is rebound by any eager backward that runs in between.
    """

    def __init__(self, fn: Callable[[], object], rasterizers: Sequence[GaussianRasterizer] = (), warmup: int = 2,
                 params: Sequence[torch.Tensor] = ()):
        for r in rasterizers:
            if r.capacity is None:
                raise ValueError("GraphedStep needs GaussianRasterizer(..., capacity=N): the synchronous mode reads "
                                 "num_rendered on the host inside every forward and cannot be captured")
        self.rasterizers = list({id(r): r for r in rasterizers}.values())   # (a rasterizer listed twice counts once)
        self._drop_stale_graph_refs()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for r in self.rasterizers:   # an overflow during warm-up: fail before capturing a graph that renders empty frames
            r.check_overflow()
        self._drop_stale_graph_refs()
        for r in self.rasterizers:
            r._cell["captured"] = []     # every forward a rasterizer enqueues while the stream captures lands here
        self.graph = torch.cuda.CUDAGraph()
        dump = os.environ.get("HS_GRAPH_DUMP")      # (diagnostics: the captured DAG as a DOT file -- scripts/repro/graph_bisect.py)
        if dump:
            self.graph.enable_debug_mode()
        try:
            # (captured on the warm-up's stream: the AccumulateGrad nodes the warm-up left alive -- any leaf whose graph is
            # still referenced somewhere -- belong to that stream; on another stream the engine would run them across a
            # fork of the capture, where a gradient buffer can be handed back to the pool and reused before the forked
            # copy has read it: replays then deliver garbage gradients now and then)
            with torch.cuda.graph(self.graph, stream=side):
                self.outputs = fn()
        finally:
            # (also when `fn` throws inside the capture: a list left behind would make the rasterizer's later eager
            # forwards believe they are being captured)
            captured = {id(r): r._cell.pop("captured", []) for r in self.rasterizers}
        # the forwards captured above left their counter copies pending (nobody may wait inside a capture): ALL of them
        # -- `fn` may call one rasterizer several times (several views, an eval render in between), and a frame that
        # overflowed renders empty whichever call it was
        if dump:
            self.graph.debug_dump(dump)
        not_called = [i for i, r in enumerate(self.rasterizers) if not captured[id(r)]]
        if not_called:
            raise ValueError(f"GraphedStep: rasterizers {not_called} of `rasterizers` were not called by `fn` during the capture")
        self._pending = [(r, p) for r in self.rasterizers for p in captured[id(r)] if p is not None]
        if any(p is None for r in self.rasterizers for p in captured[id(r)]):
            raise ValueError("GraphedStep: a captured forward left no device counters to check (was the rasterizer given "
                             "capacity=None?)")
        self._warned_helps = False
        # `params`: the leaves whose gradients the step produces -- `grads` are the static tensors every replay rewrites
        self.grads = [p.grad for p in params]

    def _drop_stale_graph_refs(self):
        """A rasterizer keeps its `raster_settings` between calls; when a step assigns settings whose tensors are COMPUTED
        (camera matrices from a trajectory, an exposure from its logarithm: non-leaves with a grad_fn), the rasterizer keeps
        the previous step's autograd graph alive through them -- and with it the AccumulateGrad nodes of the leaves behind,
        bound to the stream that step ran on.  The next step finds those nodes still alive and reuses them; if the earlier
        step ran eagerly on the default stream, the capture's backward then touches the default stream and the capture
        dies (a segmentation fault inside hipStreamEndCapture on ROCm 7).  A computed tensor left in the settings by an
        earlier step can only be stale, so it is replaced by its detached self before the warm-up and before the capture."""
        for r in self.rasterizers:
            rs = r.raster_settings
            stale = {k: v.detach() for k, v in rs._asdict().items() if isinstance(v, torch.Tensor) and v.grad_fn is not None}
            if stale:
                # (said out loud: a `fn` that does NOT assign fresh settings on every call would lose the gradients that
                # flow through these tensors -- ADVICE r5)
                import warnings
                warnings.warn(f"GraphedStep: detached the computed tensors {sorted(stale)} a rasterizer's raster_settings still held "
                              "from an earlier step (they keep that step's autograd graph alive); `fn` must assign the settings "
                              "of every call itself, as image_formation.FrameRasterizers does", RuntimeWarning, stacklevel=3)
                r.raster_settings = rs._replace(**stale)

    def step(self):
        """Enqueue one replay; returns the (static) outputs of `fn`."""
        self.graph.replay()
        return self.outputs

    __call__ = step

    def check_overflow(self) -> list:
        """Wait for the latest replay and return num_rendered of every captured forward (in call order per rasterizer, the
        rasterizers in the order given); raises BinningOverflow if ANY of them overflowed."""
        torch.cuda.current_stream().synchronize()
        counts = []
        helps = 0
        for r, p in self._pending:
            if p.host is None:
                counts.append(None)
                continue
            n, over = int(p.host[0]) & 0xFFFFFFFF, int(p.host[1])
            helps += int(p.host[6])   # hs_counters.reserved[4]: waiting workgroups did silent predecessors' counting
            if over >= 2:
                # (the captured graph holds the blockIdx-ordered passes: ticket order needs a new capture)
                from . import _lib as L
                if L.load().hs_sort_tickets(-1) == 0:
                    L.load().hs_sort_tickets(1)
                    raise RuntimeError("libhdrsplat: a radix pass of the binning stage gave up waiting (hs_counters.overflow "
                                       "= 2; another process on this GPU?) and the frame was rendered empty; the library "
                                       "now uses ticket-ordered passes -- rebuild the GraphedStep and repeat the step")
                raise RuntimeError("libhdrsplat: a radix pass of the binning stage gave up (hs_counters.overflow = 2)")
            if over:
                raise BinningOverflow(n, p.capacity, f"; rebuild the GraphedStep with capacity >= {grown_capacity(n)}")
            counts.append(n)
        if helps and not self._warned_helps:
            # the frames are right (helping keeps the look-back chains moving), but the GPU is shared with other kernels and
            # the captured graph holds the blockIdx-ordered passes: ticket order is the faster mode there, and needs a new capture
            import warnings
            self._warned_helps = True
            warnings.warn(f"casualhdrsplat_amd: the captured step's radix passes had to help {helps} silent predecessors "
                          "(GPU shared with other kernels?); call _lib.load().hs_sort_tickets(1) and rebuild the GraphedStep "
                          "for ticket-ordered passes", RuntimeWarning, stacklevel=2)
        return counts
