"""GaussianRasterizationSettings / GaussianRasterizer -- the drop-in Python boundary.

BASELINE.json's north_star fixes this API as the boundary ("drops in behind CasualHDRSplat's
GaussianRasterizer / GaussianRasterizationSettings Python API").  /root/reference holds no code
(SURVEY.md section 0), so names, argument meaning and error behaviour follow the published
diff_gaussian_rasterization package that API belongs to:

  * GaussianRasterizationSettings is a NamedTuple with the upstream field order; matrices are
    passed transposed (row-vector convention) and `projmatrix` is the full view*proj.
  * GaussianRasterizer(raster_settings).forward(means3D, means2D, opacities, shs=None,
    colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None) -> (color, radii);
    exactly one of shs/colors_precomp and one of (scales, rotations)/cov3D_precomp, else an
    Exception with the upstream message.
  * backward returns gradients for (means3D, means2D, sh, colors_precomp, opacities, scales,
    rotations, cov3Ds_precomp); grad of means2D is the NDC-scaled screen-space gradient used by
    densification.

HDR extension (image-formation order from /root/reference/assets/pipeline.png:
H --(exposure, shared CRF)--> I --(average over virtual poses)--> B; Readme.md:54): optional
settings fields `exposure`, `crf_table`, `crf_range`, `viewmatrices/projmatrices/camposes`
(N virtual poses) and `blur_domain`.  With `crf_table` set, forward returns (ldr, radii, hdr).

All compute happens in libhdrsplat.so (hand-written HIP for gfx950) through the C ABI of
include/hdrsplat.h; PyTorch only owns memory and streams.  There is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import warnings
import weakref

import torch
import torch.nn as nn

from . import _lib as L


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False
    antialiasing: bool = False
    # ---- HDR / motion-blur extension (all optional) ----
    exposure: Optional[torch.Tensor] = None       # scalar tensor, exposure time
    crf_table: Optional[torch.Tensor] = None      # [3,K] camera response on log-exposure knots
    crf_range: tuple = (-6.0, 3.0)                # (u_min, u_max) spanned by the K knots
    viewmatrices: Optional[torch.Tensor] = None   # [N,4,4] virtual poses inside the exposure window
    projmatrices: Optional[torch.Tensor] = None   # [N,4,4]
    camposes: Optional[torch.Tensor] = None       # [N,3]
    blur_domain: str = "ldr"                      # "ldr" (figure) or "hdr"
    # how the SH sum s becomes the Gaussian's linear radiance: "relu_shift" = max(s + 0.5, 0) (the published rule),
    # "exp" = e^s, "softplus" = ln(1 + e^s) (an HDR scene wants an unbounded, positive radiance)
    radiance_activation: str = "relu_shift"


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32c(t: Optional[torch.Tensor], dev) -> Optional[torch.Tensor]:
    if t is None:
        return None
    t = t.detach()
    if t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
        t = t.to(device=dev, dtype=torch.float32).contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev=None) -> int:
    """Raw HIP stream handle of the current PyTorch stream (the direct binding is ~10x cheaper per call than building a
    torch.cuda.Stream object, and a step asks twice)."""
    if _raw_stream is not None:
        idx = torch.cuda.current_device() if dev is None or dev.index is None else dev.index
        return _raw_stream(idx)
    return torch.cuda.current_stream(dev).cuda_stream


class _on_device:
    """Make `dev` the current HIP device for the duration of a launch, only if it is not already."""
    def __init__(self, dev):
        self.ctx = None
        if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
            self.ctx = torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


class _State:
    """Everything the backward needs that is not a differentiable input (upstream: geomBuffer,
    binningBuffer, imgBuffer, num_rendered)."""
    __slots__ = ("dims", "layout", "geom", "binning", "image", "num_rendered", "flags", "views", "projs",
                 "camposes", "bg", "tanfovx", "tanfovy", "scale_modifier", "crf_K", "crf_range", "W", "H",
                 "pending", "fwd_args", "keep")


class DensifyStats:
    """Per-Gaussian statistics a 3DGS trainer keeps between densification rounds, updated inside
    the backward kernel of every GaussianRasterizer call that was given this object: `grad_accum` += |means2D.grad.xy|,
    `denom` += 1 and `max_radii` = max(max_radii, radii), for the Gaussians rasterized in that call."""

    def __init__(self, P: int, device="cuda"):
        self.grad_accum = torch.zeros(P, dtype=torch.float32, device=device)
        self.denom = torch.zeros(P, dtype=torch.float32, device=device)
        self.max_radii = torch.zeros(P, dtype=torch.int32, device=device)

    def mean_grad(self) -> torch.Tensor:
        return self.grad_accum / self.denom.clamp_min(1.0)

    def reset(self) -> None:
        self.grad_accum.zero_(); self.denom.zero_(); self.max_radii.zero_()


_HELPED_FRAMES = [0]      # frames of this process whose binning stage reported look-back helps (hs_counters.reserved[4])
_PINNED_POOL: list = []  # recycled page-locked int32[8] buffers (one hs_counters each) (hipHostMalloc per step is slow)


class BinningOverflow(RuntimeError):
    """Sync-free mode: the frame needed `num_rendered` (tile, Gaussian) pairs but the binning buffers held `capacity`;
    the device rendered it empty (background only) instead of writing out of bounds."""
    def __init__(self, num_rendered: int, capacity: int, where: str = ""):
        self.num_rendered, self.capacity = num_rendered, capacity
        super().__init__(
            f"binning capacity {capacity} < num_rendered {num_rendered}: the frame was rendered empty{where}; "
            "re-run with a larger `capacity` (or capacity=None for the synchronous mode)")


class SortChainStalled(BinningOverflow):
    """A radix pass of the binning stage gave up waiting for an earlier block of its own launch (hs_counters.overflow = 2)
    and the frame was rendered empty.  Seen when several processes run these kernels on ONE GPU: two blockIdx-ordered
    passes can keep each other's blocks out (include/hdrsplat.h, hs_sort_tickets).  The library has switched to
    ticket-ordered passes for the rest of the process by the time this is raised, so -- like BinningOverflow, whose
    handlers therefore cover it -- repeating the step succeeds; forwards without a backward are repeated by the
    rasterizer itself."""
    def __init__(self):
        RuntimeError.__init__(
            self, "libhdrsplat: a radix pass of the binning stage gave up waiting for a predecessor's status word "
                  "(hs_counters.overflow = 2) and the frame was rendered empty -- is another process running the same "
                  "kernels on this GPU?  Switched to ticket-ordered passes (hs_sort_tickets(1), as HS_SORT_TICKETS=1 "
                  "would from the start); repeat the step")
        self.num_rendered, self.capacity = 0, 0


def grown_capacity(num_rendered: int) -> int:
    """Capacity to retry with after an overflow: 1.5 x the pair count the device reported."""
    return int(num_rendered * 1.5) + 4096


class _Pending:
    """Deferred overflow check for the sync-free (fixed capacity) mode: the forward copies the device counters
    {num_rendered, overflow} into a pinned buffer behind its kernels and records an event; nothing waits for it
    until someone asks."""
    def __init__(self, host, event, capacity, ticket_order=0):
        self.host, self.event, self.capacity = host, event, capacity
        self.ticket_order = ticket_order   # chain-position mode the frame was ENQUEUED under (hs_sort_tickets(-1) then)

    def resolve(self):
        """(num_rendered, overflowed) -- waits for the forward's counter copy on first use."""
        if self.host is not None:  # first call: wait for the copy, recycle the pinned buffer, remember the verdict
            self.event.synchronize()
            self.n, self.overflow = int(self.host[0]) & 0xFFFFFFFF, int(self.host[1])
            helps = int(self.host[6])   # hs_counters.reserved[4]
            slow_ranges = int(self.host[4])   # hs_counters.reserved[2]
            _PINNED_POOL.append(self.host)
            self.host = None
            if slow_ranges and L.load().hs_depth_sort(-1) == 1:
                # the counting depth sort met a depth sliver holding thousands of instances (a wall of Gaussians seen head-on):
                # its range did not fit the LDS and one workgroup sorted it through memory.  The frame is right; the look-back
                # passes do not care how the depths are distributed
                L.load().hs_depth_sort(0)
                warnings.warn(f"casualhdrsplat_amd: {slow_ranges} range(s) of the counting depth sort did not fit on chip "
                              "(many instances at one depth); using the look-back passes for the depth sort from now on",
                              RuntimeWarning, stacklevel=3)
            if helps:
                _HELPED_FRAMES[0] += 1
            # (the second such frame decides: one late block on a cold start is not a shared GPU)
            if helps and _HELPED_FRAMES[0] >= 2 and L.load().hs_sort_tickets(-1) == 0:
                # waiting workgroups of the binning stage had to do silent predecessors' counting for them: other kernels
                # (another process on this GPU) keep blocks of ours out.  The frame is right; ticket order, in which nobody
                # waits for a block that has not started, is the faster mode under those conditions
                L.load().hs_sort_tickets(1)
                warnings.warn("casualhdrsplat_amd: the GPU is shared with other kernels (the radix passes had to help "
                              f"{helps} silent predecessors); using ticket-ordered passes from now on", RuntimeWarning,
                              stacklevel=3)
        if self.overflow >= 2:
            # judged by the mode THIS frame was enqueued under, not the process-wide mode of the moment: with several
            # forwards in flight the first stalled frame switches the library to tickets, and the others that stalled in
            # blockIdx order are the same retryable event, not damaged scratch
            if not self.ticket_order:
                # blockIdx-ordered passes met a dispatch order they cannot live with (another process on the GPU):
                # ticket order from here on, and the caller repeats the step
                lib = L.load()
                if lib.hs_sort_tickets(-1) == 0:
                    lib.hs_sort_tickets(1)
                    warnings.warn("casualhdrsplat_amd: a radix pass gave up waiting (GPU shared with another process?); "
                                  "using ticket-ordered passes from now on", RuntimeWarning, stacklevel=3)
                raise SortChainStalled()
            raise RuntimeError("libhdrsplat: a radix pass of the binning stage gave up waiting for a predecessor's status "
                               "word although the passes were ticket-ordered (hs_counters.overflow = 2: damaged sort "
                               "scratch?); the frame was rendered empty")
        return self.n, self.overflow != 0

    def check(self, where: str = ""):
        n, over = self.resolve()
        if over:
            raise BinningOverflow(n, self.capacity, where)
        return n


def _run_forward(settings: GaussianRasterizationSettings, means3D, opacities, shs, colors_precomp, scales,
                 rotations, cov3D_precomp, exposure, crf_table, capacity: Optional[int], want_invdepth: bool = False):
    lib = L.load()
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("casualhdrsplat_amd rasterizes on an MI355X only: tensors must live on a cuda (HIP) device")
    P = means3D.shape[0]
    W, H = int(settings.image_width), int(settings.image_height)
    if settings.viewmatrices is not None:
        views = _f32c(settings.viewmatrices, dev).reshape(-1, 16)
        projs = _f32c(settings.projmatrices, dev).reshape(-1, 16)
        campos = _f32c(settings.camposes, dev).reshape(-1, 3)
    else:
        views = _f32c(settings.viewmatrix, dev).reshape(1, 16)
        projs = _f32c(settings.projmatrix, dev).reshape(1, 16)
        campos = _f32c(settings.campos, dev).reshape(1, 3)
    N = views.shape[0]
    if projs.shape[0] != N or campos.shape[0] != N:
        raise ValueError("viewmatrices, projmatrices and camposes must have the same leading dimension")
    bg = _f32c(settings.bg, dev).reshape(3)
    M = shs.shape[1] if shs is not None else 0
    hdr = crf_table is not None
    flags = 0
    if hdr:
        flags |= L.HS_FLAG_HDR
        if exposure is None:
            exposure = torch.ones((), device=dev)
        if settings.blur_domain == "hdr":
            flags |= L.HS_FLAG_BLUR_HDR
        elif settings.blur_domain != "ldr":
            raise ValueError("blur_domain must be 'ldr' or 'hdr'")
    if settings.debug:
        flags |= L.HS_FLAG_DEBUG
    if settings.antialiasing:
        flags |= L.HS_FLAG_ANTIALIAS
    act = getattr(settings, "radiance_activation", "relu_shift")
    if act == "exp":
        flags |= L.HS_FLAG_RADIANCE_EXP
    elif act == "softplus":
        flags |= L.HS_FLAG_RADIANCE_SOFTPLUS
    elif act != "relu_shift":
        raise ValueError("radiance_activation must be 'relu_shift', 'exp' or 'softplus'")
    exposure = None if exposure is None else _f32c(exposure, dev).reshape(1)
    crf_table = _f32c(crf_table, dev)
    crf_K = int(crf_table.shape[1]) if hdr else 0

    sync_mode = capacity is None
    dims, sizes, layout = L.plan(P, M, int(settings.sh_degree), W, H, N, 0 if sync_mode else int(capacity), crf_K)
    geom = _empty(max(int(sizes.geom_bytes), 256), torch.uint8, dev, "geom")
    out_color = _empty((3, H, W), torch.float32, dev, "out_color")
    out_hdr = _empty((3, H, W), torch.float32, dev, "out_hdr") if hdr else None
    radii = _empty(P, torch.int32, dev, "radii")
    invdepth = _empty((N, H, W), torch.float32, dev, "invdepth") if want_invdepth else None

    a = L.hs_fwd_args()
    a.dims = dims
    a.tanfovx, a.tanfovy, a.scale_modifier = float(settings.tanfovx), float(settings.tanfovy), float(settings.scale_modifier)
    a.flags = flags
    a.crf_K, a.crf_umin, a.crf_umax = crf_K, float(settings.crf_range[0]), float(settings.crf_range[1])
    a.bg, a.viewmatrices, a.projmatrices, a.camposes = _ptr(bg), _ptr(views), _ptr(projs), _ptr(campos)
    a.means3D, a.opacities, a.shs, a.colors_precomp = _ptr(means3D), _ptr(opacities), _ptr(shs), _ptr(colors_precomp)
    a.scales, a.rotations, a.cov3D_precomp = _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp)
    a.exposure, a.crf_table = _ptr(exposure), _ptr(crf_table)
    a.geom = geom.data_ptr()
    a.out_color, a.out_hdr, a.radii = out_color.data_ptr(), _ptr(out_hdr), radii.data_ptr()
    a.out_invdepth = _ptr(invdepth)

    st = _State()
    st.pending = None
    stream = _stream()
    ticket_order = lib.hs_sort_tickets(-1)   # the chain-position mode this frame's passes are enqueued under
    if sync_mode:
        # upstream semantics: one host read of num_rendered between the scan and the binning
        a.stages = L.HS_STAGE_PREPROCESS
        L.check(lib.hs_forward(C.byref(a), stream), "hs_forward[preprocess]")
        R = int(geom[:4].view(torch.int32).item()) & 0xFFFFFFFF if P > 0 else 0
        dims, sizes, layout = L.plan(P, M, int(settings.sh_degree), W, H, N, R, crf_K)
        a.dims = dims
        a.stages = L.HS_STAGE_BIN | L.HS_STAGE_RENDER
    else:
        R = -1
        a.stages = L.HS_STAGE_ALL
    binning = _empty(max(int(sizes.binning_bytes), 256), torch.uint8, dev, "binning")
    image = _empty(max(int(sizes.image_bytes), 256), torch.uint8, dev, "image")
    a.binning, a.image = binning.data_ptr(), image.data_ptr()
    # the frame's counters {num_rendered, overflow, helps} reach the host without a copy on the stream: the binning stage's
    # last kernel writes them into this page-locked buffer (hs_fwd_args.counters_host); nobody waits for them until
    # someone asks (the synchronous mode too: its sorts can report a stalled chain -- overflow = 2 -- like any other)
    if not _PINNED_POOL and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("GaussianRasterizer inside a graph capture needs one eager step first (graphs.GraphedStep "
                           "does that): page-locked memory cannot be allocated while a stream is capturing")
    host = _PINNED_POOL.pop() if _PINNED_POOL else torch.empty(8, dtype=torch.int32).pin_memory()
    a.counters_host = host.data_ptr()
    L.check(lib.hs_forward(C.byref(a), stream), "hs_forward")
    a.counters_host = None   # (the saved argument struct may be replayed by profiling helpers: never into a recycled buffer)
    ev = torch.cuda.Event()
    ev.record()
    st.pending = _Pending(host, ev, R if sync_mode else int(capacity), ticket_order)

    st.dims, st.layout, st.geom, st.binning, st.image = dims, layout, geom, binning, image
    st.num_rendered, st.flags, st.views, st.projs, st.camposes, st.bg = R, flags, views, projs, campos, bg
    st.tanfovx, st.tanfovy, st.scale_modifier = a.tanfovx, a.tanfovy, a.scale_modifier
    st.crf_K, st.crf_range, st.W, st.H = crf_K, (a.crf_umin, a.crf_umax), W, H
    st.fwd_args = a
    # inputs only: an OUTPUT here would close a cycle output -> grad_fn -> ctx.st -> output through the C++ autograd
    # node, which Python's collector cannot see, and leak the whole state (~340 MB per step at 1M Gaussians / 1080p)
    st.keep = (exposure, crf_table, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp)
    return out_color, out_hdr, radii, st, exposure, crf_table, invdepth


# HS_GUARD=1 in the environment (read at import): every tensor the library writes -- the four state workspaces, the flat
# gradient buffer, the output images -- is carved out of a larger allocation with _GUARD_BYTES of 0xA5 on either side, and
# check_guards() (after a synchronisation) reports every guard byte that changed: the GPU build's stand-in for an address
# sanitizer (there is none on this pool), tests/test_gpu_parity.py::test_no_kernel_writes_outside_its_buffers.  A kernel
# that overruns its buffer is harmless next to torch's rounded eager allocations and fatal inside a captured graph's
# packed pool, where the neighbour is somebody's live tensor.
_GUARD = os.environ.get("HS_GUARD", "") not in ("", "0")
_GUARD_BYTES = 4096
_guarded: list = []


def _empty(shape, dtype, dev, name: str) -> torch.Tensor:
    if not _GUARD:
        return torch.empty(shape, dtype=dtype, device=dev)
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    n = torch.empty((), dtype=dtype).element_size()
    for d in shape:
        n *= int(d)
    full = torch.full((n + 2 * _GUARD_BYTES,), 0xA5, dtype=torch.uint8, device=dev)
    _guarded.append((name, full, n))
    return full[_GUARD_BYTES:_GUARD_BYTES + n].view(dtype).view(shape)


def check_guards() -> list:
    """HS_GUARD=1: [(buffer name, 'before' / 'after', first damaged offset from the buffer's edge, damaged bytes)] over every
    guarded buffer handed out since the last call (waits for the device); [] = no kernel wrote outside its buffers."""
    torch.cuda.synchronize()
    bad = []
    for name, full, n in _guarded:
        for side, zone in (("before", full[:_GUARD_BYTES]), ("after", full[_GUARD_BYTES + n:])):
            hit = (zone != 0xA5).nonzero()
            if hit.numel():
                first = int(hit[0]) if side == "after" else _GUARD_BYTES - 1 - int(hit[-1])
                bad.append((name, side, first, int(hit.numel())))
    _guarded.clear()
    return bad


# id(leaf) -> [open, peak]: view-parallel rasterizer calls (reduce_group) that took the leaf and whose backward is still to
# come, and the largest that number has been since it was last zero; see _RasterizeGaussians.forward.  Keyed by id (a
# tensor's == is element-wise); a finalizer drops the entry with the tensor.  `peak`, not `open`, decides at backward time:
# of two calls sharing a leaf the LATER one's backward runs first and leaves open == 1 for the earlier one, whose gradient
# the engine nevertheless adds to the later one's.
_OPEN_CONSUMERS: dict = {}


def _open_consumers(t: torch.Tensor) -> int:
    """Peak number of simultaneously open view-parallel calls on leaf `t` in the current episode (0: none open)."""
    e = _OPEN_CONSUMERS.get(id(t))
    return e[1] if e else 0


def _open_consumers_add(t: torch.Tensor, n: int) -> None:
    k = id(t)
    e = _OPEN_CONSUMERS.get(k)
    if e is None:
        weakref.finalize(t, _OPEN_CONSUMERS.pop, k, None)
        e = _OPEN_CONSUMERS[k] = [0, 0]
    e[0] = max(0, e[0] + n)
    e[1] = max(e[1], e[0]) if e[0] else 0      # (back to zero open calls: the episode is over)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                exposure, crf_table, viewmats, projmats, camposes, raster_settings, capacity, return_alpha=False,
                deferred=None, return_invdepth=False, densify=None, aux=None):
        dev = means3D.device
        m3 = _f32c(means3D, dev)
        op = _f32c(opacities, dev)
        shs = _f32c(sh, dev) if sh is not None and sh.numel() else None
        cp = _f32c(colors_precomp, dev) if colors_precomp is not None and colors_precomp.numel() else None
        sc = _f32c(scales, dev) if scales is not None and scales.numel() else None
        ro = _f32c(rotations, dev) if rotations is not None and rotations.numel() else None
        cv = _f32c(cov3Ds_precomp, dev) if cov3Ds_precomp is not None and cov3Ds_precomp.numel() else None
        with _on_device(dev):  # kernels are launched on the tensors' GPU, whatever the current device is
            color, hdr, radii, st, exp_t, crf_t, invd = _run_forward(raster_settings, m3, op, shs, cp, sc, ro, cv,
                                                                     exposure, crf_table, capacity, return_invdepth)
        ctx.st = st
        ctx.aux = aux
        # May the chunked all-reduce of a view-parallel backward (reduce_group) still be in flight when backward() hands
        # the gradients to autograd?  Only if autograd then does nothing with them but store them: every differentiable
        # input a LEAF without a .grad yet (AccumulateGrad keeps the tensor it is given).  A non-leaf input -- the usual
        # 3DGS wiring: scales = exp(raw), opacity = sigmoid(raw), rotations = normalize(raw), shs = cat(dc, rest) -- makes
        # autograd run the activation's backward on the compute stream right away, and a leaf with a .grad is added to:
        # both would read rows RCCL is still summing.  Then backward() waits for the collectives itself.
        # ... and only if this call is the SOLE consumer of those leaves until its backward has run: a leaf that
        # feeds two rasterizer calls before one backward -- a multi-frame step, a second view -- has its two incoming
        # gradients summed by the autograd engine on the compute stream, again while RCCL may still be reducing the rows.
        # Every differentiable leaf carries the number of view-parallel calls whose backward is still to come
        # (_OPEN_CONSUMERS); a count above one at forward OR at backward time makes backward() wait.  (A forward whose
        # graph is dropped without a backward leaves its count behind: the conservative side -- later calls wait.)
        diff = [t for t in (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, exposure, crf_table)
                if isinstance(t, torch.Tensor) and t.requires_grad]
        ctx.reduce_may_stay_in_flight = all(t.is_leaf and t.grad is None for t in diff)
        ctx.open_leaves = []
        # (needs_input_grad: all False when the caller runs under torch.no_grad() -- no backward will come to close the count)
        if aux is not None and aux.get("reduce_group") is not None and any(ctx.needs_input_grad):
            for t in diff:
                if t.is_leaf:
                    _open_consumers_add(t, 1)
                    ctx.open_leaves.append(weakref.ref(t))
            if any(_open_consumers(t) > 1 for t in diff if t.is_leaf):
                ctx.reduce_may_stay_in_flight = False
        if aux is not None:  # what GaussianRasterizer keeps of the call (never the outputs: see st.keep)
            aux["pending"], aux["num_rendered"] = st.pending, st.num_rendered
            if aux.get("keep_state"):
                aux["state"] = st
        ctx.set_materialize_grads(False)  # an unused output (e.g. the radiance image) must not cost a zero image
        ctx.deferred = deferred if shs is not None else None
        ctx.densify = densify
        ctx.exp_shape = None if exposure is None else tuple(exposure.shape)
        ctx.pose_shapes = (tuple(viewmats.shape), tuple(projmats.shape), tuple(camposes.shape))
        ctx.has = (shs is not None, cp is not None, sc is not None, cv is not None, exposure is not None,
                   crf_table is not None)
        ctx.save_for_backward(m3, op, shs, cp, sc, ro, cv, exp_t, crf_t)
        ctx.mark_non_differentiable(radii)
        ctx.n_out = (hdr is not None, bool(return_alpha), bool(return_invdepth))
        outs = (color, radii) + ((hdr,) if hdr is not None else ())
        if return_alpha:
            # accumulated opacity A = 1 - mean over poses of the final transmittance (newer rasterizers expose it)
            d = st.dims
            ft = st.image[st.layout.final_T:st.layout.final_T + 4 * d.n_poses * d.W * d.H].view(torch.float32)
            outs = outs + (1.0 - ft.reshape(d.n_poses, d.H, d.W).mean(dim=0),)
        if return_invdepth:
            # expected inverse depth sum_i alpha_i T_i / z_i, averaged over the poses (newer rasterizers' third output)
            outs = outs + (invd[0] if invd.shape[0] == 1 else invd.mean(dim=0),)
        return outs

    @staticmethod
    def backward(ctx, grad_color, grad_radii=None, *more):
        has_hdr, has_alpha, has_invd = ctx.n_out
        more = list(more)
        grad_hdr = more.pop(0) if has_hdr else None
        grad_alpha = more.pop(0) if has_alpha else None
        grad_invd = more.pop(0) if has_invd else None
        st: _State = ctx.st
        saved = ctx.saved_tensors
        dev = saved[0].device
        if grad_color is None:  # the loss used only the other outputs
            grad_color = torch.zeros(3, st.H, st.W, dtype=torch.float32, device=dev)
        gcol = _f32c(grad_color, dev)
        ghdr = _f32c(grad_hdr, dev) if grad_hdr is not None else None
        galpha = _f32c(grad_alpha, dev) if grad_alpha is not None else None
        ginvd = _f32c(grad_invd, dev) if grad_invd is not None else None
        want_pose = any(ctx.needs_input_grad[10:13])
        with _on_device(dev):
            g = _launch_backward(st, saved, gcol, ghdr, L.HS_BWD_ALL, want_pose, galpha,
                                 defer_sh=ctx.deferred is not None, ginvd=ginvd, densify=ctx.densify,
                                 gather_group=None if ctx.deferred is None else ctx.deferred.get("gather_group"),
                                 gather_direct=bool(ctx.deferred.get("gather_direct")) if ctx.deferred else False,
                                 reduce_group=None if ctx.aux is None else ctx.aux.get("reduce_group"),
                                 reduce_chunks=0 if ctx.aux is None else int(ctx.aux.get("reduce_chunks") or 0))
        # this call's backward is here: its leaves have one consumer less to wait for -- but if another call took one of
        # them in the meantime (count above one NOW), the engine will add that call's gradient to ours: wait below
        shared = False
        for ref in getattr(ctx, "open_leaves", ()):
            t = ref()
            if t is not None:
                shared = shared or _open_consumers(t) > 1     # (peak of the episode: see _OPEN_CONSUMERS)
                _open_consumers_add(t, -1)
        if ctx.aux is not None and g.get("_reduce_pending") is not None:
            if ctx.reduce_may_stay_in_flight and not shared:
                # (extended, never replaced: several calls of one rasterizer may each leave collectives in flight before
                # finish_reduce() is called)
                ctx.aux["cell"].setdefault("reduce_pending", []).extend(g["_reduce_pending"])   # GaussianRasterizer.finish_reduce() waits for these
            else:
                # (see forward: autograd is about to USE these gradients.  On RCCL waiting = the compute stream waits for
                # the communication stream, the host does not block; the overlap with this backward's own chunks stays)
                from .distributed import finish_pending
                ctx.aux["cell"]["reduce_waited_in_backward"] = (ctx.aux["cell"].get("reduce_waited_in_backward", 0)
                                                                + finish_pending(g["_reduce_pending"]))
                ctx.aux["cell"].setdefault("reduce_pending", [])
        if ctx.deferred is not None:
            # view-parallel exchange: hand the per-view colour gradients to distributed.exchange_view_gradients
            ctx.deferred.update(view_colors=g["view_colors"], camposes=st.camposes, means3D=saved[0],
                                M=st.dims.M, sh_degree=st.dims.sh_degree, flat=g["_flat"], gather=g.get("_gather"))
        if st.pending is not None and not torch.cuda.is_current_stream_capturing():
            # (inside a graph capture nobody may wait: graphs.GraphedStep.check_overflow reads the counters after a replay)
            # sync-free mode: the kernels are already queued; only now look at the forward's counters.  An overflowed
            # forward handed the caller an EMPTY frame, so the loss this gradient belongs to is already wrong: the
            # step cannot be repaired here -- raise, and let the rasterizer grow its capacity for the next call
            n, over = st.pending.resolve()
            if over:
                if ctx.aux is not None and ctx.aux.get("cell") is not None:
                    # the rasterizer's own cell, not this call's bookkeeping: other forwards may have run in between
                    cell = ctx.aux["cell"]
                    cell["grow_to"] = max(int(cell.get("grow_to") or 0), grown_capacity(n))
                raise BinningOverflow(n, st.pending.capacity, " and this backward belongs to that empty frame (the "
                                      "rasterizer has grown its capacity for the following calls)")
            st.num_rendered = n
            st.pending = None
        has_sh, has_cp, has_sc, has_cv, has_exp, has_crf = ctx.has
        hdr = bool(st.flags & L.HS_FLAG_HDR)
        return (g["means3D"], g["means2D"], g["shs"] if has_sh else None, g["colors_precomp"] if has_cp else None,
                g["opacities"], g["scales"] if has_sc else None, g["rotations"] if has_sc else None,
                g["cov3D_precomp"] if has_cv else None,
                g["exposure"].reshape(ctx.exp_shape) if (hdr and has_exp) else None,
                g["crf_table"] if has_crf else None,
                g["viewmatrices"].reshape(ctx.pose_shapes[0]) if want_pose else None,
                g["projmatrices"].reshape(ctx.pose_shapes[1]) if want_pose else None,
                g["camposes"].reshape(ctx.pose_shapes[2]) if want_pose else None, None, None, None, None, None, None,
                None)


def _launch_backward(st: "_State", saved, gcol, ghdr, stages: int, want_pose: bool = False, galpha=None,
                     defer_sh: bool = False, ginvd=None, densify=None, gather_group=None, stats=None,
                     timeline=None, gather_direct: bool = False, reduce_group=None, reduce_chunks: int = 0) -> dict:
    """Enqueue hs_backward.  All per-Gaussian gradients are carved out of ONE flat fp32 buffer (the
    layout casualhdrsplat_amd.distributed all-reduces in a single RCCL call): [means3D | opacities | colors | scales |
    rotations | cov3D | exposure | crf_table | sh | means2D | pose gradients].  means2D -- the screen-space
    gradient of THIS view, a densification statistic and not a parameter gradient -- comes after everything a
    view-parallel step sums over the ranks, so the summed set is one contiguous span without it."""
    lib = L.load()
    m3, op, shs, cp, sc, ro, cv, exp_t, crf_t = saved
    dev = m3.device
    P, M = st.dims.P, st.dims.M
    _, sizes, _ = L.plan(P, M, st.dims.sh_degree, st.W, st.H, st.dims.n_poses, st.dims.capacity, st.dims.crf_K)
    bwd = _empty(max(int(sizes.bwd_bytes), 256), torch.uint8, dev, "bwd")
    hdr = bool(st.flags & L.HS_FLAG_HDR)
    # (the SH rows -- four fifths of the bytes at degree 3 -- come last of the summed span, so a chunked exchange moves the
    # other per-Gaussian rows as a few short slices and the SH rows of a chunk as ONE long one)
    spec = [("means3D", (P, 3), True), ("opacities", (P, 1), True), ("colors_precomp", (P, 3), cp is not None),
            ("scales", (P, 3), sc is not None), ("rotations", (P, 4), ro is not None),
            ("cov3D_precomp", (P, 6), cv is not None), ("exposure", (1,), hdr), ("crf_table", (3, st.crf_K), hdr),
            ("shs", (P, M, 3), shs is not None and not defer_sh),
            ("means2D", (P, 3), True),
            ("viewmatrices", (st.dims.n_poses, 16), want_pose), ("projmatrices", (st.dims.n_poses, 16), want_pose),
            ("camposes", (st.dims.n_poses, 3), want_pose)]
    total = 0
    offs = {}
    for name, shape, on in spec:
        if on:
            n = 1
            for d in shape:
                n *= d
            offs[name] = (total, n, shape)
            total += (n + 3) // 4 * 4  # keep every slice 16-byte aligned
    flat = _empty(max(total, 4), torch.float32, dev, "flat_gradients")  # fully written by the kernels (16-byte slice pads aside)
    g = {name: None for name, _, _ in spec}
    for name, (o, n, shape) in offs.items():
        g[name] = flat[o:o + n].view(shape)
    if hdr and not (stages & L.HS_BWD_CRF):
        g["exposure"].zero_()   # otherwise both are fully written by the CRF stage
        g["crf_table"].zero_()
    g["_flat"] = flat
    # deferred SH gradient: the per-view colour gradients live outside the flat (all-reduced) buffer
    g["view_colors"] = (_empty((st.dims.n_poses, P, 3), torch.float32, dev, "view_colors")
                        if defer_sh and shs is not None else None)

    a = L.hs_bwd_args()
    a.dims = st.dims
    a.tanfovx, a.tanfovy, a.scale_modifier = st.tanfovx, st.tanfovy, st.scale_modifier
    a.flags, a.stages = st.flags, stages
    a.crf_K, a.crf_umin, a.crf_umax = st.crf_K, st.crf_range[0], st.crf_range[1]
    a.bg, a.viewmatrices, a.projmatrices, a.camposes = _ptr(st.bg), _ptr(st.views), _ptr(st.projs), _ptr(st.camposes)
    a.means3D, a.opacities, a.shs, a.colors_precomp = _ptr(m3), _ptr(op), _ptr(shs), _ptr(cp)
    a.scales, a.rotations, a.cov3D_precomp = _ptr(sc), _ptr(ro), _ptr(cv)
    a.exposure, a.crf_table = _ptr(exp_t), _ptr(crf_t)
    a.geom, a.binning, a.image, a.bwd = st.geom.data_ptr(), st.binning.data_ptr(), st.image.data_ptr(), bwd.data_ptr()
    a.dL_dout_color, a.dL_dout_hdr, a.dL_dout_alpha = _ptr(gcol), _ptr(ghdr), _ptr(galpha)
    a.dL_dmeans3D, a.dL_dmeans2D, a.dL_dopacities = _ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["opacities"])
    a.dL_dshs, a.dL_dcolors_precomp, a.dL_dscales = _ptr(g["shs"]), _ptr(g["colors_precomp"]), _ptr(g["scales"])
    a.dL_drotations, a.dL_dcov3D_precomp = _ptr(g["rotations"]), _ptr(g["cov3D_precomp"])
    a.dL_dexposure, a.dL_dcrf_table = _ptr(g["exposure"]), _ptr(g["crf_table"])
    a.dL_dviewmatrices, a.dL_dprojmatrices, a.dL_dcamposes = (_ptr(g["viewmatrices"]), _ptr(g["projmatrices"]),
                                                              _ptr(g["camposes"]))
    a.dL_dview_colors = _ptr(g["view_colors"])
    a.dL_dout_invdepth = _ptr(ginvd)
    if densify is not None:
        if densify.grad_accum.shape[0] != P or densify.grad_accum.device != dev:
            raise ValueError("DensifyStats was created for a different number of Gaussians or another device")
        a.densify_grad_accum, a.densify_denom = densify.grad_accum.data_ptr(), densify.denom.data_ptr()
        a.densify_max_radii = densify.max_radii.data_ptr()
    if stats is not None:  # diagnostic instantiation of the render backward only (render_stats)
        L.check(lib.hs_render_stats(None, C.byref(a), stats.data_ptr(), _ptr(timeline), _stream()), "hs_render_stats[bwd]")
    elif P == 0:
        flat.zero_()
    elif gather_group is not None and g["view_colors"] is not None and stages == L.HS_BWD_ALL:
        # view-parallel step: the all-gather of this view's colour gradients starts as soon as the record sums
        # exist and travels while the per-Gaussian backward runs (distributed.exchange_view_gradients waits for it)
        from . import distributed as D
        a.stages = L.HS_BWD_RENDER | L.HS_BWD_CRF | L.HS_BWD_SEGSUM
        L.check(lib.hs_backward(C.byref(a), _stream()), "hs_backward[render+segsum]")
        g["_gather"] = D.start_view_gather(g["view_colors"], st.camposes, gather_group, direct=gather_direct)
        a.stages = L.HS_BWD_PROJECT
        L.check(lib.hs_backward(C.byref(a), _stream()), "hs_backward[project]")
        a.stages = stages
    elif reduce_group is not None and reduce_chunks > 0 and stages == L.HS_BWD_ALL:
        # view-parallel step with the plain (all-reduce) exchange: the per-Gaussian half runs in ascending chunks of the
        # Gaussians and the all-reduce of a chunk's gradient rows starts while the next chunk computes
        # (distributed.chunked_all_reduce; the caller waits with distributed.finish_pending before reading a gradient)
        from . import distributed as D
        a.stages = L.HS_BWD_RENDER | L.HS_BWD_CRF | L.HS_BWD_SEGSUM
        L.check(lib.hs_backward(C.byref(a), _stream()), "hs_backward[render+segsum]")
        a.stages = L.HS_BWD_PROJECT

        def project(g0, g1):
            a.g_begin, a.g_end = int(g0), int(g1)
            L.check(lib.hs_backward(C.byref(a), _stream()), "hs_backward[project chunk]")

        rows = [g[k] for k in ("means3D", "opacities", "colors_precomp", "scales", "rotations", "cov3D_precomp", "shs")
                if g[k] is not None]
        tail = [g[k] for k in ("exposure", "crf_table") if g[k] is not None]
        g["_reduce_pending"] = D.chunked_all_reduce(rows, P, reduce_chunks, project, tail=tail,
                                                    group=None if reduce_group is True else reduce_group)
        a.g_begin, a.g_end, a.stages = 0, 0, stages
    else:
        L.check(lib.hs_backward(C.byref(a), _stream()), "hs_backward")
    return g


def replay_forward(out_tensor: torch.Tensor, stages: int = L.HS_STAGE_RENDER) -> None:
    """Profiling helper: re-enqueue selected forward stages of the call that produced `out_tensor`
    (same buffers, so the results are simply rewritten)."""
    st: _State = out_tensor.grad_fn.st
    a = st.fwd_args
    a.stages = stages
    # the state keeps no reference to the outputs: colour goes back into `out_tensor`, the rest into scratch
    a.out_color = out_tensor.data_ptr()
    scratch_hdr = torch.empty_like(out_tensor) if (st.flags & L.HS_FLAG_HDR) else None
    scratch_radii = torch.empty(max(st.dims.P, 1), dtype=torch.int32, device=out_tensor.device)
    a.out_hdr, a.radii, a.out_invdepth = _ptr(scratch_hdr), scratch_radii.data_ptr(), None
    L.check(L.load().hs_forward(C.byref(a), _stream()), "hs_forward[replay]")
    a.out_hdr, a.radii = None, None  # scratch dies with this call


def replay_backward(out_tensor: torch.Tensor, grad_color: torch.Tensor, stages: int = L.HS_BWD_ALL,
                    grad_hdr: Optional[torch.Tensor] = None) -> dict:
    """Profiling helper (bench.py's per-kernel roofline leg): re-enqueue the selected half of the
    backward of the forward call that produced `out_tensor`, outside autograd."""
    fn = out_tensor.grad_fn
    dev = out_tensor.device
    return _launch_backward(fn.st, fn.saved_tensors, _f32c(grad_color, dev),
                            None if grad_hdr is None else _f32c(grad_hdr, dev), stages)


RENDER_STAT_NAMES = ("bwd_trips", "bwd_empty_trips", "bwd_active_pixels", "bwd_culled", "bwd_hist_0", "bwd_hist_1_4",
                     "bwd_hist_5_8", "bwd_hist_9_16", "bwd_hist_17_32", "bwd_hist_33_64", "bwd_staged", "bwd_batches",
                     "fwd_trips", "fwd_empty_trips", "fwd_active_pixels", "fwd_culled", "fwd_staged", "fwd_batches")


def render_stats(out_tensor: torch.Tensor, grad_color: torch.Tensor, grad_hdr: Optional[torch.Tensor] = None,
                 timeline: bool = False) -> dict:
    """Profiling helper (bench.py's lane-utilisation / VALU-roofline leg): replays the render forward and backward of
    the call that produced `out_tensor` with the counting instantiation of the kernels (hs_render_stats) and returns
    the counters by name: (wave, entry) trips of the two compositing loops, how many found no active lane, the sum
    of active pixels (<= 128 per trip), entries removed by the half-tile test, staged entries and batches."""
    fn = out_tensor.grad_fn
    st: _State = fn.st
    dev = out_tensor.device
    stats = torch.zeros(L.HS_RENDER_STATS, dtype=torch.int64, device=dev)
    a = st.fwd_args
    a.stages = L.HS_STAGE_RENDER
    a.out_color = out_tensor.data_ptr()
    scratch_hdr = torch.empty_like(out_tensor) if (st.flags & L.HS_FLAG_HDR) else None
    a.out_hdr, a.out_invdepth = _ptr(scratch_hdr), None
    L.check(L.load().hs_render_stats(C.byref(a), None, stats.data_ptr(), None, _stream()), "hs_render_stats[fwd]")
    a.out_hdr = None  # the scratch image dies with this call: never leave its address in the saved argument struct
    d = st.dims
    n_wg = ((d.W + L.HS_TILE - 1) // L.HS_TILE) * ((d.H + L.HS_TILE - 1) // L.HS_TILE) * d.n_poses
    tl = torch.zeros(n_wg, 3, dtype=torch.int64, device=dev) if timeline else None
    _launch_backward(st, fn.saved_tensors, _f32c(grad_color, dev), None if grad_hdr is None else _f32c(grad_hdr, dev),
                     L.HS_BWD_RENDER, stats=stats, timeline=tl)
    vals = stats.cpu().tolist()
    res = dict(zip(RENDER_STAT_NAMES, vals))
    if timeline:  # [workgroup] -> (start, end) on the 100 MHz device clock, (XCC id << 32 | HW_ID); padding blocks dropped
        tl = tl.cpu()
        res["bwd_timeline"] = tl[tl[:, 1] > 0]
    return res


def sh_backward_views(means3D: torch.Tensor, camposes: torch.Tensor, view_colors: torch.Tensor, M: int,
                      sh_degree: int) -> torch.Tensor:
    """dL/dsh [P,M,3] from per-view colour gradients [V,P,3] and the V camera centres [V,3] (hs_sh_backward_views):
    the local half of the view-parallel gradient exchange (distributed.exchange_view_gradients)."""
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("casualhdrsplat_amd runs on an MI355X only: tensors must live on a cuda (HIP) device")
    m3, cp, vc = _f32c(means3D.detach(), dev), _f32c(camposes.detach(), dev).reshape(-1, 3), _f32c(view_colors, dev)
    P, V = m3.shape[0], cp.shape[0]
    if tuple(vc.shape) != (V, P, 3):
        raise ValueError(f"view_colors must be [V={V}, P={P}, 3], got {tuple(vc.shape)}")
    out = torch.empty(P, M, 3, dtype=torch.float32, device=dev)
    with _on_device(dev):
        L.check(L.load().hs_sh_backward_views(P, M, sh_degree, V, _ptr(m3), _ptr(cp), _ptr(vc), _ptr(out), _stream()),
                "hs_sh_backward_views")
    return out


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, capacity=None, return_alpha=False, deferred=None, return_invdepth=False,
                        densify=None, aux=None):
    rs = raster_settings
    multi = rs.viewmatrices is not None
    # the camera tensors travel as autograd inputs so a trajectory model upstream receives pose gradients
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, rs.exposure, rs.crf_table,
                                     rs.viewmatrices if multi else rs.viewmatrix,
                                     rs.projmatrices if multi else rs.projmatrix,
                                     rs.camposes if multi else rs.campos, rs, capacity, return_alpha, deferred,
                                     return_invdepth, densify, aux)


class GaussianRasterizer(nn.Module):
    """Drop-in for diff_gaussian_rasterization.GaussianRasterizer.

    `capacity` (extension): None = upstream behaviour (one host read of num_rendered per forward); an int =
    sync-free mode with a fixed binning capacity in (tile, Gaussian) pairs: the forward is a single enqueue and the
    device itself turns a frame that does not fit into an empty (background) render plus an overflow flag.  Who
    looks at the flag, and when (SURVEY.md 8b "grows capacity and replays on overflow"):
      * a forward that cannot be followed by a backward (torch.no_grad(), or no input requires grad) checks it before
        returning -- one event wait, evaluation can afford it -- and on overflow grows `self.capacity` to 1.5 x the
        pair count the device reported and REPLAYS the forward, so the caller never sees the empty frame;
      * a training forward returns without waiting; the backward checks the flag after enqueueing its kernels.  By then
        the caller's loss was computed from the empty frame, so the step is lost: BinningOverflow is raised, and
        `self.capacity` has been grown so that repeating the step succeeds;
      * `last_num_rendered` / `check_overflow()` wait for the latest forward's counters on demand (e.g. right after
        the forward of a step whose loss is expensive).
    `keep_state=True` keeps the state buffers of the latest forward alive so `inspect_state(rasterizer)` also works
    for forwards that recorded no autograd graph.
    """

    def __init__(self, raster_settings: GaussianRasterizationSettings, capacity: Optional[int] = None,
                 return_alpha: bool = False, defer_sh_grad: bool = False, return_invdepth: bool = False,
                 densify_stats: Optional[DensifyStats] = None, gather_group=None, keep_state: bool = False,
                 reduce_group=None, reduce_chunks: int = 4):
        super().__init__()
        # view-parallel training with the plain exchange (every rank renders its own view; the per-Gaussian gradients
        # are summed over the ranks): a torch.distributed process group (or True for the default group) makes the
        # backward run its per-Gaussian half in `reduce_chunks` ascending chunks and start the all-reduce of each
        # chunk's gradient rows while the next chunk computes; call finish_reduce() before reading any gradient.
        # The collectives outlive backward() only when every differentiable input is a leaf without a .grad (autograd
        # then just stores what it is given); with activations between the parameters and the rasterizer, or gradient
        # accumulation, backward() itself waits for them before autograd touches the rows (finish_reduce() returns 0)
        self.reduce_group = reduce_group
        self.reduce_chunks = int(reduce_chunks)
        # with defer_sh_grad: a torch.distributed process group (or True for the default group) makes the backward
        # start the all-gather of the view colour gradients itself, overlapped with its per-Gaussian half
        self.gather_group = gather_group
        self.gather_direct = False  # True: that all-gather as 1-hop point-to-point sends (distributed._gather_rows)
        self.densify_stats = densify_stats  # extension: updated in place by every backward (see DensifyStats)
        # extension (newer published rasterizers return (color, radii, invdepths)): append the expected inverse
        # depth image [H,W] = sum_i alpha_i T_i / z_i to the outputs, differentiable
        self.return_invdepth = return_invdepth
        self.raster_settings = raster_settings
        self.capacity = capacity
        self.return_alpha = return_alpha  # extension: append the accumulated-opacity image [H,W] to the outputs
        # extension for view-parallel training: backward leaves `shs.grad` unset and instead fills `self.deferred`
        # with this view's per-Gaussian colour gradients; distributed.exchange_view_gradients all-gathers those
        # (12 B per Gaussian and view instead of 12*M) and forms the summed SH gradient locally
        self.defer_sh_grad = defer_sh_grad
        self.deferred: Optional[dict] = None
        self.keep_state = keep_state
        self._last: dict = {}       # bookkeeping of the latest forward (pending counters, optional state)
        self._cell: dict = {}       # shared with every call's backward: an overflow found there asks for a larger capacity
        self.overflow_replays = 0   # forwards that were replayed with a grown capacity

    @property
    def last_num_rendered(self) -> int:
        """Number of (tile, Gaussian) pairs of the latest forward.  In sync-free mode this waits for that forward's
        counters (one event) and raises BinningOverflow if it did not fit."""
        return self.check_overflow()

    def check_overflow(self) -> int:
        """Wait for the latest forward's device counters; returns num_rendered, raises BinningOverflow (after growing
        `self.capacity`) if the frame did not fit its binning capacity."""
        pend = self._last.get("pending")
        if pend is None:
            return int(self._last.get("num_rendered", 0))
        n, over = pend.resolve()
        if over:
            self.capacity = max(int(self.capacity or 0), grown_capacity(n))
            raise BinningOverflow(n, pend.capacity)
        return n

    def finish_reduce(self) -> int:
        """Wait for the chunked all-reduce the latest backward started (reduce_group=...); returns the number of
        collectives waited for (0: nothing was pending, e.g. a single-rank run)."""
        from .distributed import finish_pending
        return finish_pending(self._cell.pop("reduce_pending", None))

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        lib = L.load()
        with torch.no_grad():
            pos = _f32c(positions, positions.device)
            view = _f32c(self.raster_settings.viewmatrix, positions.device)
            vis = torch.empty(pos.shape[0], dtype=torch.uint8, device=positions.device)
            L.check(lib.hs_mark_visible(pos.shape[0], pos.data_ptr(), view.data_ptr(), vis.data_ptr(), _stream()),
                    "hs_mark_visible")
        return vis.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        given = [t for t in (means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                             rs.exposure, rs.crf_table, rs.viewmatrix, rs.projmatrix, rs.campos, rs.viewmatrices,
                             rs.projmatrices, rs.camposes) if isinstance(t, torch.Tensor)]
        backward_possible = torch.is_grad_enabled() and any(t.requires_grad for t in given)
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        grow = self._cell.pop("grow_to", None)  # an overflow found by the backward of ANY earlier call of this rasterizer
        if grow is not None and self.capacity is not None:
            self.capacity = max(int(self.capacity), int(grow))
        while True:
            self.deferred = {} if self.defer_sh_grad else None
            if self.deferred is not None and self.gather_group is not None:
                self.deferred["gather_group"] = self.gather_group
                self.deferred["gather_direct"] = bool(self.gather_direct)
            aux = {"keep_state": self.keep_state, "cell": self._cell}
            if self.reduce_group is not None and not self.defer_sh_grad:
                aux["reduce_group"], aux["reduce_chunks"] = self.reduce_group, self.reduce_chunks
            outs = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                       cov3D_precomp, rs, self.capacity, self.return_alpha, self.deferred,
                                       self.return_invdepth, self.densify_stats, aux)
            self._last = aux
            pend = aux.get("pending")
            if "captured" in self._cell and torch.cuda.is_current_stream_capturing():
                self._cell["captured"].append(pend)   # graphs.GraphedStep checks EVERY forward of the captured step
            if pend is None or backward_possible:
                return outs  # synchronous mode, or a training step (its backward looks at the counters)
            try:
                n, over = pend.resolve()
            except SortChainStalled:
                self.overflow_replays += 1   # ticket-ordered passes are on now: render the frame again
                continue
            if not over:
                return outs
            # nobody else would ever look: grow and render the frame again, transparently
            self.capacity = max(int(self.capacity), grown_capacity(n))
            self.overflow_replays += 1


def _state_of(obj) -> "_State":
    """The forward state behind an output tensor (through its autograd node) or behind a
    GaussianRasterizer(..., keep_state=True)."""
    if isinstance(obj, GaussianRasterizer):
        st = obj._last.get("state")
        if st is None:
            raise RuntimeError("inspect_state(rasterizer): create it with keep_state=True and run a forward first")
        return st
    fn = getattr(obj, "grad_fn", None)
    if fn is None or not hasattr(fn, "st"):
        raise RuntimeError("inspect_state needs a tensor returned by GaussianRasterizer.forward with grad enabled; "
                           "for torch.no_grad() forwards pass a GaussianRasterizer(..., keep_state=True) instead")
    return fn.st


def inspect_state(out_tensor) -> dict:
    """Test/profiling helper: the intermediates of the forward that produced `out_tensor` (a tensor returned by
    GaussianRasterizer.forward with grad enabled, or a GaussianRasterizer created with keep_state=True), as views
    into the state workspaces."""
    st: _State = _state_of(out_tensor)
    lay, d = st.layout, st.dims
    I = d.P * d.n_poses
    gx, gy = (d.W + L.HS_TILE - 1) // L.HS_TILE, (d.H + L.HS_TILE - 1) // L.HS_TILE
    vt = gx * gy * d.n_poses
    R = st.num_rendered if st.num_rendered >= 0 else st.pending.check()
    R = min(R, int(d.capacity))

    def view(buf, off, count, dtype):
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return buf[off:off + nbytes].view(dtype)

    if I > 0:  # instance-order offsets (a5 as published) are not part of the forward: fill them now
        a = st.fwd_args
        a.stages = L.HS_STAGE_OFFSETS
        L.check(L.load().hs_forward(C.byref(a), _stream()), "hs_forward[offsets]")
    rec = view(st.geom, lay.rec, I * 16, torch.float32).reshape(I, 16)
    depths = view(st.geom, lay.depth, I, torch.float32)
    tile_sorted = view(st.binning, lay.keys_sorted, R, torch.int32).to(torch.int64) & 0xFFFFFFFF
    pl = view(st.binning, lay.point_list, R, torch.int32)
    # the published sort key (tile << 32) | depth_bits, rebuilt from the two halves the split sort keeps
    dbits = depths.view(torch.int32).to(torch.int64)[pl.to(torch.int64)] & 0xFFFFFFFF
    keys_sorted = (tile_sorted << 32) | dbits
    # per-pose radiance images the CRF / blur average read (slot N: the pose mean, when N > 1); None when not kept
    n_img = d.n_poses + (1 if d.n_poses > 1 else 0)
    pose_hdr = (view(st.image, lay.pose_hdr, n_img * 3 * d.W * d.H, torch.float32).reshape(n_img, 3, d.H, d.W)
                if ((st.flags & L.HS_FLAG_HDR) or d.n_poses > 1) else None)
    return dict(
        pose_hdr=pose_hdr,
        num_rendered=R,
        look_back_helps=int(view(st.geom, lay.counters, 8, torch.int32)[6].item()),   # hs_counters.reserved[4]
        depth_slow_ranges=int(view(st.geom, lay.counters, 8, torch.int32)[4].item()),  # reserved[2]: ranges of the counting depth sort sorted through memory
        tile_sort=int(view(st.geom, lay.counters, 8, torch.int32)[7].item()),         # reserved[5]: 0 radix, 1 counting, 2 hierarchical
        inst_sorted=view(st.binning, lay.inst_sorted, I, torch.int32), offs_sorted=view(st.binning, lay.offs_sorted, I, torch.int32),
        rec=rec, xy=rec[:, 0:2], conic_opacity=torch.stack([rec[:, 2], rec[:, 3], rec[:, 4], rec[:, 5]], 1),
        rgb=rec[:, 6:9], depths=depths,
        radii=view(st.geom, lay.radii, I, torch.int32), tiles_touched=view(st.geom, lay.tiles_touched, I, torch.int32),
        offsets=view(st.geom, lay.offsets, I, torch.int32), cov3D=view(st.geom, lay.cov3D, d.P * 6, torch.float32).reshape(d.P, 6),
        clamped=view(st.geom, lay.clamped, I, torch.uint8),
        keys_sorted=keys_sorted, point_list=pl,
        ranges=view(st.binning, lay.ranges, vt * 2, torch.int32).reshape(vt, 2),
        final_T=view(st.image, lay.final_T, d.n_poses * d.W * d.H, torch.float32).reshape(d.n_poses, d.H, d.W),
        n_contrib=view(st.image, lay.n_contrib, d.n_poses * d.W * d.H, torch.int32).reshape(d.n_poses, d.H, d.W),
    )
