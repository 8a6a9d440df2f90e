"""The N>1 path on CPU: gloo, world size 2 (and 3 / 4 where the number of peers matters).  Each rank holds the same Gaussians, renders its own view with
the CPU oracle standing in for the MI355X kernels (the collective code is device-agnostic), packs the
gradients the way rasterizer._launch_backward lays them out and all-reduces them with
casualhdrsplat_amd.distributed; the result must equal the sum of the single-view gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _view_grads(rank, world):
    import helpers as Hh
    from casualhdrsplat_amd import synthetic as S
    from oracle import c_oracle as O
    sc = S.make_scene(400, 96, 64, 1, seed=2)
    cam = S.yaw_camera(96, 64, -5.0 + 10.0 * rank / max(world - 1, 1))
    _, b = Hh.run_oracle(O, sc, cam=cam)
    return [b["dL_dmeans3D"], b["dL_dmeans2D"], b["dL_dopacity"].reshape(-1, 1), b["dL_dshs"], b["dL_dscales"], b["dL_drots"]]


def _view_deferred(rank, world):
    """What GaussianRasterizer(..., defer_sh_grad=True) leaves after backward, built from the CPU oracle: the non-SH
    gradients plus this view's colour gradient after the SH clamp mask."""
    import helpers as Hh
    from casualhdrsplat_amd import synthetic as S
    from oracle import c_oracle as O
    sc = S.make_scene(400, 96, 64, 2, seed=2)
    cam = S.yaw_camera(96, 64, -5.0 + 10.0 * rank / max(world - 1, 1))
    f, b = Hh.run_oracle(O, sc, cam=cam)
    vc = b["dL_dcolor"] * (1 - f["clamped"].astype(np.float32))
    vc[f["radii"] <= 0] = 0
    rest = [b["dL_dmeans3D"], b["dL_dmeans2D"], b["dL_dopacity"].reshape(-1, 1), b["dL_dscales"], b["dL_drots"]]
    return sc, cam, vc, rest, b["dL_dshs"]


def _worker_views(rank, world, port, q, early_gather=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from casualhdrsplat_amd.distributed import exchange_view_gradients, init_from_env, start_view_gather
    from oracle import torch_rasterizer as TR
    init_from_env("gloo")
    sc, cam, vc, rest, _ = _view_deferred(rank, world)
    params = []
    for g in rest:
        p = torch.zeros(g.shape, requires_grad=True)
        p.grad = torch.from_numpy(g.copy())
        params.append(p)
    shs = torch.zeros(sc.shs.shape, requires_grad=True)
    deferred = dict(view_colors=torch.from_numpy(vc.copy())[None], camposes=cam.campos.reshape(1, 3).clone(),
                    means3D=sc.means3D, M=sc.shs.shape[1], sh_degree=sc.sh_degree)
    if early_gather:  # what the rasterizer's backward does with gather_group set: the all-gathers are already in flight
        deferred["gather"] = start_view_gather(deferred["view_colors"], deferred["camposes"])
        assert deferred["gather"] is not None
    n = exchange_view_gradients(params, shs, deferred, sh_backward=TR.sh_backward_views)
    assert n["all_gathered"] == world * (vc.size + 3) and n["all_reduced"] >= sum(g.size for g in rest)
    if rank == 0:
        q.put([p.grad.numpy().copy() for p in params] + [shs.grad.numpy().copy()])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("early_gather,world", [(False, 2), (True, 2), (True, 4), (True, 8)])
def test_view_exchange_equals_sum_of_single_view_gradients(oracle, early_gather, world):
    """all-gather of per-view colour gradients + local SH outer product == all-reduce of the SH gradient rows.
    (world 4: what the driver's scaling run meets first -- more ranks than the two-rank box the GPU tests can get)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_views, args=(r, world, port, q, early_gather)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    views = [_view_deferred(r, world) for r in range(world)]
    want = [sum(x) for x in zip(*[v[3] for v in views])] + [sum(v[4] for v in views)]
    for g, w in zip(got, want):
        assert np.allclose(g, w, rtol=1e-5, atol=2e-6 * np.abs(w).max())
    assert np.abs(want[-1][:, 1:]).max() > 0  # higher SH bands are exercised


def _worker(rank, world, port, shared_flat, q, algo="rccl"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from casualhdrsplat_amd.distributed import all_reduce_gradients, init_from_env
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    grads = [torch.from_numpy(g.copy()) for g in _view_grads(rank, world)]
    params = []
    if shared_flat:  # the rasterizer's layout: all gradients are views of one flat fp32 buffer
        flat = torch.zeros((sum((g.numel() + 3) // 4 * 4 for g in grads) + 1023) // 1024 * 1024)
        o = 0
        for g in grads:
            p = torch.zeros(g.shape, requires_grad=True)
            flat[o:o + g.numel()] = g.reshape(-1)
            p.grad = flat[o:o + g.numel()].view(g.shape)
            o += (g.numel() + 3) // 4 * 4
            params.append(p)
    else:
        for g in grads:
            p = torch.zeros(g.shape, requires_grad=True)
            p.grad = g
            params.append(p)
    n = all_reduce_gradients(params, algo=algo)
    assert n >= sum(g.numel() for g in grads)
    if rank == 0:
        q.put([p.grad.numpy().copy() for p in params])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shared_flat,algo,world", [(True, "rccl", 2), (False, "rccl", 2), (True, "direct", 2),
                                                    (True, "direct", 3), (True, "direct", 4), (False, "rccl", 4),
                                                    (True, "rccl", 8), (True, "direct", 8)])
def test_allreduce_equals_sum_of_single_view_gradients(oracle, shared_flat, algo, world):
    """(world 3: the 1-hop form's shards do not divide the buffer evenly -- its leftover path; world 4: several peers;
    world 8: SURVEY.md 8(c) item 10 / BASELINE config c5 -- the sum of EIGHT single-view gradients == the 8-rank exchange)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shared_flat, q, algo)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = [sum(x) for x in zip(*[_view_grads(r, world) for r in range(world)])]
    for g, w in zip(got, want):
        assert np.allclose(g, w, rtol=1e-6 * world, atol=1e-6 * np.abs(w).max())
    assert np.abs(want[0]).max() > 0


def _worker_chunked(rank, world, port, q):
    """The rasterizer's flat gradient buffer (rasterizer._launch_backward: [means3D | opacities | scales | rotations |
    exposure | crf_table | sh | means2D]) filled chunk by chunk the way hs_backward(HS_BWD_PROJECT, g_begin, g_end) fills
    it, each chunk's rows all-reduced as soon as they exist (distributed.chunked_all_reduce), against ONE all-reduce of the
    whole span afterwards (all_reduce_gradients): bit for bit the same sums.  Rows a chunk has not computed yet hold NaN: a
    collective that ran ahead of its chunk would poison the result."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from casualhdrsplat_amd.distributed import (all_reduce_gradients, chunk_bounds, chunked_all_reduce, finish_pending,
                                                init_from_env)
    init_from_env("gloo")
    src = [torch.from_numpy(g.copy()) for g in _view_grads(rank, world)]   # means3D, means2D, opacity, sh, scales, rots
    P, K = src[0].shape[0], 5
    order = [0, 2, 4, 5, 3]                                                 # summed span: means3D, opacity, scales, rots | sh
    tail_src = [torch.full((1,), 10.0 + rank), torch.arange(3 * K, dtype=torch.float32).reshape(3, K) * (rank + 1)]

    def carve(fill):
        sizes = [src[i].numel() for i in order[:-1]] + [t.numel() for t in tail_src] + [src[3].numel(), src[1].numel()]
        flat = torch.full((sum((n + 3) // 4 * 4 for n in sizes),), fill)
        views, o = [], 0
        shapes = [src[i].shape for i in order[:-1]] + [t.shape for t in tail_src] + [src[3].shape, src[1].shape]
        for n, sh in zip(sizes, shapes):
            views.append(flat[o:o + n].view(sh))
            o += (n + 3) // 4 * 4
        return flat, views

    # chunked: rows appear chunk by chunk
    flat_c, v = carve(float("nan"))
    rows_c, tail_c, m2d_c = v[:4] + [v[6]], v[4:6], v[7]
    for t, s_ in zip(tail_c, tail_src):
        t.copy_(s_)
    m2d_c.copy_(src[1])
    calls = []

    def compute(g0, g1):
        calls.append((g0, g1))
        for t, i in zip(rows_c, order):
            t[g0:g1] = src[i][g0:g1]

    pending = chunked_all_reduce(rows_c, P, 3, compute, tail=tail_c)
    assert calls == chunk_bounds(P, 3) and calls[0][0] == 0 and calls[-1][1] == P and all(a % 128 == 0 for a, _ in calls)
    # round 5: ONE coalesced collective per chunk -- the pieces of a chunk (its rows of every gradient
    # tensor; with the first chunk the exposure / CRF-table gradients) travel together -- so a step puts `chunks` collectives
    # on the wire, not chunks x tensors (3 x 5 + 2 = 17 here before)
    from casualhdrsplat_amd import distributed as D
    assert finish_pending(pending) == len(calls) <= 3 + 2
    assert D.LAST_EXCHANGE["collectives"] == len(calls)
    # unchunked: the whole span in one collective
    flat_u, v = carve(0.0)
    views_u = v
    params = []
    for t, s_ in zip(v[:4] + v[4:6] + [v[6]], [src[i] for i in order[:-1]] + tail_src + [src[3]]):
        t.copy_(s_)
        p = torch.zeros(t.shape, requires_grad=True)
        p.grad = t
        params.append(p)
    v[7].copy_(src[1])
    all_reduce_gradients(params)
    _, views_c = flat_c, rows_c[:4] + tail_c + [rows_c[4], m2d_c]
    same = all(torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))
               for a, b in zip(views_c, views_u))   # (the 16-byte pads between the slices belong to nobody)
    m2d_local = torch.equal(m2d_c, src[1])   # the screen-space gradient of THIS view stays on its rank
    if rank == 0:
        q.put((same, m2d_local, [t.numpy().copy() for t in rows_c]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_chunked_exchange_equals_one_all_reduce_bit_for_bit(oracle, world):
    """the per-Gaussian backward in K ascending chunks, each chunk's gradient rows all-reduced while the
    next computes == the sum of the single-view gradients, bit for bit the unchunked exchange (two ranks: a + b in either
    order; with more ranks the library may add a buffer's elements in an order that depends on where they lie in it, so
    four ranks are held to the sum within rounding)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_chunked, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    same, m2d_local, rows = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert m2d_local and (same or world > 2)
    views = [_view_grads(r, world) for r in range(world)]
    for got, i in zip(rows, [0, 2, 4, 5, 3]):
        want = sum(v[i] for v in views)
        if world == 2:
            assert np.array_equal(got, want)
        else:
            assert np.allclose(got, want, rtol=1e-5, atol=1e-6 * np.abs(want).max())
        assert np.abs(want).max() > 0


def test_chunk_bounds_cover_the_cloud_in_aligned_ascending_pieces():
    from casualhdrsplat_amd.distributed import chunk_bounds
    for P, K in ((1, 4), (127, 4), (128, 4), (129, 2), (1000, 3), (1_000_000, 4), (1_000_000, 1), (5, 0)):
        b = chunk_bounds(P, K)
        assert b[0][0] == 0 and b[-1][1] == P and len(b) <= max(1, K)
        assert all(g0 % 128 == 0 and g1 > g0 for g0, g1 in b) and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert chunk_bounds(0, 4) == []


def test_single_process_is_a_noop():
    from casualhdrsplat_amd.distributed import all_reduce_gradients
    p = torch.zeros(3, requires_grad=True)
    p.grad = torch.ones(3)
    assert all_reduce_gradients([p]) == 0 and torch.equal(p.grad, torch.ones(3))


def _worker_subset(rank, world, port, q):
    """The rasterizer's flat layout [per-Gaussian slices | exposure | crf_table] with an ODD Gaussian count; the
    caller reduces only the per-Gaussian parameters.  Whatever follows them in the buffer must come back untouched
    (the 1-hop form used to round its span up into the next slice)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from casualhdrsplat_amd.distributed import all_reduce_gradients, init_from_env
    init_from_env("gloo")
    P, K = 1001, 7
    shapes = [(P, 3), (P, 3), (P, 1), (P, 4, 3), (P, 3), (P, 4)]          # odd total: 3003+3003+1001+12012+3003+4004
    total = sum((int(np.prod(s)) + 3) // 4 * 4 for s in shapes)
    flat = torch.full((total + 4 + 3 * K + 5,), float("nan"))             # NaN pads: reducing them would poison sums
    params, o = [], 0
    for i, sh in enumerate(shapes):
        n = int(np.prod(sh))
        flat[o:o + n] = float(rank + 1) * (i + 1)
        flat[o + n:o + (n + 3) // 4 * 4] = 0.0 if i + 1 < len(shapes) else float("nan")  # inner pads are written, the tail is not
        p = torch.zeros(sh, requires_grad=True)
        p.grad = flat[o:o + n].view(sh)
        params.append(p)
        o += (n + 3) // 4 * 4
    flat[o:o + 1] = 100.0 + rank                                            # exposure gradient of THIS rank
    flat[o + 4:o + 4 + 3 * K] = 200.0 + rank                                # crf_table gradient of THIS rank
    for algo in ("direct", "rccl"):
        before = flat[o:].clone()
        n = all_reduce_gradients(params, algo=algo)
        assert n < total + 1, (n, total)
        assert torch.equal(torch.nan_to_num(flat[o:], nan=-1.0), torch.nan_to_num(before, nan=-1.0)), algo
    if rank == 0:
        q.put([float(p.grad.reshape(-1)[0]) for p in params] + [float(p.grad.reshape(-1)[-1]) for p in params])
    dist.barrier()
    dist.destroy_process_group()


def test_reducing_a_subset_of_the_flat_buffer_leaves_its_neighbours_alone():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_subset, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # two exchanges in a row (direct, then library): rank sums 1+2 = 3, then 3+3 = 6, times the slice's marker
    assert got == [6.0 * (i + 1) for i in range(6)] * 2, got


def _worker_validate(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import casualhdrsplat_amd.distributed as D
    D.init_from_env("gloo")
    ok = D.validate_direct(torch.device("cpu"))
    # a 1-hop form that returns wrong sums on ONE rank must be rejected on EVERY rank, and autotune must then keep the library
    real = D.all_reduce_direct

    def broken(flat, group=None):
        real(flat, group)
        if rank == 1:
            flat[0] += 1.0
    D.all_reduce_direct = broken
    bad = D.validate_direct(torch.device("cpu"))
    choice = D.autotune_all_reduce(torch.zeros(5000), iters=1)
    D.all_reduce_direct = real
    q.put((rank, ok, bad, choice))
    dist.barrier()
    dist.destroy_process_group()


def test_direct_all_reduce_is_validated_against_the_library_before_use():
    """the 1-hop reduce-scatter + all-gather is only trusted after it reproduced dist.all_reduce on a test
    vector on every rank (bench.py's probe and autotune_all_reduce both ask validate_direct first)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_validate, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == [(0, True, False, "rccl"), (1, True, False, "rccl")], got


def test_init_from_env_refuses_to_guess_a_master_port(monkeypatch):
    """No silent MASTER_PORT=29500: hand-started ranks must name the rendezvous port (torchrun and bench.py's launcher do)."""
    from casualhdrsplat_amd.distributed import init_from_env
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        init_from_env("gloo")
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert init_from_env("gloo") == (0, 1, 0)
