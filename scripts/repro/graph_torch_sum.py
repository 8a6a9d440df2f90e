"""Reproducer (VERDICT r5 next #7 / ADVICE r5): inside a torch.cuda.graph capture on ROCm, (a) does .sum() / .mean() of a
few hundred thousand elements (PyTorch's two-pass reduction clears its semaphore with a memset) return a wrong VALUE from
the second replay on?  (b) does a BLAS call (nn.Linear -> addmm: workspace in the graph's pool) overwrite a neighbouring
tensor of the pool?  torch only -- nothing of this repository is imported.
run on the box as an ordinary child process: python scripts/repro/graph_torch_sum.py"""
import torch
dev = "cuda"
torch.manual_seed(0)


def capture(fn, warm=3):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def case_sum(n):
    x = torch.randn(n, device=dev)
    g, out = capture(lambda: (x * x).mean())
    want, bad = float((x.double() ** 2).mean()), 0
    for rep in range(50):
        g.replay(); torch.cuda.synchronize()
        bad += abs(float(out) - want) > 1e-4 * abs(want)
    print(f"mean over {n} elements, 50 replays: {'WRONG in %d replays (last %.6g, want %.6g)' % (bad, float(out), want) if bad else 'ok'}")
    return bad


def case_blas(k):
    lin = torch.nn.Sequential(torch.nn.Linear(1, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(), torch.nn.Linear(32, 3)).to(dev)
    knots = torch.linspace(-6, 3, k, device=dev)[:, None]
    a = torch.randn(200_000, device=dev)

    def fn():
        left = a * 2.0                       # neighbours of the BLAS call in the graph's pool
        tab = lin(knots)
        right = a + 1.0
        return left, tab, right, (left.sum(), right.sum())
    g, out = capture(fn)
    want = [t.clone() for t in fn()[:3]]
    bad = 0
    for rep in range(50):
        g.replay(); torch.cuda.synchronize()
        bad += sum(int(not torch.allclose(o, w, rtol=1e-5, atol=1e-6)) for o, w in zip(out[:3], want))
    print(f"Linear stack on {k} knots between two pool tensors, 50 replays: {'WRONG tensors: %d' % bad if bad else 'ok'}")
    return bad


bad = sum(case_sum(n) for n in (1000, 50_000, 200_000, 2_000_000)) + sum(case_blas(k) for k in (48, 128, 256))
print("RESULT graph_torch:", "reproduced" if bad else "not reproduced in isolation")
