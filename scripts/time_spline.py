import sys, time, torch
sys.path.insert(0, '/root/repo')
from casualhdrsplat_amd import image_formation as IF
tr = IF.TrajectorySpline(IF.knots_from_lookat(7, radius=0.25), kind="cubic").cuda()
t = (1.0 + 4.0 * torch.rand(20, device="cuda")).requires_grad_(True)
def T(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
def fb():
    tr.delta.grad = None; t.grad = None
    tr.pose_at(t).sum().backward()
print("fused   pose_at fwd      ms", T(lambda: tr.pose_at(t)))
print("fused   pose_at fwd+bwd  ms", T(fb))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    tr.pose_at(t); torch.cuda.synchronize(); ev0.record()
    for _ in range(20): tr.pose_at(t)
    ev1.record(); torch.cuda.synchronize()
print("fused   GPU time per forward call (events) ms", ev0.elapsed_time(ev1) / 20)
tr.fused = False
print("tensor  pose_at fwd      ms", T(lambda: tr.pose_at(t)))
print("tensor  pose_at fwd+bwd  ms", T(fb))
