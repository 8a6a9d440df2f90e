import sys, time, torch
sys.path.insert(0, '/root/repo')
from casualhdrsplat_amd import synthetic as S
from casualhdrsplat_amd.image_formation import HDRBlurFormation, ImplicitCRF, TrajectorySpline, knots_from_lookat
dev='cuda'
P,W,H,frames,virtual,deg=20000,320,208,4,5,1
sc=S.make_scene(P,W,H,deg,seed=0,hdr=True); cam=sc.camera
m=HDRBlurFormation(TrajectorySpline(knots_from_lookat(frames+3,radius=0.25),kind='cubic'),frames,W,H,cam.tanfovx,cam.tanfovy,n_virtual=virtual,crf=ImplicitCRF(K=128),sh_degree=deg,window_from_exposure=True,window_scale=0.6).to(dev)
cloud={k:getattr(sc,k).to(dev).requires_grad_(True) for k in ("means3D","opacities","shs","scales","rotations")}
def T(fn,n=20):
    fn(); torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
print("cameras(i)      ms", T(lambda: m.cameras(1)))
print("crf.table()     ms", T(lambda: m.crf.table()))
def fwd():
    return m(1, *[cloud[k] for k in ("means3D","opacities","shs","scales","rotations")])[0]
print("forward         ms", T(fwd))
def fb():
    l=fwd().abs().mean(); l.backward()
print("forward+backward ms", T(fb))
import torch.autograd.profiler as prof
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as p:
    for _ in range(5): fb()
print(p.key_averages().table(sort_by="self_cpu_time_total", row_limit=18))
