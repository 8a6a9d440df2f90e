"""Time render fwd/bwd stages alone via replay (development aid)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.argv = [sys.argv[0], "--steps", "1"] + sys.argv[1:]
exec(open(os.path.join(ROOT, "scripts", "step_c3.py")).read().split("torch.cuda.synchronize(); t = time.time()")[0])
from casualhdrsplat_amd import _lib as L
from casualhdrsplat_amd.rasterizer import replay_backward, replay_forward
out = rast(means3D, means2D, opac, shs=shs, scales=scales, rotations=rots)
def tm(fn, n=10):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a_, b_ in e:
        a_.record(); fn(); b_.record()
    torch.cuda.synchronize()
    t = sorted(a_.elapsed_time(b_) for a_, b_ in e)
    return t[len(t)//2]
print(os.environ.get("HS_LIB_PATH", "default").split("/")[-1], "render_bwd(+memset+crf) ms", round(tm(lambda: replay_backward(out[0], dL, L.HS_BWD_RENDER)), 4),
      "render_fwd ms", round(tm(lambda: replay_forward(out[0], L.HS_STAGE_RENDER)), 4),
      "preprocess_bwd ms", round(tm(lambda: replay_backward(out[0], dL, L.HS_BWD_PREPROCESS)), 4),
      "bin ms", round(tm(lambda: replay_forward(out[0], L.HS_STAGE_BIN)), 4),
      "crf ms", round(tm(lambda: replay_backward(out[0], dL, L.HS_BWD_CRF)), 4))
