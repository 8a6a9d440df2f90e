import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HS_LIB_PATH"] = os.path.join(ROOT, "casualhdrsplat_amd", "libhdrsplat_stats.so")
sys.argv = [sys.argv[0], "--steps", "1"] + sys.argv[1:]
exec(open(os.path.join(ROOT, "scripts", "step_c3.py")).read())
from casualhdrsplat_amd import _lib
lib = _lib.load()
out = (C.c_ulonglong * 8)()
torch.cuda.synchronize()
assert lib.hs_debug_stats(out, 1) == 0
nrun = 2  # step_c3 ran the step twice (warm + 1 timed)
t, z, act, culled = out[0] / nrun, out[1] / nrun, out[2] / nrun, out[3] / nrun
print(f"bwd trips={t:.0f} zero-active={z:.0f} ({100*z/max(t,1):.1f}%) mean active lanes/trip={act/max(t,1):.2f} (non-zero trips: {act/max(t-z,1):.2f}) culled={culled:.0f} cull frac={culled/max(culled+t,1):.3f}")
