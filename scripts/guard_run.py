"""HS_GUARD=1 run (child process of tests/test_gpu_parity.py::test_no_kernel_writes_outside_its_buffers): frames of many
shapes -- every tile sort, single / N poses, LDR / HDR both blur domains, pose gradients, deferred SH, invdepth + alpha,
densification statistics, an overflowing capacity, an empty cloud -- forward + backward with guard zones around every
buffer the library writes; prints GUARD-OK or the damaged guards."""
import os, sys
os.environ["HS_GUARD"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as Hh
from casualhdrsplat_amd import synthetic as S, GaussianRasterizer, GaussianRasterizationSettings
from casualhdrsplat_amd.rasterizer import check_guards, BinningOverflow
full = len(sys.argv) > 1 and sys.argv[1] == "full"
bad = []


def run(what, sc, **kw):
    form = kw.pop("form", None)
    if form:
        os.environ["HS_TILE_SORT"] = form
    try:
        Hh.run_hip(sc, **kw)
    except BinningOverflow:
        pass
    finally:
        os.environ.pop("HS_TILE_SORT", None)
    for b in check_guards():
        bad.append((what,) + b)


rng = np.random.default_rng(5)
for form in ("radix", "hier", "count"):
    run(f"ldr 20000 {form}", S.make_scene(20000, 500, 300, 1, seed=3), form=form)
    run(f"ragged 37x23 {form}", S.make_scene(300, 37, 23, 0, seed=2), form=form)
    sc = S.make_scene(3000, 160, 96, 2, seed=6, hdr=True)
    for dom in ("ldr", "hdr"):
        run(f"hdr 8 poses {dom} {form}", sc, cameras=S.blur_poses(160, 96, 8, step=0.02), hdr=True, blur_domain=dom, form=form)
    run(f"wild {form}", Hh.make_wild(S.make_scene(2500, 200, 120, 3, seed=9), rng), form=form)
    run(f"overflow {form}", S.make_scene(5000, 256, 144, 1, seed=8), capacity=3000, form=form)
    base = S.random_camera(160, 96, 5)
    run(f"free camera {form}", S.make_scene(3000, 160, 96, 3, seed=6, hdr=True, place_in=base),
        cameras=S.perturbed_poses(base, 3, seed=1), hdr=True, form=form)
run("one gaussian", S.make_scene(1, 37, 23, 0, seed=2))
run("skewed hier", S.make_scene(30000, 640, 384, 1, seed=31), form="hier")
for case in range(12):
    c = Hh.sweep_case(rng, case)
    run(c["what"], c["sc"], cameras=c["cams"], hdr=c["hdr"], blur_domain=c["dom"], radiance_activation=c["act"],
        use_colors_precomp=c["colors"])
if full:
    sc = S.make_scene(1_000_000, 1920, 1080, 3, seed=0, hdr=True)
    for form in ("radix", "hier"):
        run(f"c3 {form}", sc, hdr=True, form=form)
    run("c4 hier", sc, cameras=S.blur_poses(1920, 1080, 8), hdr=True, form="hier")
print("GUARD-OK" if not bad else "GUARD-DAMAGED", bad, flush=True)
