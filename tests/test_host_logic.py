"""Host-side behaviour of the drop-in boundary that needs no GPU: the settings tuple, argument
validation with the reference API's error messages, loud failure on CPU tensors, synthetic scenes."""
import math
import os

import pytest
import torch

from casualhdrsplat_amd import GaussianRasterizationSettings, GaussianRasterizer, synthetic as S


def settings(W=32, H=32):
    cam = S.make_camera(W, H)
    return GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3), scale_modifier=1.0,
        viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, sh_degree=0, campos=cam.campos, prefiltered=False, debug=False)


def test_settings_field_order_matches_reference_api():
    names = GaussianRasterizationSettings._fields
    assert names[:12] == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                          "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    assert names[12] == "antialiasing"
    rs = settings()
    assert rs.exposure is None and rs.crf_table is None and rs.blur_domain == "ldr" and rs.viewmatrices is None


def test_exactly_one_of_rules():
    r = GaussianRasterizer(settings())
    P = 4
    m, o = torch.zeros(P, 3), torch.ones(P, 1)
    sh, col = torch.zeros(P, 1, 3), torch.zeros(P, 3)
    sc, ro, cov = torch.ones(P, 3), torch.zeros(P, 4), torch.zeros(P, 6)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m, o, scales=sc, rotations=ro)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m, o, shs=sh, colors_precomp=col, scales=sc, rotations=ro)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=sh)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=sh, scales=sc, rotations=ro, cov3D_precomp=cov)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=sh, scales=sc)


def test_cpu_tensors_fail_loudly_no_fallback():
    r = GaussianRasterizer(settings())
    P = 4
    with pytest.raises(RuntimeError, match="MI355X"):
        r(torch.zeros(P, 3), torch.zeros(P, 3), torch.ones(P, 1), shs=torch.zeros(P, 1, 3), scales=torch.ones(P, 3),
          rotations=torch.zeros(P, 4))


def test_product_never_imports_the_oracle():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "casualhdrsplat_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "libhs_oracle" not in txt and "c_oracle" not in txt and "torch_rasterizer" not in txt, f


def test_synthetic_scene_distribution_and_determinism():
    a = S.make_scene(5000, 320, 180, 3, seed=0, hdr=True)
    b = S.make_scene(5000, 320, 180, 3, seed=0, hdr=True)
    c = S.make_scene(5000, 320, 180, 3, seed=1, hdr=True)
    assert torch.equal(a.means3D, b.means3D) and torch.equal(a.shs, b.shs) and not torch.equal(a.means3D, c.means3D)
    assert a.shs.shape == (5000, 16, 3) and a.crf_table.shape == (3, 256)
    assert torch.allclose(a.rotations.norm(dim=1), torch.ones(5000), atol=1e-5)
    z = a.means3D[:, 2]
    assert z.min() >= 2 and z.max() <= 10
    cam = a.camera
    assert cam.tanfovx == pytest.approx(320 / (2 * 1000 * 320 / 1920))
    # every mean projects inside the image
    fx = 320 / (2 * cam.tanfovx)
    px = fx * a.means3D[:, 0] / z + (320 - 1) / 2
    py = fx * a.means3D[:, 1] / z + (180 - 1) / 2
    assert px.min() > -1 and px.max() < 320 and py.min() > -1 and py.max() < 180
    sig = a.scales.mean(dim=1) * fx / z
    assert 0.3 < sig.min() and sig.max() < 14
    assert torch.all(a.crf_table[:, 1:] > a.crf_table[:, :-1])  # monotone response
    poses = S.blur_poses(320, 180, 8)
    assert len(poses) == 8 and poses[3].campos[0] == pytest.approx(0.03)


def test_yaw_camera_keeps_cloud_centre_fixed():
    cam = S.yaw_camera(640, 360, 5.0, centre_depth=6.0)
    w2c = cam.viewmatrix.t().double()
    c = torch.tensor([0.0, 0.0, 6.0, 1.0], dtype=torch.float64)
    assert torch.allclose((w2c @ c)[:3], c[:3], atol=1e-6)
    R = w2c[:3, :3]
    assert torch.allclose(R @ R.t(), torch.eye(3, dtype=torch.float64), atol=1e-6)
    assert math.degrees(math.acos(float(R[0, 0]))) == pytest.approx(5.0, abs=1e-4)


def test_bench_launch_command_shape():
    """`python bench.py --gpus N` without a torchrun environment re-launches itself as N ranks."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd = bench.launch_command(["--gpus", "8", "--steps", "5"], 8, 12345)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    assert cmd[-5:] == [os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "5"]
    p1, p2 = bench.free_port(), bench.free_port()
    assert 1024 < p1 < 65536 and 1024 < p2 < 65536


def test_bench_self_launches_ranks_and_forwards_their_exit_code():
    """End to end on a box without GPUs: the parent starts two child ranks through torch.distributed.run, every rank
    stops with the loud no-GPU message (there is no CPU fallback), and the parent exits non-zero instead of hanging
    or printing a fake line."""
    import os
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert "bench.py needs an MI355X" in r.stderr
    assert not r.stdout.strip().startswith("{")


def test_bench_offline_counters_are_tied_to_the_kernel_source(tmp_path, monkeypatch):
    """`roofline.traffic` may only carry the committed PMC figure while render.hip is the file the counters were
    collected on (hash recorded by scripts/fold_profiles.py); the committed profile must be current."""
    import os
    import bench
    off = bench.offline_profile("c3")
    assert off is not None and off["render_bwd_kernel_hbm_bytes"] > 0
    assert off["same_kernel_source"], "render.hip changed: re-run scripts/collect_profiles.sh pmc + fold_profiles.py"
    isa = bench.isa_counts()
    assert isa is not None and not isa.get("stale"), "render.hip changed: re-run scripts/isa_loop_counts.py"
    # a different source -> the figure is not used
    root = tmp_path / "r"
    (root / "profiles").mkdir(parents=True)
    (root / "casualhdrsplat_amd" / "csrc").mkdir(parents=True)
    (root / "casualhdrsplat_amd" / "csrc" / "render.hip").write_text("// something else\n")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    (root / "profiles" / "pmc_traffic.json").write_text(open(src).read())
    monkeypatch.setattr(bench, "ROOT", str(root))
    assert bench.offline_profile("c3")["same_kernel_source"] is False


def test_isa_script_checks_the_wait_states_in_front_of_dpp_reads():
    """scripts/isa_loop_counts.py asserts, on the ISA the compiler emitted for render.hip, the wait states the hand-written
    DPP blocks assume (VALU write -> DPP read of the same VGPR as src0: two).  The checker itself: a planted hazard is
    found, the padded form passes, and the committed counts were made with zero violations."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("isa_loop_counts", os.path.join(root, "scripts", "isa_loop_counts.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = {"f": ["\tv_mul_f32_e32 v3, v1, v2", "\ts_nop 0",
                 "\tv_add_f32_dpp v5, v3, v3 row_shr:8 row_mask:0xf bank_mask:0xc"]}
    n, v = mod.check_dpp_hazards(bad)
    assert n == 1 and len(v) == 1 and "v_mul_f32_e32" in v[0]
    ok = {"f": ["\tv_mul_f32_e32 v3, v1, v2", "\ts_nop 1",
                "\tv_add_f32_dpp v5, v3, v3 row_shr:8 row_mask:0xf bank_mask:0xc",
                "\tv_mov_b32_e32 v9, v3", "\tv_mov_b32_e32 v8, v3",
                "\tv_add_f32_dpp v6, v5, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"]}
    assert mod.check_dpp_hazards(ok) == (2, [])
    rng = {"f": ["\tv_pk_mul_f32 v[2:3], v[0:1], v[0:1]", "\tv_add_f32_dpp v5, v3, v3 row_shr:4 row_mask:0xf bank_mask:0xa"]}
    assert len(mod.check_dpp_hazards(rng)[1]) == 1
    d = json.load(open(os.path.join(root, "profiles", "isa_loop_counts.json")))
    assert d["dpp_hazard_check"]["violations"] == 0 and d["dpp_hazard_check"]["dpp_instructions_checked"] >= 50


def test_bench_exchange_bytes_and_probe_order():
    """config c5 bookkeeping: with SH colours the probe tries the view forms first (5x fewer bytes, the only ones whose model
    reaches 6x), the plain library all-reduce LAST -- it is the fallback and is never dropped for being slow; the 1-hop
    all-to-all forms join only on request.  The bytes a rank sends per step follow from the tensor sizes (SURVEY.md 2.4:
    P x 59 fp32 = 236 MB all-reduced at c3; the view form all-reduces P x 11 fp32 and all-gathers 12 B per Gaussian and
    view), and the analytic model prices them on 7 x 153 GB/s links."""
    import bench
    assert bench.FALLBACK == ("allreduce", "rccl") and bench.FALLBACK in bench.EXCHANGES
    o3 = bench.probe_order(3)
    assert o3[0][0].startswith("views") and o3[-1] == bench.FALLBACK and all(a == "rccl" for _, a in o3)
    assert ("allreduce_overlap", "rccl") in o3
    assert bench.probe_order(0)[0][0].startswith("allreduce") and bench.FALLBACK in bench.probe_order(0)
    assert [e for e in bench.probe_order(3, everything=True) if e[1] == "direct"] and len(set(bench.probe_order(3, True))) == len(bench.EXCHANGES)
    m = bench.exchange_model("allreduce", 8, bench.CONFIGS["c3"], base_ms=1.18)
    assert 2.6 < m["all_reduce_ring_ms"] < 2.8 and 0.37 < m["all_reduce_one_hop_ms"] < 0.41   # 2.7 / 0.39 ms
    assert m["expected_scaling_one_hop"] < 6.1 and m["expected_scaling_ring"] < 3.0
    mv = bench.exchange_model("views_overlap", 8, bench.CONFIGS["c3"], base_ms=1.18)
    assert mv["expected_scaling_one_hop"] > 7.0 and mv["exposed_one_hop_ms"] < 0.1
    assert mv["all_reduce_ring_ms"] < 0.25 * m["all_reduce_ring_ms"]
    b = bench.exchange_bytes("allreduce", 8, bench.CONFIGS["c3"])
    assert b["all_reduced_bytes"] == 4 * (1_000_000 * 59 + 1 + 3 * 256) and b["all_gathered_bytes_per_rank"] == 0
    assert b["sent_per_rank_bytes"] == int(2 * 7 / 8 * b["all_reduced_bytes"])
    v = bench.exchange_bytes("views", 8, bench.CONFIGS["c3"])
    assert v["all_reduced_bytes"] == 4 * (1_000_000 * 11 + 1 + 3 * 256) and v["all_gathered_bytes_per_rank"] == 4 * (3_000_000 + 3)
    assert v["sent_per_rank_bytes"] < 0.55 * b["sent_per_rank_bytes"]


def test_no_memset_or_copy_on_the_paths_a_captured_step_takes():
    """The library enqueues KERNELS only on the paths a training step takes -- no memset or copy nodes in a captured step
    (graphs.GraphedStep).  (Round 5 believed memset nodes of a captured HIP graph misbehaved; round 6's reproducers --
    scripts/repro/, profiles/r06_repro_graph_hazards.txt -- show they do not, in isolation or in a graph of 2752 nodes:
    the rule stays as hygiene)  The remaining hipMemsetAsync / hipMemcpyAsync / hipMemsetD32Async calls are
    the ones listed here -- a frame without binning capacity, the stand-alone hs_sort_pairs entry point, the fault-injection
    hook of the test library -- and a new one has to be added to this list on purpose.  (An empty cloud, P == 0, is cleared
    by a kernel since round 6.)"""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "casualhdrsplat_amd", "csrc")
    allowed = {
        "api.hip": ["hipMemsetD32Async((hipDeviceptr_t)n_dev,", "hipMemsetD32Async((hipDeviceptr_t)fail_word,",               # hs_sort_pairs
                    "hipMemcpyAsync(ka, keys_in,", "hipMemcpyAsync(va, vals_in,", "hipMemcpyAsync(keys_out,", "hipMemcpyAsync(vals_out,"],
        "binning.hip": ["if (!zeroed) HS_HIP_CHECK(hipMemsetAsync(tmp, 0,",            # radix_sort_packed outside the pipeline (never: zeroed = true)
                        "HS_HIP_CHECK(hipMemsetAsync(tmp, 0, (size_t)sort_scratch_words(n_launch, passes, TILE) * 4, s));",   # hs_sort_pairs
                        "hipMemsetD32Async((hipDeviceptr_t)sc.tickets, 1, 1, s)",      # fault injection (test library)
                        "hipMemsetD32Async((hipDeviceptr_t)&counters->overflow, 2, 1, s)",   # fault injection (test library)
                        "hipMemcpyAsync(a.counters_host, counters, sizeof(hs_counters), hipMemcpyDeviceToHost, s)"],   # capacity == 0
    }
    for name in sorted(os.listdir(root)):
        if not name.endswith(".hip"):
            continue
        for ln in open(os.path.join(root, name), encoding="utf-8"):
            code = ln.split("//")[0]
            if re.search(r"hipMem(set|cpy)\w*\(", code):
                assert any(a in code for a in allowed.get(name, [])), f"{name}: {ln.strip()}"


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kernel_resources(tu, contract):
    """{demangled-ish kernel name: {field: int}} from hipcc -Rpass-analysis=kernel-resource-usage (device side only)."""
    import re
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "casualhdrsplat_amd", "csrc", tu)
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-fvisibility=hidden", "-std=c++17",
                            f"-ffp-contract={contract}", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src,
                            "-o", os.path.join(tmp, "x.o")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def test_kernel_register_budgets_hold():
    """Resource-usage guard: preprocess_bwd_kernel<3, *, *> needs 174-239 VGPRs -- the pose-gradient
    instantiation is one register-allocation change away from spilling -- and the two render kernels carry ONE spilled
    VGPR each under amdgpu_waves_per_eu(7).  A compiler or source change that adds spills fails here, on the CPU, before
    anybody times it."""
    pre = _kernel_resources("preprocess.hip", "off")
    bwd3 = {k: v for k, v in pre.items() if "preprocess_bwd_kernelILi3E" in k}
    assert len(bwd3) >= 4, sorted(pre)
    for k, v in bwd3.items():
        assert v["VGPRs Spill"] == 0 and v["ScratchSize"] == 0, (k, v)
        assert v["Occupancy"] >= 2, (k, v)
    fwd = [v for k, v in pre.items() if "preprocess_fwd_kernelILi3E" in k]
    assert fwd and all(v["VGPRs Spill"] == 0 and v["Occupancy"] >= 5 for v in fwd), fwd     # 96 VGPRs: five waves per SIMD
    ren = _kernel_resources("render.hip", "fast")
    hot = {k: v for k, v in ren.items() if ("render_fwd_kernelILb0ELb0E" in k or "render_bwd_kernelILb0ELb0E" in k)}
    assert len(hot) == 2, sorted(ren)
    for k, v in hot.items():
        assert v["VGPRs Spill"] <= 1 and v["ScratchSize"] <= 32, (k, v)
    binn = _kernel_resources("binning.hip", "off")
    for k, v in binn.items():
        assert v.get("VGPRs Spill", 0) == 0, (k, v)


def test_writelane_reads_no_sgpr_a_valu_instruction_just_wrote():
    """gfx940+: a VALU instruction reading an SGPR that the VALU instruction before it wrote needs two wait states.  The
    compiler provides them for its own instructions, NOT inside `asm` statements -- and binning.hip's hier_cover issues
    v_writelane_b32 (no builtin in this compiler) right behind the v_cmp that produced its data: without the s_nop 1 in the
    asm text the column masks arrived one ballot late.  Checked on the generated ISA: between a VALU write
    of an SGPR and a v_writelane reading it lie at least two wait states (instructions or s_nop counts)."""
    import re
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "casualhdrsplat_amd", "csrc", "binning.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "binning.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                               "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL, timeout=600)
        lines = [ln.split(";")[0].strip() for ln in open(out)]
    lines = [ln for ln in lines if ln and not ln.startswith(".") and not ln.endswith(":")]
    n_checked = 0
    last_valu_write = {}          # sgpr number -> wait states since a VALU instruction wrote it
    for ln in lines:
        op = ln.split()[0]
        for k in list(last_valu_write):
            last_valu_write[k] += 1 + (int(ln.split()[1]) if op == "s_nop" else 0)
        if op == "v_writelane_b32":
            args = [a.strip() for a in ln[len(op):].split(",")]
            m = re.fullmatch(r"s(\d+)", args[1])
            if m:
                n_checked += 1
                since = last_valu_write.get(int(m.group(1)), 99)
                assert since > 2, f"`{ln}` reads s{m.group(1)} {since - 1} wait state(s) after a VALU instruction wrote it"
        if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            m = re.match(r"\S+\s+s\[(\d+):(\d+)\]", ln) or re.match(r"\S+\s+s(\d+)\b", ln)
            if m:
                lo, hi = int(m.group(1)), int(m.group(m.lastindex))
                for k in range(lo, hi + 1):
                    last_valu_write[k] = 0
        elif op.startswith("s_") and op != "s_nop":
            m = re.match(r"\S+\s+s\[(\d+):(\d+)\]", ln) or re.match(r"\S+\s+s(\d+)\b", ln)
            if m:      # an SALU instruction rewrote the register: no VALU-write hazard any more
                for k in range(int(m.group(1)), int(m.group(m.lastindex)) + 1):
                    last_valu_write.pop(k, None)
    assert n_checked >= 64, n_checked       # 32 per instantiation of hier_cover (count + scatter kernels)
