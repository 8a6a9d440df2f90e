"""CPU-side checks of the C ABI: libhdrsplat.so builds, loads, and exports exactly what
include/hdrsplat.h declares; hs_plan (pure host code) carves sane, aligned, non-overlapping
workspaces; argument validation rejects bad calls before any HIP call is made."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "casualhdrsplat_amd", "csrc"), "-j4"])
    from casualhdrsplat_amd import _lib
    return _lib


def header_functions():
    txt = open(os.path.join(ROOT, "include", "hdrsplat.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hs_[a-z_]+)\s*\(", txt)))


def test_header_symbols_exported(lib):
    names = header_functions()
    assert set(names) == set(lib.EXPORTS)
    dll = C.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), n
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    for n in names:
        assert re.search(rf"\bT {n}\b", out), n
    # ... and nothing else: the library is built with -fvisibility=hidden (no mangled hs::launch_* internals); the only
    # other dynamic symbols are the per-translation-unit ids hipcc emits (__hip_cuid_*)
    others = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    others = [n for n in others if n not in names and not n.startswith("__hip_cuid_")]
    assert others == [], others


def test_fault_injection_exists_only_in_the_test_library(lib):
    """HS_FAULT_INJECT (a stalled chain, a late block, a missing ticket: tests/test_gpu_parity.py) is compiled into
    libhdrsplat_test.so alone (-DHS_TESTING); the product library does not read that variable -- no environment can plant a
    fault in it -- and exports the same C ABI."""
    prod = open(lib.LIB_PATH, "rb").read()
    assert b"HS_FAULT_INJECT" not in prod and b"late_block" not in prod and b"stalled_chain" not in prod
    test_path = os.path.join(os.path.dirname(lib.LIB_PATH), "libhdrsplat_test.so")
    test = open(test_path, "rb").read()
    assert b"HS_FAULT_INJECT" in test and b"late_block" in test
    out = subprocess.check_output(["nm", "-D", "--defined-only", test_path]).decode()
    for n in header_functions():
        assert re.search(rf"\bT {n}\b", out), n


def test_struct_sizes_match_c(lib, tmp_path):
    """ctypes mirrors must have the layout the C compiler gives include/hdrsplat.h."""
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "hdrsplat.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(hs_dims),sizeof(hs_sizes),sizeof(hs_counters),sizeof(hs_fwd_args),sizeof(hs_bwd_args),"
                   "sizeof(hs_layout));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(t) for t in (lib.hs_dims, lib.hs_sizes, lib.hs_counters, lib.hs_fwd_args, lib.hs_bwd_args,
                                  lib.hs_layout)]
    assert got == want


def test_version_and_plan(lib):
    L = lib.load()
    assert L.hs_version() == 308
    d, sz, lay = lib.plan(1_000_000, 16, 3, 1920, 1080, 1, 7_000_000)
    assert sz.geom_bytes > 1_000_000 * 48 and sz.binning_bytes > 7_000_000 * 16
    assert sz.image_bytes >= 1920 * 1080 * (8 + 12) and sz.bwd_bytes >= 7_000_000 * 48
    geom = [lay.counters, lay.rec, lay.depth, lay.radii, lay.tiles_touched, lay.offsets, lay.cov3D, lay.clamped,
            lay.scan_spine]
    assert geom == sorted(geom) and all(o % 256 == 0 for o in geom) and len(set(geom)) == len(geom)
    binning = [lay.keys_sorted, lay.point_list, lay.pairs_tmp, lay.ranges, lay.sort_tmp, lay.depth_pairs, lay.inst_sorted,
               lay.offs_sorted, lay.pair_sort_tmp]
    assert binning == sorted(binning) and all(o % 256 == 0 for o in binning)
    # N poses scale the per-instance arrays
    _, sz8, _ = lib.plan(1_000_000, 16, 3, 1920, 1080, 8, 7_000_000)
    assert sz8.geom_bytes > 7 * sz.geom_bytes * 0.8 and sz8.image_bytes > 8 * 1920 * 1080 * 20


def test_plan_carries_the_counting_sorts_matrices_for_small_frames_only(lib):
    """hs_layout.tile_matrix (HS_VERSION 305): frames of <= 4096 (pose, tile) keys and <= 2^21 (emission workgroup, key)
    entries get the two u32 matrices + the totals row of the counting tile sort at the end of the binning workspace; larger
    frames -- BASELINE c3, c4 -- get an empty region."""
    def tail(P, W, H, N, cap):
        _, sz, lay = lib.plan(P, 1, 0, W, H, N, cap)
        assert lay.tile_matrix % 256 == 0 and lay.tile_matrix >= lay.pair_act
        return lay.hier_ws - lay.tile_matrix      # (hier_ws, HS_VERSION 307, is carved behind it)
    rows, keys = (100_000 + 255) // 256, 50 * 50
    assert (2 * rows * keys + keys) * 4 <= tail(100_000, 800, 800, 1, 900_000) < (2 * rows * keys + keys) * 4 + 256   # c2
    assert tail(1_000_000, 1920, 1080, 1, 7_000_000) == 0                                                           # c3: 8160 tiles
    assert tail(600_000, 800, 800, 1, 4_000_000) == 0            # 2344 workgroups x 2500 keys > 2^21 entries
    assert tail(10_000, 1024, 1024, 1, 100_000) > 0 and tail(10_000, 1040, 1024, 1, 100_000) == 0   # 4096 / 4160 tiles
    assert tail(700, 256, 256, 16, 10_000) > 0 and tail(700, 256, 256, 17, 10_000) == 0             # 16 / 17 poses x 256 tiles
    assert tail(10_000, 800, 800, 1, 0) == 0                    # no binning capacity: nothing to sort


def test_plan_carries_the_hierarchical_sorts_workspace_up_to_2048_super_tile_keys(lib):
    """hs_layout.hier_ws (HS_VERSION 307): frames of <= 2048 (pose, 8 x 8-tile super-tile) keys -- BASELINE c3 (135) and
    c4 (1080) -- carry the scratch of the hierarchical tile sort at the end of the binning workspace."""
    def tail(P, W, H, N, cap):
        _, sz, lay = lib.plan(P, 1, 0, W, H, N, cap)
        assert lay.hier_ws % 256 == 0 and lay.hier_ws >= lay.tile_matrix
        return sz.binning_bytes - lay.hier_ws
    chunks = (7_000_000 + 1023) // 1024 + 135
    assert tail(1_000_000, 1920, 1080, 1, 7_000_000) >= chunks * (16 + 512) + 135 * 64 * 8      # descriptors + count rows + tiles
    assert tail(1_000_000, 1920, 1080, 8, 60_000_000) > 0 and tail(1_000_000, 1920, 1080, 16, 60_000_000) == 0   # 1080 / 2160 keys
    assert tail(10_000, 800, 800, 1, 0) == 0


def test_plan_carries_the_counting_depth_sorts_matrix_below_2_21_instances(lib):
    """hs_layout.depth_ws (HS_VERSION 308): frames of fewer than 2^21 instances carry the counting depth sort's scratch
    behind sort_tmp -- per block of 1024 (up to 2^18 instances) or 4096 instances a row of 4096 u16 counts, a row of 4096
    u32 prefixes and a word of culled instances, plus the 4096 bucket totals; larger frames (BASELINE c4) none."""
    def size(P, N):
        _, sz, lay = lib.plan(P, 1, 0, 1920, 1080, N, 1_000_000)
        assert lay.depth_ws % 256 == 0 and lay.depth_ws >= lay.sort_tmp
        return lay.depth_pairs - lay.depth_ws     # (depth_pairs is carved behind it)
    row = (4096 // 2 + 4096 + 1) * 4
    for P, N, rows in ((100_000, 1, 98), (1 << 18, 1, 256), ((1 << 18) + 1, 1, 65), (1_000_000, 1, 245), ((2 << 20) - 1, 1, 512),
                       (200_000, 4, 196)):
        want = rows * row + 4096 * 4
        assert want <= size(P, N) < want + 256, (P, N)
    assert size(2 << 20, 1) == 0 and size(1_000_000, 8) == 0


def test_depth_sort_switch_is_process_wide_and_queryable(lib):
    """hs_depth_sort: which depth sort frames below 2^21 instances get can be read and changed at run time (the Python host
    goes back to the look-back passes when a frame reports ranges that did not fit on chip); no GPU involved."""
    L = lib.load()
    before = L.hs_depth_sort(-1)
    assert before == 1          # by counting: the default
    assert L.hs_depth_sort(0) == 0 and L.hs_depth_sort(-1) == 0
    assert L.hs_depth_sort(1) == 1 and L.hs_depth_sort(-1) == 1
    L.hs_depth_sort(before)


def test_plan_rejects_bad_dims(lib):
    L = lib.load()
    d = lib.hs_dims(-1, 0, 0, 16, 16, 1, 0)
    sz = lib.hs_sizes()
    assert L.hs_plan(C.byref(d), C.byref(sz), None) == lib.HS_EINVAL
    assert b"bad dims" in L.hs_last_error()
    d = lib.hs_dims(10, 0, 0, 16, 16, 0, 0)
    assert L.hs_plan(C.byref(d), C.byref(sz), None) == lib.HS_EINVAL
    assert L.hs_plan(None, None, None) == lib.HS_EINVAL
    # a status word of the radix passes carries a 30-bit count: sorts of 2^30 or more elements are refused
    d = lib.hs_dims(1000, 0, 0, 64, 64, 1, 1 << 30)
    assert L.hs_plan(C.byref(d), C.byref(sz), None) == lib.HS_EINVAL and b"2^30" in L.hs_last_error()
    d = lib.hs_dims(1000, 0, 0, 64, 64, 1, (1 << 30) - 1)
    assert L.hs_plan(C.byref(d), C.byref(sz), None) == lib.HS_OK
    # the pair emission turns a slot into (row, column) of its rectangle with arithmetic that is exact below 2^22 tiles
    d = lib.hs_dims(1000, 0, 0, 32768, 32768, 1, 1000)
    assert L.hs_plan(C.byref(d), C.byref(sz), None) == lib.HS_EINVAL and b"2^22" in L.hs_last_error()
    d = lib.hs_dims(1000, 0, 0, 32768, 32752, 1, 1000)
    assert L.hs_plan(C.byref(d), C.byref(sz), None) == lib.HS_OK


def test_sort_ticket_switch_is_process_wide_and_queryable(lib):
    """hs_sort_tickets: the chain-position mode of the radix passes can be read and changed at run time (the Python host
    turns tickets on after a stalled chain); no GPU involved."""
    L = lib.load()
    before = L.hs_sort_tickets(-1)
    assert before in (0, 1)
    assert L.hs_sort_tickets(1) == 1 and L.hs_sort_tickets(-1) == 1
    assert L.hs_sort_tickets(0) == 0 and L.hs_sort_tickets(-1) == 0
    L.hs_sort_tickets(before)


def test_forward_backward_validate_before_touching_the_gpu(lib):
    L = lib.load()
    assert L.hs_forward(None, None) == lib.HS_EINVAL
    assert L.hs_backward(None, None) == lib.HS_EINVAL
    a = lib.hs_fwd_args()
    a.dims = lib.hs_dims(10, 1, 0, 32, 32, 1, 100)
    assert L.hs_forward(C.byref(a), None) == lib.HS_EINVAL  # null pointers
    assert b"null" in L.hs_last_error()
    # both shs and colors_precomp present -> rejected (exactly-one-of rule of the reference API)
    for f in ("means3D", "viewmatrices", "projmatrices", "camposes", "bg", "shs", "colors_precomp", "scales",
              "rotations"):
        setattr(a, f, 4096)
    assert L.hs_forward(C.byref(a), None) == lib.HS_EINVAL
    assert b"exactly one" in L.hs_last_error()
    a.colors_precomp = None
    a.dims.sh_degree = 3  # needs 16 coefficients, M = 1
    assert L.hs_forward(C.byref(a), None) == lib.HS_EINVAL
    assert b"sh_degree" in L.hs_last_error()
    assert L.hs_mark_visible(-1, None, None, None, None) == lib.HS_EINVAL
    assert L.hs_sort_pairs(None, None, None, None, 5, 40, None, None) == lib.HS_EINVAL


def test_sh_backward_views_validates_before_touching_the_gpu(lib):
    """hs_sh_backward_views (the local half of the view-parallel exchange): argument errors come back as HS_EINVAL."""
    L = lib.load()
    f = L.hs_sh_backward_views
    one = 256  # non-null dummy pointers: validation must fail before any of them is dereferenced
    assert f(10, 16, 3, 0, one, one, one, one, None) == lib.HS_EINVAL          # V < 1
    assert f(10, 4, 3, 2, one, one, one, one, None) == lib.HS_EINVAL           # M too small for degree 3
    assert f(10, 16, 4, 2, one, one, one, one, None) == lib.HS_EINVAL          # degree out of range
    assert f(10, 16, 3, 2, None, one, one, one, None) == lib.HS_EINVAL         # null means3D
    assert f(-1, 16, 3, 2, one, one, one, one, None) == lib.HS_EINVAL
    assert b"hs_sh_backward_views" in L.hs_last_error()
    assert f(0, 16, 3, 2, None, None, None, None, None) == lib.HS_OK           # nothing to do


def test_missing_library_is_a_hard_error(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libhdrsplat.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load()


def test_host_code_is_clean_under_address_and_ub_sanitizers():
    """the host side of the library (hs_plan's carving arithmetic, every entry point's argument
    validation, the thread-local error text, two threads at once) built with -fsanitize=address,undefined and driven
    by tests/native/asan_host.cpp -- no GPU involved (sanitizers for device code are not available on this pool)."""
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "casualhdrsplat_amd", "csrc"), "asan"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "asan_host: clean" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
