// C ABI of libhdrsplat.so (see include/hdrsplat.h).  Host-side only: argument validation, workspace
// carving and kernel sequencing.  Nothing here allocates device memory or synchronises.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <atomic>

#include "hs_common.h"

namespace hs {

static thread_local char g_err[512] = "";

#ifdef HS_TESTING
// libhdrsplat_test.so only (make test_lib; -DHS_TESTING): the fault injection of tests/test_gpu_parity.py.  The product
// library is built without it: no environment variable can plant a fault in it (hs_common.h: fault_injection() == 0).
int fault_injection() {
    static const int mode = [] {
        const char* e = getenv("HS_FAULT_INJECT");
        return (e && !strcmp(e, "sort_ticket")) ? 1 : (e && !strcmp(e, "stalled_chain")) ? 2 : (e && !strcmp(e, "late_block")) ? 3 : 0;
    }();
    return mode;
}
#endif

// Chain positions of the radix passes: blockIdx (default) or start-order tickets.  Process-wide, switchable at run time
// (hs_sort_tickets): the host turns tickets on when a pass reports a stalled chain (hs_counters.overflow = 2).
static std::atomic<int> g_sort_tickets{-1};
bool sort_tickets() {
    int v = g_sort_tickets.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("HS_SORT_TICKETS");
        v = (e && e[0] == '1') ? 1 : 0;
        int expected = -1;
        if (!g_sort_tickets.compare_exchange_strong(expected, v)) v = expected;
    }
    return v != 0;
}

// Depth sort of small frames: counting pass + range sorts (1) or the look-back passes (0); -1 = not set, the default.
#ifndef HS_TUNE_DEPTH_MSD_DEFAULT
#define HS_TUNE_DEPTH_MSD_DEFAULT 1
#endif
static std::atomic<int> g_depth_sort{-1};
int depth_sort_mode(int64_t I) {
    if (!depth_msd_fits(I)) return kDepthSortLsd;
    const char* e = getenv("HS_DEPTH_SORT");   // (read at every forward: the test suite switches it inside one process)
    if (e && e[0] == 'l') return kDepthSortLsd;
    if (e && e[0] == 'm') return kDepthSortMsd;
    const int v = g_depth_sort.load(std::memory_order_relaxed);
    return v >= 0 ? v : (HS_TUNE_DEPTH_MSD_DEFAULT ? kDepthSortMsd : kDepthSortLsd);
}
int depth_range_cap() {
    const char* e = getenv("HS_DEPTH_RANGE_CAP");
    const int v = e ? atoi(e) : kMsdCap;
    return v < 64 ? 64 : (v > kMsdCap ? kMsdCap : v);
}

int depth_dist_max() {
    const char* e = getenv("HS_DEPTH_DIST_MAX");
    const int v = e ? atoi(e) : 16;
    return v < 0 ? 0 : (v > 16 ? 16 : v);
}

bool scan_in_emission(int64_t I) {
    // (read at every forward, not once: the driver's test suite switches it inside one process)
    const char* e = getenv("HS_SCAN_IN_EMISSION");
    const int forced = e ? (e[0] == '1' ? 1 : 0) : -1;
    // (the scan inside the emission is a blockIdx-ordered look-back chain like the radix passes': once the process has gone
    // to ticket order -- a chain stalled, several processes share the GPU -- the offsets come from the three kernels ahead
    // of the emission, which wait for nobody)
    return forced >= 0 ? forced == 1 : (I >= (2 << 20) && !sort_tickets());
}

// A/B switch: large frames (those the counting sort does not take) get the hierarchical tile sort by default
#ifndef HS_TUNE_HIER_DEFAULT
#define HS_TUNE_HIER_DEFAULT 1
#endif
int tile_sort_mode(int64_t I, int64_t gx, int64_t gy, int64_t n_poses, int64_t capacity) {
    const char* e = getenv("HS_TILE_SORT");   // (read at every forward: the test suite switches it inside one process)
    const char f = e ? e[0] : 0;
    const bool can_count = HS_TUNE_COUNT_SORT && count_sort_fits(I, gx * gy * n_poses, capacity);
    const bool can_hier = hier_fits(I, gx, gy, n_poses, capacity);
    if (f == 'r') return kTileSortRadix;
    if (f == 'h' && can_hier) return kTileSortHier;
    if (f == 'c') return can_count ? kTileSortCount : kTileSortRadix;
    if (can_count) return kTileSortCount;
    return (HS_TUNE_HIER_DEFAULT && can_hier) ? kTileSortHier : kTileSortRadix;
}


// Stamp of a single-enqueue forward (hs_common.h, kDepthBitsAt): never 0, never the same for two calls of a process that
// could meet in the same memory (2^32 - 1 calls apart).  The only thing the library counts.
// The count starts at a hashed value with the top bit set (process id x clock): small integers 1, 2, 3 ... are what recycled
// int32 data in a fresh torch.empty workspace is most likely to hold, and a stale word that happens to carry the frame's
// tag -- harmless for the order, see hs_common.h -- would cost the first frames an extra depth pass.
static uint32_t frame_tag_seed() {
    uint64_t x = (uint64_t)getpid() * 0x9E3779B97F4A7C15ull ^ (uint64_t)time(nullptr) * 0xBF58476D1CE4E5B9ull;
    x ^= x >> 31; x *= 0x94D049BB133111EBull; x ^= x >> 29;
    return (uint32_t)x | 0x80000000u;
}
static std::atomic<uint32_t> g_frame_tag{frame_tag_seed()};
static uint32_t next_frame_tag() {
    uint32_t t;
    do { t = g_frame_tag.fetch_add(1u, std::memory_order_relaxed) + 1u; } while (t == 0u);
    return t;
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int64_t crf_partial_floats(int K, int64_t HW, int n_poses);
int64_t pose_partial_floats(int P, int N);

static int plan(const hs_dims& d, hs_sizes* sz, hs_layout* L) {
    if (d.P < 0 || d.W <= 0 || d.H <= 0 || d.n_poses < 1 || d.capacity < 0 || d.M < 0 || d.crf_K < 0 || d.crf_K > 4096 ||
        d.crf_K == 1 || d.n_poses > 21845) {
        set_error("hs_plan: bad dims P=%d W=%d H=%d N=%d capacity=%lld", d.P, d.W, d.H, d.n_poses, (long long)d.capacity);
        return HS_EINVAL;
    }
    const int64_t I = (int64_t)d.P * d.n_poses;
    const int64_t gx = (d.W + kTile - 1) / kTile, gy = (d.H + kTile - 1) / kTile;
    const int64_t vtiles = gx * gy * d.n_poses;
    // 2^30: a status word of the radix passes carries a 30-bit count next to its 2-bit flag (binning.hip)
    // 2^22 tiles per pose (a 32768 x 32768 frame): the emission's slot -> tile arithmetic (binning.hip)
    if (I >= (1ll << 30) || vtiles >= (1ll << 31) || d.capacity >= (1ll << 30) || gx * gy >= (1ll << 22)) {
        set_error("hs_plan: problem too large (instances and binning capacity must stay below 2^30, tiles per pose below 2^22)");
        return HS_EINVAL;
    }
    const int64_t HW = (int64_t)d.W * d.H;
    hs_layout l;
    int64_t o = 0;
    auto carve = [&o](int64_t bytes) { int64_t at = o; o = align_up(o + bytes, 256); return at; };
    // geometry
    l.counters = carve(sizeof(hs_counters));
    l.rec = carve(I * kRecFloats * 4);
    l.depth = carve(I * 4);
    l.radii = carve(I * 4);
    l.tiles_touched = carve(I * 4);
    l.offsets = carve(I * 4);
    l.cov3D = carve((int64_t)d.P * 6 * 4);
    l.clamped = carve(I);
    l.scan_spine = carve(((I + 255) / 256 + 1) * 4);
    l.binfo = carve(I * 8);
    sz->geom_bytes = o;
    // binning
    o = 0;
    // keys_sorted | point_list together are also packed buffer A of the tile sort (8 bytes per pair), pairs_tmp is B
    l.keys_sorted = carve(2 * align_up(d.capacity * 4, 256));
    l.point_list = l.keys_sorted + align_up(d.capacity * 4, 256);
    l.pairs_tmp = carve(d.capacity * 8);
    l.ranges = carve(vtiles * 8);
    l.sort_tmp = carve(sort_tmp_bytes(I));
    l.depth_ws = carve(depth_ws_words(I) * 4);   // frames below 2^21 instances: the counting depth sort's matrix
    l.depth_pairs = carve(2 * I * 8);
    l.inst_sorted = carve(I * 4);
    l.offs_sorted = carve(I * 4);
    l.pair_sort_tmp = carve(emit_scan_words(I) * 4 + sort_tmp_bytes(d.capacity));
    l.pair_flags = carve(d.capacity);  // cleared by the pair emission, set by the render backward
    l.pair_act = carve(d.capacity);    // written by the render forward, read by the render backward
    l.tile_matrix = carve(count_matrix_words(I, vtiles, d.capacity) * 4);   // small frames only (else empty)
    l.hier_ws = carve(hier_ws_words(I, gx, gy, d.n_poses, d.capacity) * 4);  // frames of <= 2048 (pose, super-tile) keys
    sz->binning_bytes = o;
    // image
    o = 0;
    l.final_T = carve(HW * d.n_poses * 4);
    l.n_contrib = carve(HW * d.n_poses * 4);
    l.pose_hdr = carve(HW * 3 * 4 * (d.n_poses + (d.n_poses > 1 ? 1 : 0)));
    l.tile_work = carve(vtiles * 4);
    l.tile_order = carve(vtiles * 4);
    sz->image_bytes = o;
    // backward scratch
    o = 0;
    l.pair_grads = carve(d.capacity * kPairFloats * 4);
    l.crf_partials = carve(crf_partial_floats(d.crf_K, HW, d.n_poses) * 4);
    l.inst_grads = carve(I * kInstFloats * 4);
    l.pose_partials = carve(pose_partial_floats(d.P, d.n_poses) * 4);
    sz->bwd_bytes = o;
    if (L) *L = l;
    return HS_OK;
}

// HS_FLAG_DEBUG (the `debug` field of the published settings): wait for each stage and report which one failed.  The
// only situation in which the library synchronises.
static int debug_sync(int flags, hipStream_t s, const char* stage) {
    if (!(flags & HS_FLAG_DEBUG)) return HS_OK;
    const hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        set_error("debug: stage `%s` failed: %s", stage, hipGetErrorString(e));
        return HS_EHIP;
    }
    return HS_OK;
}

static int check_common(const hs_dims& d, const float* means3D, const float* shs, const float* colors,
                        const float* scales, const float* rots, const float* cov, const float* view,
                        const float* proj, const float* campos, const float* bg, const char* who) {
    if (!means3D || !view || !proj || !campos || !bg) { set_error("%s: null required pointer", who); return HS_EINVAL; }
    if ((shs == nullptr) == (colors == nullptr)) { set_error("%s: provide exactly one of shs / colors_precomp", who); return HS_EINVAL; }
    const bool sr = scales && rots;
    if (sr == (cov != nullptr) || ((scales == nullptr) != (rots == nullptr))) {
        set_error("%s: provide exactly one of (scales, rotations) / cov3D_precomp", who);
        return HS_EINVAL;
    }
    if (shs) {
        if (d.M > 32) {  // preprocess-backward stages 128 rows of 3M + 1 floats in LDS (64 KB at M = 42)
            set_error("%s: M=%d SH coefficients per channel; at most 32 are supported", who, d.M);
            return HS_EINVAL;
        }
        if (d.sh_degree < 0 || d.sh_degree > 3 || (d.sh_degree + 1) * (d.sh_degree + 1) > d.M) {
            set_error("%s: sh_degree %d needs %d coefficients, M=%d", who, d.sh_degree, (d.sh_degree + 1) * (d.sh_degree + 1), d.M);
            return HS_EINVAL;
        }
    }
    if (rots && ((uintptr_t)rots & 15)) { set_error("%s: rotations must be 16-byte aligned", who); return HS_EINVAL; }
    return HS_OK;
}

static int check_flags(int flags, const char* who) {
    if ((flags & HS_FLAG_RADIANCE_EXP) && (flags & HS_FLAG_RADIANCE_SOFTPLUS)) {
        set_error("%s: HS_FLAG_RADIANCE_EXP and HS_FLAG_RADIANCE_SOFTPLUS are mutually exclusive", who);
        return HS_EINVAL;
    }
    return HS_OK;
}

}  // namespace hs

using namespace hs;

extern "C" {

int hs_version(void) { return HS_VERSION; }

const char* hs_last_error(void) { return g_err; }

int hs_plan(const hs_dims* dims, hs_sizes* sizes, hs_layout* layout) {
    if (!dims || !sizes) { set_error("hs_plan: null argument"); return HS_EINVAL; }
    return plan(*dims, sizes, layout);
}

int hs_forward(const hs_fwd_args* a, void* hip_stream) {
    if (!a) { set_error("hs_forward: null args"); return HS_EINVAL; }
    hipStream_t s = (hipStream_t)hip_stream;
    hs_sizes sz; hs_layout L;
    int rc = plan(a->dims, &sz, &L);
    if (rc) return rc;
    if ((rc = check_flags(a->flags, "hs_forward"))) return rc;
    if (a->dims.P > 0) {  // an empty cloud has no arrays to validate: it renders the background
        rc = check_common(a->dims, a->means3D, a->shs, a->colors_precomp, a->scales, a->rotations, a->cov3D_precomp,
                          a->viewmatrices, a->projmatrices, a->camposes, a->bg, "hs_forward");
        if (rc) return rc;
        if (!a->opacities || ((a->stages & HS_STAGE_PREPROCESS) && !a->radii)) {  // radii is written by preprocess only
            set_error("hs_forward: null opacities/radii");
            return HS_EINVAL;
        }
    }
    if (!a->geom || !a->bg) { set_error("hs_forward: null geom/bg"); return HS_EINVAL; }
    if ((a->flags & HS_FLAG_HDR) && (!a->exposure || !a->crf_table || a->crf_K < 2 || a->crf_K > 4096 || !(a->crf_umax > a->crf_umin))) {
        set_error("hs_forward: HDR needs exposure, crf_table, 2 <= crf_K <= 4096 and umax > umin");
        return HS_EINVAL;
    }
    if ((a->flags & HS_FLAG_HDR) && a->dims.crf_K != a->crf_K) {
        set_error("hs_forward: dims.crf_K (%d, sizes the workspaces) differs from crf_K (%d)", a->dims.crf_K, a->crf_K);
        return HS_EINVAL;
    }
    if (((uintptr_t)a->geom & 255) || ((uintptr_t)a->binning & 255) || ((uintptr_t)a->image & 255)) {
        set_error("hs_forward: workspaces must be 256-byte aligned");
        return HS_EINVAL;
    }
    if (a->dims.P == 0) {
        // nothing to rasterize: clear counters so the host sees R = 0, and paint the background.  (By a kernel, like every
        // other path: an empty cloud inside a captured step must not put memset / copy nodes into the graph -- ADVICE r5)
        const int64_t gx0 = (a->dims.W + kTile - 1) / kTile, gy0 = (a->dims.H + kTile - 1) / kTile;
        const bool bin = (a->stages & HS_STAGE_BIN) && !(a->stages & HS_STAGE_PREPROCESS_ONLY) && a->binning;
        rc = launch_empty_frame((a->stages & HS_STAGE_PREPROCESS) ? (hs_counters*)((char*)a->geom + L.counters) : nullptr,
                                bin ? (uint2*)((char*)a->binning + L.ranges) : nullptr, gx0 * gy0 * a->dims.n_poses,
                                bin ? (uint32_t*)a->counters_host : nullptr, (const hs_counters*)((char*)a->geom + L.counters), s);
        if (rc) return rc;
    }
    const uint32_t frame_tag = ((a->stages & HS_STAGE_PREPROCESS) && (a->stages & HS_STAGE_BIN) && a->binning) ? next_frame_tag() : 0u;
    if ((a->stages & HS_STAGE_PREPROCESS) && a->dims.P > 0) {
        rc = launch_preprocess_fwd(*a, L, s, frame_tag);
        if (rc) return rc;
        if ((rc = debug_sync(a->flags, s, "preprocess"))) return rc;
        if (!(a->stages & HS_STAGE_BIN)) {  // upstream-style call: the host reads num_rendered before binning
            rc = launch_scan(*a, L, s);
            if (rc) return rc;
        }
    }
    if ((a->stages & HS_STAGE_BIN) && !(a->stages & HS_STAGE_PREPROCESS_ONLY)) {
        if (!a->binning) { set_error("hs_forward: null binning workspace"); return HS_EINVAL; }
        if (a->dims.P > 0) {
            rc = launch_binning(*a, L, s, frame_tag);
            if (rc) return rc;
            if ((rc = debug_sync(a->flags, s, "binning"))) return rc;
        }   // (an empty cloud: launch_empty_frame above cleared the ranges and wrote the host copy of the counters)
    }
    if ((a->stages & HS_STAGE_OFFSETS) && a->dims.P > 0) {
        rc = launch_scan(*a, L, s);  // inspection only: a5 in instance order
        if (rc) return rc;
        if ((rc = launch_cov3d(*a, L, s))) return rc;   // ... and the 3-D covariances, which the pipeline does not keep
        // ... and, for a frame whose pairs were sorted by counting, the sorted tile ids (the radix path leaves them behind)
        // (the kernel asks the frame's counters which sort it had: hs_counters.reserved[5])
        if (a->binning && a->dims.capacity > 0)
            if ((rc = launch_tile_keys(*a, L, s))) return rc;
    }
    if (a->stages & HS_STAGE_RENDER) {
        if (!a->binning || !a->image || !a->out_color) { set_error("hs_forward: null binning/image/out_color"); return HS_EINVAL; }
        rc = launch_render_fwd(*a, L, s);
        if (rc) return rc;
        if ((rc = debug_sync(a->flags, s, "render forward"))) return rc;
    }
    return HS_OK;
}

int hs_backward(const hs_bwd_args* a, void* hip_stream) {
    if (!a) { set_error("hs_backward: null args"); return HS_EINVAL; }
    hipStream_t s = (hipStream_t)hip_stream;
    hs_sizes sz; hs_layout L;
    int rc = plan(a->dims, &sz, &L);
    if (rc) return rc;
    if ((rc = check_flags(a->flags, "hs_backward"))) return rc;
    rc = check_common(a->dims, a->means3D, a->shs, a->colors_precomp, a->scales, a->rotations, a->cov3D_precomp,
                      a->viewmatrices, a->projmatrices, a->camposes, a->bg, "hs_backward");
    if (rc) return rc;
    if (!a->geom || !a->binning || !a->image || !a->bwd || !a->dL_dout_color) {
        set_error("hs_backward: null workspace or dL_dout_color");
        return HS_EINVAL;
    }
    if ((a->flags & HS_FLAG_HDR) && (!a->exposure || !a->crf_table || a->crf_K < 2 || a->crf_K > 4096 || a->dims.crf_K != a->crf_K)) {
        set_error("hs_backward: HDR needs exposure, crf_table and dims.crf_K == crf_K");
        return HS_EINVAL;
    }
    const int npose_out = (a->dL_dviewmatrices != nullptr) + (a->dL_dprojmatrices != nullptr) + (a->dL_dcamposes != nullptr);
    if (npose_out != 0 && npose_out != 3) {
        set_error("hs_backward: pose gradients need dL_dviewmatrices, dL_dprojmatrices and dL_dcamposes together");
        return HS_EINVAL;
    }
    const int ndens = (a->densify_grad_accum != nullptr) + (a->densify_denom != nullptr) + (a->densify_max_radii != nullptr);
    if (ndens != 0 && ndens != 3) {
        set_error("hs_backward: densification statistics need grad_accum, denom and max_radii together");
        return HS_EINVAL;
    }
    if (a->g_begin != 0 || a->g_end != 0) {
        if (a->g_begin < 0 || a->g_end < a->g_begin || a->g_end > a->dims.P || (a->g_begin & 127) ||
            (a->stages & ~HS_BWD_PROJECT)) {
            set_error("hs_backward: [g_begin, g_end) = [%d, %d) needs 0 <= g_begin <= g_end <= P, g_begin a multiple of 128, "
                      "and stages == HS_BWD_PROJECT", a->g_begin, a->g_end);
            return HS_EINVAL;
        }
    }
    if (a->dims.P == 0) return HS_OK;
    // the CRF gradient's first stage rides at the end of the render backward's launch when both run in this call
    // (its table of K - 1 64-bit LDS words sits in the kernel's staging area: 9 KB, i.e. up to 1024 knots)
    const bool crf_in_tail = HS_TUNE_CRF_IN_RENDER_TAIL && (a->stages & HS_BWD_RENDER) && (a->stages & HS_BWD_CRF) &&
                             !(a->flags & HS_FLAG_DEBUG) && a->crf_K <= 1024;
    if (a->stages & HS_BWD_RENDER) {
        rc = launch_render_bwd(*a, L, s, nullptr, nullptr, crf_in_tail);
        if (rc) return rc;
        if ((rc = debug_sync(a->flags, s, "render backward"))) return rc;
    }
    // the second stage of the CRF-table gradient (adding the pixel blocks' partial rows up) rides on the segmented sum's
    // launch when this call goes on to it: one launch less on the critical path
    CrfReduce crf_reduce{nullptr, 0, 0, 0, nullptr, nullptr, 0};
    const bool sums_follow = (a->stages & (HS_BWD_PREPROCESS | HS_BWD_SEGSUM)) != 0 && !(a->flags & HS_FLAG_DEBUG);
    if (a->stages & HS_BWD_CRF) {
        rc = launch_crf_bwd(*a, L, s, sums_follow ? &crf_reduce : nullptr, crf_in_tail);
        if (rc) return rc;
        if ((rc = debug_sync(a->flags, s, "CRF gradient"))) return rc;
    }
    if (a->stages & (HS_BWD_PREPROCESS | HS_BWD_SEGSUM | HS_BWD_PROJECT)) {
        const bool whole = (a->stages & HS_BWD_PREPROCESS) != 0;
        rc = launch_preprocess_bwd(*a, L, s, whole || (a->stages & HS_BWD_SEGSUM), whole || (a->stages & HS_BWD_PROJECT),
                                   crf_reduce.nblocks ? &crf_reduce : nullptr);
        if (rc) return rc;
        if ((rc = debug_sync(a->flags, s, "preprocess backward"))) return rc;
    }
    return HS_OK;
}

int hs_depth_sort(int mode) {
    if (mode >= 0) hs::g_depth_sort.store(mode ? 1 : 0, std::memory_order_relaxed);
    const int v = hs::g_depth_sort.load(std::memory_order_relaxed);
    return v >= 0 ? v : (HS_TUNE_DEPTH_MSD_DEFAULT ? 1 : 0);
}

int hs_sort_tickets(int enable) {
    if (enable >= 0) { (void)hs::sort_tickets(); hs::g_sort_tickets.store(enable ? 1 : 0, std::memory_order_relaxed); }
    return hs::sort_tickets() ? 1 : 0;
}

int hs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, uint8_t* visible, void* hip_stream) {
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !visible))) { set_error("hs_mark_visible: bad argument"); return HS_EINVAL; }
    if (P == 0) return HS_OK;
    return launch_mark_visible(P, means3D, viewmatrix, visible, (hipStream_t)hip_stream);
}

int hs_sh_backward_views(int32_t P, int32_t M, int32_t sh_degree, int32_t V, const float* means3D, const float* camposes,
                         const float* dL_dview_colors, float* dL_dshs, void* hip_stream) {
    if (P < 0 || V < 1 || sh_degree < 0 || sh_degree > 3 || M < (sh_degree + 1) * (sh_degree + 1) || M > 16 ||
        (P > 0 && (!means3D || !camposes || !dL_dview_colors || !dL_dshs))) {
        set_error("hs_sh_backward_views: bad argument");
        return HS_EINVAL;
    }
    if (P == 0) return HS_OK;
    return launch_sh_backward_views(P, M, sh_degree, V, means3D, camposes, dL_dview_colors, dL_dshs, (hipStream_t)hip_stream);
}

int hs_spline_poses(int32_t n_knots, int32_t n_times, int32_t kind, const float* delta, const float* base_w2c,
                    const float* times, float* w2c, float* jacobian, int32_t* segment, void* hip_stream) {
    if (n_times < 0 || (kind != 0 && kind != 1) || n_knots < (kind == 1 ? 4 : 2) ||
        (n_times > 0 && (!delta || !base_w2c || !times || !w2c || !jacobian || !segment))) {
        set_error("hs_spline_poses: bad argument (kind 0 = linear needs >= 2 knots, 1 = cubic >= 4)");
        return HS_EINVAL;
    }
    if (n_times == 0) return HS_OK;
    return launch_spline_poses(n_knots, n_times, kind, delta, base_w2c, times, w2c, jacobian, segment, (hipStream_t)hip_stream);
}

int hs_render_stats(const hs_fwd_args* fwd, const hs_bwd_args* bwd, uint64_t* stats, uint64_t* bwd_timeline, void* hip_stream) {
    if ((!fwd && !bwd) || !stats) { set_error("hs_render_stats: null argument"); return HS_EINVAL; }
    hipStream_t s = (hipStream_t)hip_stream;
    hs_sizes sz; hs_layout L;
    if (fwd) {
        int rc = plan(fwd->dims, &sz, &L);
        if (rc) return rc;
        if (!fwd->geom || !fwd->binning || !fwd->image || !fwd->out_color || !fwd->bg) {
            set_error("hs_render_stats: forward args need geom/binning/image/out_color/bg");
            return HS_EINVAL;
        }
        if (fwd->dims.P > 0 && (rc = launch_render_fwd(*fwd, L, s, (unsigned long long*)stats))) return rc;
    }
    if (bwd) {
        int rc = plan(bwd->dims, &sz, &L);
        if (rc) return rc;
        if (!bwd->geom || !bwd->binning || !bwd->image || !bwd->bwd || !bwd->dL_dout_color || !bwd->bg) {
            set_error("hs_render_stats: backward args need geom/binning/image/bwd/dL_dout_color/bg");
            return HS_EINVAL;
        }
        if (bwd->dims.P > 0 && (rc = launch_render_bwd(*bwd, L, s, (unsigned long long*)stats, (unsigned long long*)bwd_timeline))) return rc;
    }
    return HS_OK;
}

int64_t hs_sort_tmp_bytes(int64_t n) { return sort_tmp_bytes(n) + 256 + 2 * align_up(n * 8, 256) + 2 * align_up(n * 4, 256); }

int hs_sort_pairs(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                  int64_t n, int32_t nbits, void* tmp, void* hip_stream) {
    if (n < 0 || n >= (1ll << 30) || nbits < 1 || nbits > 64 || (n > 0 && (!keys_in || !vals_in || !keys_out || !vals_out || !tmp))) {
        set_error("hs_sort_pairs: bad argument");
        return HS_EINVAL;
    }
    if (n == 0) return HS_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    char* t = (char*)tmp;
    uint32_t* n_dev = (uint32_t*)t;        // [0] element count, [1] fail word (2 = a look-back gave up: results invalid)
    uint32_t* fail_word = n_dev + 1; t += 256;
    uint64_t* kb = (uint64_t*)t; t += align_up(n * 8, 256);
    uint32_t* vb = (uint32_t*)t; t += align_up(n * 4, 256);
    uint64_t* ka = (uint64_t*)t; t += align_up(n * 8, 256);
    uint32_t* va = (uint32_t*)t; t += align_up(n * 4, 256);
    HS_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)n_dev, (int)(uint32_t)n, 1, s));
    HS_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)fail_word, 0, 1, s));
    HS_HIP_CHECK(hipMemcpyAsync(ka, keys_in, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    HS_HIP_CHECK(hipMemcpyAsync(va, vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    int rc = launch_radix_sort(ka, va, kb, vb, n_dev, n, nbits, t, fail_word, s);
    if (rc) return rc;
    const bool in_a = sort_passes(nbits) % 2 == 0;
    HS_HIP_CHECK(hipMemcpyAsync(keys_out, in_a ? ka : kb, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    HS_HIP_CHECK(hipMemcpyAsync(vals_out, in_a ? va : vb, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return HS_OK;
}

}  // extern "C"
