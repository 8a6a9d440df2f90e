/*
 * hdrsplat.h -- C ABI of libhdrsplat.so, the MI355X (gfx950) differentiable Gaussian rasterizer.
 *
 * Boundary being replaced.  BASELINE.json's north_star fixes the drop-in boundary as the
 * GaussianRasterizer / GaussianRasterizationSettings Python API.  /root/reference contains no
 * code at all (Readme.md:1-58 + two figures), so there is no reference
 * FFI file:line to cite.  The interface each entry point replaces is therefore the *published*
 * binding of the third-party package that API belongs to (diff_gaussian_rasterization, not
 * vendored/pinned by the reference -- SURVEY.md 2.3, 8b):
 *
 *   hs_forward       <-> _C.rasterize_gaussians           (pybind11, torch tensors)   [SURVEY 3.2]
 *   hs_backward      <-> _C.rasterize_gaussians_backward                              [SURVEY 3.3]
 *   hs_mark_visible  <-> _C.mark_visible                                              [SURVEY 2.3 a14]
 *   hs_plan          <-> the resize-callback carving of geomBuffer/binningBuffer/imgBuffer [a13]
 *
 * Contract: plain C structs of raw DEVICE pointers and scalars; no torch types,
 * no C++ exceptions across the ABI; the caller owns every byte (the library never allocates
 * device memory and keeps no state between calls); all work is enqueued on the caller's HIP
 * stream and nothing inside synchronises; return 0 = ok, negative = error, text through
 * hs_last_error() (thread-local).
 *
 * "Instance" below means one (pose, Gaussian) pair: with n_poses = N the library renders N
 * virtual sharp images in one launch (pose id folded into the tile sort key) and averages
 * them (motion blur as N-pose render averaging, /root/reference/assets/pipeline.png "+").
 */
#ifndef HDRSPLAT_H
#define HDRSPLAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libhdrsplat.so is built with -fvisibility=hidden: the hs_* entry points below are its only exported symbols */
#define HS_API __attribute__((visibility("default")))

#define HS_VERSION 308

#define HS_OK 0
#define HS_EINVAL (-1)    /* bad argument (null pointer, bad shape, unsupported degree ...) */
#define HS_EHIP (-2)      /* a HIP runtime call failed; see hs_last_error() */
#define HS_EOVERFLOW (-3) /* binning capacity smaller than the number of (tile, instance) pairs */

#define HS_TILE 16 /* binning tile edge in pixels (BLOCK_X = BLOCK_Y = 16 upstream) */

/* hs_fwd_args.stages */
#define HS_STAGE_PREPROCESS 1 /* preprocess; when run WITHOUT HS_STAGE_BIN (the upstream-style call, host reads
                                 num_rendered before binning) also the instance-order scan of tiles_touched */
#define HS_STAGE_BIN 2        /* depth sort, duplicateWithKeys with the scan of the depth-ordered counts inside (writes
                                 num_rendered), tile sort, tile ranges */
#define HS_STAGE_RENDER 4     /* per-tile alpha blend (+ HDR epilogue, + N-pose resolve) */
#define HS_STAGE_ALL 7
#define HS_STAGE_OFFSETS 8    /* inspection only: inclusive scan of tiles_touched in instance order into the geom
                                 workspace (`offsets`; the pipeline itself scans the depth-ordered counts) and the 3-D
                                 covariances (`cov3D`; the pipeline recomputes them in the backward instead of storing);
                                 (HS_VERSION 305) with a binning workspace of a frame sorted by counting, also keys_sorted */

#define HS_STAGE_PREPROCESS_ONLY 16 /* profiling (bench.py's per-kernel roofline leg): with PREPROCESS | BIN, enqueue the
                                 preprocess kernel exactly as a single-enqueue forward does -- carrying the binning stage's
                                 prologue: depth keys, cleared scratch and ranges -- and stop before the binning kernels */

/* hs_bwd_args.stages */
#define HS_BWD_RENDER 1      /* per-pixel backward -> one gradient record per (tile, instance) pair */
#define HS_BWD_PREPROCESS 2  /* per-instance record sum + computeCov2D/projection/SH/cov3D backward */
#define HS_BWD_CRF 4         /* CRF-table and exposure gradients (HDR only) */
#define HS_BWD_ALL 7
/* the two halves of HS_BWD_PREPROCESS on their own (a caller that all-gathers dL_dview_colors over the ranks can
 * start that exchange between them) */
#define HS_BWD_SEGSUM 8      /* per-instance record sums (+ dL_dview_colors when given) */
#define HS_BWD_PROJECT 16    /* computeCov2D/projection/SH/cov3D backward from the record sums */

/* flags */
#define HS_FLAG_HDR 1          /* exposure * CRF tone-map epilogue; out_color = LDR, out_hdr = radiance */
#define HS_FLAG_BLUR_HDR 2     /* N-pose average taken on radiance before the CRF (default: on LDR) */
#define HS_FLAG_DEBUG 4         /* wait for every stage and name the failing one (the only case of a sync) */
#define HS_FLAG_ANTIALIAS 8    /* newer published rasterizer's `antialiasing`: opacity *= sqrt(max(0.000025,
                                  det(cov2D) / det(cov2D + 0.3 I))), with its gradient */
/* radiance activation (SURVEY.md 7.3 / 8a a1 `radiance_activation`; the reference reconstructs an HDR scene,
 * /root/reference/Readme.md:54): how the SH sum s of a Gaussian becomes its linear-radiance colour.  Neither bit:
 * relu_shift = max(s + 0.5, 0), the published rule (unbounded above, clamped below with the clamp mask zeroing the
 * gradient).  Precomputed colours pass through unchanged. */
#define HS_FLAG_RADIANCE_EXP 16       /* colour = e^s            (always positive; d colour / d s = colour) */
#define HS_FLAG_RADIANCE_SOFTPLUS 32  /* colour = ln(1 + e^s)    (d colour / d s = sigmoid(s) = -expm1(-colour)) */

typedef struct hs_dims {
    int32_t P;         /* Gaussians */
    int32_t M;         /* SH coefficients stored per Gaussian and channel (0 if colors_precomp) */
    int32_t sh_degree; /* active degree, (sh_degree+1)^2 <= M */
    int32_t W, H;
    int32_t n_poses;   /* N >= 1 */
    int64_t capacity;  /* binning capacity in (tile, instance) pairs */
    int32_t crf_K;     /* knots per channel of the CRF table, 0 when the call has no HDR tone-map: sizes the scratch of
                          the CRF-gradient stage (must equal hs_fwd_args.crf_K / hs_bwd_args.crf_K under HS_FLAG_HDR) */
    int32_t reserved;  /* 0 */
} hs_dims;

typedef struct hs_sizes {
    int64_t geom_bytes;    /* per-instance geometry state (a13 GeometryState) */
    int64_t binning_bytes; /* keys/values double buffers, histograms, ranges (BinningState) */
    int64_t image_bytes;   /* final_T, n_contrib, per-pose radiance (ImageState) */
    int64_t bwd_bytes;     /* backward scratch: per-pair gradient records, CRF partials */
} hs_sizes;

/* First bytes of the geometry workspace; the host may read them after HS_STAGE_PREPROCESS. */
typedef struct hs_counters {
    uint32_t num_rendered; /* R = sum of tiles_touched over all instances */
    uint32_t overflow;     /* HS_STAGE_BIN: 1 = R > capacity; 2 = a radix pass gave up waiting for a predecessor's status
                              word (damaged scratch).  Either way the frame is rendered empty */
    uint32_t reserved[6];  /* [0] = pairs actually binned (R, or 0 on overflow); [1] = instance count of the depth sort;
                              [2] = ranges of the counting depth sort that did not fit the LDS and were sorted through
                              memory by one workgroup (correct, slow: see hs_depth_sort); [3] = tile-queue counter of the render backward (zero between launches); [4] = times a
                              waiting workgroup of HS_STAGE_BIN had to compute a silent predecessor's counts itself
                              (non-zero: other kernels kept its blocks off the GPU -- see hs_sort_tickets); [5] = the tile
                              sort HS_STAGE_BIN ran (0 radix passes, 1 counting, 2 hierarchical); others unused */
} hs_counters;

typedef struct hs_fwd_args {
    hs_dims dims;
    float tanfovx, tanfovy, scale_modifier;
    int32_t flags;
    int32_t stages;
    int32_t crf_K;               /* knots per channel of crf_table (HDR) */
    float crf_umin, crf_umax;    /* log-exposure range spanned by the table */
    /* device inputs */
    const float* bg;             /* [3] */
    const float* viewmatrices;   /* [N,16] flat "transposed": x' = m[0]x + m[4]y + m[8]z + m[12] */
    const float* projmatrices;   /* [N,16] full view*proj, same convention */
    const float* camposes;       /* [N,3] */
    const float* means3D;        /* [P,3] */
    const float* opacities;      /* [P] */
    const float* shs;            /* [P,M,3] or NULL */
    const float* colors_precomp; /* [P,3] or NULL */
    const float* scales;         /* [P,3] or NULL */
    const float* rotations;      /* [P,4] (w,x,y,z) or NULL */
    const float* cov3D_precomp;  /* [P,6] or NULL */
    const float* exposure;       /* [1] (HDR) or NULL */
    const float* crf_table;      /* [3,crf_K] (HDR) or NULL */
    /* caller-owned workspaces, sizes from hs_plan(), 256-byte aligned */
    void* geom;
    void* binning;
    void* image;
    /* device outputs */
    float* out_color;            /* [3,H,W]; LDR when HS_FLAG_HDR */
    float* out_hdr;              /* [3,H,W] linear radiance (HDR) or NULL */
    int32_t* radii;              /* [P] max over poses */
    float* out_invdepth;         /* [N,H,W] or NULL: expected inverse depth sum_i alpha_i T_i / z_i per pose
                                    (the caller averages the poses) */
    void* counters_host;         /* (HS_VERSION 304) NULL, or a host address the GPU can write (page-locked, mapped: what
                                    hipHostMalloc / torch pin_memory return): HS_STAGE_BIN leaves a copy of hs_counters
                                    (32 bytes) there -- written by its last kernel, so a sync-free caller that wants to
                                    look at num_rendered / overflow LATER needs no copy of its own on the stream */
} hs_fwd_args;

typedef struct hs_bwd_args {
    hs_dims dims;
    float tanfovx, tanfovy, scale_modifier;
    int32_t flags;
    int32_t stages;               /* HS_BWD_* bitmask; bench/profiling may run the two halves separately */
    int32_t crf_K;
    float crf_umin, crf_umax;
    const float* bg;
    const float* viewmatrices;
    const float* projmatrices;
    const float* camposes;
    const float* means3D;
    const float* opacities;
    const float* shs;
    const float* colors_precomp;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    const float* exposure;
    const float* crf_table;
    /* state produced by hs_forward (same buffers) */
    const void* geom;
    const void* binning;
    const void* image;
    void* bwd;                    /* scratch, hs_sizes.bwd_bytes */
    /* upstream gradients */
    const float* dL_dout_color;   /* [3,H,W] */
    const float* dL_dout_hdr;     /* [3,H,W] or NULL */
    const float* dL_dout_alpha;   /* [H,W] or NULL: gradient w.r.t. the accumulated-opacity image 1 - mean_k final_T_k */
    /* outputs (each may be NULL when its input is absent) */
    float* dL_dmeans3D;           /* [P,3] */
    float* dL_dmeans2D;           /* [P,3] screen-space gradient (NDC-scaled), summed over poses */
    float* dL_dopacities;         /* [P] */
    float* dL_dshs;               /* [P,M,3] */
    float* dL_dcolors_precomp;    /* [P,3] */
    float* dL_dscales;            /* [P,3] */
    float* dL_drotations;         /* [P,4] */
    float* dL_dcov3D_precomp;     /* [P,6] */
    float* dL_dexposure;          /* [1] (HDR) */
    float* dL_dcrf_table;         /* [3,crf_K] (HDR) */
    /* camera-pose gradients (the reference optimises camera motion jointly, Readme.md:54);
     * all three or none; same flat transposed layout as the inputs, unused entries are zero */
    float* dL_dviewmatrices;      /* [N,16] or NULL */
    float* dL_dprojmatrices;      /* [N,16] or NULL */
    float* dL_dcamposes;          /* [N,3]  or NULL */
    /* view-parallel exchange: when non-NULL, the colour gradient of every instance AFTER the SH clamp
     * mask (zero for culled instances) is written here, [N,P,3]; dL_dshs may then be NULL, and the SH-coefficient
     * gradient is formed later from the views of ALL ranks by hs_sh_backward_views */
    float* dL_dview_colors;
    const float* dL_dout_invdepth; /* [H,W] or NULL: gradient w.r.t. the pose-averaged inverse-depth image */
    /* densification statistics, all three or none; updated IN PLACE for every Gaussian that was
     * rasterized in at least one pose: grad_accum += |dL/dmean2D.xy| (the NDC-scaled gradient returned in
     * dL_dmeans2D), denom += 1, max_radii = max(max_radii, radius) -- what a 3DGS trainer keeps between
     * densification rounds, without re-reading the gradient tensors */
    float* densify_grad_accum;    /* [P] */
    float* densify_denom;         /* [P] */
    int32_t* densify_max_radii;   /* [P] */
    /* HS_BWD_PROJECT over the Gaussians [g_begin, g_end) only (both 0: all of them).  The per-Gaussian half of the
     * backward is one thread per Gaussian, so a view-parallel step may run it in ascending chunks and start the
     * exchange of a chunk's gradient rows while the next chunk computes (casualhdrsplat_amd.distributed.
     * chunked_all_reduce; BASELINE.json configs[4]).  g_begin must be a multiple of 128; the chunks of one backward
     * must be enqueued in ascending order, the last one ending at P (it also finishes the pose-gradient reduction). */
    int32_t g_begin, g_end;
} hs_bwd_args;

/* Byte offsets of the arrays carved out of the three state workspaces, for tests, profilers and
 * INTEGRATION.md-style bindings that want to inspect intermediates (keys, point_list, ranges...). */
typedef struct hs_layout {
    /* geom workspace; arrays are indexed by instance = pose * P + gaussian */
    int64_t counters, rec, depth, radii, tiles_touched, offsets, cov3D, clamped, scan_spine, binfo;
    /* binning workspace: keys_sorted = u32 tile id of each sorted pair, point_list = u32 instance of each sorted
     * pair (the sort key of the published algorithm is (tile << 32) | depth_bits[instance]); pairs_tmp = scratch of the
     * tile sort ((tile, instance) as 8-byte elements; keys_sorted | point_list double as its other buffer);
     * depth_pairs = scratch of the depth sort (2 x I 8-byte (depth bits, instance) elements), inst_sorted = the
     * instances in depth order (u32 x I), offs_sorted = inclusive scan of their pair counts in that order (u32 x I:
     * instance inst_sorted[i] owns the pair slots [offs_sorted[i-1], offs_sorted[i])); sort_tmp / pair_sort_tmp = scratch
     * of the depth sort / of the pair emission's scan and the tile sort (digit totals, status words) */
    int64_t keys_sorted, point_list, pairs_tmp, ranges, sort_tmp, depth_pairs, inst_sorted, offs_sorted, pair_sort_tmp;
    /* pair_flags (binning workspace): u8 per pair slot, cleared by the forward's pair emission, set to 1 by the
     * render backward for the records it wrote */
    int64_t pair_flags;
    /* pair_act (binning workspace): u8 per SORTED pair, written by the render forward for every entry it staged: bit
     * 2g + w = some pixel of lane group g (the 8 x 8 block of columns 8g .. 8g+7) of half tile w (rows 8w .. 8w+7 of the
     * tile) took the entry -- four bits, one per 8 x 8 block of the 16 x 16 tile.  The render backward walks exactly
     * those entries, one list per block */
    int64_t pair_act;
    /* image workspace */
    int64_t final_T, n_contrib, pose_hdr;
    /* tile_work: u32 per (pose, tile): (half tile, entry) trips the render forward counted on it = what the render
     * backward will replay; tile_order: u32 per render-backward workgroup: the (pose, tile) it processes (the forward
     * orders the tiles so that each XCD's lightest ones run last, see render.hip) */
    int64_t tile_work, tile_order;
    /* bwd workspace */
    /* inst_grads: 12 floats per instance, the per-instance sum of its (flagged) pair records */
    int64_t pair_grads, crf_partials, inst_grads, pose_partials;
    /* (HS_VERSION 305) tile_matrix (binning workspace; empty unless the frame is small: <= 4096 (pose, tile) keys and
     * <= 2^21 (emission workgroup, key) entries): scratch of the counting tile sort such frames get instead of radix passes
     * -- u32 [ceil(I/256)][keys] pair counts (one byte per wave) | u32 [ceil(I/256)][keys] pairs of the key in earlier
     * workgroups | u32 [keys] totals.  Such a forward writes point_list and ranges but NOT keys_sorted (nothing reads it);
     * HS_STAGE_OFFSETS fills keys_sorted from the ranges for inspection. */
    int64_t tile_matrix;
    /* (HS_VERSION 307) hier_ws (binning workspace; empty unless the frame has <= 2048 (pose, 8 x 8-tile super-tile) keys):
     * scratch of the hierarchical tile sort -- u32 header | element counts per super-tile | pairs per tile | first sorted
     * position per tile | first element / chunk per super-tile | chunk descriptors | per-chunk pair counts.  Selected with
     * HS_TILE_SORT=hier in the environment; like the counting sort it writes point_list and ranges but not keys_sorted.
     * hs_counters.reserved[5] records which tile sort a forward ran (0 radix, 1 counting, 2 hierarchical). */
    int64_t hier_ws;
    /* (HS_VERSION 308) depth_ws (binning workspace; empty unless the frame has fewer than 2^21 instances): scratch of the
     * depth sort such frames get instead of look-back passes -- per block of 1024 / 4096 instances a row of 4096 u16 bucket
     * counts (top 12 varying key bits) | a row of u32 prefixes down the columns | u32 [4096] bucket totals.  Written before
     * it is read: nothing to clear.  See hs_depth_sort. */
    int64_t depth_ws;
} hs_layout;

HS_API int hs_version(void);
HS_API const char* hs_last_error(void);
/* Limits (HS_EINVAL beyond them): P * n_poses < 2^30 instances, capacity < 2^30 pairs, fewer than 2^22 tiles per pose
 * (a 32768 x 32768 frame), n_poses <= 21845, crf_K <= 4096. */
HS_API int hs_plan(const hs_dims* dims, hs_sizes* sizes, hs_layout* layout /* may be NULL */);
HS_API int hs_forward(const hs_fwd_args* args, void* hip_stream);
HS_API int hs_backward(const hs_bwd_args* args, void* hip_stream);
HS_API int hs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, uint8_t* visible,
                    void* hip_stream);

/* SH-coefficient gradient from per-view colour gradients (the multi-GPU exchange of SURVEY.md 8e, where ranks
 * all-gather 12 bytes per Gaussian and view instead of all-reducing the 12*M-byte SH gradient rows):
 *   dL_dshs[g, k, c] = sum_{v < V} Y_k(normalize(means3D[g] - camposes[v])) * dL_dview_colors[v, g, c],
 * views added in ascending order, k < (sh_degree+1)^2, rows k >= that are zeroed.  Same basis and per-view
 * arithmetic as the SH part of hs_backward, so V = 1 reproduces its dL_dshs bit for bit. */
HS_API int hs_sh_backward_views(int32_t P, int32_t M, int32_t sh_degree, int32_t V, const float* means3D,
                         const float* camposes /* [V,3] */, const float* dL_dview_colors /* [V,P,3] */,
                         float* dL_dshs /* [P,M,3] */, void* hip_stream);

/* (HS_VERSION 306) Camera poses along the trajectory spline of the image-formation model -- the N virtual poses inside each
 * captured frame's exposure window (/root/reference/assets/pipeline.png: "camera motion spline" through the control knots
 * T_j .. T_{j+3}; Readme.md:54) -- together with their Jacobian, in one launch (spline.hip; the tensor-operation form of the
 * same arithmetic, image_formation.TrajectorySpline.pose_at, is ~1000 tiny kernels forward + backward per call).
 *   knot_j = exp(delta[j]) * base_w2c[j]   (delta: [n_knots, 6] se(3) corrections (rho, omega); base: [n_knots, 4, 4] row-major)
 *   kind 1 (cubic cumulative B-spline): times in [1, n_knots - 2]; kind 0 (geodesic between two knots): times in [0, n_knots - 1]
 * Outputs per sample time s: w2c[s] (4 x 4 row-major world-to-camera), segment[s] = j, the first knot governing the sample, and
 * jacobian[s][o][i], o = 4 * row + column over the first three rows of w2c[s] (12 values), i < 24: d / d delta[j + i / 6][i % 6]
 * (knots beyond the two of a linear segment: zeros), i = 24: d / d times[s].  Device pointers; 25 threads per sample. */
HS_API int hs_spline_poses(int32_t n_knots, int32_t n_times, int32_t kind, const float* delta, const float* base_w2c,
                    const float* times, float* w2c /* [n_times,4,4] */, float* jacobian /* [n_times,12,25] */,
                    int32_t* segment /* [n_times] */, void* hip_stream);

/* Bench/profiling only: re-runs the render stage(s) of a finished hs_forward (and hs_backward) call -- same argument
 * structs, same buffers, so the outputs are simply rewritten -- with the diagnostic instantiation of the kernels, which
 * ADDS its counts to stats[0..HS_RENDER_STATS) (device memory, zeroed by the caller).  Either struct may be NULL.
 *   [0] backward (wave, entry) trips  [1] ... with no active lane  [2] sum of active pixels over trips (<= 128 each)
 *   [3] entries rejected by the half-tile test (per wave)  [4..9] trips by active lanes: 0, 1-4, 5-8, 9-16, 17-32, 33-64
 *   [10] entries staged (per tile)  [11] staging batches  [12..17] forward: trips, empty, active pixels, culled,
 *   staged, batches.  bench.py derives lane utilisation and the VALU roofline from them. */
#define HS_RENDER_STATS 24
HS_API int hs_render_stats(const hs_fwd_args* fwd /* or NULL */, const hs_bwd_args* bwd /* or NULL */, uint64_t* stats,
                    uint64_t* bwd_timeline /* or NULL: per workgroup of the backward launch (tiles x poses of them)
                                              {start, end} on the 100 MHz device clock and (XCC id << 32 | HW_ID) */,
                    void* hip_stream);

/* The depth sort of HS_STAGE_BIN for frames of fewer than 2^21 instances, process-wide: 1 (default) = by counting -- one
 * stable counting pass over the top 12 varying bits of the depth keys (per-block bucket counts, a column scan, one
 * scatter), then every run of buckets of about 2048 instances (512 up to 2^18 instances) sorted to the end by one workgroup
 * inside its LDS; 0 = the
 * stable look-back radix passes larger frames always get.  Same result bit for bit (a stable sort has one).  The
 * counting form assumes that no 2048 consecutive positions of the bucket order spill over 4096 instances, i.e. that no
 * depth sliver of 1 / 4096 of the key range holds more than ~2048 instances; a range that does is sorted through memory by
 * its one workgroup -- correct, but a frame dominated by such a range (a wall of Gaussians at one depth seen head-on)
 * is slower than with the passes.  hs_counters.reserved[2] counts those ranges; the Python host moves the process to 0
 * when a frame reports any.  mode < 0 only queries.  Returns the setting in force.  HS_DEPTH_SORT=lsd / msd in the
 * environment overrides it per forward. */
HS_API int hs_depth_sort(int mode);

/* Chain positions of the radix passes of HS_STAGE_BIN, process-wide: 0 = blockIdx (default: relies on every XCD handing
 * its share of a grid out in increasing order), 1 = tickets drawn when a block STARTS (+3 % per step at c3; correct under
 * any dispatch order, and the faster setting when SEVERAL PROCESSES run this library on one GPU: two blockIdx-ordered
 * passes of different processes can fill the GPU with blocks that wait for blocks of their own kernel which the other
 * process' waiting blocks keep out; a waiting block then computes the silent predecessor's counts itself -- correct,
 * counted in hs_counters.reserved[4], but slower than never having to).
 * With 1 the pair emission also takes its slot offsets from kernels of their own instead of its in-launch chain.
 * enable < 0 only queries.  Returns the setting in force.  Initial value: HS_SORT_TICKETS=1 in the environment, else 0.
 * The Python host switches to 1 by itself once frames report helps (or, should a pass ever give up: overflow = 2, after
 * which it asks for the step again). */
HS_API int hs_sort_tickets(int enable);

/* Bench/test only: stable LSD radix sort of (u64 key, u32 value) pairs on bits [0, nbits), n < 2^30, using the
 * same pass kernel as HS_STAGE_BIN.  tmp must hold hs_sort_tmp_bytes(n).  Result in keys_out/vals_out.  The u32 at
 * byte 4 of tmp reads 2 afterwards if a pass gave up waiting (results invalid), else 0.  (The tests provoke exactly
 * that with HS_FAULT_INJECT=sort_ticket, which only libhdrsplat_test.so -- built with -DHS_TESTING -- reads.) */
HS_API int64_t hs_sort_tmp_bytes(int64_t n);
HS_API int hs_sort_pairs(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                  int64_t n, int32_t nbits, void* tmp, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* HDRSPLAT_H */
