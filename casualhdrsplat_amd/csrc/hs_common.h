// Internal declarations shared by the HIP translation units of libhdrsplat.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hdrsplat.h"

// gfx950 only (MI355X): the kernels count on 160 KB of LDS per workgroup (the counting tile sort's scatter takes 80 KB of
// dynamic LDS at its admitted maximum), wave64, the DPP / permlane forms and the wait states of this target.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libhdrsplat is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

namespace hs {

void set_error(const char* fmt, ...);

#define HS_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            ::hs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return HS_EHIP;                                                                        \
        }                                                                                          \
    } while (0)
#define HS_LAUNCH_CHECK() HS_HIP_CHECK(hipGetLastError())

// A/B switch: the CRF gradient's first stage at the end of the render backward's launch (render.hip, render_bwd_kernel)
#ifndef HS_TUNE_CRF_IN_RENDER_TAIL
#define HS_TUNE_CRF_IN_RENDER_TAIL 1
#endif

// A/B switch: the tile sort of small frames by counting (binning.hip)
#ifndef HS_TUNE_COUNT_SORT
#define HS_TUNE_COUNT_SORT 1
#endif

constexpr int kTile = HS_TILE;
// Gathered / scattered records occupy one aligned 64-byte memory sector each (kRecF4 float4): a 48-byte record at a
// 48-byte stride straddles two sectors half of the time, which showed up as 2-3x the algorithmic HBM traffic.
constexpr int kRecF4 = 4;                  // float4 per render-record slot (3 used)
constexpr int kPairF4 = 4;                 // float4 per (tile,instance) gradient record slot (10 floats used)
constexpr int kInstF4 = 4;                 // float4 per per-instance sum of the pair records
constexpr int kRecFloats = 4 * kRecF4;
constexpr int kPairFloats = 4 * kPairF4;
constexpr int kInstFloats = 4 * kInstF4;
// Radix passes (binning.hip): 256 threads per block, ITEMS elements per thread; LOOK = status words a thread requests at
// once during the decoupled look-back.  Measured at c3 / c4 (binning stage, ms; profiles/README.md): depth sort with
// 8 / 4 items per thread 0.29 / 0.38 against 0.275 with 16 (c4: 2.09 / 2.41 against 1.85), 16 / 32 words in flight 0.29 /
// 0.31 -- more, smaller blocks lose to their fixed cost (digit totals, block scans), and deeper look-back over-reads;
// pair sort with 12 / 20 / 24 items per thread: 0.226 / 0.232 / 0.249 against 0.225 (three blocks per CU at 24).
constexpr int kSortBlock = 256;
// depth sort (instances by depth, binning.hip "depth sort over the VARYING bits"): 512-thread blocks = 512 bins = digits of
// up to nine bits, eight items per thread -- the same 4096-element tile as the other sorts
constexpr int kDepthBins = 512, kDepthDigitBits = 9;
#ifndef HS_TUNE_DEPTH_ITEMS
#define HS_TUNE_DEPTH_ITEMS 8
#endif
constexpr int kDepthSortItems = HS_TUNE_DEPTH_ITEMS, kDepthSortLook = 8;
constexpr int kPairSortItems = 16, kPairSortLook = 8;                                       // (tile, instance) pairs by tile
constexpr int kU64SortItems = 16;                                                           // hs_sort_pairs
constexpr int kDepthTile = kDepthSortItems * kDepthBins, kPairTile = kPairSortItems * kSortBlock, kU64Tile = kU64SortItems * kSortBlock;
static_assert(kPairTile == kU64Tile && kDepthTile >= kPairTile, "the scratch of every sort is sized for the smallest tile");
constexpr int kSortTileMin = kPairTile;                     // smallest radix tile (elements per block) in use

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// index of highest set bit + 1 (number of bits needed for values < n ... as upstream getHigherMsb)
static inline int tile_bits(uint32_t n) {
    int b = 0;
    while (n) { ++b; n >>= 1; }
    return b;
}

// ---- launchers (each enqueues on `s`, returns HS_OK / HS_EHIP) ----
// `frame_tag`: stamp of this hs_forward call when preprocess and binning run in the same call (else 0: the binning stage
// then clears its own scratch): see kDepthBitsAt
int launch_preprocess_fwd(const hs_fwd_args& a, const hs_layout& L, hipStream_t s, uint32_t frame_tag);
int launch_scan(const hs_fwd_args& a, const hs_layout& L, hipStream_t s);
int launch_cov3d(const hs_fwd_args& a, const hs_layout& L, hipStream_t s);   // inspection: fills hs_layout.cov3D
int launch_binning(const hs_fwd_args& a, const hs_layout& L, hipStream_t s, uint32_t frame_tag);
// an empty cloud (P == 0): cleared counters (when given), cleared tile ranges (when given), the host copy of the counters
int launch_empty_frame(hs_counters* clear, uint2* ranges, int64_t ntiles, uint32_t* counters_host, const hs_counters* counters,
                       hipStream_t s);
// `stats` (device, render_stats_count() u64 counters, or null) selects the diagnostic instantiation of the kernel
int launch_render_fwd(const hs_fwd_args& a, const hs_layout& L, hipStream_t s, unsigned long long* stats = nullptr);
// `crf_in_tail`: the CRF gradient's first stage rides at the END of the launch (workgroups behind the tiles')
int launch_render_bwd(const hs_bwd_args& a, const hs_layout& L, hipStream_t s, unsigned long long* stats = nullptr,
                      unsigned long long* timeline = nullptr, bool crf_in_tail = false);
int render_stats_count();
struct CrfReduce;
// `defer`: non-null = do not launch the second stage; describe it there for the segmented sum's launch to run.
// `in_render_tail`: the first stage already ran at the end of the render backward's launch (two-wave workgroups: the partial
// rows are counted accordingly)
int launch_crf_bwd(const hs_bwd_args& a, const hs_layout& L, hipStream_t s, CrfReduce* defer, bool in_render_tail = false);
int launch_preprocess_bwd(const hs_bwd_args& a, const hs_layout& L, hipStream_t s, bool segsum, bool project,
                          const CrfReduce* crf_reduce);
int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* vis, hipStream_t s);
// spline.hip: camera poses along the trajectory spline with their Jacobian (hs_spline_poses)
int launch_spline_poses(int J, int T, int kind, const float* delta, const float* base, const float* times, float* w2c,
                        float* jac, int* seg, hipStream_t s);
int launch_sh_backward_views(int P, int M, int deg, int V, const float* means3D, const float* camposes,
                             const float* view_colors, float* d_shs, hipStream_t s);

// Scratch of the single-sweep radix passes (binning.hip): digit totals in kGhistCopies copies of [pass <= 8][256], one
// ticket counter per pass (a block's place in the look-back chain), then one status word per (pass, block, digit).
// sort_scratch_words(n, passes, tile) = the words a sort of n elements in blocks of `tile` needs cleared.
constexpr int kGhistCopies = 16;
constexpr int kGhistWords = kGhistCopies * 8 * 256;      // (8 passes x 256 bins, or the depth sort's 4 x 512)
// the ticket row: words [0, 8) one ticket counter per pass; words [kDepthBitsAt, kDepthBitsAt + kDepthBitsWords): OR of the
// visible depth keys and of their complements (the depth sort's digit layout, binning.hip), in kDepthBitsCopies copies of
// two 64-bit words {frame tag << 32 | bits}.  The tag makes the words self-initialising: the forward's first kernel cannot
// count on zeroed memory (it IS the kernel that zeroes the scratch, and a memset node ahead of it costs 4 us), so a
// word whose tag is not this frame's is taken as empty -- by the workgroup that ORs into it (compare-and-swap) and by the
// readers alike.  Stale or garbage bits under a matching tag could only ADD varying bits, i.e. sort a constant bit too:
// never a wrong order.
constexpr int kTicketWords = 128, kDepthBitsAt = 32, kDepthBitsCopies = 16, kDepthBitsWords = 4 * kDepthBitsCopies;
static_assert(kDepthBitsAt >= 8 && kDepthBitsAt % 2 == 0 && kDepthBitsAt + kDepthBitsWords <= kTicketWords, "depth bits live in the ticket row");
__host__ __device__ inline int64_t sweep_pass_words(int64_t nblk, int bins = 256) { return nblk * bins; }
static inline int64_t sort_scratch_words(int64_t n, int passes, int tile, int bins = 256) {
    return n > 0 ? kGhistWords + kTicketWords + (int64_t)passes * sweep_pass_words((n + tile - 1) / tile, bins) : 0;
}
// words the forward's first kernel clears for the depth sort of I instances (4 passes of 512-bin status words)
static inline int64_t depth_scratch_words(int64_t I) { return sort_scratch_words(I, 4, kDepthTile, kDepthBins); }
int64_t sort_tmp_bytes(int64_t n);
// ---- depth sort of frames below 2^21 instances: ONE counting pass over the top bits + range sorts in LDS (binning.hip,
// "depth sort by counting") instead of three or four look-back passes.  hs_layout.depth_ws, in u32 words: per block of
// depth_msd_tile(I) instances one row of kMsdBuckets u16 bucket counts | one row of u32 prefixes down the columns | the
// bucket totals | the culled instances per block.  Every word is written before it is read: nothing to clear.
constexpr int kMsdBits = 12, kMsdBuckets = 1 << kMsdBits;
#ifndef HS_TUNE_MSD_RANGE
#define HS_TUNE_MSD_RANGE 2048
#endif
constexpr int kMsdRange = HS_TUNE_MSD_RANGE, kMsdCap = 4096;   // a range-sort workgroup takes the buckets starting in its 2048 positions; <= 4096 elements stay in LDS
#ifndef HS_TUNE_MSD_SMALL_RANGE
#define HS_TUNE_MSD_SMALL_RANGE 512
#endif
// (small frames: shorter ranges, i.e. more workgroups with less to do each -- 49 workgroups of 2048 instances leave most of the
// GPU idle at BASELINE c2 while each walks its phases alone)
static inline int depth_msd_range(int64_t I) { return I <= (1 << 18) ? HS_TUNE_MSD_SMALL_RANGE : kMsdRange; }
static inline bool depth_msd_fits(int64_t I) { return I > 0 && I < (2 << 20); }
static inline int depth_msd_tile(int64_t I) { return I <= (1 << 18) ? 1024 : 4096; }   // (<= 512 rows either way)
static inline int64_t depth_msd_rows(int64_t I) { return (I + depth_msd_tile(I) - 1) / depth_msd_tile(I); }
static inline int64_t depth_ws_words(int64_t I) {
    return depth_msd_fits(I) ? depth_msd_rows(I) * (kMsdBuckets / 2 + kMsdBuckets + 1) + kMsdBuckets : 0;   // (+ culled per row)
}
// Which depth sort a forward runs: HS_DEPTH_SORT=lsd / msd in the environment forces a form (read at every forward), else
// hs_depth_sort()'s process-wide setting (the host moves it to the passes when a frame's ranges did not fit the LDS), else
// the counting form wherever it fits.
enum DepthSort { kDepthSortLsd = 0, kDepthSortMsd = 1 };
int depth_sort_mode(int64_t I);
int depth_range_cap();   // elements a range-sort workgroup keeps in LDS (kMsdCap; HS_DEPTH_RANGE_CAP lowers it: tests)
int depth_dist_max();    // members of a bucket up to which a range is sorted by distribution (16; HS_DEPTH_DIST_MAX lowers it: tests)
// Scratch behind hs_layout.pair_sort_tmp: one 64-bit status word per 256-instance block of the pair emission's chained scan
// (as u32 words), then the pair sort's scratch.  pair_scratch_words = what must be cleared before the emission runs.
static inline int64_t emit_scan_words(int64_t I) { return 2 * ((I + 255) / 256 + 2) / 64 * 64 + 64; }
static inline int64_t pair_scratch_words(int64_t I, int64_t capacity, int passes) {
    return emit_scan_words(I) + sort_scratch_words(capacity, passes, kPairSortItems * kSortBlock);
}
// Stable LSD radix sort of (u64 key, u32 value) pairs on bits [0,nbits) (hs_sort_pairs; the forward sorts packed
// (u32 key, u32 value) elements with the same pass kernel).  Ping-pongs between (k0,v0) and (k1,v1); the result lands
// in (k0,v0) when sort_passes(nbits) is even, else in (k1,v1).  `n_dev` points at the device-resident element count
// (<= n_launch); `tmp` must hold sort_tmp_bytes(n_launch); *fail_word reads 2 afterwards if a look-back gave up.
int launch_radix_sort(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, const uint32_t* n_dev,
                      int64_t n_launch, int nbits, void* tmp, uint32_t* fail_word, hipStream_t s);
// Test hook, compiled into libhdrsplat_test.so only (-DHS_TESTING; `make test_lib`): HS_FAULT_INJECT in the environment,
// read once.  0 = none; 1 = "sort_ticket": hs_sort_pairs starts its first pass with ticket 1, so chain position 0 never
// publishes and the bounded look-back must give up; 2 = "stalled_chain": the binning stage starts with the verdict of a
// stalled chain (hs_counters.overflow = 2) while the passes are blockIdx-ordered; 3 = "late_block": block 1 of every
// blockIdx-ordered radix pass starts ~3 ms late.  The product library has none of it.
#ifdef HS_TESTING
int fault_injection();
#else
constexpr int fault_injection() { return 0; }
#endif
// HS_SORT_TICKETS=1 in the environment (read once): the pipeline's radix passes take their chain positions from tickets
// instead of blockIdx (binning.hip, "Progress").
bool sort_tickets();
// Whether the pair emission computes its block offsets itself (chained scan) for a frame of I instances: from 2^21
// instances on; HS_SCAN_IN_EMISSION=1 / 0 in the environment forces it on / off (tests run both paths at small sizes).
bool scan_in_emission(int64_t I);
static inline int sort_passes(int nbits) { return (nbits + 7) / 8; }
// Tile sort of a small frame by counting instead of radix passes (binning.hip): the dims qualify when the (pose, tile) keys
// fit the per-workgroup tables and the (emission workgroup x key) matrices stay small; the binning workspace then carries
// the matrices (hs_layout.tile_matrix: counts | bases | totals).  HS_TILE_SORT=radix in the environment keeps the radix
// passes (tests run both paths; read at every forward).
constexpr int kCountTilesMax = 4096;
constexpr int64_t kCountMatrixMax = 1ll << 21;
static inline bool count_sort_fits(int64_t I, int64_t vtiles, int64_t capacity) {
    return capacity > 0 && I > 0 && vtiles <= kCountTilesMax && ((I + 255) / 256) * vtiles <= kCountMatrixMax;
}
static inline int64_t count_matrix_words(int64_t I, int64_t vtiles, int64_t capacity) {
    return count_sort_fits(I, vtiles, capacity) ? 2 * ((I + 255) / 256) * vtiles + vtiles : 0;
}
// HIERARCHICAL tile sort (binning.hip, "coarse stable pass + on-chip expansion"): the instances, walked in depth order, emit
// one (super-tile, instance) element per 8 x 8-tile SUPER-TILE their rectangle meets (carrying the rectangle clipped to
// that super-tile); ONE stable radix pass per eight bits of the super-tile id orders those; workgroups then expand chunks of
// hier_chunk(I) sorted elements into their super-tile's 64 tile lists by counting.  Fits when the (pose, super-tile) keys fit
// the per-workgroup histogram; the binning workspace then carries hs_layout.hier_ws.
constexpr int kSuper = 8;                       // tiles per super-tile side
constexpr int kHierStMax = 2048;                // (pose, super-tile) keys of a frame that qualifies
// Sorted elements per expansion workgroup (four waves x chunk / 4), chosen per frame (hier_chunk): 512 below 2^21
// instances, 1024 from there on -- measured (hier_scatter + hier_count + hier_plan, us): c3 34.9 with 1024, 31.4 with 512,
// 39.2 with 256; c4 178 with 1024, 197 with 512, 272 with 256 (more, smaller chunks cost the plan kernel and the sums over
// earlier chunks what they save in the walk).  The workspace is sized for the smaller one.
constexpr int kHierChunkMin = 512, kHierChunkMax = 1024, kHierRoundsMax = kHierChunkMax / 4 / 64;
static inline int hier_chunk(int64_t I) { return I >= (2ll << 20) ? kHierChunkMax : kHierChunkMin; }
constexpr int kHierCopies = 16;                 // copies of the per-super-tile element counts (same-address atomics)
static inline int64_t hier_super_tiles(int64_t gx, int64_t gy, int64_t n_poses) {
    return ((gx + kSuper - 1) / kSuper) * ((gy + kSuper - 1) / kSuper) * n_poses;
}
static inline bool hier_fits(int64_t I, int64_t gx, int64_t gy, int64_t n_poses, int64_t capacity) {
    return capacity > 0 && I > 0 && hier_super_tiles(gx, gy, n_poses) <= kHierStMax;
}
// hs_layout.hier_ws, in u32 words: header | element counts per super-tile (kHierCopies copies) | pairs per tile -- these
// three are cleared by the stage's first kernel -- | first sorted position of every tile | first element / first chunk of
// every super-tile | one uint4 descriptor per chunk | one row of 64 uint2 (pairs per tile and wave) per chunk
struct HierWs {
    int64_t nst, chunks_max;
    int64_t st_count, tile_total, zero_words, tile_start, coarse_first, chunk_first, desc, counts, words;
    HierWs(int64_t gx, int64_t gy, int64_t n_poses, int64_t capacity) {
        nst = hier_super_tiles(gx, gy, n_poses);
        chunks_max = (capacity + kHierChunkMin - 1) / kHierChunkMin + nst;
        const int64_t nst_pad = (nst + 63) / 64 * 64;
        int64_t o = 64;                                              // header: [0] coarse elements, [1] chunks
        st_count = o; o += kHierCopies * nst_pad;
        tile_total = o; o += nst * 64;
        zero_words = o;
        tile_start = o; o += nst * 64;
        coarse_first = o; o += nst_pad + 64;
        chunk_first = o; o += nst_pad + 64;
        desc = o; o += chunks_max * 4;
        counts = o; o += chunks_max * 64 * 2;
        words = o;
    }
    int64_t nst_pad() const { return (nst + 63) / 64 * 64; }
};
static inline int64_t hier_ws_words(int64_t I, int64_t gx, int64_t gy, int64_t n_poses, int64_t capacity) {
    return hier_fits(I, gx, gy, n_poses, capacity) ? HierWs(gx, gy, n_poses, capacity).words : 0;
}
// Which tile sort a forward of these dims runs: HS_TILE_SORT=radix / count / hier in the environment forces a form where
// the dims allow it (read at every forward: the test suite switches it inside one process); recorded in
// hs_counters.reserved[5] by the binning stage so that later inspection calls need not ask the environment again.
enum TileSort { kTileSortRadix = 0, kTileSortCount = 1, kTileSortHier = 2 };
int tile_sort_mode(int64_t I, int64_t gx, int64_t gy, int64_t n_poses, int64_t capacity);
// keys_sorted of a frame whose pairs were sorted by counting, from its tile ranges (inspection: HS_STAGE_OFFSETS)
int launch_tile_keys(const hs_fwd_args& a, const hs_layout& L, hipStream_t s);

// ---- small device helpers ----
// OR `bits` into a tagged word (see kDepthBitsAt): a word carrying another tag counts as empty
__device__ __forceinline__ void tagged_or(unsigned long long* p, uint32_t tag, uint32_t bits) {
    unsigned long long old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const unsigned long long cur = (uint32_t)(old >> 32) == tag ? old : ((unsigned long long)tag << 32);
        const unsigned long long nw = cur | bits;
        if (nw == old) return;
        const unsigned long long prev = atomicCAS(p, old, nw);
        if (prev == old) return;
        old = prev;
    }
}
// OR of the visible depth keys of a workgroup and of their complements into one of the copies of the depth-bits words (two
// atomics per workgroup; binning.hip derives the depth sort's digit layout from them).  Every thread of the block calls it.
// Sum over the 64 lanes of a wave in a fixed DPP tree; the total is valid in lane 63.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_hi_f32(float v) {
    v += dpp_f32<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);       // row_half_mirror
    v += dpp_f32<0x140>(v);       // row_mirror
    v += dpp_f32<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_f32<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
    return v;
}

// Second stage of the CRF-table / exposure gradient (render.hip, crf_grad_kernel): the partial rows of the pixel blocks
// added in a fixed order, one wave per output element (3K table entries + the exposure): lanes stride over the rows of the
// blocks that worked on that channel -- every pose, every pixel block --, then a fixed DPP tree: reproducible.  Run either
// by crf_reduce_kernel or, when the same hs_backward call goes on to the segmented sum, by the first workgroups of
// pair_segsum_kernel (one launch less on the backward's critical path).
struct CrfReduce {
    const float* partials; int bx, planes, K; float* d_table; float* d_exposure;
    int nblocks;   // 256-thread workgroups the job takes: ceil((3K + 1) / 4); 0 = nothing to do
};
__device__ __forceinline__ void crf_reduce_block(const CrfReduce& c, int block) {
    const int K = c.K;
    const int i = block * 4 + (threadIdx.x >> 6);  // 0 .. 3K: table entry ch * K + k, or 3K = exposure
    const int lane = threadIdx.x & 63;
    if (i > 3 * K) return;
    const bool expo = i == 3 * K;
    const int ch = expo ? 0 : i / K, k = expo ? K : i - ch * K;
    float acc = 0.f;
    // rows of plane p = [p * bx, (p + 1) * bx); plane p carries channel p % 3
    const int nrows = expo ? c.planes * c.bx : (c.planes / 3) * c.bx;
    for (int r = lane; r < nrows; r += 64) {
        const int row = expo ? r : ((r / c.bx) * 3 + ch) * c.bx + (r % c.bx);
        acc += c.partials[(int64_t)row * (K + 1) + k];
    }
    acc = wave_sum_hi_f32(acc);
    if (lane == 63) {
        if (!expo) { if (c.d_table) c.d_table[i] = acc; }
        else if (c.d_exposure) c.d_exposure[0] = acc;
    }
}

// OR over the 64 lanes of a wave, valid in lane 63 (six DPP steps; OR is idempotent, so rows may overlap)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_or_hi(uint32_t v) {
    v |= dpp_u32<0xB1>(v);        // quad_perm [1,0,3,2]
    v |= dpp_u32<0x4E>(v);        // quad_perm [2,3,0,1]
    v |= dpp_u32<0x141>(v);       // row_half_mirror
    v |= dpp_u32<0x140>(v);       // row_mirror
    v |= dpp_u32<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
    v |= dpp_u32<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
    return v;
}
// (`o_in` / `nz_in`: OR of this thread's visible keys / of their complements)
__device__ __forceinline__ void depth_bits_accumulate2(uint32_t o_in, uint32_t nz_in, unsigned long long* bits, uint32_t tag,
                                                       uint32_t* s_two /*LDS[2], zeroed*/) {
    // (`s_two` was zeroed before the workgroup's last barrier.  Twelve DPP instructions per lane, then ONE lane per wave
    // ORs into the two LDS words.  Measured the hard way: every lane ORing into the LDS words itself -- 64 lanes on one
    // address -- took depth_keys_kernel from 2.4 to 26 us and preprocess_fwd from 81 to 88)
    const uint32_t o = wave_or_hi(o_in), nz = wave_or_hi(nz_in);
    if ((threadIdx.x & 63) == 63) { if (o) atomicOr(&s_two[0], o); if (nz) atomicOr(&s_two[1], nz); }
    __syncthreads();
    if (threadIdx.x < 2 && s_two[threadIdx.x])
        tagged_or(bits + 2 * (blockIdx.x % kDepthBitsCopies) + threadIdx.x, tag, s_two[threadIdx.x]);
}
__device__ __forceinline__ void depth_bits_accumulate(uint32_t key, bool visible, unsigned long long* bits, uint32_t tag,
                                                      uint32_t* s_two /*LDS[2], zeroed*/) {
    depth_bits_accumulate2(visible ? key : 0u, visible ? ~key : 0u, bits, tag, s_two);
}
// d colour / d s of the radiance activation, from the stored colour (and the clamp bit for relu_shift)
__device__ __forceinline__ float radiance_dact(int act, float col, bool was_clamped) {
    if (act == 1) return col;
    if (act == 2) return -expm1f(-col);  // sigmoid(s) = 1 - e^-softplus(s); expm1: exact also where col < 6e-8 (dark Gaussians)
    return was_clamped ? 0.f : 1.f;
}

// Per-instance sum of the render-backward pair records (preprocess.hip, pair_segsum_kernel).  Thread `t` of the job: instance t >> 2 (in depth order), quad lane t & 3.
struct SegsumArgs {
    int64_t I;
    const uint32_t* inst_sorted; const uint32_t* offs_sorted;
    const float4* pair_grads; const uint8_t* pair_flags; float4* inst_grads; const hs_counters* counters;
    const int* radii_inst; const uint8_t* clamped; float* view_colors; const float4* rec; int act;
};
__device__ __forceinline__ void pair_segsum_body(const SegsumArgs& a, int64_t t) {
    // four lanes (one DPP quad) per instance: lane q adds records beg+q, beg+q+4, ...; the four partial sums are
    // combined in a fixed butterfly, so the result does not depend on timing.  Quads shorten the longest run in a
    // wave fourfold (run lengths are heavy-tailed) and make neighbouring lanes read neighbouring records.
    const int64_t i = t >> 2;
    const int q = (int)(t & 3);
    // binning overflow (sync-free mode): nothing was emitted or rendered, and slots past the capacity do not exist
    const bool valid = i < a.I && !a.counters->overflow;
    const uint32_t beg = valid ? (i == 0 ? 0u : a.offs_sorted[i - 1]) : 0u;
    const uint32_t end = valid ? a.offs_sorted[i] : 0u;
    float r[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // two slots per trip: both flags first, then both records, so two record reads are in flight per lane; the
    // sums keep the slot order (first slot added before the second)
    for (uint32_t s = beg + q; s < end; s += 8) {
        const bool two = s + 4 < end;
        const bool f0 = a.pair_flags[s] != 0;  // unflagged: never written this backward (~55 % of pairs), not read
        const bool f1 = two && a.pair_flags[s + 4] != 0;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, b0 = a0, b1 = a0;
        float2 a2 = make_float2(0.f, 0.f), b2 = a2;
        if (f0) {
            a0 = a.pair_grads[kPairF4 * (int64_t)s + 0];
            a1 = a.pair_grads[kPairF4 * (int64_t)s + 1];
            a2 = reinterpret_cast<const float2*>(a.pair_grads + kPairF4 * (int64_t)s + 2)[0];
        }
        if (f1) {
            b0 = a.pair_grads[kPairF4 * (int64_t)(s + 4) + 0];
            b1 = a.pair_grads[kPairF4 * (int64_t)(s + 4) + 1];
            b2 = reinterpret_cast<const float2*>(a.pair_grads + kPairF4 * (int64_t)(s + 4) + 2)[0];
        }
        r[0] += a0.x; r[1] += a0.y; r[2] += a0.z; r[3] += a0.w;
        r[4] += a1.x; r[5] += a1.y; r[6] += a1.z; r[7] += a1.w;
        r[8] += a2.x; r[9] += a2.y;
        r[0] += b0.x; r[1] += b0.y; r[2] += b0.z; r[3] += b0.w;
        r[4] += b1.x; r[5] += b1.y; r[6] += b1.z; r[7] += b1.w;
        r[8] += b2.x; r[9] += b2.y;
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        r[k] += __shfl_xor(r[k], 1);
        r[k] += __shfl_xor(r[k], 2);
    }
    if (i < a.I && q == 0) {
        // (a depth sort that gave up -- overflow = 2 -- left no instance list: zeros go to row i, any bijection will do)
        const int64_t inst = a.counters->overflow >= 2u ? i : (int64_t)a.inst_sorted[i];
        float4* o = a.inst_grads + kInstF4 * inst;
        o[0] = make_float4(r[0], r[1], r[2], r[3]);
        o[1] = make_float4(r[4], r[5], r[6], r[7]);
        o[2] = make_float4(r[8], r[9], 0.f, 0.f);
        if constexpr (kInstF4 == 4) o[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.view_colors) {  // colour gradient of the instance after the SH clamp mask (hs_sh_backward_views input)
            const uint8_t cl = a.clamped[inst];
            const bool on = a.radii_inst[inst] > 0;
            float col[3] = {0.f, 0.f, 0.f};
            if (a.act != 0 && on) {  // exp / softplus: the factor comes from the stored colour
                const float4 rb = a.rec[kRecF4 * inst + 1];
                col[0] = rb.z; col[1] = rb.w; col[2] = reinterpret_cast<const float*>(a.rec + kRecF4 * inst + 2)[0];
            }
            a.view_colors[3 * inst + 0] = on ? radiance_dact(a.act, col[0], cl & 1) * r[6] : 0.f;
            a.view_colors[3 * inst + 1] = on ? radiance_dact(a.act, col[1], cl & 2) * r[7] : 0.f;
            a.view_colors[3 * inst + 2] = on ? radiance_dact(a.act, col[2], cl & 4) * r[8] : 0.f;
        }
    }
}

__device__ __forceinline__ float xform_row(const float* m, int i, float x, float y, float z) {
    return ((m[i] * x + m[4 + i] * y) + m[8 + i] * z) + m[12 + i];
}

}  // namespace hs
