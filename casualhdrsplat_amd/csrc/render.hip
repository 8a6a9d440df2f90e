// Per-tile alpha-blend forward (a9) and per-pixel backward (a10) for gfx950, plus the HDR
// epilogue / prologue and N-pose resolve (a15).  Rules: SURVEY.md 8(a); no reference code exists.
//
// CDNA4 design (not a translation of the CUDA 16x16-thread tile).  Measured on MI355X the loops are bound by
// VALU issue (~4 cycles per wave64 VALU instruction, profiles/r01b_valu_rate.txt), so the design minimises vector
// instructions per (pixel, Gaussian) pair:
//  * one 128-thread workgroup (two wave64) per 16x16 binning tile; each wave owns a 16x8 half tile and each lane
//    TWO vertically adjacent pixels, so dx and every per-Gaussian term is shared by the pair and the rest is
//    packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32);
//  * the tile's sorted instance list is staged 128 entries at a time into LDS (one gather of a 64-byte-aligned record
//    per thread, the next batch prefetched into registers), the conic pre-scaled by -0.5*log2(e) so the inner
//    loop is  dx, dy -> two FMAs -> v_exp_f32;
//  * the two 32-lane halves of a wave ("lane groups") each own an 8 x 8 block of the half tile; before touching a batch
//    each wave tests the staged Gaussians against its two blocks, one Gaussian per lane, and compacts the survivors into
//    one LDS list per group (ballot + mbcnt); the compositing loop walks the two lists in lock step, each group reading
//    its own entry (ds_read_b128, two addresses per wave) -- one trip serves up to two different entries;
//  * workgroups map to tiles through an XCD strip permutation (two-tile-row strips dealt round-robin to the XCDs) so
//    that the tiles one XCD works on are neighbours and share its L2;
//  * the forward records which staged entries each lane group of each wave actually took (one byte per sorted pair) and
//    how many trips each tile cost; the backward's lane groups replay exactly those entries, each from its own list;
//  * backward: no global atomics.  Per (lane group, Gaussian) nine partial sums are formed in-lane over the pixel
//    pair, reduced across the 32 lanes of the group by halving steps ordered by price (bank-masked DPP adds first, one
//    v_permlane16_swap / ds_bpermute step, quad steps last), STORED into the group's own plane of the entry's LDS
//    record (a (wave, group) visits an entry once: no LDS atomics), the planes added in a fixed order and written as
//    ONE record (a 64-byte sector) per (tile, instance) pair at the pair's duplicateWithKeys slot; preprocess-backward
//    then sums each instance's contiguous slots.  Gradients are bitwise reproducible run to run;
//  * the backward's launch ends on light tiles taken from a queue: each XCD keeps its strip-ordered tiles except its
//    lightest 8 % (order_tiles_kernel, from the forward's trip counts), which the last workgroups of the launch take
//    one after the other -- faster XCDs take more, and the tail is one short tile long.
//
// This TU is compiled with FMA contraction on; every integer decision it makes (pair slot addressing) uses
// add/div-only expressions that contraction cannot change.
#include "hs_common.h"

namespace hs {

namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTmin = 0.0001f;
constexpr float kLogEps = 1e-8f;


// Diagnostic counters of the STATS instantiations (hs_render_stats: bench.py's lane-utilisation / VALU-roofline
// leg and profiling; never part of a timed or differentiated call).  Each wave keeps its counts in registers and adds
// them once at the end.
enum {
    kStBwdTrips = 0,      // (wave, entry) trips of the backward replay loop
    kStBwdEmpty = 1,      // ... in which no lane had an active pixel
    kStBwdActivePix = 2,  // sum over trips of active pixels (<= 128 per trip)
    kStBwdCulled = 3,     // staged entries rejected by the half-tile test (per wave)
    kStBwdHist = 4,       // [4..9] trips by active LANES: 0, 1-4, 5-8, 9-16, 17-32, 33-64
    kStBwdStaged = 10,    // entries staged into LDS (per workgroup)
    kStBwdBatches = 11,   // staging batches (per workgroup)
    kStFwdTrips = 12, kStFwdEmpty = 13, kStFwdActivePix = 14, kStFwdCulled = 15, kStFwdStaged = 16, kStFwdBatches = 17,
    kStCount = 24
};
struct WaveStats {
    unsigned long long v[kStCount];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < kStCount; ++i) v[i] = 0;
    }
    __device__ __forceinline__ void flush(unsigned long long* dst) const {
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < kStCount; ++i)
                if (v[i]) atomicAdd(dst + i, v[i]);
        }
    }
};

// ---- DPP cross-lane helpers (wave64) ----
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// Sum over the 64 lanes; the total is valid in lanes 48..63 (lane 63 is used).
__device__ __forceinline__ float wave_sum_hi(float v) {
    v += dpp<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp<0x141>(v);       // row_half_mirror
    v += dpp<0x140>(v);       // row_mirror
    v += dpp<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
    v += dpp<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
    return v;
}
// ---- the 9 (10) value reduction of the backward replay (over the 32 lanes of a lane group) ----
// Every halving step sums TWO registers over one lane bit into ONE register (x's partial sums land in the lanes
// whose bit is 0, y's in the lanes whose bit is 1), so the register count shrinks 10 -> 5 -> 3 -> 2.  The steps
// differ in price -- a bank-masked v_add_f32_dpp issues in 4 cycles, a v_permlane16_swap in 8 (plus its add) -- so
// the cheap in-row steps (lane bits 3 and 2) run FIRST, while there are many registers, and the swap (bit 4) last,
// on the few that remain.
__device__ __forceinline__ float halve16(float x, float y) {  // lane bit 4 (x -> even rows, y -> odd rows)
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// In-row steps as bank-masked DPP adds (the compiler's DPP combiner only folds full-mask moves, so these are written
// out).  pair8(x, y): lanes with (lane & 8) == 0 get x[l] + x[l+8], the others y[l] + y[l-8]; pair4 likewise on lane
// bit 2.  A VALU write followed by a DPP read of the same register needs two wait states (hipcc pads nothing inside
// asm): the ten inputs are written by ordinary code just before, hence the leading s_nop; inside the block every
// register is read at least two instructions after it was written.  The trailing s_nop covers the v_permlane16_swap
// the compiler places behind the block.
// Out: w0 = {g0, g2, g1, g3} by quad, w1 = {g4, g6, g5, g7} by quad, w2 = {g8, g8, g9, g9} by quad -- each the sum over
// lane bits 3 and 2, i.e. over the four quads of the row.
template <bool TEN>
__device__ __forceinline__ void reduce_in_rows(const float* g, float g9, float& w0, float& w1, float& w2) {
    float z0, z1, z2, z3, z4;
    if constexpr (TEN) {
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %3, %9, %9 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %8, %8 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %4, %11, %11 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %4, %10, %10 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %5, %13, %13 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %5, %12, %12 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %6, %15, %15 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %6, %14, %14 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %7, %17, %17 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %7, %16, %16 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %0, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %2, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %2, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "s_nop 1"
            : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(z0), "=&v"(z1), "=&v"(z2), "=&v"(z3), "=&v"(z4)
            : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(g[8]), "v"(g9));
    } else {
        // nine values: the odd one is summed over bits 3 and 2 by two full-row rotations (every lane gets the total)
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %3, %9, %9 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %8, %8 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %4, %11, %11 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %4, %10, %10 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %5, %13, %13 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %5, %12, %12 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %6, %15, %15 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %6, %14, %14 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %7, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %0, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %2, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1"
            : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(z0), "=&v"(z1), "=&v"(z2), "=&v"(z3), "=&v"(z4)
            : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(g[8]));
    }
}
// The two quad steps (lane bits 1 and 0): every lane of a quad ends with the quad's total.
__device__ __forceinline__ float reduce_in_quads(float t) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(t));
    return t;
}

// Nine (ten) values reduced over each 32-lane half of the wave separately (the lane groups of the render backward): the
// in-row steps, then ONE swap step over lane bit 4 for two of the three registers, the third through the LDS crossbar
// (`xor16_addr` = ((lane ^ 16) << 2): an LDS-pipe instruction plus one vector add instead of copy + swap + add -- the loop
// is bound by vector issue, the LDS pipe is not), then the quad steps on both results -- lane bit 5 is never crossed.
// Out, per group (rows 2g and 2g + 1 of the wave), all four lanes of a quad alike:
//   t0: row 2g, quads 0..3 = g0, g2, g1, g3 ; row 2g + 1, quads 0..3 = g4, g6, g5, g7      t1: every row, quads = g8, g8, g9, g9.
template <bool TEN>
__device__ __forceinline__ void group_reduce(const float* g, float g9, int xor16_addr, float& t0, float& t1) {
    float w0, w1, w2;
    reduce_in_rows<TEN>(g, g9, w0, w1, w2);
    const float u0 = halve16(w0, w1);  // even rows: w0 over bit 4, odd rows: w1 over bit 4
    const float u1 = w2 + __int_as_float(__builtin_amdgcn_ds_bpermute(xor16_addr, __float_as_int(w2)));
    t0 = reduce_in_quads(u0);
    t1 = reduce_in_quads(u1);
}

// number of set bits of a wave-uniform 64-bit mask below this lane
__device__ __forceinline__ int mask_prefix(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return v;
}

struct Crf {
    const float* table;  // [3,K]
    int K;
    float umin, umax, dt;
};

// x = H*dt ; u = ln(max(x,eps)) ; s = clamp((u-umin)/(umax-umin)*(K-1)) ; piecewise-linear lookup.
__device__ __forceinline__ void crf_locate(const Crf& c, float Hv, int& i, float& f, float& xv, bool& interior) {
    xv = Hv * c.dt;
    const float u = __logf(fmaxf(xv, kLogEps));
    float s = (u - c.umin) / (c.umax - c.umin) * (float)(c.K - 1);
    bool in = true;
    if (!(s > 0.f)) { s = 0.f; in = false; }
    if (s >= (float)(c.K - 1)) { s = (float)(c.K - 1); in = false; }
    i = (int)floorf(s);
    if (i > c.K - 2) i = c.K - 2;
    f = s - (float)i;
    interior = in && (xv > kLogEps);
}
__device__ __forceinline__ float crf_eval(const Crf& c, int ch, float Hv) {
    int i; float f, xv; bool in;
    crf_locate(c, Hv, i, f, xv, in);
    const float* t = c.table + ch * c.K;
    return t[i] * (1.f - f) + t[i + 1] * f;
}
// dL/dH given dL/dLDR
__device__ __forceinline__ float crf_grad_H(const Crf& c, int ch, float Hv, float g) {
    int i; float f, xv; bool in;
    crf_locate(c, Hv, i, f, xv, in);
    if (!in) return 0.f;
    const float* t = c.table + ch * c.K;
    const float scale = (float)(c.K - 1) / (c.umax - c.umin);
    return g * (t[i + 1] - t[i]) * scale / xv * c.dt;
}

constexpr float kLog2e = 1.4426950408889634f;

typedef float f2 __attribute__((ext_vector_type(2)));

// acc + splat(pair.y) * w and splat(pair.y) * w as ONE packed instruction (op_sel picks the pair's upper half for both
// result lanes): the compiler broadcasts a LOWER half this way by itself but copies an upper half into a fresh register
// first -- one v_mov per trip of both compositing loops (the green channel sits in the upper half of its LDS pair).
#ifndef HS_NO_OPSEL_ASM
__device__ __forceinline__ f2 pk_fma_hi(f2 pair, f2 w, f2 acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(pair), "v"(w));
    return acc;
}
__device__ __forceinline__ f2 pk_mul_hi(f2 pair, f2 w) {
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(pair), "v"(w));
    return r;
}
#else
__device__ __forceinline__ f2 pk_fma_hi(f2 pair, f2 w, f2 acc) { return acc + f2{pair.y, pair.y} * w; }
__device__ __forceinline__ f2 pk_mul_hi(f2 pair, f2 w) { return f2{pair.y, pair.y} * w; }
#endif

// Staged entry as the inner loops want it: a = {x, y, A2, B2}, b = {C2, opacity, r, g} with the conic
// pre-scaled so that log2(G) = dx*(A2*dx + B2*dy) + C2*dy*dy :  A2 = -0.5*A*log2e, B2 = -B*log2e, C2 = -0.5*C*log2e.
__device__ __forceinline__ void scale_entry(float4& a, float4& b) {
    a.z *= -0.5f * kLog2e; a.w *= -kLog2e; b.x *= -0.5f * kLog2e;
}

// One Gaussian per lane against a block of pixels [sx, sx + WIDTH - 1] x [sy, sy + 7] (pixel centres).
// Returns false only when NO point of that rectangle can pass `alpha >= 1/255`, i.e. when the maximum over the
// rectangle of  log2 G(d) = dx (A2 dx + B2 dy) + C2 dy^2  (a concave quadratic centred on the Gaussian) stays below
// log2(1/(255 o)).  For a centre outside the rectangle the maximum lies on one of the two edges facing it; both
// candidates below (optimum along the nearest vertical line, optimum along the nearest horizontal line, each
// clamped to the rectangle) are points of the rectangle and one of them is the constrained optimum, so the test is
// exact up to rounding.  The threshold carries an absolute margin of 0.05 in log2 units (+1e-4 relative), orders
// of magnitude above the fp32 evaluation error of the inner loops for variance ratios up to ~1e5, so culling
// never changes a result.
template <int WIDTH>   // rectangle [sx, sx + WIDTH - 1] x [sy, sy + 7]: 16 = a half tile, 8 = the 8 x 8 block of a lane group
__device__ __forceinline__ bool block_may_touch(const float4 a, const float4 b, float sx, float sy) {
    const float A2 = a.z, B2 = a.w, C2 = b.x, o = b.y;
    if (!(o >= kAlphaMin)) return false;
    if (!(A2 < 0.f && C2 < 0.f && 4.f * A2 * C2 - B2 * B2 > 0.f)) return true;  // not positive definite: no culling
    const float thr = -(__log2f(255.f * o) * 1.0001f + 0.05f);
    // rectangle in d = g - pixel coordinates: dx in [a.x - (sx + WIDTH - 1), a.x - sx], dy in [a.y - (sy+7), a.y - sy]
    const float dx_lo = a.x - (sx + (float)(WIDTH - 1)), dx_hi = a.x - sx, dy_lo = a.y - (sy + 7.f), dy_hi = a.y - sy;
    const float dx_e = fminf(fmaxf(0.f, dx_lo), dx_hi);  // nearest rectangle x to the centre (0 if inside)
    const float dy_e = fminf(fmaxf(0.f, dy_lo), dy_hi);
    const float dy_s = fminf(fmaxf(-0.5f * B2 * dx_e / C2, dy_lo), dy_hi);
    const float dx_s = fminf(fmaxf(-0.5f * B2 * dy_e / A2, dx_lo), dx_hi);
    const float p1 = dx_e * (A2 * dx_e + B2 * dy_s) + C2 * dy_s * dy_s;
    const float p2 = dx_s * (A2 * dx_s + B2 * dy_e) + C2 * dy_e * dy_e;
    return fmaxf(p1, p2) >= thr;
}

// Lane -> pixel pair of a wave's half tile (16 x 8 pixels at (sx, sy)): the lane owns (px, py0) and (px, py0 + 1).
// The two 32-lane halves of a wave ("lane groups") each own an 8 x 8 block -- group g = lane >> 5 the columns
// 8g .. 8g + 7 -- as 8 columns x 4 row pairs: the render backward lets each group walk its OWN list of takers (a
// Gaussian's alpha >= 1/255 footprint covers about a third of a half tile, and a compact block is missed more often
// than a 16 x 4 stripe: scripts/sim_lane_groups.py), and the forward records the takers per group.
__device__ __forceinline__ void lane_pixels(int lane, int sx, int sy, int& px, int& py0) {
    px = sx + 8 * (lane >> 5) + (lane & 7);
    py0 = sy + 2 * ((lane >> 3) & 3);
}

struct RenderFwd {
    int W, H, gx, ntiles, N, flags;
    const uint2* ranges; const uint32_t* point_list; const float4* rec; const float* bg;
    float* out_color; float* out_hdr; float* final_T; uint32_t* n_contrib; float* pose_hdr;
    float* out_invdepth;  // [N,H,W] or null
    Crf crf;
    const float* exposure;
    unsigned long long* stats;  // STATS instantiations only
    unsigned long long* timeline;
    uint8_t* pair_act;          // out: per sorted pair, bit 2g + w = some pixel of lane group g of wave w took the entry
    uint32_t* tile_work;        // out: per (pose, tile), the (half tile, entry) trips that found a taker
};

constexpr int kBatch = 128;  // staged entries per trip = threads per workgroup

// Workgroups are dealt to the 8 XCDs round-robin by blockIdx, and each XCD has its own L2.  The block -> tile map
// below gives XCD x the two-tile-row strips x, x + 8, x + 16, ... of the frame (all poses stacked): the tiles an XCD
// works on at the same time are neighbours on screen, so the records of the Gaussians they share are gathered
// through one L2, while every XCD still samples the whole image (a contiguous band per XCD would tie the launch
// time to the densest band of a real scene).
// Construction: list the strips class by class (class = strip % 8), tiles row-major inside a strip; XCD x takes the
// x-th of 8 contiguous, equally long pieces of that list -- a bijection for any grid, piece boundaries fall within
// a strip of the next class at worst.
__device__ __forceinline__ int xcd_strip_tile(int b, int nb, int gx) {
    constexpr int kXcd = 8, kRows = 2;
    const int x = b % kXcd, k = b / kXcd;
    const int per = nb / kXcd, rem = nb % kXcd;
    int p = x * per + min(x, rem) + k;               // position in the class-ordered list
    const int rows = nb / gx;                        // tile rows of all poses (nb = gx * gy * N)
    const int nstrips = (rows + kRows - 1) / kRows;
    const int strip_tiles = kRows * gx;
    const int last_short = nstrips * strip_tiles - nb;  // tiles missing from the last strip
    int c = 0;
    for (; c < kXcd - 1; ++c) {
        const int ns = (nstrips - c + kXcd - 1) / kXcd;  // strips of class c
        const int nt = ns * strip_tiles - (((nstrips - 1) % kXcd) == c ? last_short : 0);
        if (p < nt) break;
        p -= nt;
    }
    const int s_local = p / strip_tiles, r = p - s_local * strip_tiles;
    return (c + kXcd * s_local) * strip_tiles + r;   // strips are contiguous runs of kRows * gx tiles
}

__device__ __forceinline__ float hs_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Per-pixel compositing state of the forward (two of these per lane).
struct PixF {
    float T, C0, C1, C2, D;
    uint32_t last;
};
// LLVM floating-point predicates for __builtin_amdgcn_fcmpf (returns the 64-bit lane mask of the comparison)
constexpr int kFcmpOGE = 3, kFcmpOLE = 5;
// One compositing step for the two pixels of a lane (packed state: two-wide instructions where they exist).  The
// per-pixel "finished" flags of the wave live in scalar register pairs (`done0/1`), the skip / terminate decisions
// are three vector compares per pixel plus scalar mask algebra, and the selects take the masks back through
// inverse_ballot: no vector instruction is spent on flag bookkeeping (the loop is VALU-issue bound).
// DEPTH: also accumulate the expected inverse depth sum alpha T / z.
struct PairF {
    f2 T, C0, C1, C2, D;
    uint32_t last0, last1;
};
// Returns the lane mask of the lanes in which at least one of the two pixels took the entry.
template <bool DEPTH>
__device__ __forceinline__ uint64_t blend_fwd_pair(PairF& s, uint64_t& done0, uint64_t& done1, f2 pw, f2 alpha, float r,
                                                   float g, float b, float invd, uint32_t idx1, int* n_pixels = nullptr) {
    const uint64_t valid0 = ~done0 & __builtin_amdgcn_fcmpf(pw.x, 0.f, kFcmpOLE) &
                            __builtin_amdgcn_fcmpf(alpha.x, kAlphaMin, kFcmpOGE);
    const uint64_t valid1 = ~done1 & __builtin_amdgcn_fcmpf(pw.y, 0.f, kFcmpOLE) &
                            __builtin_amdgcn_fcmpf(alpha.y, kAlphaMin, kFcmpOGE);
    const f2 one = {1.f, 1.f};
    const f2 test_T = s.T * (one - alpha);
    const uint64_t cont0 = valid0 & __builtin_amdgcn_fcmpf(test_T.x, kTmin, kFcmpOGE);
    const uint64_t cont1 = valid1 & __builtin_amdgcn_fcmpf(test_T.y, kTmin, kFcmpOGE);
    done0 |= valid0 ^ cont0;  // valid, but the transmittance would drop below 1e-4: the pixel terminates here
    done1 |= valid1 ^ cont1;
    const bool upd0 = __builtin_amdgcn_inverse_ballot_w64(cont0), upd1 = __builtin_amdgcn_inverse_ballot_w64(cont1);
    const f2 aT = alpha * s.T;
    const f2 w = {upd0 ? aT.x : 0.f, upd1 ? aT.y : 0.f};
    const f2 rr = {r, r}, bb = {b, b};
    s.C0 += rr * w; s.C1 = pk_fma_hi(f2{r, g}, w, s.C1); s.C2 += bb * w;
    if constexpr (DEPTH) { const f2 dd = {invd, invd}; s.D += dd * w; }
    s.T = f2{upd0 ? test_T.x : s.T.x, upd1 ? test_T.y : s.T.y};
    s.last0 = upd0 ? idx1 : s.last0;
    s.last1 = upd1 ? idx1 : s.last1;
    if (n_pixels) *n_pixels = __popcll(cont0) + __popcll(cont1);
    return cont0 | cont1;
}

__device__ __forceinline__ void write_pixel_fwd(const RenderFwd& p, const PixF& st, int pose, int px, int py) {
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t pix = (int64_t)py * p.W + px;
    p.final_T[(int64_t)pose * HW + pix] = st.T;
    p.n_contrib[(int64_t)pose * HW + pix] = st.last;
    const float H0 = st.C0 + st.T * p.bg[0], H1 = st.C1 + st.T * p.bg[1], H2 = st.C2 + st.T * p.bg[2];
    const bool hdr = p.flags & HS_FLAG_HDR;
    if (p.pose_hdr) {
        float* ph = p.pose_hdr + (int64_t)pose * 3 * HW;
        ph[pix] = H0; ph[HW + pix] = H1; ph[2 * HW + pix] = H2;
    }
    if (p.N == 1) {
        if (hdr) {
            Crf c = p.crf;
            c.dt = p.exposure[0];
            if (p.out_hdr) { p.out_hdr[pix] = H0; p.out_hdr[HW + pix] = H1; p.out_hdr[2 * HW + pix] = H2; }
            p.out_color[pix] = crf_eval(c, 0, H0);
            p.out_color[HW + pix] = crf_eval(c, 1, H1);
            p.out_color[2 * HW + pix] = crf_eval(c, 2, H2);
        } else {
            p.out_color[pix] = H0; p.out_color[HW + pix] = H1; p.out_color[2 * HW + pix] = H2;
        }
    }
}

// LDS record of one staged entry in the forward: {x, y, A2, B2 | C2, opacity, r, g | b, 1/depth, -, -} -- 48 bytes, so
// one (scalar-computed) byte offset addresses all of it.
constexpr int kFwdEntF = 12;

// Seven waves per SIMD (72 registers).  Round 2's one-list kernel fitted 64 registers / eight waves without a spill (-6 us);
// with the two lane groups the same request spills eight registers per lane around the batch loop -- 33 MB written and
// read back per frame, visible as +30 % L2-fabric traffic -- and is no faster (0.248-0.250 ms at c3 against 0.245-0.247 at
// seven waves; c4 1.648 vs 1.614; six waves: 0.251).  (The inverse-depth and diagnostic instantiations keep what they get.)
template <bool DEPTH, bool STATS>
__global__ void __launch_bounds__(kBatch) __attribute__((amdgpu_waves_per_eu((DEPTH || STATS) ? 1 : 7)))
render_fwd_kernel(RenderFwd p) {
    constexpr int KB = kBatch;
    constexpr int kEnt = kFwdEntF * 4;          // bytes per staged record
    constexpr int kSentinel = KB * kEnt;        // record KB: opacity 0, nobody takes it (pads the shorter list of a wave)
    constexpr int kListLen = KB + 8;
    __shared__ __attribute__((aligned(16))) float s_ent[(KB + 1) * kFwdEntF];
    __shared__ int s_alive[2][2];
    // per (wave, lane group): compacted list of the staged entries that can touch the group's 8 x 8 block (byte offsets)
    __shared__ uint16_t s_list[2][2][kListLen];
    __shared__ uint8_t s_taken[2][2][KB];  // [wave][group][entry]: bit 2g + wave if a pixel of the group took the entry
    __shared__ uint32_t s_work[2];         // per wave: (group, entry) takers = what the backward's lists will hold

    const int vt = xcd_strip_tile(blockIdx.x, gridDim.x, p.gx);  // virtual tile = pose * ntiles + tile
    const int pose = vt / p.ntiles;
    const int tile = vt - pose * p.ntiles;
    const int tx = tile % p.gx, ty = tile / p.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 5;
    const int sx = tx * kTile, sy = ty * kTile + wave * 8;
    int px, py0;
    lane_pixels(lane, sx, sy, px, py0);
    const int py1 = py0 + 1;
    const bool in0 = px < p.W && py0 < p.H, in1 = px < p.W && py1 < p.H;
    const float pxf = (float)px;
    const f2 pyf = {(float)py0, (float)py1};
    const float sxf = (float)sx, syf = (float)sy;

    const uint2 range = p.ranges[vt];
    const int n = (int)(range.y - range.x);

    PairF ps;
    ps.T = f2{1.f, 1.f};
    ps.C0 = ps.C1 = ps.C2 = ps.D = f2{0.f, 0.f};
    ps.last0 = ps.last1 = 0;
    // lane masks of finished pixels (all 64 lanes of both waves run the whole kernel: exec is full)
    uint64_t done0 = __builtin_amdgcn_ballot_w64(!in0), done1 = __builtin_amdgcn_ballot_w64(!in1);

    // software pipeline of the staging gather: registers hold the NEXT batch while the current one is processed
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    float rcb = 0.f, rdepth = 1.f;
    if ((int)threadIdx.x < n) {
        const uint32_t id = p.point_list[range.x + threadIdx.x];
        const float4* r = p.rec + kRecF4 * (int64_t)id;
        ra = r[0]; rb = r[1]; rcb = reinterpret_cast<const float*>(r + 2)[0];
        if constexpr (DEPTH) rdepth = reinterpret_cast<const float*>(r + 2)[1];
    }
    WaveStats ws;
    if constexpr (STATS) ws.clear();
    const char* const ent = reinterpret_cast<const char*>(s_ent);
    float* const my_ent = s_ent + threadIdx.x * kFwdEntF;
    const uint16_t* const my_list = s_list[wave][grp];
    if (threadIdx.x < kFwdEntF) s_ent[KB * kFwdEntF + threadIdx.x] = 0.f;   // the sentinel (ordered by the loop's barriers)
    // The activity bits of a batch -- which of its entries found a taker in which lane group of which wave -- go to
    // `pair_act`, one byte per sorted pair (bit 2g + w): the backward's lane groups walk exactly those entries instead of
    // testing every staged entry again, and an entry nobody took is neither gathered nor written out.
    auto flush_activity = [&](int base_prev, int cnt_prev) {
        const int t = threadIdx.x;
        if (t < cnt_prev)
            p.pair_act[(int64_t)range.x + base_prev + t] = (uint8_t)(s_taken[0][0][t] | s_taken[0][1][t] | s_taken[1][0][t] | s_taken[1][1][t]);
        s_taken[0][0][t] = 0; s_taken[0][1][t] = 0; s_taken[1][0][t] = 0; s_taken[1][1][t] = 0;   // for the batch about to be staged
    };
    // Walks list positions [i0, i1) of the wave's two lists in lock step, front to back: the 32 lanes of group g read
    // entry i of THEIR list (an 8 x 8 block is missed by many Gaussians the whole half tile is not: one trip serves up to
    // two different entries); returns, in lane i - i0, which groups found a taker at position i (bit 0: group 0, bit 2:
    // group 1).  The lists hold LDS byte offsets, so the loop spends no vector instruction on address arithmetic; the
    // contributor number is kept scaled the same way (`last` = (index + 1) * 48, divided once at the end).
    auto walk = [&](int i0, int i1, int base48) -> uint32_t {
        uint32_t act_lo = 0u, act_hi = 0u;                  // lane i - i0: the `took` mask of position i (two halves = two groups)
        uint32_t pos = 0u;                                  // i - i0, a scalar register by construction (asm below)
        const uint32_t npos = (uint32_t)(i1 - i0);
        const uint16_t* lp = my_list + i0;
        auto trip = [&]() {
            const int jb = (int)*lp++;                      // one address per lane group
            const char* e = ent + jb;
            const float4 a = *reinterpret_cast<const float4*>(e);
            const float4 b = *reinterpret_cast<const float4*>(e + 16);
            const float2 c = *reinterpret_cast<const float2*>(e + 32);
            const float dx = a.x - pxf;
            const f2 dy = a.y - pyf;
            const float t = a.z * dx * dx, u = a.w * dx;
            const f2 pw = dy * (b.x * dy + u) + t;          // log2 of the Gaussian falloff at the two pixels
            const float al0 = fminf(kAlphaMax, b.y * hs_exp2(pw.x));
            const float al1 = fminf(kAlphaMax, b.y * hs_exp2(pw.y));
            // (keeping the index batch-relative and adding the base once per batch saves this add -- 33.5 instead of 34.5
            // vector instructions per trip -- but its two extra registers spill around the batch loop: +26 MB of scratch
            // traffic per frame and not a microsecond gained; rejected)
            const uint32_t idx48 = (uint32_t)(base48 + jb + kEnt);
            int n_pix = 0;
            const uint64_t took = blend_fwd_pair<DEPTH>(ps, done0, done1, pw, f2{al0, al1}, b.z, b.w, c.x, DEPTH ? c.y : 0.f, idx48,
                                                        STATS ? &n_pix : nullptr);
            if constexpr (STATS) ws.v[kStFwdActivePix] += n_pix;
            // the two halves of `took` go to lane i - i0 of act_lo / act_hi as they are (one scalar + two vector
            // instructions; whether a half is non-zero -- its lane group took the entry -- is asked once per walk)
            // m0 carries the lane number: a gfx9 VALU instruction reads one SGPR, m0 not counted, and this clang has no
            // writelane builtin.  m0 is on the clobber list, so the compiler keeps no value of its own in it across the
            // statement (it re-initialises m0 before each of its own uses anyway; the diagnostic about a reserved register on
            // a clobber list is switched off for this one statement).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
            asm("s_mov_b32 m0, %2\n\t"
                "v_writelane_b32 %0, %3, m0\n\t"
                "v_writelane_b32 %1, %4, m0\n\t"
                "s_add_u32 %2, %2, 1"
                : "+v"(act_lo), "+v"(act_hi), "+s"(pos) : "s"((uint32_t)took), "s"((uint32_t)(took >> 32)) : "scc", "m0");
#pragma clang diagnostic pop
            if constexpr (STATS) { ws.v[kStFwdTrips] += 1; ws.v[kStFwdEmpty] += took == 0ull; }
        };
        // two trips per test of "every pixel of the wave is finished" (a trip after that changes nothing and records
        // nothing); an odd last position on its own
        if (npos >= 2u) {
            do {
                trip();
                trip();
            } while (pos + 2u <= npos && (done0 & done1) != ~0ull);
        }
        if (pos + 1u == npos && (done0 & done1) != ~0ull) trip();
        return (act_lo != 0u ? 1u : 0u) | (act_hi != 0u ? 4u : 0u);
    };
    s_taken[0][0][threadIdx.x] = 0; s_taken[0][1][threadIdx.x] = 0; s_taken[1][0][threadIdx.x] = 0; s_taken[1][1][threadIdx.x] = 0;
    uint32_t n_taken = 0;
    int it = 0, base = 0;
    for (; base < n; base += KB, ++it) {
        const uint64_t fin = done0 & done1;      // lanes whose two pixels are both finished
        const bool wave_alive = fin != ~0ull;
        if (lane == 0) s_alive[it & 1][wave] = wave_alive;
        __syncthreads();  // also: everyone finished reading the previous batch, and its activity masks are in LDS
        if (it > 0) flush_activity(base - KB, KB);
        if (!(s_alive[it & 1][0] | s_alive[it & 1][1])) break;
        const int cnt = min(KB, n - base);
        if constexpr (STATS) { if (wave == 0) { ws.v[kStFwdStaged] += cnt; ws.v[kStFwdBatches] += 1; } }
        if ((int)threadIdx.x < cnt) {
            scale_entry(ra, rb);
            reinterpret_cast<float4*>(my_ent)[0] = ra;
            reinterpret_cast<float4*>(my_ent)[1] = rb;
            reinterpret_cast<float2*>(my_ent)[4] = make_float2(rcb, DEPTH ? 1.f / rdepth : 0.f);
        }
        if (base + KB + (int)threadIdx.x < n) {
            const uint32_t id = p.point_list[range.x + base + KB + threadIdx.x];
            const float4* r = p.rec + kRecF4 * (int64_t)id;
            ra = r[0]; rb = r[1]; rcb = reinterpret_cast<const float*>(r + 2)[0];
            if constexpr (DEPTH) rdepth = reinterpret_cast<const float*>(r + 2)[1];
        }
        __syncthreads();  // staged; every thread has read the previous batch's activity masks
        if (wave_alive) {
            // compaction: one staged Gaussian per lane against each of the wave's two 8 x 8 blocks (a block all of whose
            // pixels are finished takes nothing any more), survivors appended in order; the shorter list is padded with
            // the sentinel.  Wave-private LDS rows: no barrier needed.
            const bool alive_g[2] = {(uint32_t)fin != 0xFFFFFFFFu, (uint32_t)(fin >> 32) != 0xFFFFFFFFu};
            int n_g[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < KB / 64; ++k) {
                const int jj = k * 64 + lane;
                bool touch[2] = {false, false};
                if (jj < cnt) {
                    const float4 a = reinterpret_cast<const float4*>(s_ent + jj * kFwdEntF)[0];
                    const float4 b = reinterpret_cast<const float4*>(s_ent + jj * kFwdEntF)[1];
                    touch[0] = alive_g[0] && block_may_touch<8>(a, b, sxf, syf);
                    touch[1] = alive_g[1] && block_may_touch<8>(a, b, sxf + 8.f, syf);
                }
                if constexpr (STATS) ws.v[kStFwdCulled] += __popcll(__ballot(jj < cnt && !touch[0] && !touch[1]));
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const uint64_t mask = __ballot(touch[g]);
                    if (touch[g]) s_list[wave][g][n_g[g] + mask_prefix(mask)] = (uint16_t)(jj * kEnt);
                    n_g[g] += __popcll(mask);
                }
            }
            const int n_t = max(n_g[0], n_g[1]);
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int t = lane; t < kListLen; t += 64)
                    if (t >= n_g[g]) s_list[wave][g][t] = (uint16_t)kSentinel;
            // takers by LIST position (lane i of t0 / t1: the groups that took position i / 64 + i of their list), then one
            // lane per position marks the entries
            const uint32_t t0 = walk(0, min(n_t, 64), base * kEnt);
            const uint32_t t1 = ((done0 & done1) != ~0ull && n_t > 64) ? walk(64, n_t, base * kEnt) : 0u;

#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint32_t bit = 1u << (2 * g);
                if (t0 & bit) s_taken[wave][g][s_list[wave][g][lane] / kEnt] = (uint8_t)(bit << wave);
                if (t1 & bit) s_taken[wave][g][s_list[wave][g][64 + lane] / kEnt] = (uint8_t)(bit << wave);
                n_taken += (uint32_t)(__popcll(__ballot((t0 & bit) != 0u)) + __popcll(__ballot((t1 & bit) != 0u)));
            }
        }
    }
    if (it > 0 && base >= n) {  // the loop ran out of entries: the last batch's activity is still in LDS
        __syncthreads();
        flush_activity(base - KB, n - (base - KB));
    }
    if (lane == 0) s_work[wave] = n_taken;
    __syncthreads();
    if (threadIdx.x == 0) p.tile_work[vt] = s_work[0] + s_work[1];
    // Results leave through LDS, re-dealt to the lanes row-major (lane = column + 16 x row pair): the 8 x 8-block mapping
    // of the compositing loop would store every image row in two 32-byte pieces from two different instructions (+26 %
    // written and +30 % fetched bytes on this kernel, measured), this way a wave stores whole 64-byte row segments.
    // s_ent is free now (the barrier above orders the last batch's reads before these writes): 6 planes of 128 pixels.
    {
        float* const o = s_ent + wave * (6 * 128);
        const int pos0 = (py0 - sy) * 16 + (px - sx), pos1 = pos0 + 16;
        o[pos0] = ps.T.x; o[pos1] = ps.T.y;
        o[128 + pos0] = __uint_as_float(ps.last0); o[128 + pos1] = __uint_as_float(ps.last1);
        o[256 + pos0] = ps.C0.x; o[256 + pos1] = ps.C0.y;
        o[384 + pos0] = ps.C1.x; o[384 + pos1] = ps.C1.y;
        o[512 + pos0] = ps.C2.x; o[512 + pos1] = ps.C2.y;
        if constexpr (DEPTH) { o[640 + pos0] = ps.D.x; o[640 + pos1] = ps.D.y; }
        // (same wave writes and reads its own planes: LDS accesses of a wave complete in order, no barrier needed)
        const int qx = sx + (lane & 15), qy0 = sy + 2 * (lane >> 4);
        const int r0 = (qy0 - sy) * 16 + (lane & 15), r1 = r0 + 16;
        PixF s0, s1;
        // `last` was kept as (contributor number) * 48
        s0.T = o[r0]; s0.last = __float_as_uint(o[128 + r0]) / (kFwdEntF * 4); s0.C0 = o[256 + r0]; s0.C1 = o[384 + r0]; s0.C2 = o[512 + r0];
        s1.T = o[r1]; s1.last = __float_as_uint(o[128 + r1]) / (kFwdEntF * 4); s1.C0 = o[256 + r1]; s1.C1 = o[384 + r1]; s1.C2 = o[512 + r1];
        const bool q0 = qx < p.W && qy0 < p.H, q1 = qx < p.W && qy0 + 1 < p.H;
        if (q0) write_pixel_fwd(p, s0, pose, qx, qy0);
        if (q1) write_pixel_fwd(p, s1, pose, qx, qy0 + 1);
        if constexpr (DEPTH) {
            const int64_t HW = (int64_t)p.H * p.W;
            if (q0) p.out_invdepth[(int64_t)pose * HW + (int64_t)qy0 * p.W + qx] = o[640 + r0];
            if (q1) p.out_invdepth[(int64_t)pose * HW + (int64_t)(qy0 + 1) * p.W + qx] = o[640 + r1];
        }
    }
    if constexpr (STATS) ws.flush(p.stats);
}

// N > 1: average the per-pose images.  LDR domain (default, follows assets/pipeline.png: the blur "+" is
// drawn over the LDR images) or radiance domain (HS_FLAG_BLUR_HDR).  pose_hdr slot N receives mean radiance.
__global__ void __launch_bounds__(256) resolve_kernel(int64_t HW, int N, int flags, float* pose_hdr, Crf crf,
                                                      const float* exposure, float* out_color, float* out_hdr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over 3*HW
    if (i >= 3 * HW) return;
    const int ch = (int)(i / HW);
    const bool hdr = flags & HS_FLAG_HDR;
    if (hdr) crf.dt = exposure[0];
    float sumH = 0.f, sumL = 0.f;
    for (int k = 0; k < N; ++k) {
        const float Hv = pose_hdr[(int64_t)k * 3 * HW + i];
        sumH += Hv;
        if (hdr && !(flags & HS_FLAG_BLUR_HDR)) sumL += crf_eval(crf, ch, Hv);
    }
    const float inv = 1.f / (float)N;
    const float meanH = sumH * inv;
    pose_hdr[(int64_t)N * 3 * HW + i] = meanH;
    if (hdr) {
        if (out_hdr) out_hdr[i] = meanH;
        out_color[i] = (flags & HS_FLAG_BLUR_HDR) ? crf_eval(crf, ch, meanH) : sumL * inv;
    } else {
        out_color[i] = meanH;
    }
}

// Which tile each workgroup of the render backward processes.  Block x of 8 handles the tiles the strip map gives to
// XCD x (blocks b = x, x + 8, ... of the launch): the workgroups of the static part (b < n_static) keep that XCD's tiles
// in strip order EXCEPT its lightest ones, which go to the queue the last workgroups serve (render_bwd_kernel) -- the
// launch then ends on short tiles, and its tail is as long as one of those instead of an average one.  Weight of a
// tile = the trips the forward counted on it; "lightest" by a histogram of the weights (two trips per bin; ties inside
// the threshold bin are broken by arrival, any choice is as good).  The order is a permutation of the tiles whatever
// the weights are, and results do not depend on it.
__global__ void __launch_bounds__(1024) order_tiles_kernel(int nb, int gx, int n_static, const uint32_t* work,
                                                           uint32_t* order) {
    constexpr int kBins = 1024, kT = 1024;
    __shared__ uint32_t s_hist[kBins];
    __shared__ uint32_t s_scan[kT / 64];
    __shared__ uint32_t s_thr_bin, s_thr_take, s_cnt_thr, s_cnt_q;
    const int x = blockIdx.x, t = threadIdx.x;
    const int per = nb / 8, rem = nb % 8;
    const int cnt = per + (x < rem ? 1 : 0);                               // tiles (= blocks) of XCD x
    const int n_keep = n_static > x ? (n_static - x + 7) / 8 : 0;          // ... of which static
    const int n_light = cnt - n_keep;
    int q_base = 0;                                                        // this XCD's slice of the queue
    for (int y = 0; y < x; ++y) q_base += (per + (y < rem ? 1 : 0)) - (n_static > y ? (n_static - y + 7) / 8 : 0);
    const int E = (cnt + kT - 1) / kT;                                     // consecutive tiles per thread
    const int k0 = t * E, k1 = min(cnt, (t + 1) * E);
    if (E > 64) {   // (more than 65536 tiles per XCD: no selection, the queue takes the last tiles of the strip order)
        for (int k = k0; k < k1; ++k) order[x + 8 * k] = (uint32_t)xcd_strip_tile(x + 8 * k, nb, gx);
        return;
    }
    auto block_excl_scan = [&](uint32_t v, uint32_t* total) -> uint32_t {  // 1024 threads
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(incl, d); if ((t & 63) >= d) incl += u; }
        __syncthreads();
        if ((t & 63) == 63) s_scan[t >> 6] = incl;
        __syncthreads();
        uint32_t add = 0, tot = 0;
        for (int w = 0; w < kT / 64; ++w) { const uint32_t c = s_scan[w]; if (w < (t >> 6)) add += c; tot += c; }
        *total = tot;
        return add + incl - v;
    };
    auto bin_of = [&](int k) { return (int)min((uint32_t)(kBins - 1), work[xcd_strip_tile(x + 8 * k, nb, gx)] >> 1); };
    s_hist[t] = 0;
    if (t == 0) { s_thr_bin = 0; s_thr_take = 0; s_cnt_thr = 0; s_cnt_q = 0; }
    __syncthreads();
    for (int k = k0; k < k1; ++k) atomicAdd(&s_hist[bin_of(k)], 1u);
    __syncthreads();
    {   // threshold bin: the n_light lightest tiles = all bins below it + `take` tiles of the bin itself
        uint32_t tot;
        const uint32_t h = s_hist[t];
        const uint32_t below = block_excl_scan(h, &tot);
        if (below < (uint32_t)n_light && below + h >= (uint32_t)n_light) { s_thr_bin = t; s_thr_take = (uint32_t)n_light - below; }
    }
    __syncthreads();
    const int thr = (int)s_thr_bin;
    const uint32_t take = s_thr_take;
    uint64_t light_mask = 0;
    uint32_t kept = 0;
    for (int k = k0; k < k1; ++k) {
        const int b = bin_of(k);
        const bool light = n_light > 0 && (b < thr || (b == thr && atomicAdd(&s_cnt_thr, 1u) < take));
        light_mask |= (uint64_t)light << (k - k0);
        kept += !light;
    }
    uint32_t tot_kept;
    uint32_t kpos = block_excl_scan(kept, &tot_kept);   // kept tiles stay in strip order
    for (int k = k0; k < k1; ++k) {
        const uint32_t vt = (uint32_t)xcd_strip_tile(x + 8 * k, nb, gx);
        if ((light_mask >> (k - k0)) & 1ull) order[n_static + q_base + (int)atomicAdd(&s_cnt_q, 1u)] = vt;
        else order[x + 8 * (int)kpos++] = vt;
    }
}

// CRF-table and exposure gradients (a15 backward), bitwise reproducible.
// grid = (ceil(HW / 4096), planes): blockIdx.y selects the (pose, channel) image plane, each block owns 4096 consecutive
// pixels of it, 16 per thread, all held in registers.  A pixel whose log-exposure falls between knots i and i+1 adds
// (1 - f) g to dL/dtable[i] and f g to dL/dtable[i+1].  Both weights are added once per run of pixels that share an
// interval (neighbouring pixels of a natural image mostly do), as 32-bit FIXED-POINT integers with a power-of-two
// scale derived from the block's own max |g| (19 bits below 2^31, so 4096 addends cannot overflow a field) --
// integer adds commute, so the block's table does not depend on the order in which lanes reach the LDS, unlike the
// float atomics this replaces (and a CU retires integer LDS atomics several times faster than float or 64-bit
// ones); the two fields of an interval are turned back into floats and written as the block's partial row.  Quantisation step = 2^-19 of the block's
// largest |g|, unbiased.  Exposure gradient and the clamped ends of the table: per-thread float sums in pixel order,
// fixed DPP tree over the wave, the four waves added in wave order.  crf_reduce_kernel adds the blocks in a fixed order.
// NW waves per workgroup (4: a launch of its own or the segmented sum's; 2: the tail of the render backward's launch, whose
// workgroups are 128 threads): a workgroup owns NW * 1024 pixels; the partial rows are per workgroup either way.
constexpr int kCrfPixPerWave = 1024;

struct CrfGradArgs {
    int64_t HW; int N, flags; const float* pose_hdr; Crf crf; const float* exposure; const float* dL_dcolor; float* partials;
    int bx, planes;   // the job's workgroups: bx pixel blocks x planes (pose, channel) image planes
};

// (`bxi`, `plane`: which pixel block of which plane this workgroup takes -- blockIdx of the stand-alone launch)
// `s_tab64`: K - 1 64-bit LDS words (the stand-alone launches' dynamic LDS; a corner of the render backward's staging area)
template <int NW>
__device__ __forceinline__ void crf_grad_body(const CrfGradArgs& A, const int bxi, const int plane_in,
                                              unsigned long long* const s_tab64) {
    const int64_t HW = A.HW; const int N = A.N, flags = A.flags; const float* pose_hdr = A.pose_hdr; Crf crf = A.crf;
    const float* exposure = A.exposure; const float* dL_dcolor = A.dL_dcolor; float* partials = A.partials;
    int* const s_tab32 = reinterpret_cast<int*>(s_tab64);   // K - 1 intervals: two 32-bit fixed-point fields each
    __shared__ float s_wave[NW][4];
    __shared__ float s_max[NW];
    const int K = crf.K;
    for (int i = threadIdx.x; i < K - 1; i += NW * 64) s_tab64[i] = 0ull;
    crf.dt = exposure[0];
    const bool blur_hdr = (flags & HS_FLAG_BLUR_HDR) && N > 1;
    const float gs = blur_hdr ? 1.f : 1.f / (float)N;
    const float scale = (float)(K - 1) / (crf.umax - crf.umin);
    const int plane = plane_in;              // pose * 3 + ch
    const int pose = plane / 3, ch = plane - 3 * pose;
    const float* Hp = pose_hdr + ((int64_t)(blur_hdr ? N : pose) * 3 + ch) * HW;
    const float* gp = dL_dcolor + (int64_t)ch * HW;
    const float* t = crf.table + ch * K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    const int64_t i16 = ((int64_t)bxi * (NW * 64) + threadIdx.x) * 16;
    float Hv[16], g[16];
    if (i16 + 16 <= HW && (HW & 3) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 h4 = reinterpret_cast<const float4*>(Hp + i16)[q];
            const float4 g4 = reinterpret_cast<const float4*>(gp + i16)[q];
            Hv[4 * q] = h4.x; Hv[4 * q + 1] = h4.y; Hv[4 * q + 2] = h4.z; Hv[4 * q + 3] = h4.w;
            g[4 * q] = g4.x * gs; g[4 * q + 1] = g4.y * gs; g[4 * q + 2] = g4.z * gs; g[4 * q + 3] = g4.w * gs;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool ok = i16 + e < HW;
            Hv[e] = ok ? Hp[i16 + e] : 0.f;
            g[e] = ok ? gp[i16 + e] * gs : 0.f;
        }
    }
    // block maximum of |g| -> fixed-point scale (max is order independent)
    float mx = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(g[e]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();  // also: the table is cleared
    mx = fmaxf(s_max[0], s_max[1]);
    if constexpr (NW == 4) mx = fmaxf(mx, fmaxf(s_max[2], s_max[3]));
    // mx = m * 2^x with m in [0.5, 1)  ->  |g| * 2^(19 - x) < 2^19 ; non-finite or zero gradients: scale 1 (sums are
    // zero or garbage-in-garbage-out, as with float adds)
    int xexp = 0;
    if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &xexp);
    const float to_fix = ldexpf(1.f, 19 - xexp), from_fix = ldexpf(1.f, xexp - 19);

    float gexp = 0.f, g_lo = 0.f, g_hi = 0.f;
    int run = -1;              // knot interval of the pending run (-1: none)
    float r0 = 0.f, r1 = 0.f;  // pending contributions to table[run], table[run + 1]
    auto flush = [&]() {
        // two 32-bit integer LDS atomics (a CU retires them several times faster than one 64-bit or float atomic,
        // profiles/README.md "lane groups"); 19 bits below 2^31 leave room for the block's 4096 addends per field
        atomicAdd(&s_tab32[2 * run], __float2int_rn(r0 * to_fix));
        atomicAdd(&s_tab32[2 * run + 1], __float2int_rn(r1 * to_fix));
    };
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int idx; float f, xv; bool in;
        crf_locate(crf, Hv[e], idx, f, xv, in);
        // pixels clamped to an end of the table (black / saturated regions) are summed per thread
        const bool lo = idx == 0 && f == 0.f, hi = idx == K - 2 && f == 1.f;
        g_lo += lo ? g[e] : 0.f;
        g_hi += hi ? g[e] : 0.f;
        if (!lo && !hi) {
            if (idx != run) {
                if (run >= 0) flush();
                run = idx; r0 = 0.f; r1 = 0.f;
            }
            r0 += (1.f - f) * g[e];
            r1 += f * g[e];
        }
        if (in) gexp += g[e] * (t[idx + 1] - t[idx]) * scale * __builtin_amdgcn_rcpf(xv) * Hv[e];
    }
    if (run >= 0) flush();
    gexp = wave_sum_hi(gexp);
    g_lo = wave_sum_hi(g_lo);
    g_hi = wave_sum_hi(g_hi);
    if (lane == 63) { s_wave[wave][0] = gexp; s_wave[wave][1] = g_lo; s_wave[wave][2] = g_hi; }
    __syncthreads();
    // partial row of this block: K knots of its channel + its exposure term
    float* dst = partials + ((int64_t)plane_in * A.bx + bxi) * (K + 1);
    for (int k = threadIdx.x; k <= K; k += NW * 64) {
        float v;
        if (k == K) {
            v = NW == 4 ? ((s_wave[0][0] + s_wave[1][0]) + s_wave[2 % NW][0]) + s_wave[3 % NW][0] : s_wave[0][0] + s_wave[1][0];
        } else {
            long long acc = 0;  // field 0 of interval k + field 1 of interval k - 1
            if (k < K - 1) acc += (long long)s_tab32[2 * k];
            if (k > 0) acc += (long long)s_tab32[2 * (k - 1) + 1];
            v = (float)acc * from_fix;
            if (k == 0) v += NW == 4 ? ((s_wave[0][1] + s_wave[1][1]) + s_wave[2 % NW][1]) + s_wave[3 % NW][1] : s_wave[0][1] + s_wave[1][1];
            if (k == K - 1) v += NW == 4 ? ((s_wave[0][2] + s_wave[1][2]) + s_wave[2 % NW][2]) + s_wave[3 % NW][2] : s_wave[0][2] + s_wave[1][2];
        }
        dst[k] = v;
    }
}

struct RenderBwd {
    int W, H, gx, gy, ntiles, N, flags;
    const uint2* ranges; const uint32_t* point_list; const float4* rec; const float* bg;
    const float* final_T; const uint32_t* n_contrib; const float* pose_hdr;
    const float* dL_dcolor; const float* dL_dhdr; const float* dL_dalpha; const float* dL_dinvdepth;
    float4* pair_grads;
    uint8_t* pair_flags;
    const uint8_t* pair_act;   // per sorted pair: bit 2g + w = lane group g of wave w took the entry (written by the forward)
    Crf crf;
    const float* exposure;
    unsigned long long* stats;     // STATS instantiations only
    unsigned long long* timeline;  // STATS instantiations only, may be null: {start, end (100 MHz clock), XCC | CU ids}
                                   // per workgroup
    uint32_t* queue;               // counter of the tail queue (zeroed by the forward's first kernel, then self-resetting)
    const uint32_t* tile_order;    // workgroup -> (pose, tile), written by order_tiles_kernel after the forward
    int grid_tiles;                // workgroups of the launch that take tiles; those behind them (if any) ...
    CrfGradArgs crf_tail;          // ... take 2048-pixel blocks of the CRF gradient's first stage (bx * planes of them)
};

// Upstream gradient w.r.t. this pose's radiance H_ch at one pixel (the HDR prologue).
__device__ __forceinline__ float pixel_grad(const RenderBwd& p, const Crf& c, int pose, int ch, int64_t pix, int64_t HW) {
    const float invN = 1.f / (float)p.N;
    float g = p.dL_dcolor[ch * HW + pix];
    if (!(p.flags & HS_FLAG_HDR)) return g * invN;
    float out = 0.f;
    if (p.N == 1 || !(p.flags & HS_FLAG_BLUR_HDR)) {
        const float Hv = p.pose_hdr[((int64_t)pose * 3 + ch) * HW + pix];
        out = crf_grad_H(c, ch, Hv, g * invN);
    } else {
        const float Hm = p.pose_hdr[((int64_t)p.N * 3 + ch) * HW + pix];
        out = crf_grad_H(c, ch, Hm, g) * invN;
    }
    if (p.dL_dhdr) out += p.dL_dhdr[ch * HW + pix] * invN;
    return out;
}

// Per-pixel state of the backward replay (two per lane).
//
// The published recurrence keeps the colour accumulated behind the current entry per channel ("accum_rec") and forms
// dL/dalpha_i = T_i * sum_ch (c_i,ch - accum_rec_ch) dL_ch  -  T_final / (1 - alpha_i) * (bg . dL).  Only the dot product
// with this pixel's dL ever leaves that state, and the recurrence is linear, so the replay tracks the ONE number
//     q_i = accum_rec_i . dL  +  (T_final / T_{i+1}) * (bg . dL)
// (the background is the opaque layer behind the last contributor: q starts at bg . dL and is multiplied by
// (1 - alpha) at every step like the colour behind it), and dL/dalpha_i = T_i * (c_i . dL - q_i),
// q_{i-1} = q_i + alpha_i * (c_i . dL - q_i).  Same algebra, five vector instructions and six registers fewer per
// trip than the per-channel form; the inverse-depth channel (DEPTH) just adds invd * dLd to c . dL.
struct PixB {
    float T, dL0, dL1, dL2, q;
    float dLd;  // inverse-depth channel (DEPTH kernels only): upstream gradient
    uint32_t last;
};

__device__ __forceinline__ void load_pixel_bwd(const RenderBwd& p, PixB& s, bool inside, int pose, int px, int py) {
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t pix = (int64_t)py * p.W + px;
    s.T = 0.f; s.dL0 = s.dL1 = s.dL2 = 0.f; s.last = 0;
    if (inside) {
        s.T = p.final_T[(int64_t)pose * HW + pix];
        s.last = p.n_contrib[(int64_t)pose * HW + pix];
        Crf c = p.crf;
        if (p.flags & HS_FLAG_HDR) c.dt = p.exposure[0];
        s.dL0 = pixel_grad(p, c, pose, 0, pix, HW);
        s.dL1 = pixel_grad(p, c, pose, 1, pix, HW);
        s.dL2 = pixel_grad(p, c, pose, 2, pix, HW);
    }
    float bg_dot = (p.bg[0] * s.dL0 + p.bg[1] * s.dL1) + p.bg[2] * s.dL2;
    // accumulated opacity A = 1 - T_final: dA/dalpha_i = T_final / (1 - alpha_i), the background term with sign flipped
    if (p.dL_dalpha && inside) bg_dot -= p.dL_dalpha[pix] / (float)p.N;
    s.q = bg_dot;
    s.dLd = (p.dL_dinvdepth && inside) ? p.dL_dinvdepth[pix] / (float)p.N : 0.f;
}

// The two pixels of a lane as one packed state: every update below is a two-wide (v_pk_*) instruction where the
// hardware has one, which halves the FMA count of the replay (the loop is VALU-issue bound; v_pk_fma_f32 issues
// at the rate of one v_fma_f32).
struct PairB {
    f2 T, q, dL0, dL1, dL2, dLd;
};
__device__ __forceinline__ f2 splat(float v) { f2 r = {v, v}; return r; }
__device__ __forceinline__ PairB pack_pair(const PixB& a, const PixB& b) {
    PairB s;
    s.T = f2{a.T, b.T}; s.q = f2{a.q, b.q};
    s.dL0 = f2{a.dL0, b.dL0}; s.dL1 = f2{a.dL1, b.dL1}; s.dL2 = f2{a.dL2, b.dL2};
    s.dLd = f2{a.dLd, b.dLd};
    return s;
}
// One back-to-front step for the pixel pair.  Outputs dop = G * dL/dalpha (the opacity term), sw = o * dop (weight of
// the geometric sums) and dch = alpha * T (colour weight); inactive pixels give exact zeros and keep their state
// (alpha_eff = 0 makes every update an identity), so no per-field selects are needed.
template <bool DEPTH>
__device__ __forceinline__ void step_bwd_pair(PairB& s, bool act0, bool act1, f2 G, f2 alpha, float o, float r, float g,
                                              float b, float invd, f2& sw, f2& dop, f2& dch) {
    const f2 ae = {act0 ? alpha.x : 0.f, act1 ? alpha.y : 0.f};
    const f2 one_m = splat(1.f) - ae;
    const f2 rcp = {__builtin_amdgcn_rcpf(one_m.x), __builtin_amdgcn_rcpf(one_m.y)};
    s.T *= rcp;  // T_i = T_{i+1} / (1 - alpha_i)
    dch = ae * s.T;
    f2 cd = (splat(r) * s.dL0 + pk_mul_hi(f2{r, g}, s.dL1)) + splat(b) * s.dL2;  // c_i . dL
    if constexpr (DEPTH) cd += splat(invd) * s.dLd;                     // the inverse-depth image: a fourth channel
    const f2 diff = cd - s.q;
    const f2 gd = G * (diff * s.T);
    s.q += ae * diff;
    dop = f2{act0 ? gd.x : 0.f, act1 ? gd.y : 0.f};  // select AFTER the product (G may be inf at a skipped pixel)
    sw = splat(o) * dop;
}

// LDS record of one staged entry in the backward: the twelve floats of its render record followed by four planes of
// reduced partial sums, so ONE per-entry byte offset (what the per-group lists store) addresses everything.
//   [0..3] x, y, A2, B2   [4..7] C2, opacity, r, g   [8..11] b, 1/depth (DEPTH) or depth, radius, pair-slot start
//   [12 + NV * (2g + w) ...): the NV = nine (ten) sums of the entry over the pixels of lane group g of wave w.  A (wave,
//   group) visits an entry at most once, so it STORES its totals (no LDS atomics: a CU retires only ~0.27 lanes of
//   ds_add_f32 per clock, which is what the twenty adding lanes per trip of the two-group walk ran into --
//   profiles/README.md); planes nobody wrote are skipped at write-out by the entry's activity bits, the others are added
//   in plane order: bitwise reproducible.
constexpr int kTailPct = 8;           // share of the tiles handed out by the queue (see the kernel)
constexpr int kAccF = 12;             // first float of the sums
// staged entries per batch (<= 64: the lists are built from one ballot per group).  52 entries = 10.8 KB of LDS per
// workgroup, which -- with the kernel held to 72 registers (amdgpu_waves_per_eu 7, no spill in the loop) -- lets seven
// waves per SIMD stay resident.  Measured at c3 (whole step, ms): 64 entries / six waves 1.197, 52 / seven 1.179, 48 / seven
// 1.191, 56 / seven 1.213, 40 / eight (64 registers, spills) 1.209; 96 / 128 entries (four / three waves): 0.473 / 0.504 ms
// for the kernel against 0.468.
constexpr int kBwdKB = 52;


template <bool DEPTH, bool STATS>
__global__ void __launch_bounds__(kBatch) __attribute__((amdgpu_waves_per_eu((DEPTH || STATS) ? 1 : 7)))
render_bwd_kernel(RenderBwd p) {
    constexpr int KB = kBwdKB;              // staged entries per batch (<= threads per workgroup)
    constexpr int NV = DEPTH ? 10 : 9;      // reduced values per (tile, entry): nine published sums (+ d inverse depth)
    constexpr int kEntF = kAccF + 4 * NV;   // floats per LDS entry record: 48 (192 bytes) / 52 (208 bytes)
    constexpr int kEntB = kEntF * 4;        // ... a multiple of 16 bytes either way: every record stays 16-byte aligned
    static_assert(kEntB % 16 == 0, "LDS entry records must stay 16-byte aligned");
    constexpr int kSentinel = KB * kEntB;   // byte offset of the record no pixel takes (opacity 0)
    constexpr int kListLen = KB + 8;        // the read-ahead of the slot behind the last one stays inside the array
    __shared__ __attribute__((aligned(16))) float s_ent[(KB + 1) * kEntF];   // record KB: the sentinel
    static_assert(KB * kEntF >= 2 * 7 * 128, "the pixel-state re-deal of the prologue borrows the staging area below the sentinel");
    __shared__ uint16_t s_list[2][2][kListLen];   // [wave][lane group]: byte offsets of the group's takers, back to front
    __shared__ uint8_t s_actb[kBatch];      // activity byte of each staged entry (bit 2g + w: group g of wave w took it)
    __shared__ uint32_t s_max[2];

    // The last kTailPct per cent of the tiles are not bound to a workgroup (hence, through blockIdx, to an XCD): the
    // workgroups behind the static part take them from a queue, one after the other, until it is empty, so an XCD that
    // runs faster (the eight dies of a package finish equal work up to 8 % apart, and which ones are slow differs from
    // package to package) takes more of them.  The counter resets itself: the n_tail workers make n_tail successful
    // fetches in total and one failing fetch each, and the very last of those 2 n_tail fetches writes the zero back.
    // Measured at c3: -3 % on this kernel at 8 %, the same at 4 and 12 %, nothing at 25 %; the forward LOSES 2 % with
    // the same queue and keeps its static map.
    __shared__ uint32_t s_q;
    if constexpr (!STATS) {
        // Workgroups behind the tiles' (round 5): the first stage of the CRF-table / exposure gradient -- independent of this
        // kernel's results, bound by LDS atomics -- handed out by the dispatcher only after every tile has a workgroup, i.e.
        // into the slots the draining launch leaves empty (its last round runs at half occupancy).  Measured at c3, same
        // box: a launch of its own 1.1239 ms per step; its workgroups interleaved with the segmented sum's in one launch
        // 1.1145 (all first 1.1153, all last 1.1224); here 1.097 (c4: 7.39 -> 7.32).  A side stream for the same kernel
        // measured noise three times (rounds 1, 3, 4): what helps is WHEN its workgroups are dispatched, not the queue.
        if ((int)blockIdx.x >= p.grid_tiles) {
            const int qd = (int)blockIdx.x - p.grid_tiles;
            crf_grad_body<2>(p.crf_tail, qd % p.crf_tail.bx, qd / p.crf_tail.bx, reinterpret_cast<unsigned long long*>(s_ent));
            return;
        }
    }
    const int n_tail = (int)((int64_t)p.grid_tiles * kTailPct / 100);
    const int n_static = p.grid_tiles - n_tail;
    if (threadIdx.x < kAccF) s_ent[KB * kEntF + threadIdx.x] = 0.f;   // the sentinel record (ordered by the barriers below)
    for (;;) {
    int bidx = blockIdx.x;
    if ((int)blockIdx.x >= n_static) {
        __syncthreads();   // the previous tile of this worker is finished by every thread
        if (threadIdx.x == 0) {
            const uint32_t q = atomicAdd(p.queue, 1u);
            if (q == 2u * (uint32_t)n_tail - 1u) atomicExch(p.queue, 0u);
            s_q = q;
        }
        __syncthreads();
        if (s_q >= (uint32_t)n_tail) return;
        bidx = n_static + (int)s_q;
    }
    const int vt = p.tile_order ? (int)p.tile_order[bidx] : xcd_strip_tile(bidx, p.grid_tiles, p.gx);
    const int pose = vt / p.ntiles;
    const int tile = vt - pose * p.ntiles;
    const int tx = tile % p.gx, ty = tile / p.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 5;               // lane group: the 8 x 8 block of columns 8 grp .. 8 grp + 7 of the half tile
    const int sx = tx * kTile, sy = ty * kTile + wave * 8;
    int px, py0;
    lane_pixels(lane, sx, sy, px, py0);
    const int py1 = py0 + 1;
    const float pxf = (float)px;
    const f2 pyf = {(float)py0, (float)py1};

    unsigned long long t_start = 0;
    if constexpr (STATS) t_start = wall_clock64();
    const uint2 range = p.ranges[vt];

    // The pixel state is LOADED row-major (lane = column + 16 x row pair: every 16-lane quarter of a load instruction reads
    // one whole 64-byte row segment) and re-dealt through LDS to the 8 x 8-block mapping of the replay -- the mirror image of
    // the forward's store path; loaded in the block mapping, each row segment is fetched in two 32-byte pieces by two
    // different quarters.  The staging area is free here: the previous tile of a queue worker ended behind a barrier, and
    // the first batch below starts behind one.  Seven planes of 128 pixels per wave, wave-private (LDS accesses of a wave
    // complete in order: no barrier).
    PixB s0, s1;
    {
        const int qx = sx + (lane & 15), qy0 = sy + 2 * (lane >> 4);
        PixB t0, t1;
        load_pixel_bwd(p, t0, qx < p.W && qy0 < p.H, pose, qx, qy0);
        load_pixel_bwd(p, t1, qx < p.W && qy0 + 1 < p.H, pose, qx, qy0 + 1);
        float* const o = s_ent + wave * (7 * 128);
        const int r0 = (qy0 - sy) * 16 + (lane & 15), r1 = r0 + 16;
        o[r0] = t0.T; o[r1] = t1.T;
        o[128 + r0] = t0.dL0; o[128 + r1] = t1.dL0;
        o[256 + r0] = t0.dL1; o[256 + r1] = t1.dL1;
        o[384 + r0] = t0.dL2; o[384 + r1] = t1.dL2;
        o[512 + r0] = t0.q; o[512 + r1] = t1.q;
        o[640 + r0] = t0.dLd; o[640 + r1] = t1.dLd;
        o[768 + r0] = __uint_as_float(t0.last); o[768 + r1] = __uint_as_float(t1.last);
        // the reads below take values OTHER lanes of this wave wrote: say so to the compiler (a wavefront-scope release
        // fence + a wave barrier: no instruction is emitted, the LDS operations of a wave complete in order anyway, but
        // the order of the stores and the loads is now a stated contract instead of an alias-analysis accident)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int p0 = (py0 - sy) * 16 + (px - sx), p1 = p0 + 16;
        s0.T = o[p0]; s0.dL0 = o[128 + p0]; s0.dL1 = o[256 + p0]; s0.dL2 = o[384 + p0]; s0.q = o[512 + p0];
        s0.dLd = o[640 + p0]; s0.last = __float_as_uint(o[768 + p0]);
        s1.T = o[p1]; s1.dL0 = o[128 + p1]; s1.dL1 = o[256 + p1]; s1.dL2 = o[384 + p1]; s1.q = o[512 + p1];
        s1.dLd = o[640 + p1]; s1.last = __float_as_uint(o[768 + p1]);
    }
    PairB ps = pack_pair(s0, s1);

    const uint32_t wave_max = wave_max_u32(max(s0.last, s1.last));
    if (lane == 0) s_max[wave] = wave_max;
    __syncthreads();
    const int n_proc = (int)max(s_max[0], s_max[1]);

    // Which of its group's totals this lane adds to the entry's sums after group_reduce (see there), as a byte offset
    // inside the entry record (-1: none), and from which of the two result registers: the quad leaders of the group's two
    // rows carry t0 (eight sums), the second lanes of quads 0 and 2 of its first row carry t1 (the ninth and tenth).
    int red_off = -1;
    const bool red_t0 = (lane & 3) == 0;
    {
        const int row = lane >> 4, k = (lane & 15) >> 2;
        constexpr int kOrd[4] = {0, 2, 1, 3};
        if ((lane & 3) == 0) red_off = ((row & 1) ? 4 : 0) + kOrd[k];
        else if ((lane & 3) == 1 && !(row & 1) && k == 0) red_off = 8;
        else if (DEPTH && (lane & 3) == 1 && !(row & 1) && k == 2) red_off = 9;
        if (red_off >= 0) red_off = (kAccF + NV * (2 * grp + wave) + red_off) * 4;
    }
    const int xor16_addr = (lane ^ 16) << 2;
    char* const ent = reinterpret_cast<char*>(s_ent);
    float* const my_ent = s_ent + threadIdx.x * kEntF;
    const uint16_t* const my_list = s_list[wave][grp];
    WaveStats ws;
    if constexpr (STATS) ws.clear();
    const int nb = (n_proc + KB - 1) / KB;
    // only the instance id of the NEXT batch is prefetched (one register); its record is gathered at the
    // top of the batch -- keeping the three float4 in registers across the replay loop costs a wave of occupancy
    // The forward left one activity byte per sorted pair (bit 2g + w: some pixel of lane group g of wave w took the
    // entry).  An entry nobody took is neither gathered nor written out; the two 32-lane groups of a wave each walk
    // exactly their own takers, so one trip of the loop serves up to two different entries, and no trip is empty.
    uint32_t id_next = 0, act_next = 0;
    if (nb > 0) {
        const int base = (nb - 1) * KB;
        if ((int)threadIdx.x < n_proc - base) {
            id_next = p.point_list[range.x + base + threadIdx.x];
            act_next = p.pair_act[(int64_t)range.x + base + threadIdx.x];
        }
    }
    for (int bi = nb - 1; bi >= 0; --bi) {
        const int base = bi * KB;
        const int cnt = min(KB, n_proc - base);
        if constexpr (STATS) { if (wave == 0) { ws.v[kStBwdStaged] += cnt; ws.v[kStBwdBatches] += 1; } }
        __syncthreads();  // previous batch's write-out finished
        const bool taken = (int)threadIdx.x < cnt && act_next != 0;
        s_actb[threadIdx.x] = (uint8_t)((int)threadIdx.x < cnt ? act_next : 0u);
        if (taken) {
            const float4* r = p.rec + kRecF4 * (int64_t)id_next;
            float4 ra = r[0], rb = r[1];
            float4 rc = r[2];
            if constexpr (DEPTH) rc.y = 1.f / rc.y;  // depth -> inverse depth
            scale_entry(ra, rb);
            reinterpret_cast<float4*>(my_ent)[0] = ra;
            reinterpret_cast<float4*>(my_ent)[1] = rb;
            reinterpret_cast<float4*>(my_ent)[2] = rc;
        }
        if (bi > 0 && (int)threadIdx.x < KB) {  // batches below the top are full
            id_next = p.point_list[range.x + base - KB + threadIdx.x];
            act_next = p.pair_act[(int64_t)range.x + base - KB + threadIdx.x];
        }
        __syncthreads();
        {
            // The two lists of this wave: group g's takers among the staged entries (bit 2g + wave of the activity bytes),
            // back to front, then sentinels up to the longer list's length.  Wave-private LDS rows: no barrier needed.
            int maxn = 0;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint32_t gbit = 1u << (2 * g + wave);
                uint16_t* lst = s_list[wave][g];
                int n_g = 0;
#pragma unroll
                for (int k = (KB + 63) / 64 - 1; k >= 0; --k) {   // upper half of the batch first: it is deeper
                    const uint64_t m = __ballot((s_actb[k * 64 + lane] & gbit) != 0);
                    const int c = __popcll(m);
                    if ((m >> lane) & 1ull) lst[n_g + c - 1 - mask_prefix(m)] = (uint16_t)((k * 64 + lane) * kEntB);
                    n_g += c;
                }
#pragma unroll
                for (int t = lane; t < kListLen; t += 64)
                    if (t >= n_g) lst[t] = (uint16_t)kSentinel;
                maxn = max(maxn, n_g);
            }
            if constexpr (STATS) {
                const uint32_t wbits = 5u << wave;
                uint32_t mine = 0;
#pragma unroll
                for (int k = 0; k < (KB + 63) / 64; ++k) mine += __popcll(__ballot((s_actb[k * 64 + lane] & wbits) != 0));
                ws.v[kStBwdCulled] += (uint32_t)cnt - mine;
            }
            // a pixel takes part in entry j of this batch iff base + j < last, i.e. iff j * 128 < (last - base) * 128 (the
            // sentinel passes this test for pixels that reach past the batch: its opacity 0 is what keeps it out)
            const int lim0 = ((int)s0.last - base) * kEntB, lim1 = ((int)s1.last - base) * kEntB;
            int jb = (int)my_list[0];
            for (int ti = 0; ti < maxn; ++ti) {
                // next trip's entry, read a trip ahead -- and made a full register: carried as 16 bits it is masked again every
                // trip.  (Two trips per loop body, to drop the copy that hands it on: the compiler rolls them back into one
                // body with seven more scalar instructions per trip -- 75 + 15 against 76 + 8 -- so the copy stays.)
                int jn = (int)my_list[ti + 1];
                asm("" : "+v"(jn));
                const float4 a = *reinterpret_cast<const float4*>(ent + jb);   // (two addresses per wave: one per group)
                const float4 b = *reinterpret_cast<const float4*>(ent + jb + 16);
                const float2 c = *reinterpret_cast<const float2*>(ent + jb + 32);
                const float cb = c.x;
                const float invd = DEPTH ? c.y : 0.f;  // staged as 1 / depth in DEPTH kernels
                const float dx = a.x - pxf;
                const f2 dy = a.y - pyf;
                const float t = a.z * dx * dx, u = a.w * dx;
                const f2 pw = dy * (b.x * dy + u) + t;
                const float G0 = hs_exp2(pw.x), G1 = hs_exp2(pw.y);
                const float al0 = fminf(kAlphaMax, b.y * G0), al1 = fminf(kAlphaMax, b.y * G1);
                const bool act0 = (jb < lim0) && (pw.x <= 0.f) && (al0 >= kAlphaMin);
                const bool act1 = (jb < lim1) && (pw.y <= 0.f) && (al1 >= kAlphaMin);
                if constexpr (STATS) {
                    const int lanes = __popcll(__ballot(act0 || act1));
                    ws.v[kStBwdTrips] += 1;
                    ws.v[kStBwdEmpty] += lanes == 0;
                    ws.v[kStBwdActivePix] += __popcll(__ballot(act0)) + __popcll(__ballot(act1));
                    const int bin = lanes == 0 ? 0 : lanes <= 4 ? 1 : lanes <= 8 ? 2 : lanes <= 16 ? 3 : lanes <= 32 ? 4 : 5;
                    ws.v[kStBwdHist + bin] += 1;
                }
                f2 sw, dop, dch;
                step_bwd_pair<DEPTH>(ps, act0, act1, f2{G0, G1}, f2{al0, al1}, b.y, b.z, b.w, cb, invd, sw, dop, dch);
                // in-lane sums over the pixel pair (dx is shared):  S1 = sum w dx, S2 = sum w dy, S3 = sum w dx^2,
                // S4 = sum w dx dy, S5 = sum w dy^2 ; the conic factors are applied once per entry at write-out
                const f2 m = sw * dy;
                const f2 m2 = m * dy;
                const f2 c0 = dch * ps.dL0, c1 = dch * ps.dL1, c2 = dch * ps.dL2;
                float g[9];
                g[0] = (sw.x + sw.y) * dx;
                g[1] = m.x + m.y;
                g[2] = g[0] * dx;
                g[3] = g[1] * dx;
                g[4] = m2.x + m2.y;
                g[5] = dop.x + dop.y;
                g[6] = c0.x + c0.y;
                g[7] = c1.x + c1.y;
                g[8] = c2.x + c2.y;
                float g9 = 0.f;
                if constexpr (DEPTH) {
                    const f2 cd = dch * ps.dLd;
                    g9 = cd.x + cd.y;
                }
                float t0, t1;
                group_reduce<DEPTH>(g, g9, xor16_addr, t0, t1);
                // 2 x 9 (10) lanes, one plain LDS store: each group into its own plane of its own entry's record
                if (red_off >= 0) *reinterpret_cast<float*>(ent + jb + red_off) = red_t0 ? t0 : t1;
                jb = jn;
            }
        }
        __syncthreads();
        if (taken) {
            // the planes the entry's takers wrote (bit 2g + w of its activity byte), added in plane order
            float v[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const uint32_t ab = s_actb[threadIdx.x];
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) {
                const bool on = (ab >> pl) & 1u;
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const float x = my_ent[kAccF + NV * pl + q];
                    v[q] += on ? x : 0.f;
                }
            }
            const float4 a = reinterpret_cast<const float4*>(my_ent)[0];
            const float4 c = reinterpret_cast<const float4*>(my_ent)[2];
            // un-scale the conic: A = A2 * (-2/L), B = B2 * (-1/L), C = C2 * (-2/L)
            const float A = a.z * (-2.f / kLog2e), B = a.w * (-1.f / kLog2e), C = my_ent[4] * (-2.f / kLog2e);
            const float ddelx_dx = 0.5f * (float)p.W, ddely_dy = 0.5f * (float)p.H;
            // dL/dmean2D = -(A S1 + B S2, C S2 + B S1) (NDC-scaled); dL/dconic = (-0.5 S3, -S4, -0.5 S5)
            const float gmx = -(A * v[0] + B * v[1]) * ddelx_dx;
            const float gmy = -(C * v[1] + B * v[0]) * ddely_dy;
            // slot of the (tile, instance) pair in duplicateWithKeys order
            const int rad = __float_as_int(c.z);
            const uint32_t off = __float_as_uint(c.w);
            const int rminx = min(p.gx, max(0, (int)((a.x - (float)rad) / (float)kTile)));
            const int rminy = min(p.gy, max(0, (int)((a.y - (float)rad) / (float)kTile)));
            const int rmaxx = min(p.gx, max(0, (int)((a.x + (float)rad + (float)(kTile - 1)) / (float)kTile)));
            const int64_t slot = (int64_t)off + (int64_t)(ty - rminy) * (rmaxx - rminx) + (tx - rminx);
            p.pair_flags[slot] = 1;
            float4* o = p.pair_grads + kPairF4 * slot;
            o[0] = make_float4(gmx, gmy, -0.5f * v[2], -v[3]);
            o[1] = make_float4(-0.5f * v[4], v[5], v[6], v[7]);
            o[2] = make_float4(v[8], v[9], 0.f, 0.f);
            if constexpr (kPairF4 == 4) o[3] = make_float4(0.f, 0.f, 0.f, 0.f);  // the record fills its 64-byte sector
        }
    }
    if constexpr (STATS) {
        ws.flush(p.stats);
        if (p.timeline && threadIdx.x == 0) {
            // one row per TILE SLOT of the launch (bidx: a queue worker serves several slots, or none)
            p.timeline[3 * bidx + 0] = t_start;
            p.timeline[3 * bidx + 1] = wall_clock64();
            // HW_REG_XCC_ID (20): bits 3:0 ; HW_REG_HW_ID (4): cu_id bits 11:8, sh_id 12, se_id 15:13
            const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
            const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            p.timeline[3 * bidx + 2] = ((unsigned long long)xcc << 32) | hw;
        }
    }
    if ((int)blockIdx.x < n_static) return;
    }   // next tile of the queue
}

__global__ void __launch_bounds__(256) crf_grad_kernel(CrfGradArgs A) {
    extern __shared__ unsigned long long s_crf_tab[];
    crf_grad_body<4>(A, (int)blockIdx.x, (int)blockIdx.y, s_crf_tab);
}

// The partial rows added up in a fixed order (hs_common.h, crf_reduce_block): stand-alone launch
__global__ void __launch_bounds__(256) crf_reduce_kernel(CrfReduce c) { crf_reduce_block(c, (int)blockIdx.x); }

}  // namespace

// Ending the launch on its lightest tiles pays while the launch is a few rounds of workgroups long (c3: 8160 workgroups
// over 3072 slots, -4 % on the backward for an 8 us ordering kernel); with many rounds (c4: 65280) the tail is a small
// part of the span and the ordering costs more than it returns (measured +42 / -36 us): the strip order stays.
static bool orders_tiles(int grid) { return grid <= 6 * 3072; }


int launch_render_fwd(const hs_fwd_args& a, const hs_layout& L, hipStream_t s, unsigned long long* stats) {
    // (the forward records no timeline)
    const hs_dims& d = a.dims;
    RenderFwd p;
    p.W = d.W; p.H = d.H; p.gx = (d.W + kTile - 1) / kTile;
    const int gy = (d.H + kTile - 1) / kTile;
    p.ntiles = p.gx * gy; p.N = d.n_poses; p.flags = a.flags;
    char* bin = (char*)a.binning; char* img = (char*)a.image; char* geom = (char*)a.geom;
    p.ranges = (const uint2*)(bin + L.ranges); p.point_list = (const uint32_t*)(bin + L.point_list);
    p.rec = (const float4*)(geom + L.rec); p.bg = a.bg;
    p.out_color = a.out_color; p.out_hdr = a.out_hdr;
    p.final_T = (float*)(img + L.final_T); p.n_contrib = (uint32_t*)(img + L.n_contrib);
    const bool need_pose = (a.flags & HS_FLAG_HDR) || d.n_poses > 1;
    p.pose_hdr = need_pose ? (float*)(img + L.pose_hdr) : nullptr;
    p.crf.table = a.crf_table; p.crf.K = a.crf_K; p.crf.umin = a.crf_umin; p.crf.umax = a.crf_umax; p.crf.dt = 1.f;
    p.exposure = a.exposure;
    p.out_invdepth = a.out_invdepth;
    p.stats = stats; p.timeline = nullptr;
    p.pair_act = (uint8_t*)bin + L.pair_act;
    p.tile_work = (uint32_t*)(img + L.tile_work);
    const int grid = p.ntiles * d.n_poses;
    if (stats) {
        if (a.out_invdepth) render_fwd_kernel<true, true><<<grid, kBatch, 0, s>>>(p);
        else render_fwd_kernel<false, true><<<grid, kBatch, 0, s>>>(p);
    } else if (a.out_invdepth) render_fwd_kernel<true, false><<<grid, kBatch, 0, s>>>(p);
    else render_fwd_kernel<false, false><<<grid, kBatch, 0, s>>>(p);
    HS_LAUNCH_CHECK();
    if (orders_tiles(grid)) {   // the order in which the render backward will take the tiles (render_bwd_kernel)
        const int n_static = grid - (int)((int64_t)grid * kTailPct / 100);
        order_tiles_kernel<<<8, 1024, 0, s>>>(grid, p.gx, n_static, p.tile_work, (uint32_t*)(img + L.tile_order));
        HS_LAUNCH_CHECK();
    }
    if (d.n_poses > 1) {
        const int64_t HW = (int64_t)d.W * d.H;
        resolve_kernel<<<ceil_div(3 * HW, 256), 256, 0, s>>>(HW, d.n_poses, a.flags, p.pose_hdr, p.crf, a.exposure,
                                                            a.out_color, a.out_hdr);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}

static bool crf_grad_args(const hs_bwd_args& a, const hs_layout& L, CrfGradArgs& A, int waves = 4) {
    const hs_dims& d = a.dims;
    if (!((a.flags & HS_FLAG_HDR) && (a.dL_dcrf_table || a.dL_dexposure))) return false;
    A.crf.table = a.crf_table; A.crf.K = a.crf_K; A.crf.umin = a.crf_umin; A.crf.umax = a.crf_umax; A.crf.dt = 1.f;
    A.HW = (int64_t)d.W * d.H; A.N = d.n_poses; A.flags = a.flags;
    A.partials = (float*)((char*)a.bwd + L.crf_partials);
    A.pose_hdr = (const float*)((const char*)a.image + L.pose_hdr);
    A.exposure = a.exposure; A.dL_dcolor = a.dL_dout_color;
    const bool blur_hdr = (a.flags & HS_FLAG_BLUR_HDR) && d.n_poses > 1;
    A.planes = 3 * (blur_hdr ? 1 : d.n_poses);
    A.bx = ceil_div(A.HW, (int64_t)waves * kCrfPixPerWave);
    return true;
}

int launch_crf_bwd(const hs_bwd_args& a, const hs_layout& L, hipStream_t s, CrfReduce* defer, bool in_render_tail) {
    CrfGradArgs A;
    if (!crf_grad_args(a, L, A, in_render_tail ? 2 : 4)) return HS_OK;
    if (!in_render_tail)
        crf_grad_kernel<<<dim3(A.bx, A.planes), 256, (size_t)(a.crf_K - 1) * sizeof(unsigned long long), s>>>(A);
    CrfReduce cr;
    cr.partials = A.partials; cr.bx = A.bx; cr.planes = A.planes; cr.K = a.crf_K; cr.d_table = a.dL_dcrf_table;
    cr.d_exposure = a.dL_dexposure; cr.nblocks = ceil_div(3 * a.crf_K + 1, 4);
    if (defer) *defer = cr;   // a later launch of the call adds the rows up (its first workgroups): one launch less
    else crf_reduce_kernel<<<cr.nblocks, 256, 0, s>>>(cr);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

int launch_render_bwd(const hs_bwd_args& a, const hs_layout& L, hipStream_t s, unsigned long long* stats,
                      unsigned long long* timeline, bool crf_in_tail) {
    const hs_dims& d = a.dims;
    RenderBwd p;
    p.W = d.W; p.H = d.H; p.gx = (d.W + kTile - 1) / kTile; p.gy = (d.H + kTile - 1) / kTile;
    p.ntiles = p.gx * p.gy; p.N = d.n_poses; p.flags = a.flags;
    const char* bin = (const char*)a.binning; const char* img = (const char*)a.image; const char* geom = (const char*)a.geom;
    p.ranges = (const uint2*)(bin + L.ranges); p.point_list = (const uint32_t*)(bin + L.point_list);
    p.rec = (const float4*)(geom + L.rec); p.bg = a.bg;
    p.final_T = (const float*)(img + L.final_T); p.n_contrib = (const uint32_t*)(img + L.n_contrib);
    p.pose_hdr = (const float*)(img + L.pose_hdr);
    p.dL_dcolor = a.dL_dout_color; p.dL_dhdr = a.dL_dout_hdr; p.dL_dalpha = a.dL_dout_alpha;
    p.dL_dinvdepth = a.dL_dout_invdepth;
    p.pair_grads = (float4*)((char*)a.bwd + L.pair_grads);
    p.crf.table = a.crf_table; p.crf.K = a.crf_K; p.crf.umin = a.crf_umin; p.crf.umax = a.crf_umax; p.crf.dt = 1.f;
    p.exposure = a.exposure;
    // pairs beyond a tile's deepest contributor are never visited: only flagged records are summed later.  The flags
    // were cleared by the forward's pair emission; which records get written depends on the forward state alone,
    // so replays of this stage set the same flags again.
    p.pair_flags = (uint8_t*)a.binning + L.pair_flags;
    p.pair_act = (const uint8_t*)a.binning + L.pair_act;
    p.stats = stats; p.timeline = timeline;
    p.queue = &((hs_counters*)((char*)a.geom + L.counters))->reserved[3];
    p.tile_order = orders_tiles(p.ntiles * d.n_poses) ? (const uint32_t*)(img + L.tile_order) : nullptr;
    int grid = p.ntiles * d.n_poses;
    p.grid_tiles = grid;
    // (the CRF gradient's first stage as extra workgroups behind the tiles': see the kernel's head)
    if (crf_in_tail && !stats && crf_grad_args(a, L, p.crf_tail, 2)) grid += p.crf_tail.bx * p.crf_tail.planes;
    if (stats) {
        if (a.dL_dout_invdepth) render_bwd_kernel<true, true><<<grid, kBatch, 0, s>>>(p);
        else render_bwd_kernel<false, true><<<grid, kBatch, 0, s>>>(p);
    } else if (a.dL_dout_invdepth) render_bwd_kernel<true, false><<<grid, kBatch, 0, s>>>(p);
    else render_bwd_kernel<false, false><<<grid, kBatch, 0, s>>>(p);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

int render_stats_count() { return kStCount; }

// scratch of the CRF-gradient stage: one row of K + 1 floats per (pixel block, pose, channel)
int64_t crf_partial_floats(int K, int64_t HW, int n_poses) {
    return K > 0 ? (int64_t)ceil_div(HW, 2 * kCrfPixPerWave) * 3 * n_poses * (K + 1) : 0;   // (the smaller workgroup: two waves)
}

}  // namespace hs
