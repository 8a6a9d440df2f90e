// Per-tile alpha-blend forward (a9) and per-pixel backward (a10) for gfx950, plus the HDR
// epilogue / prologue and N-pose resolve (a15).  Rules: SURVEY.md 8(a); no reference code exists.
//
// CDNA4 design (not a translation of the CUDA 16x16-thread tile):
//  * one 256-thread workgroup per 16x16 binning tile, but each of its four wave64s owns a compact
//    8x8 sub-tile, so a wave's early termination and culling are spatially coherent;
//  * the tile's sorted instance list is staged 256 entries at a time into LDS as three float4
//    planes (one gather of a 48-byte record per thread), then read back with wave-uniform
//    (broadcast) ds_read_b128 -- no bank conflicts, no VGPRs spent on the batch;
//  * before touching a batch each wave tests the 256 staged Gaussians against its own sub-tile,
//    one Gaussian per lane, and turns the result into four 64-bit ballot masks; the compositing loop
//    then walks set bits only (s_ff1/s_flbit) -- Gaussians whose 1/255-alpha ellipse misses the
//    sub-tile cost ~0.3 instructions instead of ~30;
//  * backward: no global atomics.  Per (wave, Gaussian) the nine partial derivatives are reduced
//    across the 64 lanes with DPP row operations, parked in a per-wave LDS plane, summed over the
//    four waves in fixed order and written as ONE 48-byte record per (tile, instance) pair at the
//    pair's duplicateWithKeys slot; preprocess-backward then sums each instance's contiguous
//    slots.  Gradients are bitwise reproducible run to run.
//
// This TU is compiled with FMA contraction on (the loops are VALU-bound); every integer decision it
// makes (pair slot addressing) uses add/div-only expressions that contraction cannot change.
#include "hs_common.h"

namespace hs {

namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTmin = 0.0001f;
constexpr float kLogEps = 1e-8f;

#ifdef HS_ACCURATE_EXP
__device__ __forceinline__ float hs_exp(float x) { return expf(x); }
#else
__device__ __forceinline__ float hs_exp(float x) { return __expf(x); }
#endif

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
// Halving steps of the 9-value wave reduction (gfx950 v_permlane{32,16}_swap): the sum of TWO registers over
// one lane bit costs one swap + one add, and the result holds x's partial sums in the lanes whose bit is 0 and
// y's in the lanes whose bit is 1.
__device__ __forceinline__ float halve32(float x, float y) {  // lane bit 5
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float halve16(float x, float y) {  // lane bit 4 (x -> even rows, y -> odd rows)
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Sum over the 16 lanes of each DPP row; every lane of the row receives the row total.
__device__ __forceinline__ float row_sum(float v) {
    v += dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp<0x141>(v);  // row_half_mirror
    v += dpp<0x140>(v);  // row_mirror
    return v;
}
// Reduces g[0..8] over the wave.  Output: three registers whose 16-lane rows hold wave totals:
//   q0 rows 0..3 = g0, g2, g1, g3 ; q1 rows 0..3 = g4, g6, g5, g7 ; q2 rows 0,1 = g8 (rows 2,3 = 0).
__device__ __forceinline__ void wave_reduce9(const float* g, float& q0, float& q1, float& q2) {
    const float r0 = halve32(g[0], g[1]);
    const float r1 = halve32(g[2], g[3]);
    const float r2 = halve32(g[4], g[5]);
    const float r3 = halve32(g[6], g[7]);
    const float r4 = halve32(g[8], 0.f);
    q0 = row_sum(halve16(r0, r1));
    q1 = row_sum(halve16(r2, r3));
    q2 = row_sum(halve16(r4, r4));
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

// One Gaussian per lane against the wave's 8x8 sub-tile [sx,sx+7]x[sy,sy+7] (pixel centres).
// Returns false only when NO pixel of the sub-tile can pass `power <= 0 && alpha >= 1/255`:
// alpha >= 1/255  <=>  q(d) = A dx^2 + 2B dx dy + C dy^2 <= tau = 2 ln(255 o); the axis-aligned
// bounding box of that ellipse has half-extents sqrt(tau*C/det), sqrt(tau*A/det).  tau carries an
// absolute safety margin of 0.05 (+1e-4 relative), orders of magnitude above fp32 evaluation
// error of `power` for variance ratios up to ~1e5, so culling never changes a result.
__device__ __forceinline__ bool subtile_may_touch(const float4 a, const float4 b, float sx, float sy) {
    const float A = a.z, B = a.w, C = b.x, o = b.y;
    if (!(o >= kAlphaMin)) return false;
    const float tau = 2.f * __logf(255.f * o) * 1.0001f + 0.05f;
    const float det = A * C - B * B;
    if (!(det > 0.f)) return true;  // degenerate conic: let the exact test decide
    const float k = tau / det;
    const float hx = sqrtf(k * C) * 1.0001f, hy = sqrtf(k * A) * 1.0001f;
    return (a.x + hx >= sx) && (a.x - hx <= sx + 7.f) && (a.y + hy >= sy) && (a.y - hy <= sy + 7.f);
}

struct RenderFwd {
    int W, H, gx, ntiles, N, flags;
    const uint2* ranges; const uint32_t* point_list; const float4* rec; const float* bg;
    float* out_color; float* out_hdr; float* final_T; uint32_t* n_contrib; float* pose_hdr;
    Crf crf;
    const float* exposure;
};

// One compositing step of the forward for one staged entry (branch-free: predication instead of exec
// juggling keeps the scalar unit out of the inner loop).
struct PixF {
    float T, C0, C1, C2;
    uint32_t last;
    bool done;
};
__device__ __forceinline__ void blend_fwd(PixF& s, float power, float alpha, float r, float g, float b, uint32_t idx1) {
    const bool valid = !s.done && power <= 0.f && alpha >= kAlphaMin;
    const float test_T = s.T * (1.f - alpha);
    const bool upd = valid && !(test_T < kTmin);
    s.done = s.done || (valid && test_T < kTmin);
    const float w = upd ? alpha * s.T : 0.f;
    s.C0 += r * w; s.C1 += g * w; s.C2 += b * w;
    s.T = upd ? test_T : s.T;
    s.last = upd ? idx1 : s.last;
}

__global__ void __launch_bounds__(256) render_fwd_kernel(RenderFwd p) {
    __shared__ float4 s_a[256];
    __shared__ float4 s_b[256];
    __shared__ float s_cb[256];
    __shared__ int s_alive[2][4];

    const int vt = blockIdx.x;  // virtual tile = pose * ntiles + tile
    const int pose = vt / p.ntiles;
    const int tile = vt - pose * p.ntiles;
    const int tx = tile % p.gx, ty = tile / p.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sx = tx * kTile + (wave & 1) * 8, sy = ty * kTile + (wave >> 1) * 8;
    const int px = sx + (lane & 7), py = sy + (lane >> 3);
    const bool inside = px < p.W && py < p.H;
    const float pxf = (float)px, pyf = (float)py;
    const float sxf = (float)sx, syf = (float)sy;

    const uint2 range = p.ranges[vt];
    const int n = (int)(range.y - range.x);

    PixF st;
    st.T = 1.f; st.C0 = st.C1 = st.C2 = 0.f; st.last = 0; st.done = !inside;

    int it = 0;
    for (int base = 0; base < n; base += 256, ++it) {
        const bool wave_alive = __ballot(!st.done) != 0ull;
        if (lane == 0) s_alive[it & 1][wave] = wave_alive;
        __syncthreads();  // also: everyone finished reading the previous batch
        if (!(s_alive[it & 1][0] | s_alive[it & 1][1] | s_alive[it & 1][2] | s_alive[it & 1][3])) break;
        const int cnt = min(256, n - base);
        if ((int)threadIdx.x < cnt) {
            const uint32_t id = p.point_list[range.x + base + threadIdx.x];
            const float4* r = p.rec + 3 * (int64_t)id;
            s_a[threadIdx.x] = r[0];
            s_b[threadIdx.x] = r[1];
            s_cb[threadIdx.x] = reinterpret_cast<const float*>(r + 2)[0];
        }
        __syncthreads();
        if (!wave_alive) continue;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const int jj = k * 64 + lane;
            bool touch = false;
            if (jj < cnt) touch = subtile_may_touch(s_a[jj], s_b[jj], sxf, syf);
            uint64_t mask = __ballot(touch);
            // two staged entries per trip: their power/exp/alpha are independent (ILP), only the blend is serial
            while (mask) {
                const int j0 = k * 64 + __builtin_ctzll(mask);
                mask &= mask - 1;
                const bool two = mask != 0ull;
                const int j1 = two ? k * 64 + __builtin_ctzll(mask) : j0;
                mask &= mask - 1;
                const float4 a0 = s_a[j0], b0 = s_b[j0];
                const float4 a1 = s_a[j1], b1 = s_b[j1];
                const float c0 = s_cb[j0], c1 = s_cb[j1];
                const float dx0 = a0.x - pxf, dy0 = a0.y - pyf;
                const float dx1 = a1.x - pxf, dy1 = a1.y - pyf;
                const float pw0 = -0.5f * (a0.z * dx0 * dx0 + b0.x * dy0 * dy0) - a0.w * dx0 * dy0;
                const float pw1 = -0.5f * (a1.z * dx1 * dx1 + b1.x * dy1 * dy1) - a1.w * dx1 * dy1;
                const float al0 = fminf(kAlphaMax, b0.y * hs_exp(pw0));
                const float al1 = fminf(kAlphaMax, b1.y * hs_exp(pw1));
                blend_fwd(st, pw0, al0, b0.z, b0.w, c0, (uint32_t)(base + j0 + 1));
                if (two) blend_fwd(st, pw1, al1, b1.z, b1.w, c1, (uint32_t)(base + j1 + 1));
                if (__ballot(!st.done) == 0ull) { mask = 0; k = 4; }
            }
        }
    }
    const float T = st.T, C0 = st.C0, C1 = st.C1, C2 = st.C2;
    const uint32_t last = st.last;

    if (inside) {
        const int64_t HW = (int64_t)p.H * p.W;
        const int64_t pix = (int64_t)py * p.W + px;
        p.final_T[(int64_t)pose * HW + pix] = T;
        p.n_contrib[(int64_t)pose * HW + pix] = last;
        const float H0 = C0 + T * p.bg[0], H1 = C1 + T * p.bg[1], H2 = C2 + T * p.bg[2];
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

struct RenderBwd {
    int W, H, gx, gy, ntiles, N, flags;
    const uint2* ranges; const uint32_t* point_list; const float4* rec; const float* bg;
    const float* final_T; const uint32_t* n_contrib; const float* pose_hdr;
    const float* dL_dcolor; const float* dL_dhdr;
    float4* pair_grads;
    Crf crf;
    const float* exposure;
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

__global__ void __launch_bounds__(256) render_bwd_kernel(RenderBwd p) {
    __shared__ float4 s_a[256];
    __shared__ float4 s_b[256];
    __shared__ float4 s_c[256];
    __shared__ float s_acc[4][9][256];
    __shared__ uint64_t s_mask[4][4];
    __shared__ uint32_t s_max[4];

    const int vt = blockIdx.x;
    const int pose = vt / p.ntiles;
    const int tile = vt - pose * p.ntiles;
    const int tx = tile % p.gx, ty = tile / p.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sx = tx * kTile + (wave & 1) * 8, sy = ty * kTile + (wave >> 1) * 8;
    const int px = sx + (lane & 7), py = sy + (lane >> 3);
    const bool inside = px < p.W && py < p.H;
    const float pxf = (float)px, pyf = (float)py;
    const float sxf = (float)sx, syf = (float)sy;
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t pix = (int64_t)py * p.W + px;

    const uint2 range = p.ranges[vt];

    float T_final = 0.f, dL0 = 0.f, dL1 = 0.f, dL2 = 0.f;
    uint32_t last = 0;
    if (inside) {
        T_final = p.final_T[(int64_t)pose * HW + pix];
        last = p.n_contrib[(int64_t)pose * HW + pix];
        Crf c = p.crf;
        if (p.flags & HS_FLAG_HDR) c.dt = p.exposure[0];
        dL0 = pixel_grad(p, c, pose, 0, pix, HW);
        dL1 = pixel_grad(p, c, pose, 1, pix, HW);
        dL2 = pixel_grad(p, c, pose, 2, pix, HW);
    }
    const float bg_dot = (p.bg[0] * dL0 + p.bg[1] * dL1) + p.bg[2] * dL2;
    const float ddelx_dx = 0.5f * (float)p.W, ddely_dy = 0.5f * (float)p.H;

    const uint32_t wave_max = wave_max_u32(last);
    if (lane == 0) s_max[wave] = wave_max;
    __syncthreads();
    const int n_proc = (int)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));

    float T = T_final;
    float ar0 = 0.f, ar1 = 0.f, ar2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;

    const int nb = (n_proc + 255) / 256;
    for (int bi = nb - 1; bi >= 0; --bi) {
        const int base = bi * 256;
        const int cnt = min(256, n_proc - base);
        __syncthreads();  // previous batch's write-out finished
        if ((int)threadIdx.x < cnt) {
            const uint32_t id = p.point_list[range.x + base + threadIdx.x];
            const float4* r = p.rec + 3 * (int64_t)id;
            s_a[threadIdx.x] = r[0];
            s_b[threadIdx.x] = r[1];
            s_c[threadIdx.x] = r[2];
        }
        __syncthreads();
        uint64_t done_mask[4] = {0ull, 0ull, 0ull, 0ull};
        if (base < (int)wave_max) {
#pragma unroll 1
            for (int k = 3; k >= 0; --k) {
                const int jj = k * 64 + lane;
                bool touch = false;
                if (jj < cnt && base + jj < (int)wave_max) touch = subtile_may_touch(s_a[jj], s_b[jj], sxf, syf);
                uint64_t mask = __ballot(touch);
                uint64_t wrote = 0ull;
                while (mask) {
                    const int bit = 63 - __builtin_clzll(mask);
                    mask &= ~(1ull << bit);
                    const int j = k * 64 + bit;
                    const float4 a = s_a[j];
                    const float4 b = s_b[j];
                    const float cb = s_c[j].x;
                    const float dx = a.x - pxf, dy = a.y - pyf;
                    const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                    const float G = hs_exp(power);
                    const float alpha = fminf(kAlphaMax, b.y * G);
                    const bool act = ((uint32_t)(base + j) < last) && (power <= 0.f) && (alpha >= kAlphaMin);
                    if (__ballot(act) == 0ull) continue;
                    // branch-free per-lane update: inactive lanes keep their state and contribute zeros
                    const float one_m = 1.f - alpha;
                    const float rcp_one_m = __builtin_amdgcn_rcpf(one_m);
                    const float Tn = T * rcp_one_m;
                    T = act ? Tn : T;
                    const float dch = act ? alpha * T : 0.f;
                    const float la = act ? last_alpha : 0.f;  // la = 0 leaves accum_rec unchanged
                    const float sel = act ? 1.f : 0.f;
                    ar0 = la * lc0 + (1.f - la) * ar0;
                    ar1 = la * lc1 + (1.f - la) * ar1;
                    ar2 = la * lc2 + (1.f - la) * ar2;
                    lc0 = act ? b.z : lc0; lc1 = act ? b.w : lc1; lc2 = act ? cb : lc2;
                    last_alpha = act ? alpha : last_alpha;
                    float dL_dalpha = ((b.z - ar0) * dL0 + (b.w - ar1) * dL1) + (cb - ar2) * dL2;
                    dL_dalpha = dL_dalpha * T + (-T_final * rcp_one_m) * bg_dot;
                    dL_dalpha *= sel;
                    const float dL_dG = b.y * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * a.z - gdy * a.w;
                    const float dG_ddely = -gdy * b.x - gdx * a.w;
                    float g[9];
                    g[0] = dL_dG * dG_ddelx * ddelx_dx;
                    g[1] = dL_dG * dG_ddely * ddely_dy;
                    g[2] = -0.5f * gdx * dx * dL_dG;
                    g[3] = -gdx * dy * dL_dG;
                    g[4] = -0.5f * gdy * dy * dL_dG;
                    g[5] = G * dL_dalpha;
                    g[6] = dch * dL0; g[7] = dch * dL1; g[8] = dch * dL2;
                    float q0, q1, q2;
                    wave_reduce9(g, q0, q1, q2);
                    // rows 0..3 of q0 hold totals of g0,g2,g1,g3; of q1: g4,g6,g5,g7; rows 0,1 of q2: g8
                    if ((lane & 15) == 0) {
                        const int row = lane >> 4;
                        const int v0 = ((row & 1) << 1) | (row >> 1);
                        s_acc[wave][v0][j] = q0;
                        s_acc[wave][4 + v0][j] = q1;
                        if (row == 0) s_acc[wave][8][j] = q2;
                    }
                    wrote |= 1ull << bit;
                }
                done_mask[k] = wrote;
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_mask[wave][k] = done_mask[k];
        }
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            const int t = threadIdx.x;
            float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if ((s_mask[w][t >> 6] >> (t & 63)) & 1ull) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) v[q] += s_acc[w][q][t];
                }
            }
            // slot of the (tile, instance) pair in duplicateWithKeys order
            const float4 a = s_a[t];
            const float4 c = s_c[t];
            const int rad = __float_as_int(c.z);
            const uint32_t off = __float_as_uint(c.w);
            const int rminx = min(p.gx, max(0, (int)((a.x - (float)rad) / (float)kTile)));
            const int rminy = min(p.gy, max(0, (int)((a.y - (float)rad) / (float)kTile)));
            const int rmaxx = min(p.gx, max(0, (int)((a.x + (float)rad + (float)(kTile - 1)) / (float)kTile)));
            const int64_t slot = (int64_t)off + (int64_t)(ty - rminy) * (rmaxx - rminx) + (tx - rminx);
            float4* o = p.pair_grads + 3 * slot;
            o[0] = make_float4(v[0], v[1], v[2], v[3]);
            o[1] = make_float4(v[4], v[5], v[6], v[7]);
            o[2] = make_float4(v[8], 0.f, 0.f, 0.f);
        }
    }
}

// CRF-table and exposure gradients: fixed grid, LDS table per block, one partial row per block
// ([3K] table + [1] exposure), reduced by crf_reduce_kernel.
__global__ void __launch_bounds__(256) crf_grad_kernel(int64_t HW, int N, int flags, const float* pose_hdr, Crf crf,
                                                       const float* exposure, const float* dL_dcolor, float* partials) {
    extern __shared__ float s_tab[];  // 3K + 4
    const int K3 = 3 * crf.K;
    for (int i = threadIdx.x; i < K3 + 4; i += 256) s_tab[i] = 0.f;
    __syncthreads();
    crf.dt = exposure[0];
    const bool blur_hdr = (flags & HS_FLAG_BLUR_HDR) && N > 1;
    const int npose = blur_hdr ? 1 : N;
    const float invN = 1.f / (float)N;
    const float scale = (float)(crf.K - 1) / (crf.umax - crf.umin);
    float gexp = 0.f;
    const int64_t total = (int64_t)npose * 3 * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i / (3 * HW));
        const int64_t r = i - (int64_t)k * 3 * HW;
        const int ch = (int)(r / HW);
        const int slot = blur_hdr ? N : k;
        const float Hv = pose_hdr[(int64_t)slot * 3 * HW + r];
        const float g = dL_dcolor[r] * (blur_hdr ? 1.f : invN);
        int idx; float f, xv; bool in;
        crf_locate(crf, Hv, idx, f, xv, in);
        atomicAdd(&s_tab[ch * crf.K + idx], (1.f - f) * g);
        atomicAdd(&s_tab[ch * crf.K + idx + 1], f * g);
        if (in) {
            const float* t = crf.table + ch * crf.K;
            gexp += g * (t[idx + 1] - t[idx]) * scale / xv * Hv;
        }
    }
    gexp = wave_sum_hi(gexp);
    if ((threadIdx.x & 63) == 63) atomicAdd(&s_tab[K3], gexp);
    __syncthreads();
    for (int i = threadIdx.x; i < K3 + 1; i += 256) partials[(int64_t)blockIdx.x * (K3 + 1) + i] = s_tab[i];
}

// One wave per output element: lanes stride over the per-block partial rows (fixed order -> reproducible).
__global__ void __launch_bounds__(256) crf_reduce_kernel(const float* partials, int nblk, int K3, float* d_table,
                                                         float* d_exposure) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i > K3) return;
    float acc = 0.f;
    for (int b = lane; b < nblk; b += 64) acc += partials[(int64_t)b * (K3 + 1) + i];
    acc = wave_sum_hi(acc);
    if (lane == 63) {
        if (i < K3) { if (d_table) d_table[i] = acc; }
        else if (d_exposure) d_exposure[0] = acc;
    }
}

}  // namespace

constexpr int kCrfBlocks = 512;

int launch_render_fwd(const hs_fwd_args& a, const hs_layout& L, hipStream_t s) {
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
    render_fwd_kernel<<<p.ntiles * d.n_poses, 256, 0, s>>>(p);
    HS_LAUNCH_CHECK();
    if (d.n_poses > 1) {
        const int64_t HW = (int64_t)d.W * d.H;
        resolve_kernel<<<ceil_div(3 * HW, 256), 256, 0, s>>>(HW, d.n_poses, a.flags, p.pose_hdr, p.crf, a.exposure,
                                                            a.out_color, a.out_hdr);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}

int launch_render_bwd(const hs_bwd_args& a, const hs_layout& L, hipStream_t s) {
    const hs_dims& d = a.dims;
    RenderBwd p;
    p.W = d.W; p.H = d.H; p.gx = (d.W + kTile - 1) / kTile; p.gy = (d.H + kTile - 1) / kTile;
    p.ntiles = p.gx * p.gy; p.N = d.n_poses; p.flags = a.flags;
    const char* bin = (const char*)a.binning; const char* img = (const char*)a.image; const char* geom = (const char*)a.geom;
    p.ranges = (const uint2*)(bin + L.ranges); p.point_list = (const uint32_t*)(bin + L.point_list);
    p.rec = (const float4*)(geom + L.rec); p.bg = a.bg;
    p.final_T = (const float*)(img + L.final_T); p.n_contrib = (const uint32_t*)(img + L.n_contrib);
    p.pose_hdr = (const float*)(img + L.pose_hdr);
    p.dL_dcolor = a.dL_dout_color; p.dL_dhdr = a.dL_dout_hdr;
    p.pair_grads = (float4*)((char*)a.bwd + L.pair_grads);
    p.crf.table = a.crf_table; p.crf.K = a.crf_K; p.crf.umin = a.crf_umin; p.crf.umax = a.crf_umax; p.crf.dt = 1.f;
    p.exposure = a.exposure;
    // pairs beyond a tile's deepest contributor are never visited: their records must read as zero
    HS_HIP_CHECK(hipMemsetAsync(p.pair_grads, 0, (size_t)d.capacity * kPairFloats * sizeof(float), s));
    render_bwd_kernel<<<p.ntiles * d.n_poses, 256, 0, s>>>(p);
    HS_LAUNCH_CHECK();
    if ((a.flags & HS_FLAG_HDR) && (a.dL_dcrf_table || a.dL_dexposure)) {
        const int64_t HW = (int64_t)d.W * d.H;
        float* partials = (float*)((char*)a.bwd + L.crf_partials);
        const int K3 = 3 * a.crf_K;
        crf_grad_kernel<<<kCrfBlocks, 256, (K3 + 4) * sizeof(float), s>>>(HW, d.n_poses, a.flags, p.pose_hdr, p.crf,
                                                                         a.exposure, a.dL_dout_color, partials);
        crf_reduce_kernel<<<ceil_div(K3 + 1, 4), 256, 0, s>>>(partials, kCrfBlocks, K3, a.dL_dcrf_table,
                                                                a.dL_dexposure);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}

int crf_partial_floats(int K) { return kCrfBlocks * (3 * K + 1); }

}  // namespace hs
