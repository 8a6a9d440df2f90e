// preprocess forward / backward and markVisible for gfx950.
//
// Rules: SURVEY.md 8(a) rows a4 (preprocess fwd), a11 (computeCov2D bwd), a12 (preprocess bwd),
// a14 (markVisible).  No reference file exists to cite (/root/reference has no code).
//
// This translation unit is compiled with -ffp-contract=off: depth bits, pixel centre, conic,
// radius and the tile rectangle feed integer decisions (sort keys, tile lists) that must be
// bit-identical to the CPU oracle, so only IEEE + - * / sqrt ceil in a fixed order are used.
// One thread per instance (pose, Gaussian); the kernels are pure streaming (HBM-bound on the SH
// read/write), so loads are issued as wide as the [P,*] layouts allow.
#include "hs_common.h"


namespace hs {

namespace {

__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

struct Ewa {
    float tx, ty, tz;
    bool clamp_x, clamp_y;
    float a0[3], a1[3];
    float fx, fy;
};

__device__ __forceinline__ void quat_to_R(const float* q, float* R) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = R diag((mod*s)^2) R^T, upper triangle (xx,xy,xz,yy,yz,zz).
__device__ __forceinline__ void cov3d_from_scale_rot(const float* scale, float mod, const float* q, float* c6) {
    float R[9], M[9];
    quat_to_R(q, R);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = mod * scale[k];
#pragma unroll
        for (int i = 0; i < 3; ++i) M[3 * k + i] = s * R[3 * i + k];
    }
#define HS_SIG(i, j) ((M[0 + i] * M[0 + j] + M[3 + i] * M[3 + j]) + M[6 + i] * M[6 + j])
    c6[0] = HS_SIG(0, 0); c6[1] = HS_SIG(0, 1); c6[2] = HS_SIG(0, 2);
    c6[3] = HS_SIG(1, 1); c6[4] = HS_SIG(1, 2); c6[5] = HS_SIG(2, 2);
#undef HS_SIG
}

__device__ __forceinline__ void ewa_setup(const float* V, int W, int H, float tanfovx, float tanfovy, float pvx,
                                          float pvy, float pvz, Ewa& e) {
    float fx = (float)W / (2.f * tanfovx);
    float fy = (float)H / (2.f * tanfovy);
    float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    float txtz = pvx / pvz, tytz = pvy / pvz;
    e.clamp_x = (txtz < -limx) || (txtz > limx);
    e.clamp_y = (tytz < -limy) || (tytz > limy);
    float tx = fminf(limx, fmaxf(-limx, txtz)) * pvz;
    float ty = fminf(limy, fmaxf(-limy, tytz)) * pvz;
    float tz = pvz;
    float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        e.a0[j] = J00 * V[4 * j + 0] + J02 * V[4 * j + 2];
        e.a1[j] = J11 * V[4 * j + 1] + J12 * V[4 * j + 2];
    }
    e.tx = tx; e.ty = ty; e.tz = tz; e.fx = fx; e.fy = fy;
}

__device__ __forceinline__ void sym_mul(const float* s6, const float* v, float* out) {
    out[0] = (s6[0] * v[0] + s6[1] * v[1]) + s6[2] * v[2];
    out[1] = (s6[1] * v[0] + s6[3] * v[1]) + s6[4] * v[2];
    out[2] = (s6[2] * v[0] + s6[4] * v[1]) + s6[5] * v[2];
}
__device__ __forceinline__ float dot3(const float* a, const float* b) {
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* b) {
    b[0] = SH_C0;
    if constexpr (DEG >= 1) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    }
    if constexpr (DEG >= 2) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2.f * zz - xx - yy);
        b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
        if constexpr (DEG >= 3) {
            b[9] = SH_C3[0] * y * (3.f * xx - yy);
            b[10] = SH_C3[1] * xy * z;
            b[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
            b[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
            b[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
            b[14] = SH_C3[5] * z * (xx - yy);
            b[15] = SH_C3[6] * x * (xx - 3.f * yy);
        }
    }
}

// d b[k] / d(x,y,z), components treated as independent.
template <int DEG>
__device__ __forceinline__ void sh_basis_grad(float x, float y, float z, float (*g)[3]) {
#pragma unroll
    for (int k = 0; k < (DEG + 1) * (DEG + 1); ++k) g[k][0] = g[k][1] = g[k][2] = 0.f;
    if constexpr (DEG >= 1) {
        g[1][1] = -SH_C1; g[2][2] = SH_C1; g[3][0] = -SH_C1;
    }
    if constexpr (DEG >= 2) {
        float xx = x * x, yy = y * y, zz = z * z;
        g[4][0] = SH_C2[0] * y; g[4][1] = SH_C2[0] * x;
        g[5][1] = SH_C2[1] * z; g[5][2] = SH_C2[1] * y;
        g[6][0] = SH_C2[2] * -2.f * x; g[6][1] = SH_C2[2] * -2.f * y; g[6][2] = SH_C2[2] * 4.f * z;
        g[7][0] = SH_C2[3] * z; g[7][2] = SH_C2[3] * x;
        g[8][0] = SH_C2[4] * 2.f * x; g[8][1] = SH_C2[4] * -2.f * y;
        if constexpr (DEG >= 3) {
            g[9][0] = SH_C3[0] * 6.f * x * y; g[9][1] = SH_C3[0] * (3.f * xx - 3.f * yy);
            g[10][0] = SH_C3[1] * y * z; g[10][1] = SH_C3[1] * x * z; g[10][2] = SH_C3[1] * x * y;
            g[11][0] = SH_C3[2] * -2.f * x * y; g[11][1] = SH_C3[2] * (4.f * zz - xx - 3.f * yy);
            g[11][2] = SH_C3[2] * 8.f * y * z;
            g[12][0] = SH_C3[3] * -6.f * x * z; g[12][1] = SH_C3[3] * -6.f * y * z;
            g[12][2] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
            g[13][0] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); g[13][1] = SH_C3[4] * -2.f * x * y;
            g[13][2] = SH_C3[4] * 8.f * x * z;
            g[14][0] = SH_C3[5] * 2.f * x * z; g[14][1] = SH_C3[5] * -2.f * y * z; g[14][2] = SH_C3[5] * (xx - yy);
            g[15][0] = SH_C3[6] * (3.f * xx - 3.f * yy); g[15][1] = SH_C3[6] * -6.f * x * y;
        }
    }
}

struct PreFwd {
    int P, M, W, H, N;
    float tanfovx, tanfovy, mod;
    const float* view; const float* proj; const float* campos;
    const float* means; const float* opac; const float* shs; const float* colors; const float* scales;
    const float* rots; const float* cov_pre;
    float4* rec; float* depth; int* radii_inst; uint32_t* tiles; float* cov3D; uint8_t* clamped;
    uint2* binfo;  // tile rectangle {min_x | min_y << 16, width | height << 16} for the pair emission
    int* radii_out;
    // single-enqueue forward (binning workspace already there): the start of the binning stage rides along --
    // depth-sort keys/values of the instances, the instance count word, cleared tile ranges; else null
    uint2* depth_pairs; hs_counters* counters; uint2* ranges; int64_t n_vtiles;
    uint32_t* sort_zero; int64_t n_sort_zero;   // scratch of the depth sort that follows (binning.hip), cleared here ...
    int64_t bits_at;                            // ... except the depth-bits words [bits_at, bits_at + kDepthBitsWords): this
    uint32_t frame_tag;                         // kernel ORs the depth keys into them, stamped with the frame's tag
    uint32_t* pair_zero; int64_t n_pair_zero;   // ... and of the pair emission's scan + the tile sort
    bool antialias;
    int act;  // radiance activation: 0 relu_shift, 1 exp, 2 softplus
};

#ifndef HS_TUNE_PF_REC_LDS
#define HS_TUNE_PF_REC_LDS 1
#endif
#ifndef HS_TUNE_PF_EARLY
#define HS_TUNE_PF_EARLY 1
#endif
// a4.  instance = pose * P + g.
// Occupancy the register allocator may be held to (0: none): with its SH row in flight the degree-3 instantiation takes 98
// registers, which the allocation granule of 8 turns into FOUR waves per SIMD; asked for five it fits into 96 unspilled.
#ifndef HS_TUNE_PF_WAVES
#define HS_TUNE_PF_WAVES 5
#endif
#if HS_TUNE_PF_WAVES > 0
#define HS_PF_OCC __attribute__((amdgpu_waves_per_eu(HS_TUNE_PF_WAVES, HS_TUNE_PF_WAVES)))
#else
#define HS_PF_OCC
#endif
template <int DEG>
__global__ void __launch_bounds__(256) HS_PF_OCC preprocess_fwd_kernel(PreFwd p) {
    __shared__ uint32_t s_two[2];       // OR of the workgroup's visible depth keys / of their complements
    if (p.depth_pairs) {                // (uniform)
        if (threadIdx.x < 2) s_two[threadIdx.x] = 0u;
        __syncthreads();
    }
    uint32_t depth_key = 0xFFFFFFFFu;   // this thread's key of the depth sort (all ones: culled, or no instance)
    // Block -> (chunk of 256 Gaussians, pose): workgroups are dealt to the 8 XCDs round-robin by blockIdx, so XCD x runs
    // its blocks x, x + 8, x + 16, ... in order; those are made the N poses of chunk x, then of chunk x + 8, ... -- the
    // poses of a chunk follow each other on ONE XCD and read the chunk's SH rows, scales and rotations (236 bytes
    // per Gaussian at degree 3) through that XCD's L2 once instead of N times from memory.  N = 1: chunk = blockIdx.
    const int64_t lin = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int pose = seq % p.N;
    const int64_t g64 = ((int64_t)(seq / p.N) * 8 + xcd) * 256 + threadIdx.x;
    uint32_t ntiles = 0;
#if HS_TUNE_PF_REC_LDS
    // The 64-byte render records leave through a wave-private LDS stage, so a store instruction of the wave covers 1 KB of
    // consecutive addresses instead of 16 bytes in each of 64 sectors (round 5: preprocess_fwd + binning -7 us at c3)
    __shared__ float4 s_rec[4][64 * 4];
    float4 rec_a = make_float4(0.f, 0.f, 0.f, 0.f), rec_b = rec_a, rec_c = rec_a;
#endif
    if (g64 < (int64_t)p.P) {
    const int g = (int)g64;
    const int64_t idx = (int64_t)pose * p.P + g;
    const float* V = p.view + 16 * pose;
    const float* PM = p.proj + 16 * pose;

    int my_radius = 0;
    uint2 bi = make_uint2(0u, 0u);
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
    float depth = 0.f;
    uint8_t clampbits = 0;

    // Loads first (round 5): position, scale, rotation and opacity are requested together, and the SH row of a Gaussian
    // whose centre projects into (or near) the frame right behind them -- 192 bytes at degree 3 whose latency then runs
    // under the projection / covariance arithmetic (a few hundred dependent instructions with their IEEE divisions)
    // instead of after it.  The test is a HINT: a Gaussian that turns out visible without having passed it loads its row
    // where it always did; one that passed it and is culled after all has read a row for nothing.
    const float x = p.means[3 * g], y = p.means[3 * g + 1], z = p.means[3 * g + 2];
    float sc[3] = {0.f, 0.f, 0.f};
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float s6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.cov_pre) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s6[k] = p.cov_pre[6 * g + k];
    } else {
        sc[0] = p.scales[3 * g]; sc[1] = p.scales[3 * g + 1]; sc[2] = p.scales[3 * g + 2];
        q4 = reinterpret_cast<const float4*>(p.rots)[g];
    }
    const float opac_in = p.opac[g];
    const float pvx = xform_row(V, 0, x, y, z), pvy = xform_row(V, 1, x, y, z), pvz = xform_row(V, 2, x, y, z);
    const float phx = xform_row(PM, 0, x, y, z), phy = xform_row(PM, 1, x, y, z), phw = xform_row(PM, 3, x, y, z);
    const float pw = 1.0f / (phw + 0.0000001f);
    const float ppx = phx * pw, ppy = phy * pw;
    constexpr int NCF = (DEG + 1) * (DEG + 1);
    float shr[3 * NCF];
#pragma unroll
    for (int k = 0; k < 3 * NCF; ++k) shr[k] = 0.f;
    bool have_sh = false;
    bool vis = false;
#if HS_TUNE_PF_EARLY
    const bool early = !p.colors && pvz > 0.2f && fabsf(ppx) < 1.5f && fabsf(ppy) < 1.5f;
    if (early) {
        const float* sh = p.shs + (int64_t)g * p.M * 3;
#pragma unroll
        for (int k = 0; k < 3 * NCF; ++k) shr[k] = sh[k];
        have_sh = true;
    }
#endif
    if (!p.cov_pre && (pose == 0 || pvz > 0.2f)) {
        float q[4] = {q4.x, q4.y, q4.z, q4.w};
        cov3d_from_scale_rot(sc, p.mod, q, s6);
        // (the 3-D covariance is NOT kept: the backward recomputes it from the scale and the rotation it loads anyway -- 24
        // bytes per Gaussian less to write here and to read there; HS_STAGE_OFFSETS fills the array for inspection)
    }

    if (pvz > 0.2f) {

        Ewa e;
        ewa_setup(V, p.W, p.H, p.tanfovx, p.tanfovy, pvx, pvy, pvz, e);
        float u0[3], u1[3];
        sym_mul(s6, e.a0, u0);
        sym_mul(s6, e.a1, u1);
        const float ca = dot3(e.a0, u0) + 0.3f;
        const float cb = dot3(e.a1, u0);
        const float cc = dot3(e.a1, u1) + 0.3f;
        const float det = ca * cc - cb * cb;
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            const float conA = cc * det_inv, conB = -cb * det_inv, conC = ca * det_inv;
            const float mid = 0.5f * (ca + cc);
            const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lam1 = mid + disc, lam2 = mid - disc;
            const int rad = (int)ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
            const float pix_x = ((ppx + 1.0f) * (float)p.W - 1.0f) * 0.5f;
            const float pix_y = ((ppy + 1.0f) * (float)p.H - 1.0f) * 0.5f;
            const int gx = (p.W + kTile - 1) / kTile, gy = (p.H + kTile - 1) / kTile;
            const int rminx = min(gx, max(0, (int)((pix_x - (float)rad) / (float)kTile)));
            const int rminy = min(gy, max(0, (int)((pix_y - (float)rad) / (float)kTile)));
            const int rmaxx = min(gx, max(0, (int)((pix_x + (float)rad + (float)(kTile - 1)) / (float)kTile)));
            const int rmaxy = min(gy, max(0, (int)((pix_y + (float)rad + (float)(kTile - 1)) / (float)kTile)));
            const int area = (rmaxx - rminx) * (rmaxy - rminy);
            if (area != 0) {
                vis = true;
                my_radius = rad;
                ntiles = (uint32_t)area;
                bi = make_uint2((uint32_t)rminx | ((uint32_t)rminy << 16),
                                (uint32_t)(rmaxx - rminx) | ((uint32_t)(rmaxy - rminy) << 16));
                depth = pvz;
                ra = make_float4(pix_x, pix_y, conA, conB);
                float opac = opac_in;
                if (p.antialias) {  // energy compensation of the 0.3-pixel dilation (newer published rasterizer)
                    const float det0 = (ca - 0.3f) * (cc - 0.3f) - cb * cb;
                    opac = opac * sqrtf(fmaxf(0.000025f, det0 / det));
                }
                rb = make_float4(conC, opac, 0.f, 0.f);
                rc = make_float4(0.f, pvz, __int_as_float(rad), 0.f);
            }
        }
    }
    // (round 5, measured and removed: the wave loading the SH rows of its 64 Gaussians TOGETHER -- 48-byte pieces of ~21
    // rows per load instruction instead of 16 bytes of 64 rows -- and handing them to their owners through this LDS stage
    // chunk by chunk: preprocess_fwd + binning +6 us at c3, +80 us at c4.  The per-thread row loads are not what holds
    // the kernel back once they are issued early; the 64-byte record STORES were, see s_rec)
    if (vis) {   // colour of the survivors
        float col[3];
        if (p.colors) {
            col[0] = p.colors[3 * g]; col[1] = p.colors[3 * g + 1]; col[2] = p.colors[3 * g + 2];
        } else {
            const float* cp = p.campos + 3 * pose;
            const float dx = x - cp[0], dy = y - cp[1], dz = z - cp[2];
            const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float ux = dx / len, uy = dy / len, uz = dz / len;
            float b[(DEG + 1) * (DEG + 1)];
            sh_basis<DEG>(ux, uy, uz, b);
            if (!have_sh) {
                const float* sh = p.shs + (int64_t)g * p.M * 3;
#pragma unroll
                for (int k = 0; k < 3 * NCF; ++k) shr[k] = sh[k];
            }
            float acc[3] = {b[0] * shr[0], b[0] * shr[1], b[0] * shr[2]};
#pragma unroll
            for (int k = 1; k < (DEG + 1) * (DEG + 1); ++k) {
                acc[0] = acc[0] + b[k] * shr[3 * k + 0];
                acc[1] = acc[1] + b[k] * shr[3 * k + 1];
                acc[2] = acc[2] + b[k] * shr[3 * k + 2];
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                if (p.act == 1) {
                    col[ch] = expf(acc[ch]);
                } else if (p.act == 2) {
                    col[ch] = acc[ch] > 20.f ? acc[ch] : log1pf(expf(acc[ch]));
                } else {
                    float v = acc[ch] + 0.5f;
                    if (v < 0.f) clampbits |= (uint8_t)(1u << ch);
                    col[ch] = fmaxf(v, 0.f);
                }
            }
        }
        rb.z = col[0]; rb.w = col[1]; rc.x = col[2];
    }
#if HS_TUNE_PF_REC_LDS
    rec_a = ra; rec_b = rb; rec_c = rc;
#else
    p.rec[kRecF4 * idx + 0] = ra;
    p.rec[kRecF4 * idx + 1] = rb;
    p.rec[kRecF4 * idx + 2] = rc;
    p.rec[kRecF4 * idx + 3] = make_float4(0.f, 0.f, 0.f, 0.f);  // whole sector written: no partial-line write-back
#endif
    p.depth[idx] = depth;
    p.radii_inst[idx] = my_radius;
    p.tiles[idx] = ntiles;
    p.binfo[idx] = bi;
    p.clamped[idx] = clampbits;
    if (p.N == 1) p.radii_out[g] = my_radius;
    else if (my_radius > 0) atomicMax(p.radii_out + g, my_radius);
    if (p.depth_pairs) {  // (depth bits, instance) for the depth sort; culled instances get the largest key: they sort to the end
        if (my_radius > 0) depth_key = __float_as_uint(depth);
        p.depth_pairs[idx] = make_uint2(depth_key, (uint32_t)idx);
    }
    }  // g < P
#if HS_TUNE_PF_REC_LDS
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float4* w = s_rec[wave];
        const int sw = (lane >> 2) & 3;       // swizzle: the 16 lanes of a quarter-wave land on 16 different bank groups
        w[lane * 4 + (0 ^ sw)] = rec_a;
        w[lane * 4 + (1 ^ sw)] = rec_b;
        w[lane * 4 + (2 ^ sw)] = rec_c;
        w[lane * 4 + (3 ^ sw)] = make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the wave's 64 instances are consecutive (same pose, consecutive Gaussians): one 4 KB stretch of records
        const int64_t g_w0 = g64 - lane;
        const int nvalid = (int)min((int64_t)64, (int64_t)p.P - g_w0);
        float4* dst = p.rec + kRecF4 * ((int64_t)pose * p.P + g_w0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = j * 64 + lane, r = f >> 2, k = f & 3;
            if (r < nvalid) dst[f] = w[r * 4 + (k ^ ((r >> 2) & 3))];
        }
    }
#endif
    if (p.depth_pairs) {
        // which bits of the visible depth keys vary (the depth sort sorts only those: binning.hip)
        depth_bits_accumulate(depth_key, depth_key != 0xFFFFFFFFu,
                              reinterpret_cast<unsigned long long*>(p.sort_zero + p.bits_at), p.frame_tag, s_two);
        for (int64_t t = lin; t < p.n_vtiles; t += (int64_t)gridDim.x * 256) p.ranges[t] = make_uint2(0u, 0u);
        for (int64_t t = lin; t < p.n_sort_zero; t += (int64_t)gridDim.x * 256)
            if (t < p.bits_at || t >= p.bits_at + kDepthBitsWords) p.sort_zero[t] = 0u;
        for (int64_t t = lin; t < p.n_pair_zero; t += (int64_t)gridDim.x * 256) p.pair_zero[t] = 0u;
        if (lin == 0) {
            p.counters->overflow = 0u;
            p.counters->reserved[1] = (uint32_t)((int64_t)p.P * p.N);
            p.counters->reserved[2] = 0u; p.counters->reserved[3] = 0u;   // tile-queue counters of the render kernels
            p.counters->reserved[4] = 0u;                                 // look-back helps of this frame's binning
        }
    }
}

// Inspection only (HS_STAGE_OFFSETS): the 3-D covariances as the forward computed them, into the geometry workspace
__global__ void __launch_bounds__(256) cov3d_kernel(int P, float mod, const float* scales, const float* rots, const float* cov_pre,
                                                    float* cov3D) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= P) return;
    float s6[6];
    if (cov_pre) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s6[k] = cov_pre[6 * (int64_t)g + k];
    } else {
        float sc[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
        const float4 q4 = reinterpret_cast<const float4*>(rots)[g];
        float q[4] = {q4.x, q4.y, q4.z, q4.w};
        cov3d_from_scale_rot(sc, mod, q, s6);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) cov3D[6 * (int64_t)g + k] = s6[k];
}

__global__ void mark_visible_kernel(int P, const float* means, const float* V, uint8_t* vis) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    vis[i] = xform_row(V, 2, means[3 * i], means[3 * i + 1], means[3 * i + 2]) > 0.2f;
}

// ------------------------------------------------------------------------------------------------
// a11 + a12 + the per-Gaussian segmented sum of the render-backward pair records.
// One thread per Gaussian, looping over poses; sums are formed in a fixed order (pair slots
// ascending, poses ascending) so the gradients are bitwise reproducible run to run.
// ------------------------------------------------------------------------------------------------
// Per-instance sum of the render-backward pair records.  Threads walk the instances in DEPTH order, the order
// in which duplicateWithKeys laid the pair slots out, so neighbouring lanes read neighbouring slot runs (coalesced
// through L1/L2) and each thread adds its own run front to back: a fixed order, hence bitwise reproducible sums.
// Unflagged records were never written this backward (their tile's replay stopped before them): skipped by select.
__global__ void __launch_bounds__(256) pair_segsum_kernel(SegsumArgs sg, CrfReduce crf_reduce) {
    // (the second stage of the CRF-table gradient rides on this launch's first workgroups when the same call computed the
    // first: hs_common.h, CrfReduce)
    for (int b = blockIdx.x; b < crf_reduce.nblocks; b += gridDim.x) crf_reduce_block(crf_reduce, b);
    pair_segsum_body(sg, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

struct PreBwd {
    int P, M, W, H, N;
    float tanfovx, tanfovy, mod;
    const float* view; const float* proj; const float* campos;
    const float* means; const float* shs; const float* scales; const float* rots; const float* opac;
    bool has_colors_precomp, has_cov_pre, antialias;
    int act;  // radiance activation: 0 relu_shift, 1 exp, 2 softplus
    const float4* rec; const int* radii_inst; const uint32_t* tiles; const uint32_t* offsets; const float* cov_pre;
    const uint8_t* clamped;
    const float4* inst_grads;
    float* d_means3D; float* d_means2D; float* d_opac; float* d_shs; float* d_colors; float* d_scales;
    float* d_rots; float* d_cov;
    float* dens_grad; float* dens_denom; int* dens_radii;  // densification statistics, updated in place, or null
    float* pose_partials;  // [blocks][N][kPoseVals] or null
    int blk0;              // first block of this launch (hs_bwd_args.g_begin / kPreBwdBlock: a chunk of the Gaussians)
};

// Camera-pose gradient terms per pose: 12 view-matrix entries (flat 4j+i, i<3), 12 projection entries (rows 0,1,3),
// 3 camera-centre entries, padded to 28.
constexpr int kPoseVals = 28;

constexpr int kPreBwdBlock = 128;

// Block-cooperative copies between `rows` consecutive [M3]-float rows in HBM and LDS rows of stride `ld` (= M3 + 1,
// conflict-free for one-row-per-thread access).  16-byte global accesses when a row is a multiple of four floats
// (M = 4, 16: SH degree 1, 3) -- four times fewer memory instructions in flight per byte; rows start 16-byte
// aligned because the block's first row index is a multiple of kPreBwdBlock.
__device__ __forceinline__ void stage_rows_in(float* s_rows, const float* src, int rows, int M3, int ld) {
    if ((M3 & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int q = M3 >> 2;
        const float4* src4 = reinterpret_cast<const float4*>(src);
        for (int i = threadIdx.x; i < rows * q; i += kPreBwdBlock) {
            const float4 v = src4[i];
            const int r = i / q;
            float* d = s_rows + r * ld + ((i - r * q) << 2);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < rows * M3; i += kPreBwdBlock) s_rows[(i / M3) * ld + (i % M3)] = src[i];
    }
}
// The same for a FULL block whose rows are Q float4 long, Q known at compile time (M = (DEG + 1)^2 with 3M a multiple of
// four: degree 3 -> Q = 12, degree 1 -> Q = 3): every thread issues its Q loads back to back and only then writes the LDS.
// The run-time loop above compiles to one load -> wait -> four LDS writes per trip, i.e. ONE 16-byte load in flight per
// thread: at three waves per SIMD that is 12 KB in flight per CU while the block does nothing else (round 5; same-box A/B
// in profiles/README.md).
#ifndef HS_TUNE_PB_UNROLL
#define HS_TUNE_PB_UNROLL 1
#endif
template <int Q>
__device__ __forceinline__ void stage_rows_issue(float4 (&v)[Q > 0 ? Q : 1], const float* src) {
    const float4* src4 = reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int j = 0; j < Q; ++j) v[j] = src4[j * kPreBwdBlock + threadIdx.x];
}
template <int Q>
__device__ __forceinline__ void stage_rows_commit(float* s_rows, const float4 (&v)[Q > 0 ? Q : 1], int ld) {
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int i = j * kPreBwdBlock + threadIdx.x;
        const int r = i / Q;
        float* d = s_rows + r * ld + ((i - r * Q) << 2);
        d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
    }
}
__device__ __forceinline__ void stage_rows_out(float* dst, const float* s_rows, int rows, int M3, int ld) {
    if ((M3 & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const int q = M3 >> 2;
        float4* dst4 = reinterpret_cast<float4*>(dst);
        for (int i = threadIdx.x; i < rows * q; i += kPreBwdBlock) {
            const int r = i / q;
            const float* sr = s_rows + r * ld + ((i - r * q) << 2);
            // non-temporal: the gradient rows are this library's last word on them (the optimiser reads them much later),
            // and keeping 192 MB of them out of the L2 leaves it to the rows being READ: stage -9 us at c3 (same-box A/B;
            // the same hint on the render backward's pair records or on the segmented sum's reads of them costs time)
            typedef float v4f __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(v4f{sr[0], sr[1], sr[2], sr[3]}, reinterpret_cast<v4f*>(dst4) + i);
        }
    } else {
        for (int i = threadIdx.x; i < rows * M3; i += kPreBwdBlock) dst[i] = s_rows[(i / M3) * ld + (i % M3)];
    }
}

// The [P, M, 3] SH tensors (input coefficients and their gradient) are the bulk of this kernel's bytes.  A thread
// owns one Gaussian = one 12*M-byte row, so direct per-thread access would touch 64 different rows per wave
// instruction; instead the block's rows are moved between HBM and LDS with fully coalesced accesses and each thread
// works on its row in LDS (row stride M*3+1 words: conflict-free).
// SHG = false: the SH-coefficient gradient is not formed here (view-parallel exchange, hs_sh_backward_views).
// Occupancy the register allocator may be held to (0: none, the default): 128 rows of 3M + 1 floats in LDS (25 KB at degree
// 3) allow six workgroups = three waves per SIMD; the degree-3 instantiation takes 183 registers = two.  Measured at c3
// (round 5): forced to three waves (168 registers, 14 spilled) the stage is 1 us FASTER at best -- the kernel streams at
// 4.9 TB/s either way.
#ifndef HS_TUNE_PB_WAVES
#define HS_TUNE_PB_WAVES 0
#endif
#if HS_TUNE_PB_WAVES > 0
#define HS_PB_OCC __attribute__((amdgpu_waves_per_eu(HS_TUNE_PB_WAVES, HS_TUNE_PB_WAVES)))
#else
#define HS_PB_OCC
#endif
template <int DEG, bool POSE, bool SHG>
__global__ void __launch_bounds__(kPreBwdBlock) HS_PB_OCC preprocess_bwd_kernel(PreBwd p) {
    extern __shared__ float s_sh[];  // [kPreBwdBlock][M*3 + 1]
    __shared__ float s_pose[kPreBwdBlock / 64][kPoseVals];
    constexpr int NC = (DEG + 1) * (DEG + 1);
    const int blk = (int)blockIdx.x + p.blk0;
    const int g0 = blk * kPreBwdBlock;
    const int g = g0 + threadIdx.x;
    const int M3 = p.M * 3, ld = M3 + 1;
    const int rows = min(kPreBwdBlock, p.P - g0);
    const bool stage_in = !p.has_colors_precomp && DEG >= 1;
    // Loads FIRST, all of them (round 5): the block's SH rows (Q float4 per thread, back to back), then this thread's own
    // inputs -- position, scale, rotation, pose 0's summed pair record -- and only then the LDS writes and the barrier:
    // every latency of the kernel's head runs at the same time instead of one after the other.
    constexpr int Q = (3 * NC) % 4 == 0 && HS_TUNE_PB_UNROLL ? 3 * NC / 4 : 0;
    float4 staged[Q > 0 ? Q : 1];
    bool fast_stage = false;
    if constexpr (Q > 0) {
        const float* src = p.shs + (int64_t)g0 * M3;
        fast_stage = stage_in && M3 == 3 * NC && rows == kPreBwdBlock && (reinterpret_cast<uintptr_t>(src) & 15) == 0;   // (uniform)
        if (fast_stage) stage_rows_issue<Q>(staged, src);
    }
    const bool valid = g < p.P;
    const float x = valid ? p.means[3 * g] : 0.f, y = valid ? p.means[3 * g + 1] : 0.f, z = valid ? p.means[3 * g + 2] : 0.f;
    float s6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 rot4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float scl[3] = {0.f, 0.f, 0.f};
    if (valid) {   // the forward's 3-D covariance, bit for bit: the same function of the same inputs (or the input itself)
        if (p.cov_pre) {
#pragma unroll
            for (int k = 0; k < 6; ++k) s6[k] = p.cov_pre[6 * (int64_t)g + k];
        } else {
            scl[0] = p.scales[3 * g]; scl[1] = p.scales[3 * g + 1]; scl[2] = p.scales[3 * g + 2];
            rot4 = reinterpret_cast<const float4*>(p.rots)[g];
        }
    }
    const int rad0 = valid ? p.radii_inst[g] : 0;                       // pose 0
    float4 pre_q0 = make_float4(0.f, 0.f, 0.f, 0.f), pre_q1 = pre_q0;
    float2 pre_q2 = make_float2(0.f, 0.f);
    uint8_t pre_cl = 0;
    // (round 5, measured and removed: the wave loading its 64 sum rows together -- four instructions of 1 KB instead of three
    // touching 64 sectors each -- and handing them out through a wave-private LDS stage, the mirror image of preprocess_fwd's
    // record stores: stage +16 us.  The exchange waits for the rows at the kernel's head, where the per-thread loads just sit in
    // flight beside the SH staging)
    if (rad0 > 0) {
        pre_q0 = p.inst_grads[kInstF4 * (int64_t)g + 0]; pre_q1 = p.inst_grads[kInstF4 * (int64_t)g + 1];
        pre_q2 = reinterpret_cast<const float2*>(p.inst_grads + kInstF4 * (int64_t)g + 2)[0];
        if (!p.has_colors_precomp) pre_cl = p.clamped[g];
    }
    if (stage_in) {
        if (fast_stage) { if constexpr (Q > 0) stage_rows_commit<Q>(s_sh, staged, ld); }
        else stage_rows_in(s_sh, p.shs + (int64_t)g0 * M3, rows, M3, ld);
    }
    if (stage_in) __syncthreads();
    float gm[3] = {0.f, 0.f, 0.f};
    float gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gm2d[2] = {0.f, 0.f};
    float gop = 0.f;
    float gcol_pre[3] = {0.f, 0.f, 0.f};

    if (valid && !p.cov_pre) {
        const float q[4] = {rot4.x, rot4.y, rot4.z, rot4.w};
        cov3d_from_scale_rot(scl, p.mod, q, s6);
    }

    int max_radius = 0;  // over poses
    for (int pose = 0; pose < p.N; ++pose) {
        const int64_t idx = (int64_t)pose * p.P + g;
        const int rad = pose == 0 ? rad0 : (valid ? p.radii_inst[idx] : 0);
        const bool on = rad > 0;
        if (on) max_radius = max(max_radius, rad);
        float pg[POSE ? kPoseVals : 1];
        if constexpr (POSE) {
#pragma unroll
            for (int k = 0; k < kPoseVals; ++k) pg[k] = 0.f;
        }
        if (on) {
        // ---- this instance's summed pair records (pair_segsum_kernel) ----
        float r[9];
        float r9;
        {
            float4 q0 = pre_q0, q1 = pre_q1;
            float2 q2 = pre_q2;
            if (pose != 0) {
                q0 = p.inst_grads[kInstF4 * idx + 0]; q1 = p.inst_grads[kInstF4 * idx + 1];
                q2 = reinterpret_cast<const float2*>(p.inst_grads + kInstF4 * idx + 2)[0];
            }
            r[0] = q0.x; r[1] = q0.y; r[2] = q0.z; r[3] = q0.w; r[4] = q1.x; r[5] = q1.y; r[6] = q1.z; r[7] = q1.w;
            r[8] = q2.x;
            // r = {dmean2D.x, dmean2D.y, dconic A, B, C, dopacity, dcolor r,g,b}; r9 = d(inverse depth)
            r9 = q2.y;
        }
        gm2d[0] += r[0]; gm2d[1] += r[1];
        if (!p.antialias) gop += r[5];

        const float* V = p.view + 16 * pose;
        const float* PM = p.proj + 16 * pose;
        const float pvx = xform_row(V, 0, x, y, z), pvy = xform_row(V, 1, x, y, z), pvz = xform_row(V, 2, x, y, z);
        Ewa e;
        ewa_setup(V, p.W, p.H, p.tanfovx, p.tanfovy, pvx, pvy, pvz, e);
        float u0[3], u1[3];
        sym_mul(s6, e.a0, u0);
        sym_mul(s6, e.a1, u1);
        const float a = dot3(e.a0, u0) + 0.3f, b = dot3(e.a1, u0), cc = dot3(e.a1, u1) + 0.3f;
        const float denom = a * cc - b * b;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        const float gA = r[2], gB = r[3], gC = r[4];
        // antialiasing: opacity_eff = opacity * s, s = sqrt(max(eps, q)), q = det(cov - 0.3 I) / det(cov)
        float aa_da = 0.f, aa_db = 0.f, aa_dc = 0.f;
        if (p.antialias) {
            const float det0 = (a - 0.3f) * (cc - 0.3f) - b * b;
            const float q = det0 / denom;
            const float sfac = sqrtf(fmaxf(0.000025f, q));
            gop += r[5] * sfac;
            if (q > 0.000025f) {
                const float k = r[5] * p.opac[g] * 0.5f / sfac / (denom * denom);  // dL/dq / denom^2
                aa_da = k * ((cc - 0.3f) * denom - det0 * cc);
                aa_dc = k * ((a - 0.3f) * denom - det0 * a);
                aa_db = k * (2.f * b * (det0 - denom));
            }
        }
        if (denom2inv != 0.f) {
            const float dLda = denom2inv * (-cc * cc * gA + b * cc * gB + (denom - a * cc) * gC) + aa_da;
            const float dLdc = denom2inv * (-a * a * gC + a * b * gB + (denom - a * cc) * gA) + aa_dc;
            const float dLdb = denom2inv * (2.f * b * cc * gA - (denom + 2.f * b * b) * gB + 2.f * a * b * gC) + aa_db;
            const float* pp = e.a0; const float* qq = e.a1;
            gcov[0] += pp[0] * pp[0] * dLda + pp[0] * qq[0] * dLdb + qq[0] * qq[0] * dLdc;
            gcov[3] += pp[1] * pp[1] * dLda + pp[1] * qq[1] * dLdb + qq[1] * qq[1] * dLdc;
            gcov[5] += pp[2] * pp[2] * dLda + pp[2] * qq[2] * dLdb + qq[2] * qq[2] * dLdc;
            gcov[1] += 2.f * pp[0] * pp[1] * dLda + (pp[0] * qq[1] + pp[1] * qq[0]) * dLdb + 2.f * qq[0] * qq[1] * dLdc;
            gcov[2] += 2.f * pp[0] * pp[2] * dLda + (pp[0] * qq[2] + pp[2] * qq[0]) * dLdb + 2.f * qq[0] * qq[2] * dLdc;
            gcov[4] += 2.f * pp[1] * pp[2] * dLda + (pp[1] * qq[2] + pp[2] * qq[1]) * dLdb + 2.f * qq[1] * qq[2] * dLdc;
            float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
            float ga0v[3], ga1v[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float ga0 = 2.f * dLda * u0[j] + dLdb * u1[j];
                const float ga1 = 2.f * dLdc * u1[j] + dLdb * u0[j];
                ga0v[j] = ga0; ga1v[j] = ga1;
                dJ00 += ga0 * V[4 * j + 0]; dJ02 += ga0 * V[4 * j + 2];
                dJ11 += ga1 * V[4 * j + 1]; dJ12 += ga1 * V[4 * j + 2];
            }
            const float tz = 1.f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
            const float dtx = e.clamp_x ? 0.f : -e.fx * tz2 * dJ02;
            const float dty = e.clamp_y ? 0.f : -e.fy * tz2 * dJ12;
            const float dtz = -e.fx * tz2 * dJ00 - e.fy * tz2 * dJ11 + (2.f * e.fx * e.tx) * tz3 * dJ02 +
                              (2.f * e.fy * e.ty) * tz3 * dJ12;
#pragma unroll
            for (int j = 0; j < 3; ++j) gm[j] += V[4 * j + 0] * dtx + V[4 * j + 1] * dty + V[4 * j + 2] * dtz;
            if constexpr (POSE) {
                // t = Wv m + tv and A = J Wv:  dL/dWv_ij = dL/dt_i m_j + (row of A that uses Wv_i.)
                const float J00 = e.fx * tz, J02 = -(e.fx * e.tx) * tz2, J11 = e.fy * tz, J12 = -(e.fy * e.ty) * tz2;
                const float mm[3] = {x, y, z};
#pragma unroll
                for (int j = 0; j < 3; ++j) {  // pg[3j+i] <-> viewmatrix flat[4j+i]
                    pg[3 * j + 0] = dtx * mm[j] + ga0v[j] * J00;
                    pg[3 * j + 1] = dty * mm[j] + ga1v[j] * J11;
                    pg[3 * j + 2] = dtz * mm[j] + ga0v[j] * J02 + ga1v[j] * J12;
                }
                pg[9] = dtx; pg[10] = dty; pg[11] = dtz;
            }
        }
        if (r9 != 0.f) {  // inverse depth 1 / z, z = row 2 of the view transform
            const float dz = -r9 / (pvz * pvz);
#pragma unroll
            for (int j = 0; j < 3; ++j) gm[j] += V[4 * j + 2] * dz;
            if constexpr (POSE) {
                const float mm[3] = {x, y, z};
#pragma unroll
                for (int j = 0; j < 3; ++j) pg[3 * j + 2] += dz * mm[j];
                pg[11] += dz;
            }
        }
        {
            const float phx = xform_row(PM, 0, x, y, z), phy = xform_row(PM, 1, x, y, z),
                        phw = xform_row(PM, 3, x, y, z);
            const float mw = 1.0f / (phw + 0.0000001f);
            const float mul1 = phx * mw * mw, mul2 = phy * mw * mw;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                gm[j] += (PM[4 * j + 0] * mw - PM[4 * j + 3] * mul1) * r[0] +
                         (PM[4 * j + 1] * mw - PM[4 * j + 3] * mul2) * r[1];
            if constexpr (POSE) {
                // ndc = (ph.x, ph.y) * mw :  d/dPV_row0 = g.x mw m~, d/dPV_row1 = g.y mw m~, d/dPV_row3 = -(g.ndc) mw m~
                const float gw = -(r[0] * phx + r[1] * phy) * mw * mw;
                const float mt[4] = {x, y, z, 1.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pg[12 + 3 * j + 0] = r[0] * mw * mt[j];
                    pg[12 + 3 * j + 1] = r[1] * mw * mt[j];
                    pg[12 + 3 * j + 2] = gw * mt[j];
                }
            }
        }
        if (!p.has_colors_precomp) {
            const float* cp = p.campos + 3 * pose;
            const float dx = x - cp[0], dy = y - cp[1], dz = z - cp[2];
            const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float ux = dx / len, uy = dy / len, uz = dz / len;
            const uint8_t cl = pose == 0 ? pre_cl : p.clamped[idx];
            float gc[3];
            {
                float col[3] = {0.f, 0.f, 0.f};
                if (p.act != 0) {
                    const float4 rb = p.rec[kRecF4 * idx + 1];
                    col[0] = rb.z; col[1] = rb.w; col[2] = reinterpret_cast<const float*>(p.rec + kRecF4 * idx + 2)[0];
                }
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) gc[ch] = radiance_dact(p.act, col[ch], (cl >> ch) & 1) * r[6 + ch];
            }
            if constexpr (DEG >= 1) {
                float gb[NC][3];
                sh_basis_grad<DEG>(ux, uy, uz, gb);
                const float* sh = s_sh + threadIdx.x * ld;
                float gdir[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 1; k < NC; ++k) {
                    const float w = (sh[3 * k] * gc[0] + sh[3 * k + 1] * gc[1]) + sh[3 * k + 2] * gc[2];
                    gdir[0] += gb[k][0] * w; gdir[1] += gb[k][1] * w; gdir[2] += gb[k][2] * w;
                }
                const float dd = (ux * gdir[0] + uy * gdir[1]) + uz * gdir[2];
                const float inv = 1.f / len;
                const float gv0 = (gdir[0] - ux * dd) * inv, gv1 = (gdir[1] - uy * dd) * inv, gv2 = (gdir[2] - uz * dd) * inv;
                gm[0] += gv0; gm[1] += gv1; gm[2] += gv2;
                if constexpr (POSE) { pg[24] = -gv0; pg[25] = -gv1; pg[26] = -gv2; }  // v = m - campos
            }
        } else {
            gcol_pre[0] += r[6]; gcol_pre[1] += r[7]; gcol_pre[2] += r[8];
        }
    }  // if (on)
        if constexpr (POSE) {
            // workgroup sum of the 27 pose terms (all threads take part: `on` is per thread, `want_pose` uniform)
            __syncthreads();  // s_sh rows are still being read above; the scratch below aliases nothing but keep order simple
#pragma unroll
            for (int k = 0; k < kPoseVals - 1; ++k) {
                float v = pg[k];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
                pg[k] = v;
            }
            if ((threadIdx.x & 63) == 0) {
#pragma unroll
                for (int k = 0; k < kPoseVals - 1; ++k) s_pose[threadIdx.x >> 6][k] = pg[k];
            }
            __syncthreads();
            if (threadIdx.x < kPoseVals - 1)
                p.pose_partials[((int64_t)blk * p.N + pose) * kPoseVals + threadIdx.x] =
                    s_pose[0][threadIdx.x] + s_pose[1][threadIdx.x];
        }
    }

    // ---- Sigma -> scale / rotation (pose independent, applied once to the pose-summed gradient) ----
    if (valid && !p.has_cov_pre) {
        float R[9], Mx[9], s[3];
        const float q[4] = {rot4.x, rot4.y, rot4.z, rot4.w};
        quat_to_R(q, R);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s[k] = p.mod * scl[k];
#pragma unroll
            for (int j = 0; j < 3; ++j) Mx[3 * k + j] = s[k] * R[3 * j + k];
        }
        const float G[9] = {gcov[0], 0.5f * gcov[1], 0.5f * gcov[2], 0.5f * gcov[1], gcov[3],
                            0.5f * gcov[4], 0.5f * gcov[2], 0.5f * gcov[4], gcov[5]};
        float dR[9], dsc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float ds = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float dM = 2.f * ((Mx[3 * k + 0] * G[0 + j] + Mx[3 * k + 1] * G[3 + j]) + Mx[3 * k + 2] * G[6 + j]);
                ds += dM * R[3 * j + k];
                dR[3 * j + k] = s[k] * dM;
            }
            dsc[k] = p.mod * ds;
        }
        const float r = q[0], qx = q[1], qy = q[2], qz = q[3];
        if (p.d_scales) { p.d_scales[3 * g] = dsc[0]; p.d_scales[3 * g + 1] = dsc[1]; p.d_scales[3 * g + 2] = dsc[2]; }
        if (p.d_rots) {
            float4 o;
            o.x = 2.f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
            o.y = 2.f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.f * qx * dR[4] - r * dR[5] + qz * dR[6] + r * dR[7] - 2.f * qx * dR[8]);
            o.z = 2.f * (-2.f * qy * dR[0] + qx * dR[1] + r * dR[2] + qx * dR[3] + qz * dR[5] - r * dR[6] + qz * dR[7] - 2.f * qy * dR[8]);
            o.w = 2.f * (-2.f * qz * dR[0] - r * dR[1] + qx * dR[2] + r * dR[3] - 2.f * qz * dR[4] + qy * dR[5] + qx * dR[6] + qy * dR[7]);
            reinterpret_cast<float4*>(p.d_rots)[g] = o;
        }
    } else if (valid && p.d_cov) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p.d_cov[6 * (int64_t)g + k] = gcov[k];
    }
    if (valid && p.dens_grad && max_radius > 0) {
        p.dens_grad[g] += sqrtf(gm2d[0] * gm2d[0] + gm2d[1] * gm2d[1]);
        p.dens_denom[g] += 1.f;
        p.dens_radii[g] = max(p.dens_radii[g], max_radius);
    }
    if (valid) {
        if (p.d_means3D) { p.d_means3D[3 * g] = gm[0]; p.d_means3D[3 * g + 1] = gm[1]; p.d_means3D[3 * g + 2] = gm[2]; }
        if (p.d_means2D) { p.d_means2D[3 * g] = gm2d[0]; p.d_means2D[3 * g + 1] = gm2d[1]; p.d_means2D[3 * g + 2] = 0.f; }
        if (p.d_opac) p.d_opac[g] = gop;
        if (p.has_colors_precomp && p.d_colors) {
            p.d_colors[3 * g] = gcol_pre[0]; p.d_colors[3 * g + 1] = gcol_pre[1]; p.d_colors[3 * g + 2] = gcol_pre[2];
        }
    }
    if constexpr (SHG) {
        if (!p.has_colors_precomp && p.d_shs) {
            // dL/dsh = sum over poses of basis(dir) (x) masked colour gradient, accumulated in this thread's LDS row
            // (the row held the SH input, which only this thread read and is done with): a second, cheap walk over
            // the poses instead of 3*NC accumulator registers alive through the whole kernel -- those registers
            // cost the kernel a wave of occupancy per SIMD, and it is bound by memory latency
            float* row = s_sh + threadIdx.x * ld;
            for (int k = 0; k < M3; ++k) row[k] = 0.f;
            if (valid) {
                for (int pose = 0; pose < p.N; ++pose) {
                    const int64_t idx = (int64_t)pose * p.P + g;
                    if ((pose == 0 ? rad0 : p.radii_inst[idx]) <= 0) continue;
                    const float4 q1 = pose == 0 ? pre_q1 : p.inst_grads[kInstF4 * idx + 1];
                    const float q2 = pose == 0 ? pre_q2.x : reinterpret_cast<const float*>(p.inst_grads + kInstF4 * idx + 2)[0];
                    const uint8_t cl = pose == 0 ? pre_cl : p.clamped[idx];
                    float col[3] = {0.f, 0.f, 0.f};
                    if (p.act != 0) {
                        const float4 rb = p.rec[kRecF4 * idx + 1];
                        col[0] = rb.z; col[1] = rb.w; col[2] = reinterpret_cast<const float*>(p.rec + kRecF4 * idx + 2)[0];
                    }
                    const float gc[3] = {radiance_dact(p.act, col[0], cl & 1) * q1.z, radiance_dact(p.act, col[1], cl & 2) * q1.w,
                                         radiance_dact(p.act, col[2], cl & 4) * q2};
                    const float* cp = p.campos + 3 * pose;
                    const float dx = x - cp[0], dy = y - cp[1], dz = z - cp[2];
                    const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
                    float bs[NC];
                    sh_basis<DEG>(dx / len, dy / len, dz / len, bs);
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        row[3 * k + 0] += bs[k] * gc[0];
                        row[3 * k + 1] += bs[k] * gc[1];
                        row[3 * k + 2] += bs[k] * gc[2];
                    }
                }
            }
            __syncthreads();
            float* dst = p.d_shs + (int64_t)g0 * M3;
            stage_rows_out(dst, s_sh, rows, M3, ld);
        }
    }
}

// dL/dsh[g] = sum over views of basis(dir_v(g)) (x) colour gradient of view v (hs_sh_backward_views): one thread per
// Gaussian, views in ascending order, rows leave through LDS so the [P, M, 3] store is coalesced.
template <int DEG>
__global__ void __launch_bounds__(kPreBwdBlock) sh_views_kernel(int P, int M, int V, const float* means, const float* campos,
                                                                const float* view_colors, float* d_shs) {
    extern __shared__ float s_sh[];  // [kPreBwdBlock][M*3 + 1]
    constexpr int NC = (DEG + 1) * (DEG + 1);
    const int g0 = blockIdx.x * kPreBwdBlock;
    const int g = g0 + threadIdx.x;
    const int M3 = M * 3, ld = M3 + 1;
    const int rows = min(kPreBwdBlock, P - g0);
    float gsh[NC * 3];
#pragma unroll
    for (int k = 0; k < NC * 3; ++k) gsh[k] = 0.f;
    if (g < P) {
        const float x = means[3 * g], y = means[3 * g + 1], z = means[3 * g + 2];
        for (int v = 0; v < V; ++v) {
            const float* gcp = view_colors + 3 * ((int64_t)v * P + g);
            const float gc[3] = {gcp[0], gcp[1], gcp[2]};
            const float* cp = campos + 3 * v;
            const float dx = x - cp[0], dy = y - cp[1], dz = z - cp[2];
            const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float ux = dx / len, uy = dy / len, uz = dz / len;
            float bs[NC];
            sh_basis<DEG>(ux, uy, uz, bs);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                gsh[3 * k + 0] += bs[k] * gc[0];
                gsh[3 * k + 1] += bs[k] * gc[1];
                gsh[3 * k + 2] += bs[k] * gc[2];
            }
        }
    }
    float* row = s_sh + threadIdx.x * ld;
#pragma unroll
    for (int k = 0; k < NC * 3; ++k) row[k] = gsh[k];
    for (int k = NC * 3; k < M3; ++k) row[k] = 0.f;
    __syncthreads();
    float* dst = d_shs + (int64_t)g0 * M3;
    stage_rows_out(dst, s_sh, rows, M3, ld);
}

// Pose-gradient partials [nblk][N*kPoseVals] -> column sums, two fixed-order stages (deterministic).
constexpr int kPoseChunks = 64;
__global__ void __launch_bounds__(256) pose_reduce1_kernel(const float* partials, int nblk, int cols, float* stage) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y;
    if (col >= cols) return;
    const int per = (nblk + kPoseChunks - 1) / kPoseChunks;
    const int r0 = chunk * per, r1 = min(nblk, r0 + per);
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += partials[(int64_t)r * cols + col];
    stage[(int64_t)chunk * cols + col] = acc;
}
// One thread per OUTPUT element -- 16 + 16 + 3 per pose -- so that the matrix entries no pose term maps to (column 3 of the view
// matrix rows, column 2 of the projection rows) are written as zeros by this kernel.  (They used to be cleared by two
// hipMemsetAsync ahead of it; a captured step -- graphs.GraphedStep -- holds kernels only since round 5, when gradients of a
// captured formation step differed from replay to replay.  Round 6's reproducers exonerate HIP's memset nodes, DESIGN.md
// 4.11; the rule stays: one node kind in the graph, nothing to clear that a kernel does not clear itself.)
__global__ void __launch_bounds__(256) pose_reduce2_kernel(const float* stage, int N, float* d_view, float* d_proj,
                                                           float* d_campos) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= N * 35) return;
    const int pose = o / 35, e = o % 35;
    int k = -1;   // the pose term behind this element (none: zero)
    if (e < 16) {            // pg[3j+i] <-> viewmatrix flat[4j+i], i < 3
        const int j = e / 4, i = e % 4;
        if (i < 3) k = 3 * j + i;
    } else if (e < 32) {     // pg[12+3j+r] <-> projmatrix flat[4j + {0,1,3}[r]]
        const int q = e - 16, j = q / 4, c = q % 4;
        if (c != 2) k = 12 + 3 * j + (c == 3 ? 2 : c);
    } else {
        k = 24 + (e - 32);
    }
    float acc = 0.f;
    if (k >= 0) {
        const int cols = N * kPoseVals, col = pose * kPoseVals + k;
        for (int c = 0; c < kPoseChunks; ++c) acc += stage[(int64_t)c * cols + col];
    }
    if (e < 16) d_view[16 * pose + e] = acc;
    else if (e < 32) d_proj[16 * pose + (e - 16)] = acc;
    else d_campos[3 * pose + (e - 32)] = acc;
}

__global__ void __launch_bounds__(256) fill_i32_kernel(int* dst, int64_t n, int v) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n) dst[i0 + k] = v;
}

}  // namespace

int launch_preprocess_fwd(const hs_fwd_args& a, const hs_layout& L, hipStream_t s, uint32_t frame_tag) {
    const hs_dims& d = a.dims;
    char* geom = (char*)a.geom;
    PreFwd p;
    p.P = d.P; p.M = d.M; p.W = d.W; p.H = d.H; p.N = d.n_poses;
    p.tanfovx = a.tanfovx; p.tanfovy = a.tanfovy; p.mod = a.scale_modifier;
    p.view = a.viewmatrices; p.proj = a.projmatrices; p.campos = a.camposes;
    p.means = a.means3D; p.opac = a.opacities; p.shs = a.shs; p.colors = a.colors_precomp; p.scales = a.scales;
    p.rots = a.rotations; p.cov_pre = a.cov3D_precomp;
    p.rec = (float4*)(geom + L.rec); p.depth = (float*)(geom + L.depth); p.radii_inst = (int*)(geom + L.radii);
    p.tiles = (uint32_t*)(geom + L.tiles_touched); p.cov3D = (float*)(geom + L.cov3D);
    p.clamped = (uint8_t*)(geom + L.clamped); p.radii_out = a.radii;
    p.binfo = (uint2*)(geom + L.binfo);
    p.antialias = (a.flags & HS_FLAG_ANTIALIAS) != 0;
    p.act = (a.flags & HS_FLAG_RADIANCE_EXP) ? 1 : (a.flags & HS_FLAG_RADIANCE_SOFTPLUS) ? 2 : 0;
    p.depth_pairs = nullptr; p.counters = nullptr; p.ranges = nullptr; p.n_vtiles = 0;
    p.sort_zero = nullptr; p.n_sort_zero = 0; p.bits_at = 0; p.frame_tag = 0u;
    p.pair_zero = nullptr; p.n_pair_zero = 0;
    if ((a.stages & HS_STAGE_BIN) && a.binning) {
        char* bin = (char*)a.binning;
        p.depth_pairs = (uint2*)(bin + L.depth_pairs);
        p.counters = (hs_counters*)(geom + L.counters); p.ranges = (uint2*)(bin + L.ranges);
        p.n_vtiles = (int64_t)((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile) * d.n_poses;
        p.sort_zero = (uint32_t*)(bin + L.sort_tmp);
        p.n_sort_zero = depth_scratch_words((int64_t)d.P * d.n_poses);
        p.bits_at = kGhistWords + kDepthBitsAt;   // (tagged words: no memset ahead of the frame's first kernel)
        p.frame_tag = frame_tag;
        p.pair_zero = (uint32_t*)(bin + L.pair_sort_tmp);
        p.n_pair_zero = pair_scratch_words((int64_t)d.P * d.n_poses, d.capacity,
                                           sort_passes(tile_bits((uint32_t)p.n_vtiles)));
    }
    // (N poses: the per-Gaussian output radius is the max over the poses, by atomicMax -- cleared by a kernel, not by
    // hipMemsetAsync: a captured step holds kernels only, see pose_reduce2_kernel)
    if (d.n_poses > 1 && d.P > 0) fill_i32_kernel<<<ceil_div((int64_t)d.P, 1024), 256, 0, s>>>(a.radii, (int64_t)d.P, 0);
    // chunks of 256 Gaussians, padded to a multiple of the 8 XCDs, times the poses (see the kernel's block map)
    const int grid = (int)(ceil_div(ceil_div((int64_t)d.P, 256), 8) * 8 * d.n_poses);
    const int deg = a.colors_precomp ? 0 : d.sh_degree;
    switch (deg) {
        case 0: preprocess_fwd_kernel<0><<<grid, 256, 0, s>>>(p); break;
        case 1: preprocess_fwd_kernel<1><<<grid, 256, 0, s>>>(p); break;
        case 2: preprocess_fwd_kernel<2><<<grid, 256, 0, s>>>(p); break;
        default: preprocess_fwd_kernel<3><<<grid, 256, 0, s>>>(p); break;
    }
    HS_LAUNCH_CHECK();
    return HS_OK;
}

int launch_preprocess_bwd(const hs_bwd_args& a, const hs_layout& L, hipStream_t s, bool segsum, bool project,
                          const CrfReduce* crf_reduce) {
    const hs_dims& d = a.dims;
    const char* geom = (const char*)a.geom;
    PreBwd p;
    p.P = d.P; p.M = d.M; p.W = d.W; p.H = d.H; p.N = d.n_poses;
    p.tanfovx = a.tanfovx; p.tanfovy = a.tanfovy; p.mod = a.scale_modifier;
    p.view = a.viewmatrices; p.proj = a.projmatrices; p.campos = a.camposes;
    p.means = a.means3D; p.shs = a.shs; p.scales = a.scales; p.rots = a.rotations; p.opac = a.opacities;
    p.antialias = (a.flags & HS_FLAG_ANTIALIAS) != 0;
    p.act = (a.flags & HS_FLAG_RADIANCE_EXP) ? 1 : (a.flags & HS_FLAG_RADIANCE_SOFTPLUS) ? 2 : 0;
    p.has_colors_precomp = a.colors_precomp != nullptr; p.has_cov_pre = a.cov3D_precomp != nullptr;
    p.rec = (const float4*)(geom + L.rec); p.radii_inst = (const int*)(geom + L.radii);
    p.tiles = (const uint32_t*)(geom + L.tiles_touched); p.offsets = (const uint32_t*)(geom + L.offsets);
    p.cov_pre = a.cov3D_precomp; p.clamped = (const uint8_t*)(geom + L.clamped);
    p.inst_grads = (const float4*)((const char*)a.bwd + L.inst_grads);
    const CrfReduce no_reduce{nullptr, 0, 0, 0, nullptr, nullptr, 0};
    if (segsum) {
        const int64_t I = (int64_t)d.P * d.n_poses;
        const char* bin = (const char*)a.binning;
        SegsumArgs sg;
        sg.I = I; sg.inst_sorted = (const uint32_t*)(bin + L.inst_sorted); sg.offs_sorted = (const uint32_t*)(bin + L.offs_sorted);
        sg.pair_grads = (const float4*)((const char*)a.bwd + L.pair_grads); sg.pair_flags = (const uint8_t*)bin + L.pair_flags;
        sg.inst_grads = (float4*)((char*)a.bwd + L.inst_grads); sg.counters = (const hs_counters*)(geom + L.counters);
        sg.radii_inst = p.radii_inst; sg.clamped = p.clamped;
        sg.view_colors = a.colors_precomp ? nullptr : a.dL_dview_colors; sg.rec = p.rec; sg.act = p.act;
        pair_segsum_kernel<<<ceil_div(4 * I, 256), 256, 0, s>>>(sg, crf_reduce ? *crf_reduce : no_reduce);
        HS_LAUNCH_CHECK();
    }
    if (!project) return HS_OK;
    p.d_means3D = a.dL_dmeans3D; p.d_means2D = a.dL_dmeans2D; p.d_opac = a.dL_dopacities; p.d_shs = a.dL_dshs;
    p.d_colors = a.dL_dcolors_precomp; p.d_scales = a.dL_dscales; p.d_rots = a.dL_drotations;
    p.d_cov = a.dL_dcov3D_precomp;
    p.dens_grad = a.densify_grad_accum; p.dens_denom = a.densify_denom; p.dens_radii = a.densify_max_radii;
    const bool shg = p.d_shs != nullptr;
    // hs_bwd_args.g_begin / g_end: this launch covers the Gaussians [g_lo, g_hi) (all of them by default); a step that
    // exchanges gradients chunk by chunk enqueues the chunks in ascending order
    const bool ranged = a.g_begin != 0 || a.g_end != 0;
    const int g_lo = ranged ? a.g_begin : 0, g_hi = ranged ? a.g_end : d.P;
    const int nblk_all = ceil_div(d.P, kPreBwdBlock);
    const int grid = ceil_div(g_hi - g_lo, kPreBwdBlock);
    p.blk0 = g_lo / kPreBwdBlock;
    const int deg = a.colors_precomp ? 0 : d.sh_degree;
    const size_t lds = (size_t)kPreBwdBlock * (d.M * 3 + 1) * sizeof(float);
    float* pose_partials = a.dL_dviewmatrices ? (float*)((char*)a.bwd + L.pose_partials) : nullptr;
    p.pose_partials = pose_partials;
#define HS_LAUNCH_PRE_BWD(DEG_)                                                                              \
    if (pose_partials && shg) preprocess_bwd_kernel<DEG_, true, true><<<grid, kPreBwdBlock, lds, s>>>(p);       \
    else if (pose_partials) preprocess_bwd_kernel<DEG_, true, false><<<grid, kPreBwdBlock, lds, s>>>(p);        \
    else if (shg) preprocess_bwd_kernel<DEG_, false, true><<<grid, kPreBwdBlock, lds, s>>>(p);                  \
    else preprocess_bwd_kernel<DEG_, false, false><<<grid, kPreBwdBlock, lds, s>>>(p)
    if (grid > 0) switch (deg) {
        case 0: HS_LAUNCH_PRE_BWD(0); break;
        case 1: HS_LAUNCH_PRE_BWD(1); break;
        case 2: HS_LAUNCH_PRE_BWD(2); break;
        default: HS_LAUNCH_PRE_BWD(3); break;
    }
#undef HS_LAUNCH_PRE_BWD
    HS_LAUNCH_CHECK();
    if (pose_partials && g_hi >= d.P) {   // (the last chunk: every block's partial row is written by now)
        const int grid = nblk_all;
        const int cols = d.n_poses * kPoseVals;
        float* stage = pose_partials + (int64_t)grid * cols;
        pose_reduce1_kernel<<<dim3(ceil_div(cols, 256), kPoseChunks), 256, 0, s>>>(pose_partials, grid, cols, stage);
        pose_reduce2_kernel<<<ceil_div(d.n_poses * 35, 256), 256, 0, s>>>(stage, d.n_poses, a.dL_dviewmatrices,
                                                                        a.dL_dprojmatrices, a.dL_dcamposes);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}

int64_t pose_partial_floats(int P, int N) {
    return ((int64_t)ceil_div(P > 0 ? P : 1, kPreBwdBlock) + kPoseChunks) * N * kPoseVals;
}

int launch_sh_backward_views(int P, int M, int deg, int V, const float* means3D, const float* camposes,
                             const float* view_colors, float* d_shs, hipStream_t s) {
    const int grid = ceil_div(P, kPreBwdBlock);
    const size_t lds = (size_t)kPreBwdBlock * (M * 3 + 1) * sizeof(float);
    switch (deg) {
        case 0: sh_views_kernel<0><<<grid, kPreBwdBlock, lds, s>>>(P, M, V, means3D, camposes, view_colors, d_shs); break;
        case 1: sh_views_kernel<1><<<grid, kPreBwdBlock, lds, s>>>(P, M, V, means3D, camposes, view_colors, d_shs); break;
        case 2: sh_views_kernel<2><<<grid, kPreBwdBlock, lds, s>>>(P, M, V, means3D, camposes, view_colors, d_shs); break;
        default: sh_views_kernel<3><<<grid, kPreBwdBlock, lds, s>>>(P, M, V, means3D, camposes, view_colors, d_shs); break;
    }
    HS_LAUNCH_CHECK();
    return HS_OK;
}

int launch_cov3d(const hs_fwd_args& a, const hs_layout& L, hipStream_t s) {
    cov3d_kernel<<<ceil_div(a.dims.P, 256), 256, 0, s>>>(a.dims.P, a.scale_modifier, a.scales, a.rotations, a.cov3D_precomp,
                                                        (float*)((char*)a.geom + L.cov3D));
    HS_LAUNCH_CHECK();
    return HS_OK;
}

int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* vis, hipStream_t s) {
    mark_visible_kernel<<<ceil_div(P, 256), 256, 0, s>>>(P, means3D, view, vis);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

}  // namespace hs
