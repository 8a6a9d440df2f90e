/*
 * hs_oracle.c -- CPU restatement of the differentiable 3D-Gaussian tile rasterizer
 * (the hot path named by BASELINE.json north_star).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped path is the HIP library
 * (casualhdrsplat_amd/csrc) and never routes through this file.
 *
 * PARITY UNPINNED.  /root/reference holds only Readme.md + two figures (no code, no
 * tests, no golden vectors -- SURVEY.md section 0).  The rasterizer the north-star API
 * belongs to is the third-party package diff_gaussian_rasterization
 * (graphdeco-inria/diff-gaussian-rasterization); the reference neither vendors it nor
 * pins a version, and it is not installed here.  This file therefore restates the
 * *published* algorithm of that package as frozen in SURVEY.md section 8(a), rows
 * a4..a12 (each function below names its row).  It is pinned by closed-form
 * known-answer tests and by an independent float64 PyTorch-autograd implementation
 * (oracle/torch_rasterizer.py), see tests/test_oracle_*.py.  Also restated: the two extras of
 * newer versions of that package that SURVEY.md 8(f) n3 lists -- the `antialiasing` opacity
 * compensation (hso_camera.antialias) and the expected inverse-depth image with its gradient
 * (the depths / invdepth arguments of hso_render_fwd / hso_render_bwd / hso_preprocess_bwd).
 *
 * Reference anchors (the only ones that exist): /root/reference/Readme.md:54
 * ("we train 3DGS to reconstruct an HDR scene ... jointly estimating camera motion,
 * exposure time, and camera response curve") and assets/pipeline.png (H -> CRF -> I
 * -> blur-average -> B), which fix the ORDER of the HDR epilogue implemented in
 * hso_tonemap_*.
 *
 * Numerics: all arithmetic is IEEE binary32, evaluated in exactly the operation order
 * written here; build with -ffp-contract=off so no FMA is formed.  Everything that
 * feeds an integer decision (depth bits, radius, tile rectangle, sort keys) uses only
 * + - * / sqrt ceil, so the HIP kernels can (and must) reproduce those bit for bit.
 * Per-Gaussian gradient sums are accumulated in double and rounded once, so the
 * oracle is at least as accurate as any fp32 summation order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HSO_TILE 16

typedef struct {
    int P;          /* number of Gaussians */
    int sh_degree;  /* active SH degree 0..3 */
    int M;          /* SH coefficients stored per Gaussian and channel */
    int W, H;
    float tanfovx, tanfovy;
    float scale_modifier;
    float bg[3];
    float viewmatrix[16]; /* flat, "transposed": p_view.x = m[0]x+m[4]y+m[8]z+m[12] */
    float projmatrix[16]; /* full view*proj, same convention */
    float campos[3];
    int antialias;        /* newer published rasterizer's flag: opacity *= sqrt(max(0.000025, det(cov2D)/det(cov2D + 0.3 I))) */
    int radiance_activation; /* how the SH sum s becomes the (linear, HDR) colour -- SURVEY.md 7.3 / 8a a1:
                                0 relu_shift: max(s + 0.5, 0) (the published rule, unbounded above),
                                1 exp: e^s,  2 softplus: ln(1 + e^s).  Precomputed colours pass through unchanged. */
} hso_camera;

/* d colour / d s of the radiance activation, from the stored colour (and the clamp bit for relu_shift) */
static inline float radiance_dact(int act, float col, int was_clamped) {
    if (act == 1) return col;                       /* d e^s = e^s */
    if (act == 2) return -expm1f(-col);             /* sigmoid(s) = 1 - e^{-softplus(s)}; expm1 keeps it exact for dark
                                                       Gaussians (col < 6e-8, where 1 - expf(-col) rounds to 0) */
    return was_clamped ? 0.f : 1.f;
}

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* p_view_i = ((m[i]*x + m[4+i]*y) + m[8+i]*z) + m[12+i]   (SURVEY 8a a1 convention) */
static inline float xform_row(const float* m, int i, float x, float y, float z) {
    return ((m[i] * x + m[4 + i] * y) + m[8 + i] * z) + m[12 + i];
}

/* Rotation matrix of quaternion (r,x,y,z), row-major R[3*i+j]. */
static void quat_to_R(const float* q, float* R) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

/* Sigma = R diag((mod*s)^2) R^T ; six upper-triangular entries
 * (xx, xy, xz, yy, yz, zz).  M_ki = s_k * R_ik ; Sigma_ij = (M_0i M_0j + M_1i M_1j) + M_2i M_2j. */
static void cov3d_from_scale_rot(const float* scale, float mod, const float* q, float* cov6) {
    float R[9], Mx[9];
    quat_to_R(q, R);
    for (int k = 0; k < 3; ++k) {
        float s = mod * scale[k];
        for (int i = 0; i < 3; ++i) Mx[3 * k + i] = s * R[3 * i + k];
    }
#define SIG(i, j) ((Mx[0 + i] * Mx[0 + j] + Mx[3 + i] * Mx[3 + j]) + Mx[6 + i] * Mx[6 + j])
    cov6[0] = SIG(0, 0); cov6[1] = SIG(0, 1); cov6[2] = SIG(0, 2);
    cov6[3] = SIG(1, 1); cov6[4] = SIG(1, 2); cov6[5] = SIG(2, 2);
#undef SIG
}

/* Shared by forward and backward: the 2x3 matrix A = J * Wv and the clamped view-space t. */
typedef struct {
    float tx, ty, tz;      /* clamped (tx,ty), original tz */
    int clamp_x, clamp_y;  /* 1 if the 1.3*tanfov clamp was active */
    float a0[3], a1[3];    /* rows of A */
    float fx, fy;
} hso_ewa;

static void ewa_setup(const hso_camera* c, float pvx, float pvy, float pvz, hso_ewa* e) {
    const float* V = c->viewmatrix;
    float fx = (float)c->W / (2.f * c->tanfovx);
    float fy = (float)c->H / (2.f * c->tanfovy);
    float limx = 1.3f * c->tanfovx, limy = 1.3f * c->tanfovy;
    float txtz = pvx / pvz, tytz = pvy / pvz;
    e->clamp_x = (txtz < -limx) || (txtz > limx);
    e->clamp_y = (tytz < -limy) || (tytz > limy);
    float tx = fminf_(limx, fmaxf_(-limx, txtz)) * pvz;
    float ty = fminf_(limy, fmaxf_(-limy, tytz)) * pvz;
    float tz = pvz;
    float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    /* Wv_ij (std rotation, row i col j) = V[4*j + i] */
    for (int j = 0; j < 3; ++j) {
        e->a0[j] = J00 * V[4 * j + 0] + J02 * V[4 * j + 2];
        e->a1[j] = J11 * V[4 * j + 1] + J12 * V[4 * j + 2];
    }
    e->tx = tx; e->ty = ty; e->tz = tz; e->fx = fx; e->fy = fy;
}

static inline void sym_mul(const float* s6, const float* v, float* out) {
    out[0] = (s6[0] * v[0] + s6[1] * v[1]) + s6[2] * v[2];
    out[1] = (s6[1] * v[0] + s6[3] * v[1]) + s6[4] * v[2];
    out[2] = (s6[2] * v[0] + s6[4] * v[1]) + s6[5] * v[2];
}
static inline float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* Real SH basis values b[0..15] at unit direction (x,y,z); SURVEY 8a a4. */
static void sh_basis(int deg, float x, float y, float z, float* b) {
    b[0] = SH_C0;
    if (deg < 1) return;
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2.f * zz - xx - yy);
    b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
    if (deg < 3) return;
    b[9] = SH_C3[0] * y * (3.f * xx - yy);
    b[10] = SH_C3[1] * xy * z;
    b[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
    b[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
    b[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
    b[14] = SH_C3[5] * z * (xx - yy);
    b[15] = SH_C3[6] * x * (xx - 3.f * yy);
}

/* d b[k] / d(x,y,z), the three components treated as independent. */
static void sh_basis_grad(int deg, float x, float y, float z, float (*g)[3]) {
    for (int k = 0; k < 16; ++k) g[k][0] = g[k][1] = g[k][2] = 0.f;
    if (deg < 1) return;
    g[1][1] = -SH_C1; g[2][2] = SH_C1; g[3][0] = -SH_C1;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z;
    g[4][0] = SH_C2[0] * y; g[4][1] = SH_C2[0] * x;
    g[5][1] = SH_C2[1] * z; g[5][2] = SH_C2[1] * y;
    g[6][0] = SH_C2[2] * -2.f * x; g[6][1] = SH_C2[2] * -2.f * y; g[6][2] = SH_C2[2] * 4.f * z;
    g[7][0] = SH_C2[3] * z; g[7][2] = SH_C2[3] * x;
    g[8][0] = SH_C2[4] * 2.f * x; g[8][1] = SH_C2[4] * -2.f * y;
    if (deg < 3) return;
    g[9][0] = SH_C3[0] * 6.f * x * y;          g[9][1] = SH_C3[0] * (3.f * xx - 3.f * yy);
    g[10][0] = SH_C3[1] * y * z;               g[10][1] = SH_C3[1] * x * z; g[10][2] = SH_C3[1] * x * y;
    g[11][0] = SH_C3[2] * -2.f * x * y;        g[11][1] = SH_C3[2] * (4.f * zz - xx - 3.f * yy); g[11][2] = SH_C3[2] * 8.f * y * z;
    g[12][0] = SH_C3[3] * -6.f * x * z;        g[12][1] = SH_C3[3] * -6.f * y * z; g[12][2] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
    g[13][0] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); g[13][1] = SH_C3[4] * -2.f * x * y; g[13][2] = SH_C3[4] * 8.f * x * z;
    g[14][0] = SH_C3[5] * 2.f * x * z;         g[14][1] = SH_C3[5] * -2.f * y * z; g[14][2] = SH_C3[5] * (xx - yy);
    g[15][0] = SH_C3[6] * (3.f * xx - 3.f * yy); g[15][1] = SH_C3[6] * -6.f * x * y;
}

/* ------------------------------------------------------------------------------------------
 * a4  preprocess forward (one Gaussian per iteration).
 * Outputs are zero-initialised by this function for culled Gaussians.
 * rect is [P,4] = (min_x, min_y, max_x, max_y) in tiles, max exclusive.
 * ------------------------------------------------------------------------------------------ */
int hso_preprocess_fwd(const hso_camera* c, const float* means3D, const float* opacities,
                       const float* shs, const float* colors_precomp, const float* scales,
                       const float* rotations, const float* cov3D_precomp,
                       float* depths, float* xy, float* conic_opacity, float* rgb, int* radii,
                       uint32_t* tiles_touched, int* rect, float* cov3D, uint8_t* clamped) {
    const int P = c->P;
    const int gx = (c->W + HSO_TILE - 1) / HSO_TILE, gy = (c->H + HSO_TILE - 1) / HSO_TILE;
    const int ncoef = (c->sh_degree + 1) * (c->sh_degree + 1);
    for (int i = 0; i < P; ++i) {
        depths[i] = 0.f; xy[2 * i] = xy[2 * i + 1] = 0.f;
        for (int k = 0; k < 4; ++k) conic_opacity[4 * i + k] = 0.f;
        for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = 0.f; clamped[3 * i + k] = 0; }
        for (int k = 0; k < 4; ++k) rect[4 * i + k] = 0;
        for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = 0.f;
        radii[i] = 0; tiles_touched[i] = 0;

        float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
        float pvx = xform_row(c->viewmatrix, 0, x, y, z);
        float pvy = xform_row(c->viewmatrix, 1, x, y, z);
        float pvz = xform_row(c->viewmatrix, 2, x, y, z);
        if (pvz <= 0.2f) continue; /* near-plane cull */

        float phx = xform_row(c->projmatrix, 0, x, y, z);
        float phy = xform_row(c->projmatrix, 1, x, y, z);
        float phw = xform_row(c->projmatrix, 3, x, y, z);
        float pw = 1.0f / (phw + 0.0000001f);
        float ppx = phx * pw, ppy = phy * pw;

        float s6[6];
        if (cov3D_precomp) memcpy(s6, cov3D_precomp + 6 * i, sizeof s6);
        else cov3d_from_scale_rot(scales + 3 * i, c->scale_modifier, rotations + 4 * i, s6);
        memcpy(cov3D + 6 * i, s6, sizeof s6);

        hso_ewa e;
        ewa_setup(c, pvx, pvy, pvz, &e);
        float u0[3], u1[3];
        sym_mul(s6, e.a0, u0);
        sym_mul(s6, e.a1, u1);
        float ca = dot3(e.a0, u0) + 0.3f;
        float cb = dot3(e.a1, u0);
        float cc = dot3(e.a1, u1) + 0.3f;

        float det = ca * cc - cb * cb;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conA = cc * det_inv, conB = -cb * det_inv, conC = ca * det_inv;

        float mid = 0.5f * (ca + cc);
        float disc = sqrtf(fmaxf_(0.1f, mid * mid - det));
        float lam1 = mid + disc, lam2 = mid - disc;
        float rad_f = ceilf(3.f * sqrtf(fmaxf_(lam1, lam2)));
        int my_radius = (int)rad_f;

        float pix_x = ((ppx + 1.0f) * (float)c->W - 1.0f) * 0.5f;
        float pix_y = ((ppy + 1.0f) * (float)c->H - 1.0f) * 0.5f;

        int rminx = imin(gx, imax(0, (int)((pix_x - (float)my_radius) / (float)HSO_TILE)));
        int rminy = imin(gy, imax(0, (int)((pix_y - (float)my_radius) / (float)HSO_TILE)));
        int rmaxx = imin(gx, imax(0, (int)((pix_x + (float)my_radius + (float)(HSO_TILE - 1)) / (float)HSO_TILE)));
        int rmaxy = imin(gy, imax(0, (int)((pix_y + (float)my_radius + (float)(HSO_TILE - 1)) / (float)HSO_TILE)));
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;

        if (colors_precomp) {
            for (int ch = 0; ch < 3; ++ch) rgb[3 * i + ch] = colors_precomp[3 * i + ch];
        } else {
            float dx = x - c->campos[0], dy = y - c->campos[1], dz = z - c->campos[2];
            float len = sqrtf((dx * dx + dy * dy) + dz * dz);
            float ux = dx / len, uy = dy / len, uz = dz / len;
            float b[16];
            sh_basis(c->sh_degree, ux, uy, uz, b);
            const float* sh = shs + (size_t)i * c->M * 3;
            for (int ch = 0; ch < 3; ++ch) {
                float acc = b[0] * sh[ch];
                for (int k = 1; k < ncoef; ++k) acc = acc + b[k] * sh[3 * k + ch];
                if (c->radiance_activation == 1) {
                    rgb[3 * i + ch] = expf(acc);
                } else if (c->radiance_activation == 2) {
                    rgb[3 * i + ch] = acc > 20.f ? acc : log1pf(expf(acc));
                } else {
                    acc = acc + 0.5f;
                    clamped[3 * i + ch] = acc < 0.f;
                    rgb[3 * i + ch] = fmaxf_(acc, 0.f);
                }
            }
        }
        depths[i] = pvz;
        radii[i] = my_radius;
        xy[2 * i] = pix_x; xy[2 * i + 1] = pix_y;
        conic_opacity[4 * i + 0] = conA; conic_opacity[4 * i + 1] = conB;
        conic_opacity[4 * i + 2] = conC; conic_opacity[4 * i + 3] = opacities[i];
        if (c->antialias) {
            float det0 = (ca - 0.3f) * (cc - 0.3f) - cb * cb;
            conic_opacity[4 * i + 3] = opacities[i] * sqrtf(fmaxf_(0.000025f, det0 / det));
        }
        rect[4 * i + 0] = rminx; rect[4 * i + 1] = rminy; rect[4 * i + 2] = rmaxx; rect[4 * i + 3] = rmaxy;
        tiles_touched[i] = (uint32_t)((rmaxx - rminx) * (rmaxy - rminy));
    }
    return 0;
}

/* a14 markVisible */
int hso_mark_visible(const hso_camera* c, const float* means3D, uint8_t* visible) {
    for (int i = 0; i < c->P; ++i) {
        float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
        visible[i] = xform_row(c->viewmatrix, 2, x, y, z) > 0.2f;
    }
    return 0;
}

/* a5 inclusive scan; returns R. */
int64_t hso_scan(const uint32_t* tiles_touched, int P, uint32_t* offsets) {
    uint32_t acc = 0;
    for (int i = 0; i < P; ++i) { acc += tiles_touched[i]; offsets[i] = acc; }
    return (int64_t)acc;
}

/* index of highest set bit + 1 (0 for n == 0); number of tile-id bits the sort covers (a7). */
int hso_key_tile_bits(uint32_t n) {
    int b = 0;
    while (n) { ++b; n >>= 1; }
    return b;
}

/* a6 duplicateWithKeys: key = (tile_id << 32) | bits(depth); value = Gaussian index. */
int hso_duplicate_with_keys(const hso_camera* c, const float* depths, const int* rect, const int* radii,
                            const uint32_t* offsets, uint64_t* keys, uint32_t* vals) {
    const int gx = (c->W + HSO_TILE - 1) / HSO_TILE;
    for (int i = 0; i < c->P; ++i) {
        if (radii[i] <= 0) continue;
        uint32_t off = i == 0 ? 0u : offsets[i - 1];
        uint32_t dbits;
        memcpy(&dbits, &depths[i], 4);
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; ++x) {
                uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
                key = (key << 32) | dbits;
                keys[off] = key; vals[off] = (uint32_t)i; ++off;
            }
    }
    return 0;
}

/* a7 stable LSD radix sort on bits [0, nbits). */
int hso_sort_pairs(const uint64_t* keys_in, const uint32_t* vals_in, int64_t R, int nbits,
                   uint64_t* keys_out, uint32_t* vals_out) {
    if (R == 0) return 0;
    uint64_t* kb = (uint64_t*)malloc((size_t)R * 8 * 2);
    uint32_t* vb = (uint32_t*)malloc((size_t)R * 4 * 2);
    if (!kb || !vb) { free(kb); free(vb); return -1; }
    uint64_t* k0 = kb; uint64_t* k1 = kb + R; uint32_t* v0 = vb; uint32_t* v1 = vb + R;
    memcpy(k0, keys_in, (size_t)R * 8); memcpy(v0, vals_in, (size_t)R * 4);
    for (int shift = 0; shift < nbits; shift += 8) {
        int64_t cnt[257];
        memset(cnt, 0, sizeof cnt);
        int w = nbits - shift < 8 ? nbits - shift : 8;
        uint64_t mask = (1ull << w) - 1;
        for (int64_t i = 0; i < R; ++i) cnt[((k0[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < R; ++i) {
            int64_t pos = cnt[(k0[i] >> shift) & mask]++;
            k1[pos] = k0[i]; v1[pos] = v0[i];
        }
        uint64_t* tk = k0; k0 = k1; k1 = tk;
        uint32_t* tv = v0; v0 = v1; v1 = tv;
    }
    memcpy(keys_out, k0, (size_t)R * 8); memcpy(vals_out, v0, (size_t)R * 4);
    free(kb); free(vb);
    return 0;
}

/* a8 identifyTileRanges: ranges[tile] = [first, last+1), (0,0) when empty. */
int hso_tile_ranges(const uint64_t* keys_sorted, int64_t R, int ntiles, uint32_t* ranges) {
    memset(ranges, 0, (size_t)ntiles * 8);
    for (int64_t i = 0; i < R; ++i) {
        uint32_t t = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0) ranges[2 * t] = 0;
        else {
            uint32_t tp = (uint32_t)(keys_sorted[i - 1] >> 32);
            if (t != tp) { ranges[2 * tp + 1] = (uint32_t)i; ranges[2 * t] = (uint32_t)i; }
        }
        if (i == R - 1) ranges[2 * t + 1] = (uint32_t)R;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a9  render forward.  out_color is [3,H,W]; final_T, n_contrib are [H,W].
 * ------------------------------------------------------------------------------------------ */
/* depths / out_invdepth (both or neither, may be NULL): expected inverse depth image sum_i alpha_i T_i / depth_i
 * (third output of newer published rasterizers). */
int hso_render_fwd(const hso_camera* c, const uint32_t* ranges, const uint32_t* point_list,
                   const float* xy, const float* conic_opacity, const float* rgb,
                   float* out_color, float* final_T, uint32_t* n_contrib,
                   const float* depths, float* out_invdepth) {
    const int W = c->W, H = c->H;
    const int gx = (W + HSO_TILE - 1) / HSO_TILE;
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int tile = (py / HSO_TILE) * gx + (px / HSO_TILE);
            uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
            float pxf = (float)px, pyf = (float)py;
            float T = 1.0f, C[3] = {0.f, 0.f, 0.f}, D = 0.f;
            uint32_t contributor = 0, last = 0;
            for (uint32_t k = beg; k < end; ++k) {
                ++contributor;
                uint32_t id = point_list[k];
                float dx = xy[2 * id] - pxf, dy = xy[2 * id + 1] - pyf;
                const float* co = conic_opacity + 4 * id;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                float alpha = fminf_(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1.f - alpha);
                if (test_T < 0.0001f) break; /* pixel done */
                for (int ch = 0; ch < 3; ++ch) C[ch] += rgb[3 * id + ch] * alpha * T;
                if (out_invdepth) D += (1.f / depths[id]) * alpha * T;
                T = test_T;
                last = contributor;
            }
            size_t pix = (size_t)py * W + px;
            final_T[pix] = T;
            n_contrib[pix] = last;
            if (out_invdepth) out_invdepth[pix] = D;
            for (int ch = 0; ch < 3; ++ch) out_color[(size_t)ch * H * W + pix] = C[ch] + T * c->bg[ch];
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Threshold guard band of a rendered frame (test infrastructure for SURVEY.md 7.4-3).  The forward takes three
 * piecewise-constant decisions per (pixel, entry): `power > 0`, `alpha < 1/255` and `T (1 - alpha) < 1e-4`.  Any other
 * fp32 implementation (a GPU's exp2-based falloff, a different multiplication order of T) may land on the other side
 * of a threshold when the value sits within rounding distance of it, which changes a whole contribution and not the
 * last bit.  This walks the frame exactly like hso_render_fwd and reports
 *   pix_risk[H*W]   1 where some decision of that pixel had |alpha * 255 - 1| < guard_alpha, |test_T * 1e4 - 1| < guard_T
 *                   or 0 < |power| < 1e-6,
 *   gauss_risk[P]   1 for every Gaussian visited by such a pixel (a flipped contributor changes T for all of them),
 *   min_margin[3]   the smallest |alpha * 255 - 1|, |test_T * 1e4 - 1| and non-zero |power| seen.
 * Fixtures are reject-sampled until no pixel is at risk; comparators of larger scenes require every difference in the
 * decisions to sit on a pixel at risk and apply the strict gradient tolerance to the Gaussians that are not.
 * ------------------------------------------------------------------------------------------ */
int64_t hso_threshold_risk(const hso_camera* c, const uint32_t* ranges, const uint32_t* point_list, const float* xy,
                           const float* conic_opacity, float guard_alpha, float guard_T, uint8_t* pix_risk,
                           uint8_t* gauss_risk, double* min_margin) {
    const int W = c->W, H = c->H;
    const int gx = (W + HSO_TILE - 1) / HSO_TILE;
    int64_t n_risky = 0;
    min_margin[0] = min_margin[1] = min_margin[2] = 1e30;
    if (gauss_risk) memset(gauss_risk, 0, (size_t)c->P);
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int tile = (py / HSO_TILE) * gx + (px / HSO_TILE);
            uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
            float pxf = (float)px, pyf = (float)py;
            float T = 1.0f;
            int risky = 0, risky_T = 0;
            uint32_t k_end = end;
            for (uint32_t k = beg; k < end; ++k) {
                uint32_t id = point_list[k];
                float dx = xy[2 * id] - pxf, dy = xy[2 * id + 1] - pyf;
                const float* co = conic_opacity + 4 * id;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                double ap = fabs((double)power);
                if (ap > 0.0 && ap < min_margin[2]) min_margin[2] = ap;
                if (ap > 0.0 && ap < 1e-6) risky = 1;
                if (power > 0.0f) continue;
                float alpha = fminf_(0.99f, co[3] * expf(power));
                double ma = fabs((double)alpha * 255.0 - 1.0);
                if (ma < min_margin[0]) min_margin[0] = ma;
                if (ma < guard_alpha) risky = 1;
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1.f - alpha);
                double mt = fabs((double)test_T * 1e4 - 1.0);
                if (mt < min_margin[1]) min_margin[1] = mt;
                if (mt < guard_T) risky = risky_T = 1;
                if (test_T < 0.0001f) { k_end = k + 1; break; }
                T = test_T;
            }
            pix_risk[(size_t)py * W + px] = (uint8_t)risky;
            if (risky) {
                ++n_risky;
                /* the Gaussians whose gradient a flipped decision of this pixel reaches: every entry that contributes
                 * (or is within the guard band of contributing) up to the entry that ends the pixel -- up to the end
                 * of the list when it is the terminating decision that may flip */
                if (gauss_risk) {
                    const uint32_t stop = risky_T ? end : k_end;
                    for (uint32_t k = beg; k < stop; ++k) {
                        uint32_t id = point_list[k];
                        float dx = xy[2 * id] - pxf, dy = xy[2 * id + 1] - pyf;
                        const float* co = conic_opacity + 4 * id;
                        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 1e-6f) continue;
                        float alpha = fminf_(0.99f, co[3] * expf(power));
                        if ((double)alpha * 255.0 >= 1.0 - guard_alpha) gauss_risk[id] = 1;
                    }
                }
            }
        }
    return n_risky;
}

/* Test aid: the Gaussians that contribute to the pixels of `pix_mask` (u8 [H,W], non-zero = selected) -- the rows of
 * the gradient that a per-pixel decision taken elsewhere (e.g. which interval of the piecewise-linear CRF the pixel's
 * radiance falls into, a15) can reach.  whole_list = 0: the contributors under THIS implementation's decisions (up to
 * the terminating entry).  whole_list != 0: every entry of the pixel's tile list that contributes under ANY admissible
 * decision -- alpha within `guard_alpha` (relative) of 1/255 or above, no termination -- i.e. the rows another fp32
 * implementation that decided differently on this pixel may have touched.  Marks gauss_out[id] = 1; returns the number
 * of selected pixels. */
int64_t hso_pixel_reach(const hso_camera* c, const uint32_t* ranges, const uint32_t* point_list, const float* xy,
                        const float* conic_opacity, const uint8_t* pix_mask, uint8_t* gauss_out, int whole_list,
                        float guard_alpha) {
    const int W = c->W, H = c->H;
    const int gx = (W + HSO_TILE - 1) / HSO_TILE;
    int64_t n = 0;
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            if (!pix_mask[(size_t)py * W + px]) continue;
            ++n;
            int tile = (py / HSO_TILE) * gx + (px / HSO_TILE);
            uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
            float pxf = (float)px, pyf = (float)py;
            float T = 1.0f;
            for (uint32_t k = beg; k < end; ++k) {
                uint32_t id = point_list[k];
                float dx = xy[2 * id] - pxf, dy = xy[2 * id + 1] - pyf;
                const float* co = conic_opacity + 4 * id;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (whole_list) {
                    if (power > 1e-6f) continue;
                    float alpha = fminf_(0.99f, co[3] * expf(power));
                    if ((double)alpha * 255.0 >= 1.0 - guard_alpha) gauss_out[id] = 1;
                    continue;
                }
                if (power > 0.0f) continue;
                float alpha = fminf_(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1.f - alpha);
                if (test_T < 0.0001f) break;
                gauss_out[id] = 1;
                T = test_T;
            }
        }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * a10  render backward (per pixel, back to front).  Outputs per Gaussian:
 *   dL_dmean2D [P,2]  in NDC-scaled units (pixel gradient * 0.5*W, 0.5*H),
 *   dL_dconic  [P,3]  true partials w.r.t. (A,B,C) of power = -0.5(A dx^2 + C dy^2) - B dx dy,
 *   dL_dopacity[P], dL_dcolor [P,3].
 * abs_terms (optional, may be NULL; [P, 11]) receives, next to every one of the ten per-Gaussian sums (order: mean2D x, y;
 * conic A, B, C; opacity; colour r, g, b; inverse depth), the sum over the pixels of  w * |term|:  |term| = the per-pixel
 * term evaluated with every difference inside it replaced by the sum of the absolute values of its operands
 * (|c - accum| -> |c| + |accum|, the background term's sum of |bg_ch dL_ch|, |gdx A| + |gdy B| ...), i.e. the scale against
 * which an fp32 evaluation of that term, in any operation order, is accurate to a few ulp; w = 4 + the number of
 * T <- T / (1 - alpha) steps the replay has taken on that pixel before this contributor -- every step adds an ulp to T's
 * relative error, and T multiplies the whole term (the conditioning DESIGN.md's numerical contract speaks of) -- and in
 * column 10 the number of terms n.  Tests bound |other implementation - this one| per element by
 * 1e-4 |ref| + c 2^-24 sum w |term| (tests/helpers.py, assert_grads_bounded).
 * ------------------------------------------------------------------------------------------ */
int hso_render_bwd(const hso_camera* c, const uint32_t* ranges, const uint32_t* point_list,
                   const float* xy, const float* conic_opacity, const float* rgb,
                   const float* final_T, const uint32_t* n_contrib, const float* dL_dpix,
                   float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                   float* abs_terms,
                   const float* depths, const float* dL_dinvdepth_pix, float* dL_dinvdepth) {
    /* depths, dL_dinvdepth_pix [H*W], dL_dinvdepth [P] (all or none): the inverse-depth image is a fourth blended
     * channel without a background term; dL_dinvdepth receives sum_pix alpha T dL/dD(pix) per Gaussian */
    const int W = c->W, H = c->H, P = c->P;
    const int gx = (W + HSO_TILE - 1) / HSO_TILE;
    double* acc = (double*)calloc((size_t)P * 11, sizeof(double));
    if (!acc) return -1;
    double* aab = abs_terms ? (double*)calloc((size_t)P * 11, sizeof(double)) : NULL;
    if (abs_terms && !aab) { free(acc); return -1; }
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int tile = (py / HSO_TILE) * gx + (px / HSO_TILE);
            uint32_t beg = ranges[2 * tile];
            size_t pix = (size_t)py * W + px;
            uint32_t last = n_contrib[pix];
            const float T_final = final_T[pix];
            float T = T_final;
            float dLp[3];
            for (int ch = 0; ch < 3; ++ch) dLp[ch] = dL_dpix[(size_t)ch * H * W + pix];
            float bg_dot = (c->bg[0] * dLp[0] + c->bg[1] * dLp[1]) + c->bg[2] * dLp[2];
            float accum_rec[3] = {0.f, 0.f, 0.f}, last_color[3] = {0.f, 0.f, 0.f};
            float accum_invd = 0.f, last_invd = 0.f;
            const float dLd = dL_dinvdepth_pix ? dL_dinvdepth_pix[pix] : 0.f;
            float last_alpha = 0.f;
            float pxf = (float)px, pyf = (float)py;
            double wdepth = 4.0;   /* abs_terms weight: 4 + the T / (1 - alpha) steps behind this contributor (see below) */
            for (uint32_t kk = last; kk-- > 0;) {
                uint32_t id = point_list[beg + kk];
                float dx = xy[2 * id] - pxf, dy = xy[2 * id + 1] - pyf;
                const float* co = conic_opacity + 4 * id;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                float G = expf(power);
                float alpha = fminf_(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                T = T / (1.f - alpha);
                float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.f;
                double mag_alpha = 0.0;   /* magnitude of dL_dalpha: sums of absolute values instead of differences */
                double* a = acc + (size_t)id * 11;
                double* ab = aab ? aab + (size_t)id * 11 : NULL;
                if (dL_dinvdepth_pix) {
                    float invd = 1.f / depths[id];
                    accum_invd = last_alpha * last_invd + (1.f - last_alpha) * accum_invd;
                    last_invd = invd;
                    dL_dalpha += (invd - accum_invd) * dLd;
                    mag_alpha += (fabs((double)invd) + fabs((double)accum_invd)) * fabs((double)dLd);
                    a[10] += (double)(dchannel_dcolor * dLd);
                    if (ab) ab[9] += wdepth * fabs((double)(dchannel_dcolor * dLd));
                }
                for (int ch = 0; ch < 3; ++ch) {
                    float col = rgb[3 * id + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = col;
                    dL_dalpha += (col - accum_rec[ch]) * dLp[ch];
                    mag_alpha += (fabs((double)col) + fabs((double)accum_rec[ch])) * fabs((double)dLp[ch]);
                    a[6 + ch] += (double)(dchannel_dcolor * dLp[ch]);
                    if (ab) ab[6 + ch] += wdepth * fabs((double)(dchannel_dcolor * dLp[ch]));
                }
                dL_dalpha *= T;
                mag_alpha *= (double)T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                mag_alpha += fabs((double)(T_final / (1.f - alpha))) *
                             ((fabs((double)(c->bg[0] * dLp[0])) + fabs((double)(c->bg[1] * dLp[1]))) + fabs((double)(c->bg[2] * dLp[2])));
                float dL_dG = co[3] * dL_dalpha;
                float gdx = G * dx, gdy = G * dy;
                float dG_ddelx = -gdx * co[0] - gdy * co[1];
                float dG_ddely = -gdy * co[2] - gdx * co[1];
                float tmx = dL_dG * dG_ddelx * ddelx_dx;
                a[0] += (double)tmx;
                a[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                a[2] += (double)(-0.5f * gdx * dx * dL_dG);
                a[3] += (double)(-gdx * dy * dL_dG);
                a[4] += (double)(-0.5f * gdy * dy * dL_dG);
                a[5] += (double)(G * dL_dalpha);
                if (ab) {
                    const double mG = wdepth * fabs((double)co[3]) * mag_alpha;        /* weighted magnitude of dL_dG */
                    const double mx = fabs((double)(gdx * co[0])) + fabs((double)(gdy * co[1]));
                    const double my = fabs((double)(gdy * co[2])) + fabs((double)(gdx * co[1]));
                    ab[0] += mG * mx * (double)ddelx_dx;
                    ab[1] += mG * my * (double)ddely_dy;
                    ab[2] += fabs((double)(0.5f * gdx * dx)) * mG;
                    ab[3] += fabs((double)(gdx * dy)) * mG;
                    ab[4] += fabs((double)(0.5f * gdy * dy)) * mG;
                    ab[5] += wdepth * fabs((double)G) * mag_alpha;
                    ab[10] += 1.0;
                }
                wdepth += 1.0;
            }
        }
    for (int i = 0; i < P; ++i) {
        const double* a = acc + (size_t)i * 11;
        if (dL_dinvdepth) dL_dinvdepth[i] = (float)a[10];
        dL_dmean2D[2 * i] = (float)a[0]; dL_dmean2D[2 * i + 1] = (float)a[1];
        dL_dconic[3 * i] = (float)a[2]; dL_dconic[3 * i + 1] = (float)a[3]; dL_dconic[3 * i + 2] = (float)a[4];
        dL_dopacity[i] = (float)a[5];
        for (int ch = 0; ch < 3; ++ch) dL_dcolor[3 * i + ch] = (float)a[6 + ch];
        if (abs_terms) for (int k = 0; k < 11; ++k) abs_terms[(size_t)i * 11 + k] = (float)aab[(size_t)i * 11 + k];
    }
    free(acc);
    free(aab);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a11 + a12  preprocess backward (computeCov2D backward, projection backward, SH backward,
 * cov3D backward).  Any output pointer may be NULL.  Culled Gaussians (radii <= 0) get zeros.
 * ------------------------------------------------------------------------------------------ */
int hso_preprocess_bwd(const hso_camera* c, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const int* radii, const float* cov3D,
                       const uint8_t* clamped, const float* rgb, const float* dL_dmean2D, const float* dL_dconic,
                       const float* dL_dcolor,
                       float* dL_dmeans3D, float* dL_dshs, float* dL_dcolors_precomp,
                       float* dL_dscales, float* dL_drots, float* dL_dcov3D,
                       const float* opacities, float* dL_dopacity, const float* dL_dinvdepth) {
    /* opacities + dL_dopacity (in: gradient w.r.t. the opacity the render used; out: w.r.t. the input opacity) are
     * needed when c->antialias; dL_dinvdepth [P] (may be NULL) is the gradient w.r.t. 1/depth of each Gaussian */
    const int P = c->P;
    const int ncoef = (c->sh_degree + 1) * (c->sh_degree + 1);
    const float* V = c->viewmatrix;
    const float* PM = c->projmatrix;
    (void)colors_precomp; (void)cov3D_precomp;
    for (int i = 0; i < P; ++i) {
        float gm[3] = {0.f, 0.f, 0.f};
        float gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (dL_dshs) memset(dL_dshs + (size_t)i * c->M * 3, 0, sizeof(float) * c->M * 3);
        if (dL_dcolors_precomp) for (int k = 0; k < 3; ++k) dL_dcolors_precomp[3 * i + k] = 0.f;
        if (dL_dscales) for (int k = 0; k < 3; ++k) dL_dscales[3 * i + k] = 0.f;
        if (dL_drots) for (int k = 0; k < 4; ++k) dL_drots[4 * i + k] = 0.f;
        if (dL_dcov3D) for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = 0.f;
        if (dL_dmeans3D) for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = 0.f;
        if (radii[i] <= 0) continue;

        float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
        float pvx = xform_row(V, 0, x, y, z), pvy = xform_row(V, 1, x, y, z), pvz = xform_row(V, 2, x, y, z);
        const float* s6 = cov3D + 6 * i;

        /* ---- a11: conic -> cov2D -> (Sigma, t) ---- */
        hso_ewa e;
        ewa_setup(c, pvx, pvy, pvz, &e);
        float u0[3], u1[3];
        sym_mul(s6, e.a0, u0);
        sym_mul(s6, e.a1, u1);
        float a = dot3(e.a0, u0) + 0.3f, b = dot3(e.a1, u0), cc = dot3(e.a1, u1) + 0.3f;
        float denom = a * cc - b * b;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float gA = dL_dconic[3 * i], gB = dL_dconic[3 * i + 1], gC = dL_dconic[3 * i + 2];
        float aa_da = 0.f, aa_db = 0.f, aa_dc = 0.f;
        if (c->antialias && opacities && dL_dopacity) {
            /* opacity_eff = opacity * s, s = sqrt(max(eps, q)), q = det(cov - 0.3 I) / det(cov) */
            float det0 = (a - 0.3f) * (cc - 0.3f) - b * b;
            float q = det0 / denom;
            float sfac = sqrtf(fmaxf_(0.000025f, q));
            float g_eff = dL_dopacity[i];
            dL_dopacity[i] = g_eff * sfac;
            if (q > 0.000025f) {
                float k = g_eff * opacities[i] * 0.5f / sfac / (denom * denom);
                aa_da = k * ((cc - 0.3f) * denom - det0 * cc);
                aa_dc = k * ((a - 0.3f) * denom - det0 * a);
                aa_db = k * (2.f * b * (det0 - denom));
            }
        }
        if (dL_dinvdepth && dL_dinvdepth[i] != 0.f) {
            float dz = -dL_dinvdepth[i] / (pvz * pvz);
            for (int j = 0; j < 3; ++j) gm[j] += V[4 * j + 2] * dz;
        }
        if (denom2inv != 0.f) {
            float dLda = denom2inv * (-cc * cc * gA + b * cc * gB + (denom - a * cc) * gC) + aa_da;
            float dLdc = denom2inv * (-a * a * gC + a * b * gB + (denom - a * cc) * gA) + aa_dc;
            float dLdb = denom2inv * (2.f * b * cc * gA - (denom + 2.f * b * b) * gB + 2.f * a * b * gC) + aa_db;
            const float* p = e.a0; const float* q = e.a1;
            gcov[0] = p[0] * p[0] * dLda + p[0] * q[0] * dLdb + q[0] * q[0] * dLdc;
            gcov[3] = p[1] * p[1] * dLda + p[1] * q[1] * dLdb + q[1] * q[1] * dLdc;
            gcov[5] = p[2] * p[2] * dLda + p[2] * q[2] * dLdb + q[2] * q[2] * dLdc;
            gcov[1] = 2.f * p[0] * p[1] * dLda + (p[0] * q[1] + p[1] * q[0]) * dLdb + 2.f * q[0] * q[1] * dLdc;
            gcov[2] = 2.f * p[0] * p[2] * dLda + (p[0] * q[2] + p[2] * q[0]) * dLdb + 2.f * q[0] * q[2] * dLdc;
            gcov[4] = 2.f * p[1] * p[2] * dLda + (p[1] * q[2] + p[2] * q[1]) * dLdb + 2.f * q[1] * q[2] * dLdc;
            /* dL/da0, dL/da1 */
            float ga0[3], ga1[3];
            for (int j = 0; j < 3; ++j) {
                ga0[j] = 2.f * dLda * u0[j] + dLdb * u1[j];
                ga1[j] = 2.f * dLdc * u1[j] + dLdb * u0[j];
            }
            float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
            for (int j = 0; j < 3; ++j) {
                dJ00 += ga0[j] * V[4 * j + 0]; dJ02 += ga0[j] * V[4 * j + 2];
                dJ11 += ga1[j] * V[4 * j + 1]; dJ12 += ga1[j] * V[4 * j + 2];
            }
            float tz = 1.f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
            float dtx = e.clamp_x ? 0.f : -e.fx * tz2 * dJ02;
            float dty = e.clamp_y ? 0.f : -e.fy * tz2 * dJ12;
            float dtz = -e.fx * tz2 * dJ00 - e.fy * tz2 * dJ11 + (2.f * e.fx * e.tx) * tz3 * dJ02 +
                        (2.f * e.fy * e.ty) * tz3 * dJ12;
            for (int j = 0; j < 3; ++j)
                gm[j] += V[4 * j + 0] * dtx + V[4 * j + 1] * dty + V[4 * j + 2] * dtz;
        }

        /* ---- a12: screen-space mean -> mean3D ---- */
        {
            float phx = xform_row(PM, 0, x, y, z), phy = xform_row(PM, 1, x, y, z), phw = xform_row(PM, 3, x, y, z);
            float mw = 1.0f / (phw + 0.0000001f);
            float mul1 = phx * mw * mw, mul2 = phy * mw * mw;
            float gx2 = dL_dmean2D[2 * i], gy2 = dL_dmean2D[2 * i + 1];
            for (int j = 0; j < 3; ++j)
                gm[j] += (PM[4 * j + 0] * mw - PM[4 * j + 3] * mul1) * gx2 +
                         (PM[4 * j + 1] * mw - PM[4 * j + 3] * mul2) * gy2;
        }

        /* ---- a12: colour -> SH coefficients and view direction ---- */
        if (shs) {
            float dx = x - c->campos[0], dy = y - c->campos[1], dz = z - c->campos[2];
            float len = sqrtf((dx * dx + dy * dy) + dz * dz);
            float ux = dx / len, uy = dy / len, uz = dz / len;
            float bs[16], gb[16][3];
            sh_basis(c->sh_degree, ux, uy, uz, bs);
            sh_basis_grad(c->sh_degree, ux, uy, uz, gb);
            const float* sh = shs + (size_t)i * c->M * 3;
            float gdir[3] = {0.f, 0.f, 0.f};
            for (int ch = 0; ch < 3; ++ch) {
                float g = radiance_dact(c->radiance_activation, rgb[3 * i + ch], clamped[3 * i + ch]) * dL_dcolor[3 * i + ch];
                for (int k = 0; k < ncoef; ++k) {
                    if (dL_dshs) dL_dshs[((size_t)i * c->M + k) * 3 + ch] = bs[k] * g;
                    for (int d = 0; d < 3; ++d) gdir[d] += gb[k][d] * sh[3 * k + ch] * g;
                }
            }
            float dd = (ux * gdir[0] + uy * gdir[1]) + uz * gdir[2];
            float inv = 1.f / len;
            gm[0] += (gdir[0] - ux * dd) * inv;
            gm[1] += (gdir[1] - uy * dd) * inv;
            gm[2] += (gdir[2] - uz * dd) * inv;
        } else if (dL_dcolors_precomp) {
            for (int ch = 0; ch < 3; ++ch) dL_dcolors_precomp[3 * i + ch] = dL_dcolor[3 * i + ch];
        }

        /* ---- a12: Sigma -> scale, rotation ---- */
        if (scales && rotations) {
            float R[9], Mx[9];
            const float* q = rotations + 4 * i;
            quat_to_R(q, R);
            float s[3];
            for (int k = 0; k < 3; ++k) {
                s[k] = c->scale_modifier * scales[3 * i + k];
                for (int j = 0; j < 3; ++j) Mx[3 * k + j] = s[k] * R[3 * j + k];
            }
            /* full symmetric gradient matrix, off-diagonals halved */
            float G[9] = {gcov[0], 0.5f * gcov[1], 0.5f * gcov[2],
                          0.5f * gcov[1], gcov[3], 0.5f * gcov[4],
                          0.5f * gcov[2], 0.5f * gcov[4], gcov[5]};
            float dM[9]; /* dL/dM = 2 M G */
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 3; ++j)
                    dM[3 * k + j] = 2.f * ((Mx[3 * k + 0] * G[0 + j] + Mx[3 * k + 1] * G[3 + j]) + Mx[3 * k + 2] * G[6 + j]);
            float dR[9]; /* dL/dR_ik = s_k * dM_ki */
            for (int k = 0; k < 3; ++k) {
                float ds = 0.f;
                for (int j = 0; j < 3; ++j) { ds += dM[3 * k + j] * R[3 * j + k]; dR[3 * j + k] = s[k] * dM[3 * k + j]; }
                if (dL_dscales) dL_dscales[3 * i + k] = c->scale_modifier * ds;
            }
            float r = q[0], qx = q[1], qy = q[2], qz = q[3];
            if (dL_drots) {
                dL_drots[4 * i + 0] = 2.f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
                dL_drots[4 * i + 1] = 2.f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.f * qx * dR[4] - r * dR[5] + qz * dR[6] + r * dR[7] - 2.f * qx * dR[8]);
                dL_drots[4 * i + 2] = 2.f * (-2.f * qy * dR[0] + qx * dR[1] + r * dR[2] + qx * dR[3] + qz * dR[5] - r * dR[6] + qz * dR[7] - 2.f * qy * dR[8]);
                dL_drots[4 * i + 3] = 2.f * (-2.f * qz * dR[0] - r * dR[1] + qx * dR[2] + r * dR[3] - 2.f * qz * dR[4] + qy * dR[5] + qx * dR[6] + qy * dR[7]);
            }
        } else if (dL_dcov3D) {
            for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = gcov[k];
        }
        if (dL_dmeans3D) for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = gm[k];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Magnitude shadow of hso_preprocess_bwd (test infrastructure of the per-element gradient bound, tests/helpers.py
 * assert_grads_bounded): the same chain -- conic -> cov2D -> (Sigma, t); screen-space mean -> mean3D; colour -> SH
 * coefficients and view direction; Sigma -> scale, rotation -- evaluated on MAGNITUDES: the inputs are the weighted
 * sums of |term| that hso_render_bwd returns next to its ten per-Gaussian sums (abs_terms), every product enters with
 * the absolute value of its coefficient and every difference becomes a sum.  The outputs bound, element by element,
 * what an fp32 evaluation of a11 / a12 in any association order can be off by per unit of relative rounding: a net
 * Jacobian entry that is tiny because the chain cancels (the scale gradient of a nearly isotropic Gaussian) still
 * carries the rounding of the large intermediates it was cancelled from.
 * A_* inputs >= 0; any output pointer may be NULL.  A_opacity is in/out like dL_dopacity (antialiasing scales it).
 * ------------------------------------------------------------------------------------------ */
int hso_preprocess_bwd_abs(const hso_camera* c, const float* means3D, const float* shs,
                           const float* scales, const float* rotations, const int* radii, const float* cov3D,
                           const uint8_t* clamped, const float* rgb, const float* A_mean2D, const float* A_conic,
                           const float* A_color, float* B_means3D, float* B_shs, float* B_colors_precomp,
                           float* B_scales, float* B_rots, float* B_cov3D,
                           const float* opacities, float* A_opacity, const float* A_invdepth) {
    const int P = c->P;
    const int ncoef = (c->sh_degree + 1) * (c->sh_degree + 1);
    const float* V = c->viewmatrix;
    const float* PM = c->projmatrix;
    for (int i = 0; i < P; ++i) {
        double gm[3] = {0., 0., 0.};
        double gcov[6] = {0., 0., 0., 0., 0., 0.};
        if (B_shs) memset(B_shs + (size_t)i * c->M * 3, 0, sizeof(float) * c->M * 3);
        if (B_colors_precomp) for (int k = 0; k < 3; ++k) B_colors_precomp[3 * i + k] = 0.f;
        if (B_scales) for (int k = 0; k < 3; ++k) B_scales[3 * i + k] = 0.f;
        if (B_rots) for (int k = 0; k < 4; ++k) B_rots[4 * i + k] = 0.f;
        if (B_cov3D) for (int k = 0; k < 6; ++k) B_cov3D[6 * i + k] = 0.f;
        if (B_means3D) for (int k = 0; k < 3; ++k) B_means3D[3 * i + k] = 0.f;
        if (radii[i] <= 0) continue;
        float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
        float pvx = xform_row(V, 0, x, y, z), pvy = xform_row(V, 1, x, y, z), pvz = xform_row(V, 2, x, y, z);
        const float* s6 = cov3D + 6 * i;
        hso_ewa e;
        ewa_setup(c, pvx, pvy, pvz, &e);
        float u0[3], u1[3];
        sym_mul(s6, e.a0, u0);
        sym_mul(s6, e.a1, u1);
        double a = dot3(e.a0, u0) + 0.3f, b = dot3(e.a1, u0), cc = dot3(e.a1, u1) + 0.3f;
        double denom = a * cc - b * b;
        double denom2inv = 1.0 / ((denom * denom) + 0.0000001);
        double gA = A_conic[3 * i], gB = A_conic[3 * i + 1], gC = A_conic[3 * i + 2];
        double aa_da = 0., aa_db = 0., aa_dc = 0.;
        if (c->antialias && opacities && A_opacity) {
            double det0 = (a - 0.3) * (cc - 0.3) - b * b;
            double q = det0 / denom;
            double sfac = sqrt(q > 0.000025 ? q : 0.000025);
            double g_eff = A_opacity[i];
            A_opacity[i] = (float)(g_eff * sfac);
            if (q > 0.000025) {
                double k = fabs(g_eff * opacities[i] * 0.5 / sfac / (denom * denom));
                aa_da = k * (fabs((cc - 0.3) * denom) + fabs(det0 * cc));
                aa_dc = k * (fabs((a - 0.3) * denom) + fabs(det0 * a));
                aa_db = k * (fabs(2. * b * det0) + fabs(2. * b * denom));
            }
        }
        if (A_invdepth && A_invdepth[i] != 0.f) {
            double dz = A_invdepth[i] / ((double)pvz * pvz);
            for (int j = 0; j < 3; ++j) gm[j] += fabs((double)V[4 * j + 2]) * dz;
        }
        if (denom2inv != 0.) {
            double dLda = denom2inv * (cc * cc * gA + fabs(b * cc) * gB + (fabs(denom) + fabs(a * cc)) * gC) + aa_da;
            double dLdc = denom2inv * (a * a * gC + fabs(a * b) * gB + (fabs(denom) + fabs(a * cc)) * gA) + aa_dc;
            double dLdb = denom2inv * (2. * fabs(b * cc) * gA + (fabs(denom) + 2. * b * b) * gB + 2. * fabs(a * b) * gC) + aa_db;
            double p[3], q[3];
            for (int j = 0; j < 3; ++j) { p[j] = fabs((double)e.a0[j]); q[j] = fabs((double)e.a1[j]); }
            gcov[0] = p[0] * p[0] * dLda + p[0] * q[0] * dLdb + q[0] * q[0] * dLdc;
            gcov[3] = p[1] * p[1] * dLda + p[1] * q[1] * dLdb + q[1] * q[1] * dLdc;
            gcov[5] = p[2] * p[2] * dLda + p[2] * q[2] * dLdb + q[2] * q[2] * dLdc;
            gcov[1] = 2. * p[0] * p[1] * dLda + (p[0] * q[1] + p[1] * q[0]) * dLdb + 2. * q[0] * q[1] * dLdc;
            gcov[2] = 2. * p[0] * p[2] * dLda + (p[0] * q[2] + p[2] * q[0]) * dLdb + 2. * q[0] * q[2] * dLdc;
            gcov[4] = 2. * p[1] * p[2] * dLda + (p[1] * q[2] + p[2] * q[1]) * dLdb + 2. * q[1] * q[2] * dLdc;
            double ga0[3], ga1[3];
            for (int j = 0; j < 3; ++j) {
                ga0[j] = 2. * dLda * fabs((double)u0[j]) + dLdb * fabs((double)u1[j]);
                ga1[j] = 2. * dLdc * fabs((double)u1[j]) + dLdb * fabs((double)u0[j]);
            }
            double dJ00 = 0., dJ02 = 0., dJ11 = 0., dJ12 = 0.;
            for (int j = 0; j < 3; ++j) {
                dJ00 += ga0[j] * fabs((double)V[4 * j + 0]); dJ02 += ga0[j] * fabs((double)V[4 * j + 2]);
                dJ11 += ga1[j] * fabs((double)V[4 * j + 1]); dJ12 += ga1[j] * fabs((double)V[4 * j + 2]);
            }
            double tz = 1. / e.tz, tz2 = tz * tz, tz3 = fabs(tz2 * tz);
            double dtx = e.clamp_x ? 0. : fabs(e.fx * tz2) * dJ02;
            double dty = e.clamp_y ? 0. : fabs(e.fy * tz2) * dJ12;
            double dtz = fabs(e.fx * tz2) * dJ00 + fabs(e.fy * tz2) * dJ11 + fabs(2. * e.fx * e.tx) * tz3 * dJ02 +
                         fabs(2. * e.fy * e.ty) * tz3 * dJ12;
            for (int j = 0; j < 3; ++j)
                gm[j] += fabs((double)V[4 * j + 0]) * dtx + fabs((double)V[4 * j + 1]) * dty + fabs((double)V[4 * j + 2]) * dtz;
        }
        {
            double phx = xform_row(PM, 0, x, y, z), phy = xform_row(PM, 1, x, y, z), phw = xform_row(PM, 3, x, y, z);
            double mw = 1.0 / (phw + 0.0000001);
            double mul1 = phx * mw * mw, mul2 = phy * mw * mw;
            double gx2 = A_mean2D[2 * i], gy2 = A_mean2D[2 * i + 1];
            for (int j = 0; j < 3; ++j)
                gm[j] += (fabs(PM[4 * j + 0] * mw) + fabs(PM[4 * j + 3] * mul1)) * gx2 +
                         (fabs(PM[4 * j + 1] * mw) + fabs(PM[4 * j + 3] * mul2)) * gy2;
        }
        if (shs) {
            float dx = x - c->campos[0], dy = y - c->campos[1], dz = z - c->campos[2];
            float len = sqrtf((dx * dx + dy * dy) + dz * dz);
            float ux = dx / len, uy = dy / len, uz = dz / len;
            float bs[16], gb[16][3];
            sh_basis(c->sh_degree, ux, uy, uz, bs);
            sh_basis_grad(c->sh_degree, ux, uy, uz, gb);
            const float* sh = shs + (size_t)i * c->M * 3;
            double gdir[3] = {0., 0., 0.};
            for (int ch = 0; ch < 3; ++ch) {
                double g = fabs((double)radiance_dact(c->radiance_activation, rgb[3 * i + ch], clamped[3 * i + ch])) * A_color[3 * i + ch];
                for (int k = 0; k < ncoef; ++k) {
                    if (B_shs) B_shs[((size_t)i * c->M + k) * 3 + ch] = (float)(fabs((double)bs[k]) * g);
                    for (int d = 0; d < 3; ++d) gdir[d] += fabs((double)gb[k][d] * sh[3 * k + ch]) * g;
                }
            }
            double dd = (fabs((double)ux) * gdir[0] + fabs((double)uy) * gdir[1]) + fabs((double)uz) * gdir[2];
            double inv = 1. / len;
            gm[0] += (gdir[0] + fabs((double)ux) * dd) * inv;
            gm[1] += (gdir[1] + fabs((double)uy) * dd) * inv;
            gm[2] += (gdir[2] + fabs((double)uz) * dd) * inv;
        } else if (B_colors_precomp) {
            for (int ch = 0; ch < 3; ++ch) B_colors_precomp[3 * i + ch] = A_color[3 * i + ch];
        }
        if (scales && rotations) {
            float R[9];
            double Mx[9], sabs[3];
            const float* q = rotations + 4 * i;
            quat_to_R(q, R);
            for (int k = 0; k < 3; ++k) {
                sabs[k] = fabs((double)c->scale_modifier * scales[3 * i + k]);
                for (int j = 0; j < 3; ++j) Mx[3 * k + j] = sabs[k] * fabs((double)R[3 * j + k]);
            }
            double G[9] = {gcov[0], 0.5 * gcov[1], 0.5 * gcov[2], 0.5 * gcov[1], gcov[3], 0.5 * gcov[4],
                           0.5 * gcov[2], 0.5 * gcov[4], gcov[5]};
            double dM[9], dR[9];
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 3; ++j)
                    dM[3 * k + j] = 2. * ((Mx[3 * k + 0] * G[0 + j] + Mx[3 * k + 1] * G[3 + j]) + Mx[3 * k + 2] * G[6 + j]);
            for (int k = 0; k < 3; ++k) {
                double ds = 0.;
                for (int j = 0; j < 3; ++j) { ds += dM[3 * k + j] * fabs((double)R[3 * j + k]); dR[3 * j + k] = sabs[k] * dM[3 * k + j]; }
                if (B_scales) B_scales[3 * i + k] = (float)(fabs((double)c->scale_modifier) * ds);
            }
            double r = fabs((double)q[0]), qx = fabs((double)q[1]), qy = fabs((double)q[2]), qz = fabs((double)q[3]);
            if (B_rots) {
                B_rots[4 * i + 0] = (float)(2. * (qz * dR[1] + qy * dR[2] + qz * dR[3] + qx * dR[5] + qy * dR[6] + qx * dR[7]));
                B_rots[4 * i + 1] = (float)(2. * (qy * dR[1] + qz * dR[2] + qy * dR[3] + 2. * qx * dR[4] + r * dR[5] + qz * dR[6] + r * dR[7] + 2. * qx * dR[8]));
                B_rots[4 * i + 2] = (float)(2. * (2. * qy * dR[0] + qx * dR[1] + r * dR[2] + qx * dR[3] + qz * dR[5] + r * dR[6] + qz * dR[7] + 2. * qy * dR[8]));
                B_rots[4 * i + 3] = (float)(2. * (2. * qz * dR[0] + r * dR[1] + qx * dR[2] + r * dR[3] + 2. * qz * dR[4] + qy * dR[5] + qx * dR[6] + qy * dR[7]));
            }
        } else if (B_cov3D) {
            for (int k = 0; k < 6; ++k) B_cov3D[6 * i + k] = (float)gcov[k];
        }
        if (B_means3D) for (int k = 0; k < 3; ++k) B_means3D[3 * i + k] = (float)gm[k];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a15  HDR epilogue / prologue ([DESIGN], order fixed by assets/pipeline.png: H -> dt,CRF -> I).
 *   x = H*dt ; u = ln(max(x, 1e-8)) ; s = clamp((u-umin)/(umax-umin)*(K-1), 0, K-1)
 *   i = min(floor(s), K-2) ; f = s - i ; LDR = tab[c][i]*(1-f) + tab[c][i+1]*f
 * hdr, ldr are [3, n]; table is [3, K].
 * ------------------------------------------------------------------------------------------ */
#define HSO_LOG_EPS 1e-8f
static inline void crf_locate(float Hv, float dt, int K, float umin, float umax, int* idx, float* frac,
                              float* xo, int* interior) {
    float xv = Hv * dt;
    float u = logf(fmaxf_(xv, HSO_LOG_EPS));
    float s = (u - umin) / (umax - umin) * (float)(K - 1);
    int in = 1;
    if (!(s > 0.f)) { s = 0.f; in = 0; }
    if (s >= (float)(K - 1)) { s = (float)(K - 1); in = 0; }
    int i = (int)floorf(s);
    if (i > K - 2) i = K - 2;
    *idx = i; *frac = s - (float)i; *xo = xv; *interior = in && (xv > HSO_LOG_EPS);
}

int hso_tonemap_fwd(const float* hdr, int64_t n, float exposure, const float* table, int K,
                    float umin, float umax, float* ldr) {
    for (int ch = 0; ch < 3; ++ch)
        for (int64_t p = 0; p < n; ++p) {
            int i, in; float f, xv;
            crf_locate(hdr[ch * n + p], exposure, K, umin, umax, &i, &f, &xv, &in);
            const float* t = table + (size_t)ch * K;
            ldr[ch * n + p] = t[i] * (1.f - f) + t[i + 1] * f;
        }
    return 0;
}

/* dL_dhdr [3,n] (overwritten), dL_dtable [3,K] and dL_dexposure[1] are accumulated in double. */
int hso_tonemap_bwd(const float* hdr, int64_t n, float exposure, const float* table, int K,
                    float umin, float umax, const float* dL_dldr, float* dL_dhdr, float* dL_dtable,
                    float* dL_dexposure) {
    double* gt = (double*)calloc((size_t)3 * K, sizeof(double));
    if (!gt) return -1;
    double gexp = 0.0;
    float scale = (float)(K - 1) / (umax - umin);
    for (int ch = 0; ch < 3; ++ch)
        for (int64_t p = 0; p < n; ++p) {
            int i, in; float f, xv;
            float Hv = hdr[ch * n + p];
            crf_locate(Hv, exposure, K, umin, umax, &i, &f, &xv, &in);
            const float* t = table + (size_t)ch * K;
            float g = dL_dldr[ch * n + p];
            gt[ch * K + i] += (double)((1.f - f) * g);
            gt[ch * K + i + 1] += (double)(f * g);
            float gx = 0.f;
            if (in) gx = g * (t[i + 1] - t[i]) * scale / xv;
            dL_dhdr[ch * n + p] = gx * exposure;
            gexp += (double)(gx * Hv);
        }
    for (int k = 0; k < 3 * K; ++k) dL_dtable[k] = (float)gt[k];
    dL_dexposure[0] = (float)gexp;
    free(gt);
    return 0;
}
