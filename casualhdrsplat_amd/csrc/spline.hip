// Camera poses along the trajectory spline, with their Jacobian, in ONE launch (hs_spline_poses).
//
// The image-formation model of the reference (/root/reference/assets/pipeline.png, Readme.md:54) samples N virtual camera poses
// inside every captured frame's exposure window from a spline through learnable se(3) control knots.  Written with tensor
// operations (image_formation.TrajectorySpline.pose_at: exponential / logarithm maps, products) that is ~300 tiny kernels
// forward and ~700 backward whatever the number of poses: 9 ms of host time per call, 4 ms of GPU time even inside a
// captured graph -- several times the rasterizer's own step at BASELINE c3.  The arithmetic is a few thousand flops per pose:
// here one thread per (sample time, input) evaluates it in forward-mode dual numbers (float64: the value and ONE directional
// derivative per thread -- 25 threads per sample, one for each of the 4 x 6 knot corrections that govern the sample's segment
// and one for the sample time; a thread that carried all 25 partials itself spent 1.4 ms in scratch memory), so the launch
// returns the poses AND d pose / d (knots, time); the backward pass in Python is one multiply-sum with that Jacobian.
//
//   knot_j = exp(delta_j) * base_j                                 (left-multiplied se(3) correction; xi = (rho, omega))
//   cubic  (cumulative uniform B-spline, Lovegrove et al. 2013): t in [j + 1, j + 2):  u = t - (j + 1),
//          pose = exp(B3(u) x3) exp(B2(u) x2) exp(B1(u) x1) knot_j,  x_k = log(knot_{j+k} knot_{j+k-1}^-1)
//   linear (geodesic between two knots):                          t in [j, j + 1):  pose = exp(u x1) knot_j
// Same formulas, thresholds and series as the tensor implementation (which stays as the CPU path and as this kernel's
// oracle: tests/test_image_formation.py compares values and gradients).
#include "hs_common.h"

namespace hs {
namespace {

constexpr int NI = 25;   // inputs of a sample: knots j .. j+3 (6 corrections each), then the sample time
constexpr int ND = 1;    // directional derivatives a thread carries

struct Dual {
    double v;
    double d[ND];
};
__device__ __forceinline__ Dual mk(double v) { Dual r; r.v = v; for (int i = 0; i < ND; ++i) r.d[i] = 0.0; return r; }
// input number `i`, seen by the thread that differentiates with respect to input `dir`
__device__ __forceinline__ Dual seed(double v, int i, int dir) { Dual r = mk(v); r.d[0] = i == dir ? 1.0 : 0.0; return r; }
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ Dual operator-(const Dual& a) { Dual r; r.v = -a.v; for (int i = 0; i < ND; ++i) r.d[i] = -a.d[i]; return r; }
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
    Dual r; const double inv = 1.0 / b.v; r.v = a.v * inv;
    for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
__device__ __forceinline__ Dual operator*(double s, const Dual& a) { Dual r; r.v = s * a.v; for (int i = 0; i < ND; ++i) r.d[i] = s * a.d[i]; return r; }
__device__ __forceinline__ Dual operator+(double s, const Dual& a) { Dual r = a; r.v += s; return r; }
__device__ __forceinline__ Dual operator-(double s, const Dual& a) { Dual r = -a; r.v += s; return r; }
__device__ __forceinline__ Dual chain(const Dual& a, double f, double df) { Dual r; r.v = f; for (int i = 0; i < ND; ++i) r.d[i] = df * a.d[i]; return r; }
__device__ __forceinline__ Dual dsin(const Dual& a) { return chain(a, sin(a.v), cos(a.v)); }
__device__ __forceinline__ Dual dcos(const Dual& a) { return chain(a, cos(a.v), -sin(a.v)); }
__device__ __forceinline__ Dual dsqrt(const Dual& a) { const double s = sqrt(a.v); return chain(a, s, 0.5 / s); }
__device__ __forceinline__ Dual dacos(const Dual& a) { return chain(a, acos(a.v), -1.0 / sqrt(1.0 - a.v * a.v)); }

struct Rigid { Dual R[9]; Dual t[3]; };   // [R t; 0 1]

__device__ void mul3(const Dual* A, const Dual* B, Dual* C) {   // 3 x 3 products (C may not alias A or B)
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
__device__ void hat(const Dual* w, Dual* K) {
    const Dual z = mk(0.0);
    K[0] = z; K[1] = -w[2]; K[2] = w[1];
    K[3] = w[2]; K[4] = z; K[5] = -w[0];
    K[6] = -w[1]; K[7] = w[0]; K[8] = z;
}
__device__ void rigid_mul(const Rigid& A, const Rigid& B, Rigid& C) {
    mul3(A.R, B.R, C.R);
    for (int r = 0; r < 3; ++r) C.t[r] = A.R[3 * r] * B.t[0] + A.R[3 * r + 1] * B.t[1] + A.R[3 * r + 2] * B.t[2] + A.t[r];
}
__device__ void rigid_inv(const Rigid& A, Rigid& C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C.R[3 * r + c] = A.R[3 * c + r];
    for (int r = 0; r < 3; ++r) C.t[r] = -(C.R[3 * r] * A.t[0] + C.R[3 * r + 1] * A.t[1] + C.R[3 * r + 2] * A.t[2]);
}

// image_formation.se3_exp
__device__ void se3_exp(const Dual* xi, Rigid& T) {
    const Dual* rho = xi; const Dual* om = xi + 3;
    const Dual th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    const Dual th = dsqrt(1e-20 + th2);
    Dual A, B, C;
    if (th2.v < 1e-8) {
        A = 1.0 - (1.0 / 6.0) * th2; B = 0.5 - (1.0 / 24.0) * th2; C = 1.0 / 6.0 - (1.0 / 120.0) * th2;
    } else {
        A = dsin(th) / th; B = (1.0 - dcos(th)) / (1e-20 + th2); C = (1.0 - A) / (1e-20 + th2);
    }
    Dual K[9], K2[9];
    hat(om, K);
    mul3(K, K, K2);
    Dual V[9];
    for (int i = 0; i < 9; ++i) {
        const double e = (i % 4 == 0) ? 1.0 : 0.0;
        T.R[i] = e + (A * K[i] + B * K2[i]);
        V[i] = e + (B * K[i] + C * K2[i]);
    }
    for (int r = 0; r < 3; ++r) T.t[r] = V[3 * r] * rho[0] + V[3 * r + 1] * rho[1] + V[3 * r + 2] * rho[2];
}

// image_formation.se3_log
__device__ void se3_log(const Rigid& T, Dual* xi) {
    Dual c = 0.5 * ((T.R[0] + T.R[4] + T.R[8]) - mk(1.0));
    if (c.v > 1.0 - 1e-7) c = mk(1.0 - 1e-7);            // (a clamped value has no derivative: torch.clamp)
    else if (c.v < -1.0 + 1e-7) c = mk(-1.0 + 1e-7);
    const Dual th = dacos(c);
    const bool small = th.v < 1e-4;
    const Dual th2 = th * th;
    const Dual k = small ? (0.5 + (1.0 / 12.0) * th2) : th / (1e-20 + 2.0 * dsin(th));
    Dual om[3] = {k * (T.R[7] - T.R[5]), k * (T.R[2] - T.R[6]), k * (T.R[3] - T.R[1])};
    Dual K[9], K2[9];
    hat(om, K);
    mul3(K, K, K2);
    Dual coef;
    if (small) {
        coef = mk(1.0 / 12.0);
    } else {
        const Dual A = dsin(th) / (1e-20 + th);
        const Dual B = (1.0 - dcos(th)) / (1e-20 + th2);
        coef = (1.0 - A / (2.0 * B)) / (1e-20 + th2);
    }
    for (int r = 0; r < 3; ++r) {
        Dual acc = mk(0.0);
        for (int cidx = 0; cidx < 3; ++cidx) {
            const double e = (r == cidx) ? 1.0 : 0.0;
            const Dual vinv = e + (coef * K2[3 * r + cidx] - 0.5 * K[3 * r + cidx]);
            acc = acc + vinv * T.t[cidx];
        }
        xi[r] = acc;
    }
    xi[3] = om[0]; xi[4] = om[1]; xi[5] = om[2];
}

// 32 lanes per sample (25 of them work: lane = the input they differentiate with respect to), two samples per wave
__global__ void __launch_bounds__(64) spline_poses_kernel(int J, int T, int kind, const float* delta, const float* base,
                                                          const float* times, float* w2c, float* jac, int* seg) {
    const int s = blockIdx.x * 2 + (threadIdx.x >> 5);
    const int dir = threadIdx.x & 31;
    if (s >= T || dir >= NI) return;
    const double t = (double)times[s];
    const int fl = (int)floor(t);
    const bool cubic = kind == 1;
    const int j = cubic ? min(max(fl - 1, 0), J - 4) : min(max(fl, 0), J - 2);
    const Dual u = seed(t - (double)(cubic ? j + 1 : j), 24, dir);
    const int nk = cubic ? 4 : 2;
    Rigid Tk[4];
    for (int k = 0; k < nk; ++k) {
        Dual xi[6];
        for (int c = 0; c < 6; ++c) xi[c] = seed((double)delta[(j + k) * 6 + c], 6 * k + c, dir);
        Rigid E, Bk;
        se3_exp(xi, E);
        const float* b = base + (int64_t)(j + k) * 16;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Bk.R[3 * r + c] = mk((double)b[4 * r + c]);
            Bk.t[r] = mk((double)b[4 * r + 3]);
        }
        rigid_mul(E, Bk, Tk[k]);
    }
    Rigid pose = Tk[0];
    const Dual u2 = u * u, u3 = u2 * u;
    for (int k = 0; k + 1 < nk; ++k) {
        Rigid inv, rel, E, next;
        rigid_inv(Tk[k], inv);
        rigid_mul(Tk[k + 1], inv, rel);
        Dual x[6];
        se3_log(rel, x);
        Dual bk;
        if (!cubic) bk = u;
        else if (k == 0) bk = (1.0 / 6.0) * (5.0 + (3.0 * u - 3.0 * u2 + u3));
        else if (k == 1) bk = (1.0 / 6.0) * (1.0 + (3.0 * u + 3.0 * u2 - 2.0 * u3));
        else bk = (1.0 / 6.0) * u3;
        for (int c = 0; c < 6; ++c) x[c] = bk * x[c];
        se3_exp(x, E);
        rigid_mul(E, pose, next);
        pose = next;
    }
    float* o = w2c + (int64_t)s * 16;
    float* jo = jac + (int64_t)s * 12 * NI;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 4; ++c) {
            const Dual& e = c < 3 ? pose.R[3 * r + c] : pose.t[r];
            if (dir == 0) o[4 * r + c] = (float)e.v;
            jo[(4 * r + c) * NI + dir] = (float)e.d[0];
        }
    }
    if (dir == 0) {
        o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
        seg[s] = j;
    }
}

}  // namespace

int launch_spline_poses(int J, int T, int kind, const float* delta, const float* base, const float* times, float* w2c,
                        float* jac, int* seg, hipStream_t s) {
    spline_poses_kernel<<<ceil_div(T, 2), 64, 0, s>>>(J, T, kind, delta, base, times, w2c, jac, seg);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

}  // namespace hs
