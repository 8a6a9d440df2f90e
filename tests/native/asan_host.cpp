// Host-side sanitizer driver ("-fsanitize=address on the host-side C++ in CI on the CPU box").
// Exercises every C-ABI entry point's HOST code -- hs_plan's carving arithmetic, the argument validation of
// hs_forward / hs_backward / hs_mark_visible / hs_sh_backward_views / hs_sort_pairs / hs_render_stats / hs_spline_poses, the
// thread-local error text -- with AddressSanitizer + UBSan on the host objects of libhdrsplat (built by
// `make -C casualhdrsplat_amd/csrc asan`).  No GPU is needed or touched: every call here must return before its first
// HIP call (bad arguments) or is pure host code (hs_plan).  Exit code 0 = clean.
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "hdrsplat.h"

#define CHECK(cond)                                                           \
    do {                                                                      \
        if (!(cond)) { std::fprintf(stderr, "FAILED: %s (line %d): %s\n", #cond, __LINE__, hs_last_error()); return 1; } \
    } while (0)

static int run() {
    CHECK(hs_version() == HS_VERSION);
    hs_sizes sz;
    hs_layout lay;
    // hs_plan over a sweep of shapes, including the awkward ones (empty cloud, ragged image, many poses, huge capacity)
    const int Ps[] = {0, 1, 7, 1001, 1000000};
    const int Ws[] = {1, 17, 1920, 7680};
    const int Ns[] = {1, 3, 8, 64};
    for (int P : Ps)
        for (int W : Ws)
            for (int N : Ns)
                for (int K : {0, 2, 256, 4096}) {
                    hs_dims d{P, 16, 3, W, (W * 9 + 15) / 16, N, (int64_t)P * 8 + 5, K, 0};
                    CHECK(hs_plan(&d, &sz, &lay) == HS_OK);
                    CHECK(sz.geom_bytes > 0 && sz.binning_bytes >= 0 && sz.image_bytes > 0 && sz.bwd_bytes >= 0);
                    CHECK(lay.rec % 256 == 0 && lay.point_list % 256 == 0 && lay.pose_hdr % 256 == 0 && lay.inst_grads % 256 == 0);
                    CHECK(lay.pair_grads < sz.bwd_bytes || sz.bwd_bytes == 0 || d.capacity == 0);
                }
    {   // rejected dims
        hs_dims d{-1, 0, 0, 16, 16, 1, 0, 0, 0};
        CHECK(hs_plan(&d, &sz, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "bad dims"));
        d = hs_dims{10, 0, 0, 16, 16, 0, 0, 0, 0};
        CHECK(hs_plan(&d, &sz, nullptr) == HS_EINVAL);
        d = hs_dims{10, 0, 0, 16, 16, 1, 0, 1, 0};       // a CRF table needs two knots
        CHECK(hs_plan(&d, &sz, nullptr) == HS_EINVAL);
        d = hs_dims{2000000000, 0, 0, 16, 16, 2, 0, 0, 0};  // instance index would overflow 32 bits
        CHECK(hs_plan(&d, &sz, nullptr) == HS_EINVAL);
        CHECK(hs_plan(nullptr, nullptr, nullptr) == HS_EINVAL);
    }
    // validation of the launch entry points: every one of these must fail before touching HIP
    CHECK(hs_forward(nullptr, nullptr) == HS_EINVAL);
    CHECK(hs_backward(nullptr, nullptr) == HS_EINVAL);
    hs_fwd_args f;
    std::memset(&f, 0, sizeof f);
    f.dims = hs_dims{10, 1, 0, 32, 32, 1, 100, 0, 0};
    CHECK(hs_forward(&f, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "null"));
    float* fake = reinterpret_cast<float*>(4096);  // never dereferenced on the host
    f.means3D = f.viewmatrices = f.projmatrices = f.camposes = f.bg = f.shs = f.colors_precomp = f.scales = f.rotations = fake;
    CHECK(hs_forward(&f, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "exactly one"));
    f.colors_precomp = nullptr;
    f.dims.sh_degree = 3;
    CHECK(hs_forward(&f, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "sh_degree"));
    f.dims.sh_degree = 0;
    f.dims.M = 40;
    CHECK(hs_forward(&f, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "at most 32"));
    f.dims.M = 1;
    f.flags = HS_FLAG_RADIANCE_EXP | HS_FLAG_RADIANCE_SOFTPLUS;
    CHECK(hs_forward(&f, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "mutually exclusive"));
    f.flags = HS_FLAG_HDR;  // HDR without exposure / table
    f.opacities = fake; f.radii = reinterpret_cast<int32_t*>(fake); f.geom = fake;
    CHECK(hs_forward(&f, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "HDR"));
    hs_bwd_args b;
    std::memset(&b, 0, sizeof b);
    b.dims = f.dims;
    CHECK(hs_backward(&b, nullptr) == HS_EINVAL);
    b.means3D = b.viewmatrices = b.projmatrices = b.camposes = b.bg = b.shs = b.scales = b.rotations = fake;
    CHECK(hs_backward(&b, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "null workspace"));
    // a Gaussian range is for HS_BWD_PROJECT alone, starts on a workgroup boundary and lies inside [0, P]
    b.geom = b.binning = b.image = b.bwd = fake; b.dL_dout_color = fake;
    b.stages = HS_BWD_PROJECT; b.g_begin = 64; b.g_end = b.dims.P;
    CHECK(hs_backward(&b, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "g_begin"));
    b.g_begin = 0; b.g_end = b.dims.P + 1;
    CHECK(hs_backward(&b, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "g_begin"));
    b.g_end = b.dims.P; b.stages = HS_BWD_ALL;
    CHECK(hs_backward(&b, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "HS_BWD_PROJECT"));
    b.g_begin = b.g_end = 0; b.geom = b.binning = b.image = b.bwd = nullptr; b.dL_dout_color = nullptr;
    CHECK(hs_mark_visible(-1, nullptr, nullptr, nullptr, nullptr) == HS_EINVAL);
    CHECK(hs_mark_visible(0, nullptr, nullptr, nullptr, nullptr) == HS_OK);
    CHECK(hs_sort_pairs(nullptr, nullptr, nullptr, nullptr, 5, 40, nullptr, nullptr) == HS_EINVAL);
    CHECK(hs_sort_pairs(nullptr, nullptr, nullptr, nullptr, 0, 40, nullptr, nullptr) == HS_OK);
    CHECK(hs_sort_tmp_bytes(1000000) > 1000000 * 24);
    CHECK(hs_sh_backward_views(10, 4, 3, 2, fake, fake, fake, fake, nullptr) == HS_EINVAL);
    CHECK(hs_sh_backward_views(0, 16, 3, 2, nullptr, nullptr, nullptr, nullptr, nullptr) == HS_OK);
    CHECK(hs_render_stats(nullptr, nullptr, nullptr, nullptr, nullptr) == HS_EINVAL);
    CHECK(hs_spline_poses(3, 5, 1, fake, fake, fake, fake, fake, (int32_t*)fake, nullptr) == HS_EINVAL);   // a cubic segment needs four knots
    CHECK(hs_spline_poses(4, 5, 2, fake, fake, fake, fake, fake, (int32_t*)fake, nullptr) == HS_EINVAL && std::strstr(hs_last_error(), "kind"));
    CHECK(hs_spline_poses(4, 5, 1, nullptr, fake, fake, fake, fake, (int32_t*)fake, nullptr) == HS_EINVAL);
    CHECK(hs_spline_poses(2, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == HS_OK);
    return 0;
}

int main() {
    // the error text is thread-local: two threads validating at once must not trample each other's message
    int rc[2] = {1, 1};
    std::thread t0([&] { rc[0] = run(); }), t1([&] { rc[1] = run(); });
    t0.join();
    t1.join();
    if (rc[0] || rc[1]) return 1;
    std::puts("asan_host: clean");
    return 0;
}
