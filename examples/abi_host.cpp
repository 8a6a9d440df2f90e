// Stand-alone host for the C ABI of include/hdrsplat.h: no PyTorch, no Python -- hipMalloc'ed buffers, plain structs,
// one stream.  This is what a C / C++ / Go(cgo) / Rust(FFI) caller of libhdrsplat.so writes; the Python layer
// (casualhdrsplat_amd/rasterizer.py) does the same through ctypes with torch-owned memory.
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/abi_host.cpp -L casualhdrsplat_amd -lhdrsplat \
//         -Wl,-rpath,$PWD/casualhdrsplat_amd -o examples/abi_host
//   examples/abi_host scene.bin out.bin
//
// scene.bin (little endian): int32 P, M, deg, W, H; float tanfovx, tanfovy; float bg[3], view[16], proj[16], campos[3];
// then float means[P*3], opac[P], shs[P*M*3], scales[P*3], rots[P*4], dL_dcolor[3*H*W].
// out.bin: uint32 num_rendered; float color[3*H*W]; int32 radii[P]; float dmeans3D[P*3], dmeans2D[P*3], dopac[P],
// dshs[P*M*3], dscales[P*3], drots[P*4].
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hdrsplat.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
#define HS(x) do { int r_ = (x); if (r_ != HS_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #x, r_, hs_last_error()); std::exit(3); } } while (0)

template <typename T>
static T* to_device(const std::vector<T>& h) {
    T* d = nullptr;
    CK(hipMalloc(&d, h.size() * sizeof(T) + 16));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
template <typename T>
static T* device_alloc(size_t n) {
    T* d = nullptr;
    CK(hipMalloc(&d, n * sizeof(T) + 256));
    return d;
}
template <typename T>
static std::vector<T> read_vec(std::FILE* f, size_t n) {
    std::vector<T> v(n);
    if (std::fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "scene file too short\n"); std::exit(1); }
    return v;
}
template <typename T>
static void write_dev(std::FILE* f, const T* d, size_t n) {
    std::vector<T> h(n);
    CK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
    std::fwrite(h.data(), sizeof(T), n, f);
}

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]); return 1; }
    std::FILE* in = std::fopen(argv[1], "rb");
    if (!in) { std::perror(argv[1]); return 1; }
    const auto hd = read_vec<int32_t>(in, 5);
    const int P = hd[0], M = hd[1], deg = hd[2], W = hd[3], H = hd[4];
    const auto tf = read_vec<float>(in, 2);
    const auto bg = read_vec<float>(in, 3), view = read_vec<float>(in, 16), proj = read_vec<float>(in, 16), campos = read_vec<float>(in, 3);
    const auto means = read_vec<float>(in, (size_t)P * 3), opac = read_vec<float>(in, P), shs = read_vec<float>(in, (size_t)P * M * 3);
    const auto scales = read_vec<float>(in, (size_t)P * 3), rots = read_vec<float>(in, (size_t)P * 4);
    const auto dL = read_vec<float>(in, (size_t)3 * H * W);
    std::fclose(in);

    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    hs_fwd_args f = {};
    f.dims = hs_dims{P, M, deg, W, H, 1, 0, 0, 0};
    f.tanfovx = tf[0]; f.tanfovy = tf[1]; f.scale_modifier = 1.f;
    f.bg = to_device(bg); f.viewmatrices = to_device(view); f.projmatrices = to_device(proj); f.camposes = to_device(campos);
    f.means3D = to_device(means); f.opacities = to_device(opac); f.shs = to_device(shs);
    f.scales = to_device(scales); f.rotations = to_device(rots);
    float* out_color = device_alloc<float>((size_t)3 * H * W);
    int32_t* radii = device_alloc<int32_t>(P);
    f.out_color = out_color; f.radii = radii;

    // 1. preprocess with a geometry workspace sized for P; read num_rendered like the published host code does
    hs_sizes sz; hs_layout lay;
    HS(hs_plan(&f.dims, &sz, &lay));
    f.geom = device_alloc<char>(sz.geom_bytes);
    f.stages = HS_STAGE_PREPROCESS;
    HS(hs_forward(&f, stream));
    hs_counters ctr;
    CK(hipMemcpyAsync(&ctr, (char*)f.geom + lay.counters, sizeof ctr, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    // 2. binning + render with workspaces sized for exactly R pairs
    f.dims.capacity = ctr.num_rendered;
    HS(hs_plan(&f.dims, &sz, &lay));
    f.binning = device_alloc<char>(sz.binning_bytes);
    f.image = device_alloc<char>(sz.image_bytes);
    f.stages = HS_STAGE_BIN | HS_STAGE_RENDER;
    HS(hs_forward(&f, stream));

    // 3. backward
    hs_bwd_args b = {};
    b.dims = f.dims; b.tanfovx = f.tanfovx; b.tanfovy = f.tanfovy; b.scale_modifier = 1.f; b.stages = HS_BWD_ALL;
    b.bg = f.bg; b.viewmatrices = f.viewmatrices; b.projmatrices = f.projmatrices; b.camposes = f.camposes;
    b.means3D = f.means3D; b.opacities = f.opacities; b.shs = f.shs; b.scales = f.scales; b.rotations = f.rotations;
    b.geom = f.geom; b.binning = f.binning; b.image = f.image;
    b.bwd = device_alloc<char>(sz.bwd_bytes);
    b.dL_dout_color = to_device(dL);
    float* dm3 = device_alloc<float>((size_t)P * 3); float* dm2 = device_alloc<float>((size_t)P * 3);
    float* dop = device_alloc<float>(P); float* dsh = device_alloc<float>((size_t)P * M * 3);
    float* dsc = device_alloc<float>((size_t)P * 3); float* dro = device_alloc<float>((size_t)P * 4);
    b.dL_dmeans3D = dm3; b.dL_dmeans2D = dm2; b.dL_dopacities = dop; b.dL_dshs = dsh; b.dL_dscales = dsc; b.dL_drotations = dro;
    HS(hs_backward(&b, stream));
    CK(hipStreamSynchronize(stream));

    std::FILE* out = std::fopen(argv[2], "wb");
    if (!out) { std::perror(argv[2]); return 1; }
    std::fwrite(&ctr.num_rendered, 4, 1, out);
    write_dev(out, out_color, (size_t)3 * H * W);
    write_dev(out, radii, P);
    write_dev(out, dm3, (size_t)P * 3); write_dev(out, dm2, (size_t)P * 3); write_dev(out, dop, P);
    write_dev(out, dsh, (size_t)P * M * 3); write_dev(out, dsc, (size_t)P * 3); write_dev(out, dro, (size_t)P * 4);
    std::fclose(out);
    std::printf("abi_host: P=%d %dx%d num_rendered=%u version=%d\n", P, W, H, ctr.num_rendered, hs_version());
    return 0;
}
