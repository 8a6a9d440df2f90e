// Reproducer, second form (VERDICT r5 next #7a): a LONG captured stream -- thousands of kernel nodes, as a training step
// has -- with small memsets in between (PyTorch's two-pass reduction clears a few-byte semaphore with one ahead of its
// kernel; this library cleared 64 B ... 48 MB).  Pattern per memset m: K_inc (S[m] += 1: what the previous user left
// behind) ; memset S[m] = 0 ; K_use (OUT[m] = S[m] == 0 ? good : bad ; S[m] = 7).  A memset node that does not run between
// its neighbours on some replay leaves OUT[m] == bad.
// build + run: hipcc --offload-arch=gfx950 -O2 scripts/repro/graph_memset_long.hip -o /tmp/graph_memset_long && /tmp/graph_memset_long
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
__global__ void k_fill(float* a, int n, float v) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) a[i] = a[i] * 0.5f + v; }
__global__ void k_inc(unsigned* s, int words) { if (threadIdx.x < words) s[threadIdx.x] += 1u; }
__global__ void k_use(unsigned* s, int words, unsigned* out) {
    if (threadIdx.x == 0) { unsigned bad = 0; for (int w = 0; w < words; ++w) bad |= s[w]; *out = bad ? 0xBADu : 0x600Du; }
    __syncthreads();
    if (threadIdx.x < words) s[threadIdx.x] = 7u;
}
int main() {
    const int M = 64, FILL = 40, NF = 1 << 16;        // 64 memsets, 40 filler kernels around each: ~5000 nodes
    int total_bad = 0;
    for (int words : {1, 16, 1024}) {
        unsigned *S, *OUT; float* A;
        CK(hipMalloc(&S, M * words * 4)); CK(hipMalloc(&OUT, M * 4)); CK(hipMalloc(&A, NF * 4));
        hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        CK(hipMemsetAsync(S, 0, M * words * 4, s)); CK(hipMemsetAsync(A, 0, NF * 4, s));
        auto enqueue = [&]() -> int {
            for (int m = 0; m < M; ++m) {
                for (int f = 0; f < FILL; ++f) k_fill<<<NF / 256, 256, 0, s>>>(A, NF, (float)f);
                k_inc<<<1, 1024, 0, s>>>(S + m * words, words);
                CK(hipMemsetAsync(S + m * words, 0, words * 4, s));
                k_use<<<1, 1024, 0, s>>>(S + m * words, words, OUT + m);
            }
            return 0;
        };
        hipGraph_t g; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        if (enqueue()) return 2;
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
        std::vector<unsigned> out(M);
        int bad = 0;
        for (int rep = 0; rep < 50; ++rep) {
            CK(hipGraphLaunch(exec, s));
            CK(hipMemcpyAsync(out.data(), OUT, M * 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            int b = 0; for (int m = 0; m < M; ++m) b += out[m] != 0x600Du;
            if (b && bad < 3) printf("  words=%d replay %d: %d of %d memsets did not run between their neighbours\n", words, rep, b, M);
            bad += b;
        }
        printf("memsets of %d B in a graph of %zu nodes, 50 replays: %s\n", words * 4, nn, bad ? "WRONG" : "ok");
        total_bad += bad;
        CK(hipFree(S)); CK(hipFree(OUT)); CK(hipFree(A)); CK(hipStreamDestroy(s));
    }
    printf("RESULT graph_memset_long: %s\n", total_bad ? "memset node misplaced (reproduced)" : "memset nodes keep their place (not reproduced)");
    return 0;
}
