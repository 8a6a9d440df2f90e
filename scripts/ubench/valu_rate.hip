// Microbenchmark: issue rate of the VALU instruction kinds the render kernels are made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));

#define ITERS 4096
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
    float v[8];
    float2_ w[8];
    for (int i = 0; i < 8; ++i) { v[i] = seed + threadIdx.x * 0.001f + i; w[i] = float2_{v[i], v[i] + 1.f}; }
    const float a = seed * 0.5f, b = seed * 0.25f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[i]) : "v"(float2_{a, a}), "v"(float2_{b, b}));
            if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a));
            if (KIND == 5) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(v[i]), "v"(a) : "vcc");
            if (KIND == 6) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
            if (KIND == 7) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 1) & 7]));
            if (KIND == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            if (KIND == 9) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            if (KIND == 10) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[i]) : "s"(seed));
            if (KIND == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[i]) : "v"(float2_{a, a}));
            if (KIND == 12) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 13) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            if (KIND == 14) asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(v[i]) : "v"(a) : "s10", "s11");
            if (KIND == 15) asm volatile("v_cmp_lt_f32 s[10:11], %0, %1" :: "v"(v[i]), "v"(a) : "s10", "s11");
            if (KIND == 16) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i]) : "v"(float2_{a, a}));
            if (KIND == 17) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i]));
            if (KIND == 18) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            if (KIND == 19) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(a));
            if (KIND == 20) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a)); asm volatile("v_exp_f32 %0, %0" : "+v"(w[i].x)); }
            if (KIND == 21) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            if (KIND == 22) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + w[i].x + w[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
double run(const char* name, float* d, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = blocks*4 / 1024 ; instrs per wave = ITERS*8
    double instr_per_simd = (double)blocks * 4 / 1024.0 * ITERS * 8;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-22s blocks=%5d  %.3f ms  -> %.2f cycles/instr/SIMD (at 2.4 GHz nominal)\n", name, blocks, ms, cyc / instr_per_simd);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, 8192 * 256 * 4);
    for (int blocks : {2048, 4096}) {  // 1 and 8 waves per SIMD
        run<0>("v_fma_f32", d, blocks); run<12>("v_fmac_f32", d, blocks); run<1>("v_pk_fma_f32", d, blocks); run<2>("v_mul_f32", d, blocks);
        run<11>("v_pk_mul_f32", d, blocks); run<3>("v_exp_f32", d, blocks); run<8>("v_rcp_f32", d, blocks);
        run<4>("v_cndmask_b32 vcc", d, blocks); run<5>("v_cmp_lt_f32 vcc", d, blocks); run<9>("v_min_f32", d, blocks);
        run<10>("v_sub_f32 sgpr", d, blocks); run<6>("v_add_f32_dpp quad", d, blocks); run<7>("v_permlane32_swap", d, blocks);
        run<13>("v_add_f32", d, blocks); run<14>("v_cndmask sgprmask", d, blocks); run<15>("v_cmp -> sgpr", d, blocks);
        run<16>("v_pk_add_f32", d, blocks); run<17>("v_mul_f32 x,x,x", d, blocks); run<18>("v_max_f32", d, blocks);
        run<19>("v_mov_b32", d, blocks); run<20>("v_mul + v_exp pair", d, blocks); run<21>("v_and_b32", d, blocks); run<22>("v_add_u32", d, blocks);
        run<2>("v_mul_f32 (again)", d, blocks); run<0>("v_fma_f32 (again)", d, blocks);
    }
    return 0;
}
