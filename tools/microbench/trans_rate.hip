// issue-rate probe: cycles per instruction of f32 / f16 transcendentals on one wave per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(float* out, long long* cyc) {
    float x0 = threadIdx.x * 1e-3f + 0.5f, x1 = x0 + 0.1f, x2 = x0 + 0.2f, x3 = x0 + 0.3f;
    _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1, h2 = (_Float16)x2, h3 = (_Float16)x3;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 1024; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            if (MODE == 1) { asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            if (MODE == 2) { asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3)); }
            if (MODE == 3) { asm volatile("v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3)); }
            if (MODE == 4) { asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            if (MODE == 5) { asm volatile("v_cvt_f16_f32 %0, %0\n v_cvt_f16_f32 %1, %1\n v_cvt_f16_f32 %2, %2\n v_cvt_f16_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            if (MODE == 6) { asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %3, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %0, %1 op_sel_hi:[1,0,0]" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            if (MODE == 7) { asm volatile("v_pk_add_f16 %0, %0, %1\n v_pk_add_f16 %1, %1, %2\n v_pk_add_f16 %2, %2, %3\n v_pk_add_f16 %3, %3, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (float)h0 + (float)h1 + (float)h2 + (float)h3;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    float* d; long long* c; hipMalloc(&d, 1 << 20); hipMalloc(&c, 64); hipMemset(c, 0, 64);
    // one wave per SIMD on one CU: 256 threads, 1 block
    k<0><<<1, 256>>>(d, c); k<1><<<1, 256>>>(d, c); k<2><<<1, 256>>>(d, c); k<3><<<1, 256>>>(d, c);
    k<4><<<1, 256>>>(d, c); k<5><<<1, 256>>>(d, c); k<6><<<1, 256>>>(d, c); k<7><<<1, 256>>>(d, c);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
    const char* n[8] = {"v_exp_f32", "v_rcp_f32", "v_exp_f16", "v_rcp_f16", "v_fma_f32", "v_cvt_f16_f32", "v_fma_mix_f32", "v_pk_add_f16"};
    for (int i = 0; i < 8; ++i) printf("%-14s %.2f ticks per instruction (s_memtime ticks; fma = reference)\n", n[i], h[i] / (1024.0 * 32));
    return 0;
}
