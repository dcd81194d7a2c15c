// micro-benchmark: v_mfma_f32_32x32x16_bf16 issue rate, one or two waves per SIMD, NACC accumulator tiles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* ticks) {
    f32x16_t acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8_t a[4], b[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) { a[j][i] = (__bf16)(float)(threadIdx.x + i + j); b[j][i] = (__bf16)(float)(threadIdx.x * 3 + i - j); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[(j >> 2) & 3], acc[j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int NACC>
void run(int threads, int blocks, const char* name) {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 4 * 512 * 4096); hipMalloc(&ticks, 8);
    const int iters = 4096 * 16 / NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(out, 16, ticks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(out, iters, ticks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * NACC;            // MFMAs per wave
    const double waves_per_simd = threads / 256.0 * (blocks / 256.0);
    const double flops = mf * 32768.0 * (threads / 64) * blocks;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s  ns/MFMA/SIMD=%.2f  ticks/MFMA(wave)=%.1f  tick_ns=%.3f\n", name, ms, flops / ms * 1e-9,
           ms * 1e6 / (mf * waves_per_simd), (double)t / mf, ms * 1e6 / (double)t);
}
int main() {
    run<16>(256, 256, "1 wave/SIMD, 16 acc");
    run<16>(256, 256, "1 wave/SIMD, 16 acc (again)");
    run<4>(256, 256, "1 wave/SIMD, 4 acc");
    run<1>(256, 256, "1 wave/SIMD, 1 acc");
    run<8>(256, 256, "1 wave/SIMD, 8 acc");
    run<4>(256, 512, "2 wg x 1 wave/SIMD, 4 acc");
    run<4>(256, 128, "half the CUs, 4 acc");
    return 0;
}
