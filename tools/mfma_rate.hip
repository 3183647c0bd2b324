// Issue-rate microbenchmark for the fp32 MFMAs on gfx950 (one or two waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (float)(iters * 4 * NACC);
}
__global__ void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (float)(iters * 8);
}
template <typename K>
void run(const char* name, K kern, int threads, double flop_per_mfma, int mf_per_iter) {
    float* d; hipMalloc(&d, (1 << 20) * 4 + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float cyc; hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);
    double waves = blocks * threads / 64.0;
    double tf = waves * iters * mf_per_iter * flop_per_mfma / (ms * 1e-3) / 1e12;
    printf("%-28s threads/WG=%4d: %.1f clock64-ticks/MFMA/wave, %.1f TFLOP/s chip, %.3f ms\n", name, threads, cyc, tf, ms);
    hipFree(d);
}
int main() {
    run("16x16x4 f32, 8 acc", k16<8>, 256, 2048, 32);
    run("16x16x4 f32, 8 acc", k16<8>, 512, 2048, 32);
    run("16x16x4 f32, 2 acc", k16<2>, 256, 2048, 8);
    run("16x16x4 f32, 1 acc", k16<1>, 256, 2048, 4);
    run("32x32x2 f32, 2 acc", k32, 256, 4096, 8);
    run("32x32x2 f32, 2 acc", k32, 512, 4096, 8);
    return 0;
}
