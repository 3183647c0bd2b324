// In-kernel timing of fused_common.h's stream_gemm: ticks per k-block for one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../diffusionsat_amd/csrc/fused_common.h"
using namespace dsat;
template <int RB, int TPW>
__global__ __launch_bounds__(256) void kern(const float4* wp, int kblocks, int ntiles, float* out, long long* ticks) {
    __shared__ __attribute__((aligned(16))) float sIn[16 * RB * 520];
    for (int i = threadIdx.x; i < 16 * RB * 520; i += 256) sIn[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc[RB][TPW];
    for (int r = 0; r < RB; ++r) for (int j = 0; j < TPW; ++j) acc[r][j] = f32x4{0, 0, 0, 0};
    long long t0 = clock64();
    stream_gemm<RB, TPW>(sIn, 520, kblocks, wp, wave * TPW, ntiles, lane, acc);
    long long t1 = clock64();
    float s = 0;
    for (int r = 0; r < RB; ++r) for (int j = 0; j < TPW; ++j) s += acc[r][j][0] + acc[r][j][1] + acc[r][j][2] + acc[r][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int RB, int TPW>
void run(const float4* wp, int kblocks, int ntiles, int blocks) {
    float* out; long long* ticks;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&ticks, blocks * 4 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kern<RB, TPW>), dim3(blocks), dim3(256), 0, 0, wp, kblocks, ntiles, out, ticks);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), ticks, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double avg = 0; long long mx = 0;
    for (auto v : h) { avg += v; mx = v > mx ? v : mx; }
    avg /= h.size();
    printf("RB=%d TPW=%d blocks=%3d kblocks=%d: avg %.0f ticks/kblock (max wave %.0f), ideal %d\n", RB, TPW, blocks, kblocks,
           avg / kblocks, (double)mx / kblocks, RB * TPW * 4 * 32);
    (void)hipFree(out); (void)hipFree(ticks);
}
int main() {
    const int kblocks = 32, ntiles = 32;
    std::vector<float> w((size_t)ntiles * kblocks * 256);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (float)((i * 2654435761u) >> 20) * 1e-4f - 0.2f;
    float4* wp; (void)hipMalloc(&wp, w.size() * 4);
    (void)hipMemcpy(wp, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {1, 194, 256}) {
        run<1, 8>(wp, kblocks, ntiles, blocks);
        run<1, 4>(wp, kblocks, ntiles, blocks);
        run<2, 4>(wp, kblocks, ntiles, blocks);
        run<1, 2>(wp, kblocks, ntiles, blocks);
    }
    return 0;
}
