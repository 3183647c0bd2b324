// What does the L2 -> CU path deliver when every CU streams the same weight image, as the 16-row tiles of P1 / P3 do?
// One 1024-thread workgroup per CU; every wave streams its 1/16 of a W-byte buffer with global_load_dwordx4 (1 KB per
// instruction, DEPTH instructions in flight), REPS passes; thread 0 times the workgroup with s_memtime.  No arithmetic
// beyond an xor that keeps the loads alive.  Prints bytes per shader clock and CU for several grid sizes and depths.
//   hipcc --offload-arch=gfx950 -O3 -o tools/libs/l2_stream_probe tools/l2_stream_probe.hip && tools/libs/l2_stream_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const uint4* __restrict__ w, int n16_per_wave, int reps, long long* cyc,
                                                      unsigned* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint4* p = w + (size_t)wave * n16_per_wave + lane;   // a wave's share is contiguous: 64 lanes x 16 B per instruction
    unsigned acc = 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < n16_per_wave; i += 64 * DEPTH) {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = p[i + d * 64];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
        }
    }
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int DEPTH>
int run(const uint4* w, size_t bytes, int blocks, int reps, long long* dcyc, unsigned* sink) {
    const int n16_per_wave = (int)(bytes / 16 / 16);   // 16 waves
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(blocks), dim3(1024), 0, 0, w, n16_per_wave, 2, dcyc, sink);   // warm L2
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(blocks), dim3(1024), 0, 0, w, n16_per_wave, reps, dcyc, sink);
    CK(hipDeviceSynchronize());
    std::vector<long long> c(blocks);
    CK(hipMemcpy(c.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    const double per = (double)bytes * reps;
    printf("  %7.2f MB image, %3d workgroups, %2d KB in flight per wave: median %5.1f B/clk/CU (slowest CU %5.1f, fastest %5.1f)\n",
           bytes / 1e6, blocks, DEPTH, per / c[blocks / 2], per / c[blocks - 1], per / c[0]);
    return 0;
}

int main() {
    const size_t maxb = 8u << 20;
    uint4* w; long long* dcyc; unsigned* sink;
    CK(hipMalloc(&w, maxb)); CK(hipMalloc(&dcyc, 4096 * sizeof(long long))); CK(hipMalloc(&sink, 64));
    CK(hipMemset(w, 1, maxb));
    printf("L2 -> CU stream, every workgroup reads the SAME image (s_memtime shader clocks; 64 B/clk/CU is the vector-memory path's width)\n");
    const size_t img[] = {2850816, 1351680, 16u * 1024 * 16};   // lit_mlp planes, update_gate planes, 256 KB (16 KB per wave)
    for (size_t b : img) {
        const size_t bytes = b / (16 * 16 * 64 * 8) * (16 * 16 * 64 * 8);   // whole 8-deep groups per wave
        for (int blocks : {1, 32, 194, 256}) {
            if (run<4>(w, bytes, blocks, 8, dcyc, sink)) return 1;
            if (run<8>(w, bytes, blocks, 8, dcyc, sink)) return 1;
        }
    }
    return 0;
}
