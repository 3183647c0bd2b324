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

// The weight ring of the split chains, alone: per wave a ring of PD slots of NL 1-KB loads (inline asm, SGPR base + VGPR
// offset); a step waits for its slot (s_waitcnt vmcnt(NL * (PD - 1))), touches nothing, asks for the slot again.
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gl(f4& d, unsigned voff, const uint4* sb) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
}
template <int PD, int NL>
__global__ __launch_bounds__(1024) void ring_kernel(const uint4* __restrict__ w, int n16_per_wave, int reps, long long* cyc,
                                                    unsigned* sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint4* base = w + (size_t)wave * n16_per_wave;
    const unsigned voff = lane * 16u;
    const int steps = n16_per_wave / (64 * NL);
    f4 ring[PD][NL];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int p = 0; p < PD; ++p)
#pragma unroll
            for (int j = 0; j < NL; ++j) gl(ring[p][j], voff, base + (size_t)(p * NL + j) * 64);
        for (int s = 0; s < steps; s += PD) {
#pragma unroll
            for (int p = 0; p < PD; ++p) {
                if (NL == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(ring[p][0]), "+v"(ring[p][1]), "+v"(ring[p][2]) : "n"(NL * (PD - 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ring[p][0]), "+v"(ring[p][1]), "+v"(ring[p][2]), "+v"(ring[p][NL - 1]) : "n"(NL * (PD - 1)) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                const int sn = min(s + p + PD, steps - 1);
#pragma unroll
                for (int j = 0; j < NL; ++j) gl(ring[p][j], voff, base + (size_t)(sn * NL + j) * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < PD; ++p)
#pragma unroll
            for (int j = 0; j < NL; ++j) asm volatile("" : "+v"(ring[p][j]));
    }
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (ring[0][0][0] == 1.2345f) sink[0] = 1;
}
template <int PD, int NL>
int run_ring(const uint4* w, size_t bytes, int blocks, int reps, long long* dcyc, unsigned* sink) {
    const int n16_per_wave = (int)(bytes / 16 / 16) / (64 * NL * PD) * (64 * NL * PD);
    hipLaunchKernelGGL((ring_kernel<PD, NL>), dim3(blocks), dim3(1024), 0, 0, w, n16_per_wave, 2, dcyc, sink);
    hipLaunchKernelGGL((ring_kernel<PD, NL>), dim3(blocks), dim3(1024), 0, 0, w, n16_per_wave, reps, dcyc, sink);
    CK(hipDeviceSynchronize());
    std::vector<long long> c(blocks);
    CK(hipMemcpy(c.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    const double per = (double)n16_per_wave * 16 * 16 * reps;
    printf("  ring: %d slots of %d KB, %3d workgroups: median %5.1f B/clk/CU (slowest CU %5.1f)\n", PD, NL, blocks,
           per / c[blocks / 2], per / c[blocks - 1]);
    return 0;
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
    printf("the split chains' ring alone (2.85 MB image)\n");
    for (int blocks : {1, 194}) {
        if (run_ring<2, 3>(w, 2850816, blocks, 8, dcyc, sink)) return 1;
        if (run_ring<3, 3>(w, 2850816, blocks, 8, dcyc, sink)) return 1;
        if (run_ring<4, 3>(w, 2850816, blocks, 8, dcyc, sink)) return 1;
        if (run_ring<2, 4>(w, 2850816, blocks, 8, dcyc, sink)) return 1;
        if (run_ring<2, 6>(w, 2850816, blocks, 8, dcyc, sink)) return 1;
        if (run_ring<1, 6>(w, 2850816, blocks, 8, dcyc, sink)) return 1;
    }
    return 0;
}
