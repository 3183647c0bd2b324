// Microbenchmark: per-kernel cost of a dependent chain of tiny kernels, eager vs hipGraph replay.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
// the round kernels' shape: 1024 threads and all of a CU's LDS per workgroup, one workgroup per CU
__global__ __launch_bounds__(1024) void fat(float* p, int n) {
    __shared__ float big[40000];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    big[threadIdx.x] = (float)i;
    __syncthreads();
    if (i < n) p[i] += big[(threadIdx.x + 1) & 1023];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    float* d; CK(hipMalloc(&d, 1 << 24));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int K = 2000;
    for (int blocks : {1, 256, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < K; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st, d, blocks * 256);
            auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(st));
            auto t2 = std::chrono::steady_clock::now();
            if (rep) printf("eager  blocks=%4d: host enqueue %.2f us/kernel, total %.2f us/kernel\n", blocks,
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / K,
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / K);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < K; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st, d, blocks * 256);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            auto t2 = std::chrono::steady_clock::now();
            if (rep) printf("graph  blocks=%4d: total %.2f us/kernel\n", blocks,
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / K);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    {   // dependent launches of the fat shape in a graph: the floor of a round-kernel boundary
        const int K2 = 600;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < K2; ++i) hipLaunchKernelGGL(fat, dim3(256), dim3(1024), 0, st, d, 256 * 1024);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            auto t2 = std::chrono::steady_clock::now();
            if (rep) printf("graph  256 workgroups x 1024 threads x 160 KB LDS: total %.2f us/kernel\n",
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / K2);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
