// Feasibility probe for DESIGN.md §7b: one weight-streaming layer  out[16*RB][N] = X[16*RB][K] @ W[K][N]
// with every fp32 product formed from bf16 slices on v_mfma_f32_16x16x32_bf16 (x = h + m + l, three 8-bit
// slices per operand; the six products of weight >= 2^-16 relative to h*h, fp32 accumulation) against the same
// layer on v_mfma_f32_16x16x4_f32.  Reports cycles per workgroup and the error of both against an fp64 result.
//   split_bf16_probe [blocks] [K] [N]        (16 waves, RB = 4: the shape of P2's layers)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/split_bf16_probe.hip -o tools/libs/split_bf16_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int NW = 16, RB = 4, BM = 16 * RB;

static inline uint16_t bf16_rn(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static inline float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline void split3(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
    h = bf16_rn(x); const float r = x - bf16_f(h);
    m = bf16_rn(r); const float r2 = r - bf16_f(m);
    l = bf16_rn(r2);
}

__device__ __forceinline__ unsigned short d_bf16(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float d_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// ---- split path: weights [tile][kstep][plane][lane] uint4 (8 bf16 = W[32s + 8(l>>4) + j][16t + (l&15)])
template <int TERMS>
__global__ __launch_bounds__(NW * 64) void k_split(const uint4* __restrict__ wf, const float* __restrict__ X, int K, int ksteps,
                                                   int ntiles, float* __restrict__ out, long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned short sx[];   // [3][BM][ldk]
    const int ldk = ksteps * 32 + 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < BM * ksteps * 32; i += NW * 64) {
        const int r = i / (ksteps * 32), k = i % (ksteps * 32);
        const float x = k < K ? X[r * K + k] : 0.f;
        const unsigned short h = d_bf16(x); const float r1 = x - d_f(h);
        const unsigned short m = d_bf16(r1); const float r2 = r1 - d_f(m);
        const unsigned short l = d_bf16(r2);
        sx[(0 * BM + r) * ldk + k] = h; sx[(1 * BM + r) * ldk + k] = m; sx[(2 * BM + r) * ldk + k] = l;
    }
    __syncthreads();
    const long long t0 = clock64();
    const int t = wave;   // one column tile per wave
    f32x4 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < ntiles) {
        const uint4* wp = wf + ((size_t)t * ksteps * 3) * 64 + lane;
        uint4 w0 = wp[0], w1 = wp[64], w2 = wp[128];
        const unsigned short* px = sx + (lane & 15) * ldk + 8 * (lane >> 4);
        for (int s = 0; s < ksteps; ++s) {
            const int sn = s + 1 < ksteps ? s + 1 : s;
            const uint4 n0 = wp[(size_t)sn * 192], n1 = wp[(size_t)sn * 192 + 64], n2 = wp[(size_t)sn * 192 + 128];
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&w0), wm = *reinterpret_cast<const bf16x8*>(&w1),
                         wl = *reinterpret_cast<const bf16x8*>(&w2);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(px + (0 * BM + rb * 16) * ldk + s * 32);
                const bf16x8 xm = *reinterpret_cast<const bf16x8*>(px + (1 * BM + rb * 16) * ldk + s * 32);
                const bf16x8 xl = *reinterpret_cast<const bf16x8*>(px + (2 * BM + rb * 16) * ldk + s * 32);
                if (TERMS >= 6) {
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, acc[rb], 0, 0, 0);
                }
                if (TERMS >= 3) {
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, acc[rb], 0, 0, 0);
                }
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[rb], 0, 0, 0);
            }
            w0 = n0; w1 = n1; w2 = n2;
        }
        const int ld = ntiles * 16;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            *reinterpret_cast<float4*>(out + ((size_t)blockIdx.x * BM + rb * 16 + (lane & 15)) * ld + t * 16 + 4 * (lane >> 4)) =
                make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
}

// ---- fp32 path: weights [tile][kblock(16)][lane] float4 = W[16kb + 4(l>>4) + j][16t + (l&15)]
__global__ __launch_bounds__(NW * 64) void k_f32(const float4* __restrict__ wf, const float* __restrict__ X, int K, int kblocks,
                                                 int ntiles, float* __restrict__ out, long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(16))) float sxf[];   // [BM][ld]
    const int ldk = kblocks * 16 + 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < BM * kblocks * 16; i += NW * 64) {
        const int r = i / (kblocks * 16), k = i % (kblocks * 16);
        sxf[r * ldk + k] = k < K ? X[r * K + k] : 0.f;
    }
    __syncthreads();
    const long long t0 = clock64();
    const int t = wave;
    f32x4 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < ntiles) {
        const float4* wp = wf + ((size_t)t * kblocks) * 64 + lane;
        float4 w = wp[0];
        const float* px = sxf + (lane & 15) * ldk + 4 * (lane >> 4);
        for (int kb = 0; kb < kblocks; ++kb) {
            const float4 wn = wp[(size_t)(kb + 1 < kblocks ? kb + 1 : kb) * 64];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float4 a = *reinterpret_cast<const float4*>(px + rb * 16 * ldk + kb * 16);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a.x, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a.y, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a.z, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a.w, acc[rb], 0, 0, 0);
            }
            w = wn;
        }
        const int ld = ntiles * 16;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            *reinterpret_cast<float4*>(out + ((size_t)blockIdx.x * BM + rb * 16 + (lane & 15)) * ld + t * 16 + 4 * (lane >> 4)) =
                make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 208;
    const int K = argc > 2 ? atoi(argv[2]) : 204;
    const int N = argc > 3 ? atoi(argv[3]) : 256;
    const int ntiles = (N + 15) / 16, ksteps = (K + 31) / 32, kblocks = (K + 15) / 16;
    if (ntiles > NW) { printf("N <= %d\n", NW * 16); return 1; }
    std::vector<float> X((size_t)BM * K), W((size_t)K * N);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    for (auto& v : X) v = 4.f * rnd();                    // activations O(1)
    for (auto& v : W) v = 0.2f * rnd();                   // Glorot-sized weights
    std::vector<double> ref((size_t)BM * N, 0.0), refabs((size_t)BM * N, 0.0);
    for (int r = 0; r < BM; ++r)
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < N; ++n) {
                ref[(size_t)r * N + n] += (double)X[(size_t)r * K + k] * W[(size_t)k * N + n];
                refabs[(size_t)r * N + n] += std::fabs((double)X[(size_t)r * K + k] * W[(size_t)k * N + n]);
            }
    // fragment-ordered weights
    std::vector<uint16_t> ws((size_t)ntiles * ksteps * 3 * 64 * 8, 0);
    for (int t = 0; t < ntiles; ++t)
        for (int st = 0; st < ksteps; ++st)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * st + 8 * (l >> 4) + j, n = 16 * t + (l & 15);
                    const float w = (k < K && n < N) ? W[(size_t)k * N + n] : 0.f;
                    uint16_t h, m, lo; split3(w, h, m, lo);
                    const size_t base = (((size_t)t * ksteps + st) * 3) * 64 * 8;
                    ws[base + (0 * 64 + l) * 8 + j] = h; ws[base + (1 * 64 + l) * 8 + j] = m; ws[base + (2 * 64 + l) * 8 + j] = lo;
                }
    std::vector<float> wf((size_t)ntiles * kblocks * 64 * 4, 0.f);
    for (int t = 0; t < ntiles; ++t)
        for (int kb = 0; kb < kblocks; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 4; ++j) {
                    const int k = 16 * kb + 4 * (l >> 4) + j, n = 16 * t + (l & 15);
                    wf[(((size_t)t * kblocks + kb) * 64 + l) * 4 + j] = (k < K && n < N) ? W[(size_t)k * N + n] : 0.f;
                }
    void *dws, *dwf; float *dX, *dout; long long* dcyc;
    CK(hipMalloc(&dws, ws.size() * 2)); CK(hipMalloc(&dwf, wf.size() * 4)); CK(hipMalloc(&dX, X.size() * 4));
    CK(hipMalloc(&dout, (size_t)blocks * BM * ntiles * 16 * 4)); CK(hipMalloc(&dcyc, (size_t)blocks * NW * 8));
    CK(hipMemcpy(dws, ws.data(), ws.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwf, wf.data(), wf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    const size_t lds_split = (size_t)3 * BM * (ksteps * 32 + 8) * 2, lds_f32 = (size_t)BM * (kblocks * 16 + 8) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_split<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_split<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_f32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f32));
    std::vector<float> out((size_t)BM * ntiles * 16);
    std::vector<long long> cyc((size_t)blocks * NW);
    auto report = [&](const char* name, double mfma_floor) {
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(cyc.data(), dcyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        std::vector<long long> wg(blocks, 0);
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < NW; ++w) wg[b] = std::max(wg[b], cyc[(size_t)b * NW + w]);
        std::sort(wg.begin(), wg.end());
        double emax = 0, erel = 0;
        for (int r = 0; r < BM; ++r)
            for (int n = 0; n < N; ++n) {
                const double e = std::fabs((double)out[(size_t)r * ntiles * 16 + n] - ref[(size_t)r * N + n]);
                emax = std::max(emax, e);
                erel = std::max(erel, e / refabs[(size_t)r * N + n]);   // relative to sum |x w|: the natural fp32 error scale
            }
        printf("%-22s k-loop %6lld cycles/WG median (max %6lld), MFMA issue floor %6.0f;  max |err| %.3e, max err / sum|xw| %.3e\n",
               name, wg[blocks / 2], wg.back(), mfma_floor, emax, erel);
        return 0;
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(NW * 64), lds_f32, 0, (const float4*)dwf, dX, K, kblocks, ntiles, dout, dcyc);
        CK(hipDeviceSynchronize());
    }
    report("fp32 16x16x4", kblocks * 4.0 * RB * 32 * (NW / 4));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_split<6>, dim3(blocks), dim3(NW * 64), lds_split, 0, (const uint4*)dws, dX, K, ksteps, ntiles, dout, dcyc);
        CK(hipDeviceSynchronize());
    }
    report("bf16 slices, 6 products", ksteps * 6.0 * RB * 16 * (NW / 4));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_split<3>, dim3(blocks), dim3(NW * 64), lds_split, 0, (const uint4*)dws, dX, K, ksteps, ntiles, dout, dcyc);
        CK(hipDeviceSynchronize());
    }
    report("bf16 slices, 3 products", ksteps * 3.0 * RB * 16 * (NW / 4));
    printf("LDS: split %zu B, fp32 %zu B per %d-row tile; weight bytes per tile: split %zu, fp32 %zu\n", lds_split, lds_f32, BM,
           ws.size() * 2, wf.size() * 4);
    return 0;
}
