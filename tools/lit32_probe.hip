// Feasibility probe: lit_mlp (137 -> 512 -> 512 -> 256) on a 32-ROW tile with fp32 products from
// bf16 slices.  Two 100 KB hidden tiles do not fit LDS, so layer 1 is computed in two column halves and layer 2
// accumulates over the matching halves of its K range:
//   L0 (two passes of 16 column tiles) -> h1 planes | L1 cols 0-255 -> h2 | L2 k 0-255 | L1 cols 256-511 -> h2 | L2 k 256-511
// LDS: h1 [3][32][528] + max(X [3][32][176], h2 half [3][32][272]) = 153.6 KB.  Reports cycles per workgroup and the
// error against an fp64 result.   lit32_probe [blocks]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Idiffusionsat_amd/csrc tools/lit32_probe.hip -o tools/libs/lit32_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "split_bf16.h"
using namespace dsat;

constexpr int NW = 16, NTH = NW * 64, ROWS = 32, PD = 2;
constexpr int K0 = 137, KS0 = 5, N0 = 512, KS1 = 16, N1 = 512, KS2 = 16, N2 = 256;
constexpr int LDX = KS0 * 32 + 16, LD1 = 512 + 16, LD2 = 256 + 16;

struct Args { const float4 *w0, *w1, *w2; const float *b0, *b1, *b2; const float* x; float* out; long long* cyc; };

__global__ __launch_bounds__(NTH) void kern(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* h1 = lds;                       // [3][32][LD1]
    unsigned short* r2 = lds + 3 * ROWS * LD1;      // X planes, then the h2 half
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    for (int i = tid; i < ROWS * (KS0 * 32 / 4); i += NTH) {
        const int r = i / (KS0 * 8), c = 4 * (i % (KS0 * 8));
        float4 v;
        v.x = c + 0 < K0 ? a.x[r * K0 + c + 0] : 0.f; v.y = c + 1 < K0 ? a.x[r * K0 + c + 1] : 0.f;
        v.z = c + 2 < K0 ? a.x[r * K0 + c + 2] : 0.f; v.w = c + 3 < K0 ? a.x[r * K0 + c + 3] : 0.f;
        split_st4(r2, ROWS * LDX, r * LDX + c, v);
    }
    __syncthreads();
    const long long t0 = clock64();
    const unsigned voff = (unsigned)lane * 16u;
    const int col4 = 4 * (lane >> 4), row = lane & 15;
    auto wb = [&](const float4* w, int tile, int ks) { return w + ((size_t)tile * ks * 3) * 64; };
    auto leaky = [](sf32x4 acc, float4 b) {
        float4 v = make_float4(acc[0] + b.x, acc[1] + b.y, acc[2] + b.z, acc[3] + b.w);
        v.x = fmaxf(v.x, 0.2f * v.x); v.y = fmaxf(v.y, 0.2f * v.y); v.z = fmaxf(v.z, 0.2f * v.z); v.w = fmaxf(v.w, 0.2f * v.w);
        return v;
    };
    const float4 b0a = *reinterpret_cast<const float4*>(a.b0 + wv * 16 + col4), b0b = *reinterpret_cast<const float4*>(a.b0 + (wv + 16) * 16 + col4);
    const float4 b1a = *reinterpret_cast<const float4*>(a.b1 + wv * 16 + col4), b1b = *reinterpret_cast<const float4*>(a.b1 + (wv + 16) * 16 + col4);
    const float4 b2 = *reinterpret_cast<const float4*>(a.b2 + wv * 16 + col4);
    const unsigned short* xin = r2 + row * LDX + 8 * (lane >> 4);
    const unsigned short* h1in = h1 + row * LD1 + 8 * (lane >> 4);
    const unsigned short* h2in = r2 + row * LD2 + 8 * (lane >> 4);
    sf32x4 z = {0.f, 0.f, 0.f, 0.f};
    sf32x4 a0[2] = {z, z}, a0b[2] = {z, z}, a1[2] = {z, z}, a1b[2] = {z, z}, a2[2] = {z, z};
    // L0, two passes
    split_layer<2, PD>(voff, wb(a.w0, wv, KS0), KS0, xin, ROWS * LDX, LDX, [] {}, a0);
    auto ep0a = [&]() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) split_st4(h1, ROWS * LD1, (rb * 16 + row) * LD1 + wv * 16 + col4, leaky(a0[rb], b0a));
    };
    split_layer<2, PD, false>(voff, wb(a.w0, wv + 16, KS0), KS0, xin, ROWS * LDX, LDX, ep0a, a0b);
    auto ep0b = [&]() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) split_st4(h1, ROWS * LD1, (rb * 16 + row) * LD1 + (wv + 16) * 16 + col4, leaky(a0b[rb], b0b));
    };
    // L1 columns 0..255 (tile wv)
    split_layer<2, PD>(voff, wb(a.w1, wv, KS1), KS1, h1in, ROWS * LD1, LD1, ep0b, a1);
    auto ep1a = [&]() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) split_st4(r2, ROWS * LD2, (rb * 16 + row) * LD2 + wv * 16 + col4, leaky(a1[rb], b1a));
    };
    // L2, k 0..255
    split_layer<2, PD>(voff, wb(a.w2, wv, KS2), KS2 / 2, h2in, ROWS * LD2, LD2, ep1a, a2);
    // L1 columns 256..511 (tile wv + 16): reads h1 only
    split_layer<2, PD, false>(voff, wb(a.w1, wv + 16, KS1), KS1, h1in, ROWS * LD1, LD1, [] {}, a1b);
    auto ep1b = [&]() {
        lds_barrier();   // every wave is done with the first half of h2
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) split_st4(r2, ROWS * LD2, (rb * 16 + row) * LD2 + wv * 16 + col4, leaky(a1b[rb], b1b));
    };
    // L2, k 256..511
    split_layer<2, PD>(voff, wb(a.w2, wv, KS2) + (size_t)(KS2 / 2) * 192, KS2 / 2, h2in, ROWS * LD2, LD2, ep1b, a2);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
        *reinterpret_cast<float4*>(a.out + ((size_t)blockIdx.x * ROWS + rb * 16 + row) * N2 + wv * 16 + col4) =
            make_float4(a2[rb][0] + b2.x, a2[rb][1] + b2.y, a2[rb][2] + b2.z, a2[rb][3] + b2.w);
    const long long t1 = clock64();
    if (lane == 0) a.cyc[blockIdx.x * NW + wave] = t1 - t0;
}

static inline uint16_t bf16_rn(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static std::vector<uint16_t> pack(const std::vector<float>& W, int K, int N, int ks) {   // W[k][n]
    const int nt = N / 16;
    std::vector<uint16_t> ws((size_t)nt * ks * 3 * 64 * 8, 0);
    for (int t = 0; t < nt; ++t) for (int s = 0; s < ks; ++s) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const int k = 32 * s + 8 * (l >> 4) + j, n = 16 * t + (l & 15);
        const float w = k < K ? W[(size_t)k * N + n] : 0.f;
        const uint16_t h = bf16_rn(w); const float r1 = w - bf16_f(h); const uint16_t m = bf16_rn(r1); const uint16_t lo = bf16_rn(r1 - bf16_f(m));
        const size_t base = (((size_t)t * ks + s) * 3) * 64 * 8;
        ws[base + (0 * 64 + l) * 8 + j] = h; ws[base + (1 * 64 + l) * 8 + j] = m; ws[base + (2 * 64 + l) * 8 + j] = lo;
    }
    return ws;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 97;
    uint32_t s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    std::vector<float> X((size_t)ROWS * K0), W0((size_t)K0 * N0), W1((size_t)N0 * N1), W2((size_t)N1 * N2), B0(N0), B1(N1), B2(N2);
    for (auto& v : X) v = 2.f * rnd();
    for (auto& v : W0) v = 0.2f * rnd(); for (auto& v : W1) v = 0.15f * rnd(); for (auto& v : W2) v = 0.15f * rnd();
    for (auto& v : B0) v = 0.1f * rnd(); for (auto& v : B1) v = 0.1f * rnd(); for (auto& v : B2) v = 0.1f * rnd();
    auto mlp = [&](int r, std::vector<double>& o) {
        std::vector<double> h1(N0), h2(N1);
        for (int n = 0; n < N0; ++n) { double t = B0[n]; for (int k = 0; k < K0; ++k) t += (double)X[(size_t)r * K0 + k] * W0[(size_t)k * N0 + n]; h1[n] = std::max(t, 0.2 * t); }
        for (int n = 0; n < N1; ++n) { double t = B1[n]; for (int k = 0; k < N0; ++k) t += h1[k] * W1[(size_t)k * N1 + n]; h2[n] = std::max(t, 0.2 * t); }
        o.assign(N2, 0.0);
        for (int n = 0; n < N2; ++n) { double t = B2[n]; for (int k = 0; k < N1; ++k) t += h2[k] * W2[(size_t)k * N2 + n]; o[n] = t; }
    };
    auto p0 = pack(W0, K0, N0, KS0), p1 = pack(W1, N0, N1, KS1), p2 = pack(W2, N1, N2, KS2);
    Args a{};
    void *d0, *d1, *d2; float *db0, *db1, *db2, *dx, *dout; long long* dcyc;
    CK(hipMalloc(&d0, p0.size() * 2)); CK(hipMalloc(&d1, p1.size() * 2)); CK(hipMalloc(&d2, p2.size() * 2));
    CK(hipMalloc(&db0, N0 * 4)); CK(hipMalloc(&db1, N1 * 4)); CK(hipMalloc(&db2, N2 * 4)); CK(hipMalloc(&dx, X.size() * 4));
    CK(hipMalloc(&dout, (size_t)blocks * ROWS * N2 * 4)); CK(hipMalloc(&dcyc, (size_t)blocks * NW * 8));
    CK(hipMemcpy(d0, p0.data(), p0.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d1, p1.data(), p1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d2, p2.data(), p2.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db0, B0.data(), N0 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db1, B1.data(), N1 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db2, B2.data(), N2 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    a.w0 = (const float4*)d0; a.w1 = (const float4*)d1; a.w2 = (const float4*)d2; a.b0 = db0; a.b1 = db1; a.b2 = db2; a.x = dx; a.out = dout; a.cyc = dcyc;
    const size_t ldsb = (size_t)(3 * ROWS * LD1 + 3 * ROWS * std::max(LDX, LD2)) * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTH), ldsb, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<float> out((size_t)ROWS * N2);
    std::vector<long long> cyc((size_t)blocks * NW);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cyc.data(), dcyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    std::vector<long long> wg(blocks, 0);
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < NW; ++w) wg[b] = std::max(wg[b], cyc[(size_t)b * NW + w]);
    std::sort(wg.begin(), wg.end());
    double emax = 0, scale = 0;
    std::vector<double> o;
    for (int r = 0; r < ROWS; ++r) { mlp(r, o); for (int n = 0; n < N2; ++n) { emax = std::max(emax, std::fabs(out[(size_t)r * N2 + n] - o[n])); scale = std::max(scale, std::fabs(o[n])); } }
    printf("blocks %d: %lld cycles per 32-row tile (median; max %lld), kernel %.1f us; MFMA issue floor %d; LDS %zu B; weights %.2f MB per tile; max |err| %.2e (|out| max %.2f)\n",
           blocks, wg[blocks / 2], wg.back(), ms * 1e3, (2 * KS0 + 2 * KS1 + KS2) * 2 * 6 * 16 * 4, ldsb, (p0.size() + p1.size() + p2.size()) * 2 / 1e6, emax, scale);
    return 0;
}
