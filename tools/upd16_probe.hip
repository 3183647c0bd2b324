// Probe: update_gate (528 -> 240 -> 240 -> 128) on a 16-row tile as split products (split_bf16.h), one column tile per
// wave (15 / 15 / 8 of 16 waves busy), with clock stamps between the layers: where do the chain's cycles go?
//   upd16_probe [blocks]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Idiffusionsat_amd/csrc tools/upd16_probe.hip -o tools/libs/upd16_probe
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

#ifndef PDU
#define PDU 3
#endif
constexpr int NW = 16, NTH = NW * 64, ROWS = 16;
constexpr int K0 = 528, KS0 = 17, N0 = 240, KS1 = 8, N1 = 240, KS2 = 8, N2 = 128;
constexpr int NT0 = 15, NT1 = 15, NT2 = 8;
constexpr int LDX = KS0 * 32 + 16, LD1 = 256 + 16;

struct Args { const float4 *w0, *w1, *w2; const float *b0, *b1, *b2; const float* x; float* out; long long* cyc; };

__device__ __forceinline__ float4 leaky(sf32x4 acc, float4 b) {
    float4 v = make_float4(acc[0] + b.x, acc[1] + b.y, acc[2] + b.z, acc[3] + b.w);
    v.x = fmaxf(v.x, 0.2f * v.x); v.y = fmaxf(v.y, 0.2f * v.y); v.z = fmaxf(v.z, 0.2f * v.z); v.w = fmaxf(v.w, 0.2f * v.w);
    return v;
}

__global__ __launch_bounds__(NTH) void kern(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* xs = lds;                        // [3][16][LDX]
    unsigned short* h1 = xs + 3 * ROWS * LDX;        // [3][16][LD1]
    unsigned short* h2 = h1 + 3 * ROWS * LD1;        // [3][16][LD1]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    for (int i = tid; i < ROWS * (KS0 * 32 / 4); i += NTH) {
        const int r = i / (KS0 * 8), c = 4 * (i % (KS0 * 8));
        float4 v;
        v.x = c + 0 < K0 ? a.x[r * K0 + c + 0] : 0.f; v.y = c + 1 < K0 ? a.x[r * K0 + c + 1] : 0.f;
        v.z = c + 2 < K0 ? a.x[r * K0 + c + 2] : 0.f; v.w = c + 3 < K0 ? a.x[r * K0 + c + 3] : 0.f;
        split_st4(xs, ROWS * LDX, r * LDX + c, v);
    }
    for (int i = tid; i < 3 * ROWS * LD1 / 2; i += NTH) { reinterpret_cast<unsigned*>(h1)[i] = 0u; reinterpret_cast<unsigned*>(h2)[i] = 0u; }
    __syncthreads();
    const long long t0 = clock64();
    const unsigned voff = (unsigned)lane * 16u;
    const int col4 = 4 * (lane >> 4), row = lane & 15;
    const bool on0 = wv < NT0, on1 = wv < NT1, on2 = wv < NT2;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b0 = on0 ? *reinterpret_cast<const float4*>(a.b0 + wv * 16 + col4) : z4;
    const float4 b1 = on1 ? *reinterpret_cast<const float4*>(a.b1 + wv * 16 + col4) : z4;
    const float4 b2 = on2 ? *reinterpret_cast<const float4*>(a.b2 + wv * 16 + col4) : z4;
    const unsigned short* xin = xs + row * LDX + 8 * (lane >> 4);
    const unsigned short* h1in = h1 + row * LD1 + 8 * (lane >> 4);
    const unsigned short* h2in = h2 + row * LD1 + 8 * (lane >> 4);
    const sf32x4 z = {0.f, 0.f, 0.f, 0.f};
    sf32x4 a0[1] = {z}, a1[1] = {z}, a2[1] = {z};
    auto wb = [&](const float4* w, int tile, int ks) { return w + ((size_t)tile * ks * 3) * 64; };
    long long t1 = 0, t2 = 0;
    if (on0) split_layer<1, PDU, false>(voff, wb(a.w0, wv, KS0), KS0, xin, ROWS * LDX, LDX, [] {}, a0);
    auto ep0 = [&]() { if (on0) split_st4(h1, ROWS * LD1, row * LD1 + wv * 16 + col4, leaky(a0[0], b0)); t1 = clock64(); };
    if (on1) split_layer<1, PDU>(voff, wb(a.w1, wv, KS1), KS1, h1in, ROWS * LD1, LD1, ep0, a1);
    else split_layer<0, PDU>(voff, nullptr, 0, nullptr, 0, 0, ep0, a1);
    auto ep1 = [&]() { if (on1) split_st4(h2, ROWS * LD1, row * LD1 + wv * 16 + col4, leaky(a1[0], b1)); t2 = clock64(); };
    if (on2) split_layer<1, PDU>(voff, wb(a.w2, wv, KS2), KS2, h2in, ROWS * LD1, LD1, ep1, a2);
    else split_layer<0, PDU>(voff, nullptr, 0, nullptr, 0, 0, ep1, a2);
    if (on2)
        *reinterpret_cast<float4*>(a.out + ((size_t)blockIdx.x * ROWS + row) * N2 + wv * 16 + col4) =
            make_float4(a2[0][0] + b2.x, a2[0][1] + b2.y, a2[0][2] + b2.z, a2[0][3] + b2.w);
    lds_barrier();
    const long long t3 = clock64();
    if (lane == 0) {
        long long* c = a.cyc + ((size_t)blockIdx.x * NW + wave) * 4;
        c[0] = t3 - t0; c[1] = t1 - t0; c[2] = t2 - t1; c[3] = t3 - t2;
    }
}
static inline uint16_t bf16_rn(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static std::vector<uint16_t> pack_planes(const std::vector<float>& W, int K, int N, int ks) {   // W[k][n]
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
    const int blocks = argc > 1 ? atoi(argv[1]) : 194;
    uint32_t s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    std::vector<float> X((size_t)ROWS * K0), W0((size_t)K0 * 256), W1((size_t)256 * 256), W2((size_t)256 * N2), B0(256), B1(256), B2(N2);
    for (auto& v : X) v = 2.f * rnd();
    // 240-wide layers padded to 256 columns / rows with zeros
    for (int k = 0; k < K0; ++k) for (int n = 0; n < 256; ++n) W0[(size_t)k * 256 + n] = n < N0 ? 0.2f * rnd() : 0.f;
    for (int k = 0; k < 256; ++k) for (int n = 0; n < 256; ++n) W1[(size_t)k * 256 + n] = (k < N0 && n < N1) ? 0.15f * rnd() : 0.f;
    for (int k = 0; k < 256; ++k) for (int n = 0; n < N2; ++n) W2[(size_t)k * N2 + n] = k < N1 ? 0.15f * rnd() : 0.f;
    for (int n = 0; n < 256; ++n) { B0[n] = n < N0 ? 0.1f * rnd() : 0.f; B1[n] = n < N1 ? 0.1f * rnd() : 0.f; }
    for (auto& v : B2) v = 0.1f * rnd();
    std::vector<double> ref((size_t)ROWS * N2);
    double scale = 0;
    for (int r = 0; r < ROWS; ++r) {
        std::vector<double> h1(256), h2(256);
        for (int n = 0; n < 256; ++n) { double t = B0[n]; for (int k = 0; k < K0; ++k) t += (double)X[(size_t)r * K0 + k] * W0[(size_t)k * 256 + n]; h1[n] = std::max(t, 0.2 * t); }
        for (int n = 0; n < 256; ++n) { double t = B1[n]; for (int k = 0; k < 256; ++k) t += h1[k] * W1[(size_t)k * 256 + n]; h2[n] = std::max(t, 0.2 * t); }
        for (int n = 0; n < N2; ++n) { double t = B2[n]; for (int k = 0; k < 256; ++k) t += h2[k] * W2[(size_t)k * N2 + n]; ref[(size_t)r * N2 + n] = t; scale = std::max(scale, std::fabs(t)); }
    }
    auto p0 = pack_planes(W0, K0, 256, KS0), p1 = pack_planes(W1, 256, 256, KS1), p2 = pack_planes(W2, 256, N2, KS2);
    Args a{};
    void *d0, *d1, *d2; float *db0, *db1, *db2, *dx, *dout; long long* dcyc;
    CK(hipMalloc(&d0, p0.size() * 2)); CK(hipMalloc(&d1, p1.size() * 2)); CK(hipMalloc(&d2, p2.size() * 2));
    CK(hipMemcpy(d0, p0.data(), p0.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d1, p1.data(), p1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d2, p2.data(), p2.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&db0, 256 * 4)); CK(hipMalloc(&db1, 256 * 4)); CK(hipMalloc(&db2, N2 * 4)); CK(hipMalloc(&dx, X.size() * 4));
    CK(hipMalloc(&dout, (size_t)blocks * ROWS * N2 * 4)); CK(hipMalloc(&dcyc, (size_t)blocks * NW * 4 * 8));
    CK(hipMemcpy(db0, B0.data(), 256 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db1, B1.data(), 256 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db2, B2.data(), N2 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    a.w0 = (const float4*)d0; a.w1 = (const float4*)d1; a.w2 = (const float4*)d2; a.b0 = db0; a.b1 = db1; a.b2 = db2; a.x = dx; a.out = dout; a.cyc = dcyc;
    const size_t ldsb = (size_t)(3 * ROWS * LDX + 2 * 3 * ROWS * LD1) * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    for (int rep = 0; rep < 4; ++rep) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTH), ldsb, 0, a); CK(hipDeviceSynchronize()); }
    std::vector<float> out((size_t)ROWS * N2);
    std::vector<long long> cyc((size_t)blocks * NW * 4);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cyc.data(), dcyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double emax = 0;
    for (size_t i = 0; i < out.size(); ++i) emax = std::max(emax, std::fabs(out[i] - ref[i]));
    const char* nm[4] = {"chain", "layer 0 (17 k-steps, 15 tiles)", "layer 1 (epilogue, barrier, 8 k-steps, 15 tiles)", "layer 2 (epilogue, barrier, 8 k-steps, 8 tiles)"};
    for (int q = 0; q < 4; ++q) {
        std::vector<long long> v(blocks, 0);
        for (int b = 0; b < blocks; ++b) v[b] = cyc[((size_t)b * NW + 0) * 4 + q];   // wave 0 has a tile in every layer
        std::sort(v.begin(), v.end());
        printf("%-52s median %6lld  max %6lld cycles (wave 0)\n", nm[q], v[blocks / 2], v.back());
    }
    printf("blocks %d, PD %d, weights %.2f MB per tile, max |err| %.2e (|out| max %.2f)\n", blocks, PDU, (p0.size() + p1.size() + p2.size()) * 2 / 1e6, emax, scale);
    return 0;
}
