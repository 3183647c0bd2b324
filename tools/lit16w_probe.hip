// Feasibility probe: lit_mlp (137 -> 512 -> 512 -> 256) on a 16-ROW tile with fp32 products from bf16 slices where
// the WEIGHTS travel as fp32 (4 B instead of the 6 B of three bf16 planes) and are sliced in registers right before
// the MFMAs (v_cvt_pk_bf16_f32, unpack, subtract: VALU work that runs beside the matrix pipe).  A 16-row tile is
// bound by the weight stream from L2, so fewer bytes per weight may buy more than the slicing
// costs.  MODE 0: planes pre-sliced (6 B, split_bf16.h as shipped, one column tile per wave and pass); MODE 1:
// fp32 weights sliced in registers, two column tiles per wave in the 512-wide layers; MODE 2: pre-sliced planes,
// two column tiles per wave sharing the activation fragments.
// 1x MI355X, 194 workgroups: 57.4 k / 66.1 k / 56.0 k cycles per tile (v_mfma_f32_16x16x4_f32 chain in the round
// kernel: 68.7 k) -- mode 0 became role_a1s of round_p1_kernel; mode 2 made no difference inside the kernel.
//   lit16w_probe [blocks]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Idiffusionsat_amd/csrc tools/lit16w_probe.hip -o tools/libs/lit16w_probe
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

#ifndef PDW
#define PDW 3
#endif
#ifndef PDP
#define PDP 2
#endif
constexpr int NW = 16, NTH = NW * 64, ROWS = 16;
constexpr int K0 = 137, KS0 = 5, N0 = 512, KS1 = 16, N1 = 512, KS2 = 16, N2 = 256;
constexpr int LDX = KS0 * 32 + 16, LD1 = 512 + 16;

struct Args { const float4 *w0, *w1, *w2; const float *b0, *b1, *b2; const float* x; float* out; long long* cyc; };

// ---- fp32 weight slots: per tile and k-step two 16-byte fragments (k = 8(l>>4) + 0..3 | + 4..7 of column l&15)
struct WSlot { sf32x4 a, b; };

__device__ __forceinline__ void slice8(const WSlot& w, bf16x8& wh, bf16x8& wm, bf16x8& wl) {
    unsigned h[4], m[4], l[4];
    split_pair(w.a[0], w.a[1], h[0], m[0], l[0]);
    split_pair(w.a[2], w.a[3], h[1], m[1], l[1]);
    split_pair(w.b[0], w.b[1], h[2], m[2], l[2]);
    split_pair(w.b[2], w.b[3], h[3], m[3], l[3]);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    wh = __builtin_bit_cast(bf16x8, (u4){h[0], h[1], h[2], h[3]});
    wm = __builtin_bit_cast(bf16x8, (u4){m[0], m[1], m[2], m[3]});
    wl = __builtin_bit_cast(bf16x8, (u4){l[0], l[1], l[2], l[3]});
}

template <int NT> struct Tie;
template <> struct Tie<1> {
    template <int N> static __device__ __forceinline__ void wait(WSlot (&s)[1]) {
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(s[0].a), "+v"(s[0].b) : "n"(N) : "memory");
    }
};
template <> struct Tie<2> {
    template <int N> static __device__ __forceinline__ void wait(WSlot (&s)[2]) {
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s[0].a), "+v"(s[0].b), "+v"(s[1].a), "+v"(s[1].b) : "n"(N) : "memory");
    }
};

// One layer, NT column tiles per wave (tile bases wb[j]; 128 float4 per k-step and tile), 16 rows.
template <int NT, int PD, bool BARRIER = true, typename Prep>
__device__ __forceinline__ void wf_layer(unsigned voff, const float4* const (&wb)[NT], int stot, const unsigned short* px,
                                         int plane, Prep&& prep, sf32x4 (&acc)[NT]) {
    WSlot ring[PD][NT];
    const int last = stot - 1;
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float4* sb = wb[j] + (size_t)(p < last ? p : last) * 128;
            gload_b128(ring[p][j].a, voff, sb);
            gload_b128(ring[p][j].b, voff, sb + 64);
        }
    prep();
    if (BARRIER) lds_barrier();
    const int main = stot / PD * PD;
    for (int s = 0; s < main; s += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const unsigned short* q = px + (s + p) * 32;
            split_rotate_prio(s + p);
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q), xm = *reinterpret_cast<const bf16x8*>(q + plane),
                         xl = *reinterpret_cast<const bf16x8*>(q + 2 * plane);
            Tie<NT>::template wait<2 * NT * (PD - 1)>(ring[p]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x8 wh, wm, wl;
                slice8(ring[p][j], wh, wm, wl);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int sn = s + p + PD;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4* sb = wb[j] + (size_t)(sn < last ? sn : last) * 128;
                gload_b128(ring[p][j].a, voff, sb);
                gload_b128(ring[p][j].b, voff, sb + 64);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    Tie<NT>::template wait<0>(ring[0]);
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int p = 1; p < PD; ++p)
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(ring[p][j].a), "+v"(ring[p][j].b));
#pragma unroll
    for (int p = 0; p + 1 < PD; ++p) {
        if (main + p < stot) {
            const unsigned short* q = px + (main + p) * 32;
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q), xm = *reinterpret_cast<const bf16x8*>(q + plane),
                         xl = *reinterpret_cast<const bf16x8*>(q + 2 * plane);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x8 wh, wm, wl;
                slice8(ring[p][j], wh, wm, wl);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[j], 0, 0, 0);
            }
        }
    }
}


// ---- pre-sliced planes, NT column tiles per wave sharing the activation fragments (LDS reads per MFMA halve)
template <int NT> struct TieP;
template <> struct TieP<1> {
    template <int N> static __device__ __forceinline__ void wait(SplitSlot (&s)[1]) {
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(s[0].h), "+v"(s[0].m), "+v"(s[0].l) : "n"(N) : "memory");
    }
};
template <> struct TieP<2> {
    template <int N> static __device__ __forceinline__ void wait(SplitSlot (&s)[2]) {
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"(s[0].h), "+v"(s[0].m), "+v"(s[0].l), "+v"(s[1].h), "+v"(s[1].m), "+v"(s[1].l) : "n"(N) : "memory");
    }
};
template <int NT>
__device__ __forceinline__ void pl_mfma(const SplitSlot (&w)[NT], bf16x8 xh, bf16x8 xm, bf16x8 xl, sf32x4 (&acc)[NT]) {
    // interleaved over the tiles: consecutive MFMAs never share an accumulator
#define PLM(WP, XP) _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[j].WP), XP, acc[j], 0, 0, 0);
    PLM(l, xh) PLM(h, xl) PLM(m, xm) PLM(h, xm) PLM(m, xh) PLM(h, xh)
#undef PLM
}
template <int NT, int PD, bool BARRIER = true, typename Prep>
__device__ __forceinline__ void pl_layer(unsigned voff, const float4* const (&wb)[NT], int stot, const unsigned short* px,
                                         int plane, Prep&& prep, sf32x4 (&acc)[NT]) {
    SplitSlot ring[PD][NT];
    const int last = stot - 1;
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float4* sb = wb[j] + (size_t)(p < last ? p : last) * 192;
            gload_b128(ring[p][j].h, voff, sb);
            gload_b128(ring[p][j].m, voff, sb + 64);
            gload_b128(ring[p][j].l, voff, sb + 128);
        }
    prep();
    if (BARRIER) lds_barrier();
    const int main = stot / PD * PD;
    for (int s = 0; s < main; s += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const unsigned short* q = px + (s + p) * 32;
            split_rotate_prio(s + p);
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q), xm = *reinterpret_cast<const bf16x8*>(q + plane),
                         xl = *reinterpret_cast<const bf16x8*>(q + 2 * plane);
#ifdef NO_REDUNDANT
            if (s + p + PD > last + PD - 1 + 0 && false) {}
            if (s + p > last - PD) TieP<NT>::template wait<0>(ring[p]);   // the layer's last PD steps: nothing younger is
            else                                                           // guaranteed in flight (over-waits: probe only)
#endif
            TieP<NT>::template wait<3 * NT * (PD - 1)>(ring[p]);
            pl_mfma<NT>(ring[p], xh, xm, xl, acc);
            __builtin_amdgcn_sched_barrier(0);
            const int sn = s + p + PD;
#ifdef NO_REDUNDANT
            if (sn <= last)
#endif
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4* sb = wb[j] + (size_t)(sn < last ? sn : last) * 192;
                gload_b128(ring[p][j].h, voff, sb);
                gload_b128(ring[p][j].m, voff, sb + 64);
                gload_b128(ring[p][j].l, voff, sb + 128);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    TieP<NT>::template wait<0>(ring[0]);
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int p = 1; p < PD; ++p)
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(ring[p][j].h), "+v"(ring[p][j].m), "+v"(ring[p][j].l));
#pragma unroll
    for (int p = 0; p + 1 < PD; ++p) {
        if (main + p < stot) {
            const unsigned short* q = px + (main + p) * 32;
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q), xm = *reinterpret_cast<const bf16x8*>(q + plane),
                         xl = *reinterpret_cast<const bf16x8*>(q + 2 * plane);
            pl_mfma<NT>(ring[p], xh, xm, xl, acc);
        }
    }
}

__device__ __forceinline__ float4 leaky(sf32x4 acc, float4 b) {
    float4 v = make_float4(acc[0] + b.x, acc[1] + b.y, acc[2] + b.z, acc[3] + b.w);
    v.x = fmaxf(v.x, 0.2f * v.x); v.y = fmaxf(v.y, 0.2f * v.y); v.z = fmaxf(v.z, 0.2f * v.z); v.w = fmaxf(v.w, 0.2f * v.w);
    return v;
}


// MODE 3: ONE ring for the whole chain.  The wave's fragments form one stream: pass i (a layer's column tile) covers
// virtual steps [B, B + ks); slot = V % PD; after step V is consumed the slot asks for step V + PD, which may already
// belong to the next pass (weights do not depend on activations) -- no redundant refills at the end of a pass, no
// empty pipe at the start of the next.  `wnext`: the next pass's fragments (the chain's last pass: its own last PD steps).
#ifndef PDV
#define PDV 3
#endif
template <int PD>
__device__ __forceinline__ void vpass(SplitSlot (&ring)[PD], unsigned voff, const float4* wb, int B, int ks, const float4* wnext,
                                      const unsigned short* px, int plane, int ld, sf32x4 (&acc)[1]) {
    const int E = B + ks;
    for (int c = B / PD * PD; c < E; c += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const int V = c + p;
            if (V >= B && V < E) {
                const int k = V - B;
                const unsigned short* q = px + k * 32;
#ifndef X_NO_ROT
                split_rotate_prio(k);
#endif
#ifdef X_NO_LDS
                bf16x8 xh = __builtin_bit_cast(bf16x8, ring[(p + 1) % PD].h), xm = xh, xl = xh; (void)q;
                asm volatile("" : "+v"(xh));
#else
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q), xm = *reinterpret_cast<const bf16x8*>(q + plane),
                             xl = *reinterpret_cast<const bf16x8*>(q + 2 * plane);
#endif
                asm volatile("s_waitcnt vmcnt(%3)" : "+v"(ring[p].h), "+v"(ring[p].m), "+v"(ring[p].l) : "n"(3 * (PD - 1)) : "memory");
#ifdef X_NO_MFMA
                asm volatile("" :: "v"(xh), "v"(xm), "v"(xl));   // the LDS reads stay, nothing consumes the weights
#elif defined(X_TWO_MFMA)
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[p].h), xh, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[p].m), xm, acc[0], 0, 0, 0);
                asm volatile("" :: "v"(xl), "v"(ring[p].l));
#else
                split_step<1>(ring[p], xh, xm, xl, q, plane, ld, acc);
#endif
                __builtin_amdgcn_sched_barrier(0);
                const int Vn = V + PD;
                const float4* sb = Vn < E ? wb + (size_t)(Vn - B) * 192 : wnext + (size_t)(Vn - E) * 192;
                gload_b128(ring[p].h, voff, sb);
                gload_b128(ring[p].m, voff, sb + 64);
                gload_b128(ring[p].l, voff, sb + 128);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <int MODE>
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
    __syncthreads();
    const long long t0 = clock64();
    const unsigned voff = (unsigned)lane * 16u;
    const int col4 = 4 * (lane >> 4), row = lane & 15;
    const float4 b0a = *reinterpret_cast<const float4*>(a.b0 + wv * 16 + col4), b0b = *reinterpret_cast<const float4*>(a.b0 + (wv + 16) * 16 + col4);
    const float4 b1a = *reinterpret_cast<const float4*>(a.b1 + wv * 16 + col4), b1b = *reinterpret_cast<const float4*>(a.b1 + (wv + 16) * 16 + col4);
    const float4 b2 = *reinterpret_cast<const float4*>(a.b2 + wv * 16 + col4);
    const unsigned short* xin = xs + row * LDX + 8 * (lane >> 4);
    const unsigned short* h1in = h1 + row * LD1 + 8 * (lane >> 4);
    const unsigned short* h2in = h2 + row * LD1 + 8 * (lane >> 4);
    const sf32x4 z = {0.f, 0.f, 0.f, 0.f};
    sf32x4 a0[2] = {z, z}, a1[2] = {z, z}, a2[1] = {z};
    if constexpr (MODE == 1) {
        auto wb = [&](const float4* w, int tile, int ks) { return w + ((size_t)tile * ks * 2) * 64; };
        const float4* const w0[2] = {wb(a.w0, wv, KS0), wb(a.w0, wv + 16, KS0)};
        wf_layer<2, PDW, false>(voff, w0, KS0, xin, ROWS * LDX, [] {}, a0);
        auto ep0 = [&]() {
            split_st4(h1, ROWS * LD1, row * LD1 + wv * 16 + col4, leaky(a0[0], b0a));
            split_st4(h1, ROWS * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(a0[1], b0b));
        };
        const float4* const w1[2] = {wb(a.w1, wv, KS1), wb(a.w1, wv + 16, KS1)};
        wf_layer<2, PDW>(voff, w1, KS1, h1in, ROWS * LD1, ep0, a1);
        auto ep1 = [&]() {
            split_st4(h2, ROWS * LD1, row * LD1 + wv * 16 + col4, leaky(a1[0], b1a));
            split_st4(h2, ROWS * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(a1[1], b1b));
        };
        const float4* const w2[1] = {wb(a.w2, wv, KS2)};
        wf_layer<1, PDW + 1>(voff, w2, KS2, h2in, ROWS * LD1, ep1, a2);
    } else if constexpr (MODE == 3) {
        auto wb = [&](const float4* w, int tile, int ks) { return w + ((size_t)tile * ks * 3) * 64; };
        const float4* const base[5] = {wb(a.w0, wv, KS0), wb(a.w0, wv + 16, KS0), wb(a.w1, wv, KS1), wb(a.w1, wv + 16, KS1), wb(a.w2, wv, KS2)};
        sf32x4 t0a[1] = {z}, t0b[1] = {z}, t1a[1] = {z}, t1b[1] = {z};
        SplitSlot ring[PDV];
#pragma unroll
        for (int p = 0; p < PDV; ++p) {
            gload_b128(ring[p].h, voff, base[0] + (size_t)p * 192);
            gload_b128(ring[p].m, voff, base[0] + (size_t)p * 192 + 64);
            gload_b128(ring[p].l, voff, base[0] + (size_t)p * 192 + 128);
        }
        vpass<PDV>(ring, voff, base[0], 0, KS0, base[1], xin, ROWS * LDX, LDX, t0a);
        vpass<PDV>(ring, voff, base[1], KS0, KS0, base[2], xin, ROWS * LDX, LDX, t0b);
        split_st4(h1, ROWS * LD1, row * LD1 + wv * 16 + col4, leaky(t0a[0], b0a));
        split_st4(h1, ROWS * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(t0b[0], b0b));
#ifndef X_NO_BARRIER
        lds_barrier();
#endif
        vpass<PDV>(ring, voff, base[2], 2 * KS0, KS1, base[3], h1in, ROWS * LD1, LD1, t1a);
        vpass<PDV>(ring, voff, base[3], 2 * KS0 + KS1, KS1, base[4], h1in, ROWS * LD1, LD1, t1b);
        split_st4(h2, ROWS * LD1, row * LD1 + wv * 16 + col4, leaky(t1a[0], b1a));
        split_st4(h2, ROWS * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(t1b[0], b1b));
#ifndef X_NO_BARRIER
        lds_barrier();
#endif
        vpass<PDV>(ring, voff, base[4], 2 * KS0 + 2 * KS1, KS2, base[4] + (size_t)(KS2 - PDV) * 192, h2in, ROWS * LD1, LD1, a2);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[0].h), "+v"(ring[0].m), "+v"(ring[0].l) :: "memory");
#pragma unroll
        for (int p = 1; p < PDV; ++p) asm volatile("" : "+v"(ring[p].h), "+v"(ring[p].m), "+v"(ring[p].l));
        __builtin_amdgcn_s_setprio(0);
    } else if constexpr (MODE == 2) {
        auto wb = [&](const float4* w, int tile, int ks) { return w + ((size_t)tile * ks * 3) * 64; };
        const float4* const w0[2] = {wb(a.w0, wv, KS0), wb(a.w0, wv + 16, KS0)};
        pl_layer<2, PDP, false>(voff, w0, KS0, xin, ROWS * LDX, [] {}, a0);
        auto ep0 = [&]() {
            split_st4(h1, ROWS * LD1, row * LD1 + wv * 16 + col4, leaky(a0[0], b0a));
            split_st4(h1, ROWS * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(a0[1], b0b));
        };
        const float4* const w1[2] = {wb(a.w1, wv, KS1), wb(a.w1, wv + 16, KS1)};
        pl_layer<2, PDP>(voff, w1, KS1, h1in, ROWS * LD1, ep0, a1);
        auto ep1 = [&]() {
            split_st4(h2, ROWS * LD1, row * LD1 + wv * 16 + col4, leaky(a1[0], b1a));
            split_st4(h2, ROWS * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(a1[1], b1b));
        };
        const float4* const w2[1] = {wb(a.w2, wv, KS2)};
        pl_layer<1, 3>(voff, w2, KS2, h2in, ROWS * LD1, ep1, a2);
    } else {
        auto wb = [&](const float4* w, int tile, int ks) { return w + ((size_t)tile * ks * 3) * 64; };
        sf32x4 t0a[1] = {z}, t0b[1] = {z}, t1a[1] = {z}, t1b[1] = {z};
        split_layer<1, 3, false>(voff, wb(a.w0, wv, KS0), KS0, xin, ROWS * LDX, LDX, [] {}, t0a);
        split_layer<1, 3, false>(voff, wb(a.w0, wv + 16, KS0), KS0, xin, ROWS * LDX, LDX, [] {}, t0b);
        auto ep0 = [&]() {
            split_st4(h1, ROWS * LD1, row * LD1 + wv * 16 + col4, leaky(t0a[0], b0a));
            split_st4(h1, ROWS * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(t0b[0], b0b));
        };
        split_layer<1, 3>(voff, wb(a.w1, wv, KS1), KS1, h1in, ROWS * LD1, LD1, ep0, t1a);
        split_layer<1, 3, false>(voff, wb(a.w1, wv + 16, KS1), KS1, h1in, ROWS * LD1, LD1, [] {}, t1b);
        auto ep1 = [&]() {
            split_st4(h2, ROWS * LD1, row * LD1 + wv * 16 + col4, leaky(t1a[0], b1a));
            split_st4(h2, ROWS * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(t1b[0], b1b));
        };
        split_layer<1, 3>(voff, wb(a.w2, wv, KS2), KS2, h2in, ROWS * LD1, LD1, ep1, a2);
    }
    *reinterpret_cast<float4*>(a.out + ((size_t)blockIdx.x * ROWS + row) * N2 + wv * 16 + col4) =
        make_float4(a2[0][0] + b2.x, a2[0][1] + b2.y, a2[0][2] + b2.z, a2[0][3] + b2.w);
    const long long t1 = clock64();
    if (lane == 0) a.cyc[blockIdx.x * NW + wave] = t1 - t0;
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
static std::vector<float> pack_f32(const std::vector<float>& W, int K, int N, int ks) {
    const int nt = N / 16;
    std::vector<float> ws((size_t)nt * ks * 2 * 64 * 4, 0.f);
    for (int t = 0; t < nt; ++t) for (int s = 0; s < ks; ++s) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const int k = 32 * s + 8 * (l >> 4) + j, n = 16 * t + (l & 15);
        const float w = k < K ? W[(size_t)k * N + n] : 0.f;
        ws[((((size_t)t * ks + s) * 2 + (j >> 2)) * 64 + l) * 4 + (j & 3)] = w;
    }
    return ws;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
static int run(int blocks, const std::vector<float>& X, const std::vector<float>& W0, const std::vector<float>& W1,
               const std::vector<float>& W2, const std::vector<float>& B0, const std::vector<float>& B1,
               const std::vector<float>& B2, const std::vector<double>& ref, double scale) {
    Args a{};
    void *d0, *d1, *d2; float *db0, *db1, *db2, *dx, *dout; long long* dcyc;
    size_t bytes = 0;
    if (MODE == 1) {
        auto p0 = pack_f32(W0, K0, N0, KS0), p1 = pack_f32(W1, N0, N1, KS1), p2 = pack_f32(W2, N1, N2, KS2);
        bytes = (p0.size() + p1.size() + p2.size()) * 4;
        CK(hipMalloc(&d0, p0.size() * 4)); CK(hipMalloc(&d1, p1.size() * 4)); CK(hipMalloc(&d2, p2.size() * 4));
        CK(hipMemcpy(d0, p0.data(), p0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d1, p1.data(), p1.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d2, p2.data(), p2.size() * 4, hipMemcpyHostToDevice));
    } else {
        auto p0 = pack_planes(W0, K0, N0, KS0), p1 = pack_planes(W1, N0, N1, KS1), p2 = pack_planes(W2, N1, N2, KS2);
        bytes = (p0.size() + p1.size() + p2.size()) * 2;
        CK(hipMalloc(&d0, p0.size() * 2)); CK(hipMalloc(&d1, p1.size() * 2)); CK(hipMalloc(&d2, p2.size() * 2));
        CK(hipMemcpy(d0, p0.data(), p0.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d1, p1.data(), p1.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d2, p2.data(), p2.size() * 2, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&db0, N0 * 4)); CK(hipMalloc(&db1, N1 * 4)); CK(hipMalloc(&db2, N2 * 4)); CK(hipMalloc(&dx, X.size() * 4));
    CK(hipMalloc(&dout, (size_t)blocks * ROWS * N2 * 4)); CK(hipMalloc(&dcyc, (size_t)blocks * NW * 8));
    CK(hipMemcpy(db0, B0.data(), N0 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db1, B1.data(), N1 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db2, B2.data(), N2 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    a.w0 = (const float4*)d0; a.w1 = (const float4*)d1; a.w2 = (const float4*)d2; a.b0 = db0; a.b1 = db1; a.b2 = db2; a.x = dx; a.out = dout; a.cyc = dcyc;
    const size_t ldsb = (size_t)(3 * ROWS * LDX + 2 * 3 * ROWS * LD1) * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(NTH), ldsb, 0, a);
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
    double emax = 0;
    for (size_t i = 0; i < out.size(); ++i) emax = std::max(emax, std::fabs(out[i] - ref[i]));
    printf("mode %d (%s) blocks %d: %lld cycles per 16-row tile (median; max %lld), kernel %.1f us; weights %.2f MB per tile; max |err| %.2e (|out| max %.2f)\n",
           MODE, MODE == 1 ? "fp32 weights sliced in registers" : MODE == 2 ? "pre-sliced planes, two tiles per wave" : MODE == 3 ? "pre-sliced planes, one ring across the passes" : "pre-sliced planes", blocks, wg[blocks / 2], wg.back(), ms * 1e3, bytes / 1e6, emax, scale);
    return 0;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 194;
    uint32_t s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    std::vector<float> X((size_t)ROWS * K0), W0((size_t)K0 * N0), W1((size_t)N0 * N1), W2((size_t)N1 * N2), B0(N0), B1(N1), B2(N2);
    for (auto& v : X) v = 2.f * rnd();
    for (auto& v : W0) v = 0.2f * rnd(); for (auto& v : W1) v = 0.15f * rnd(); for (auto& v : W2) v = 0.15f * rnd();
    for (auto& v : B0) v = 0.1f * rnd(); for (auto& v : B1) v = 0.1f * rnd(); for (auto& v : B2) v = 0.1f * rnd();
    std::vector<double> ref((size_t)ROWS * N2);
    double scale = 0;
    for (int r = 0; r < ROWS; ++r) {
        std::vector<double> h1(N0), h2(N1);
        for (int n = 0; n < N0; ++n) { double t = B0[n]; for (int k = 0; k < K0; ++k) t += (double)X[(size_t)r * K0 + k] * W0[(size_t)k * N0 + n]; h1[n] = std::max(t, 0.2 * t); }
        for (int n = 0; n < N1; ++n) { double t = B1[n]; for (int k = 0; k < N0; ++k) t += h1[k] * W1[(size_t)k * N1 + n]; h2[n] = std::max(t, 0.2 * t); }
        for (int n = 0; n < N2; ++n) { double t = B2[n]; for (int k = 0; k < N1; ++k) t += h2[k] * W2[(size_t)k * N2 + n]; ref[(size_t)r * N2 + n] = t; scale = std::max(scale, std::fabs(t)); }
    }
#ifndef ONLY_MODE3
    if (run<0>(blocks, X, W0, W1, W2, B0, B1, B2, ref, scale)) return 1;
    if (run<1>(blocks, X, W0, W1, W2, B0, B1, B2, ref, scale)) return 1;
    if (run<2>(blocks, X, W0, W1, W2, B0, B1, B2, ref, scale)) return 1;
#endif
    if (run<3>(blocks, X, W0, W1, W2, B0, B1, B2, ref, scale)) return 1;
    return 0;
}
