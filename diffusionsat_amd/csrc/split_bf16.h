// fp32 products from bf16 slices on the bf16 matrix pipe.
//
// gfx950 has no reduced-precision fp32 MFMA; v_mfma_f32_16x16x4_f32 runs at the fp32 vector rate, 1/16 of
// v_mfma_f32_16x16x32_bf16.  Every fp32 number is exactly  h + m + l  with three bf16 values (8 significant bits
// each, round to nearest even at every step, the residuals are exact in fp32), and a product of two bf16 values is
// exact in fp32.  Of the nine slice products of  x * w  the six of relative weight >= 2^-16,
//     xh*wl + xl*wh + xm*wm + xm*wh + xh*wm + xh*wh        (small terms first),
// are accumulated in fp32 by six MFMAs; the dropped ones are below 2^-24 of the product, i.e. below fp32's own
// rounding.  Measured against fp64 on the clause layers' shapes (tools/split_bf16_probe.hip): max error / sum|xw|
// 1.55e-7 for the six-product form, 1.69e-7 for v_mfma_f32_16x16x4_f32 -- and 2.3-2.6x fewer cycles.
//
// Operand images: activations live in LDS as three planes of bf16 [rows][ld] with ld = K + 16 elements, i.e. a row
// stride of 32 bytes mod 64: a lane's ds_read_b128 (row l&15, k = 32s + 8(l>>4) .. +7) has the byte pattern of
// fused_common.h's fp32 fragments, and the hardware serves it eight rows of two k-groups at a time -- with K + 8
// (16 bytes mod 64) rows r, k-group 1 and r + 1, k-group 0 share banks (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// was 0.46 then).  Weights come from HBM/L2 as [tile][k-step][plane][lane] 16-byte fragments (dsat_api.hip packs
// them), straight into registers.
// As in fused_common.h the weight fragment is the MFMA's A operand, so a lane ends up with four consecutive
// output columns of row l&15.
#pragma once
#include <hip/hip_runtime.h>
#include "fused_common.h"   // gload_b128

namespace dsat {

typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// Two floats -> two bf16 (round to nearest even) in one instruction, v_cvt_pk_bf16_f32; low half = a.
typedef __bf16 bf16pair __attribute__((ext_vector_type(2)));
typedef float f32pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pk(float a, float b) {
    const f32pair v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16pair));
}
__device__ __forceinline__ float bf16_lo(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf16_hi(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }
// (a, b) -> packed slices h, m, l of both: 3 conversions, 4 unpacks, 4 subtractions
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_pk(a, b);
    const float ra = a - bf16_lo(h), rb = b - bf16_hi(h);
    m = bf16_pk(ra, rb);
    l = bf16_pk(ra - bf16_lo(m), rb - bf16_hi(m));
}

// Four consecutive columns of one row -> the three planes (8-byte stores); `off` = row * ld + col in elements,
// a multiple of 4; `plane` = elements per plane.
__device__ __forceinline__ void split_st4(unsigned short* planes, int plane, int off, float4 v) {
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair(v.x, v.y, h0, m0, l0);
    split_pair(v.z, v.w, h1, m1, l1);
    *reinterpret_cast<uint2*>(planes + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(planes + plane + off) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(planes + 2 * plane + off) = make_uint2(l0, l1);
}

// The weight ring.  Plain loads one step ahead were enough for a first version of P2 (four row blocks share a
// fragment: 384 MFMA cycles per wave and k-step); a 16-row tile has 96 cycles of MFMA work per 3 KiB of weights, is
// bound by the weight stream and needs several steps in flight -- and hipcc drains every load at a loop's back edge
// (s_waitcnt vmcnt(0)), so a compiler-scheduled ring degenerates to one step.  Hence, everywhere, the same device as
// fused_common.h: the loads are inline asm (gload_b128: SGPR base + 32-bit VGPR offset), completion is counted by
// hand, every wait names the registers it releases, and a slot is refilled only after the last MFMA that reads it
// (tools/audit_ring.py replays the vmcnt FIFO over the ISA).  A slot = the three slices of one k-step; refills
// are clamped to the last step, so exactly 3*PD loads are in flight at every wait of the main loop.
struct SplitSlot { sf32x4 h, m, l; };

// A layer may skip a run of k-steps (`gap_len` steps from step `gap_at` on: their products were formed elsewhere and
// enter through the accumulators' start value): step j of the `stot` executed ones is k-step j + (j >= gap_at ?
// gap_len : 0) of the weight image and of the activation planes.
__device__ __forceinline__ int split_kstep(int j, int gap_at, int gap_len) { return j + (j >= gap_at ? gap_len : 0); }

template <int PD, int P0 = 0, int P1 = PD>   // slots [P0, P1): the prime may be issued in two parts, in slot order
__device__ __forceinline__ void split_prime(SplitSlot (&ring)[PD], unsigned voff, const float4* __restrict__ wbase,
                                            int stot, int gap_at = 0, int gap_len = 0) {
    static_assert(3 * PD <= 63, "vmcnt field is 6 bits");
    const int last = stot - 1;
#pragma unroll
    for (int p = P0; p < P1; ++p) {
        const float4* sb = wbase + (size_t)split_kstep(p < last ? p : last, gap_at, gap_len) * 192;
        gload_b128(ring[p].h, voff, sb);
        gload_b128(ring[p].m, voff, sb + 64);
        gload_b128(ring[p].l, voff, sb + 128);
    }
}

// One k-step of one column tile: the slot's three weight slices against NRB row blocks.  x0*: the fragments of
// row block 0 (read before the caller's wait, so their LDS latency hides behind it); the next row block's
// fragments are read while the current one's six MFMAs issue.
template <int NRB>
__device__ __forceinline__ void split_step(const SplitSlot& w, bf16x8 xh, bf16x8 xm, bf16x8 xl, const unsigned short* q,
                                           int plane, int ld, sf32x4 (&acc)[NRB]) {
    const bf16x8 wh = __builtin_bit_cast(bf16x8, w.h), wm = __builtin_bit_cast(bf16x8, w.m),
                 wl = __builtin_bit_cast(bf16x8, w.l);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        bf16x8 nh = xh, nm = xm, nl = xl;
        if (rb + 1 < NRB) {
            const unsigned short* qn = q + (rb + 1) * 16 * ld;
            nh = *reinterpret_cast<const bf16x8*>(qn);
            nm = *reinterpret_cast<const bf16x8*>(qn + plane);
            nl = *reinterpret_cast<const bf16x8*>(qn + 2 * plane);
        }
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[rb], 0, 0, 0);
        xh = nh; xm = nm; xl = nl;
    }
}

// The matrix pipe serves the oldest of a SIMD's waves first, so four waves with the same work finish a k-loop up to
// 9 k cycles apart and the last one runs its tail alone, latencies exposed.  Rotating the issue priority among the
// SIMD's four waves from k-step to k-step keeps them together: 359.7 -> 363.8 steps/s at config 2 (P3 22.2 -> 21.8
// us; the same rotation in the fp32 chains of fused_common.h cost P1 4 us and is not used there).
// -DDSAT_SPLIT_NO_ROTPRIO turns it off.
__device__ __forceinline__ void split_rotate_prio(int step) {
#ifndef DSAT_SPLIT_NO_ROTPRIO
#ifndef DSAT_ROT_SHIFT
#define DSAT_ROT_SHIFT 1   // a turn lasts two k-steps (0: 359.0, 1: 360.9, 2: 358.2 steps/s on one box)
#endif
    const int x = ((step >> DSAT_ROT_SHIFT) + (int)(threadIdx.x >> 8)) & 3;
    if (x == 0) __builtin_amdgcn_s_setprio(0);
    else if (x == 1) __builtin_amdgcn_s_setprio(1);
    else if (x == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
#else
    (void)step;
#endif
}

// k-steps [s0, s1) of a primed ring (s0 and s1 - s0 multiples of PD; slot p holds step s0 + p on entry and
// min(s1 + p, stot - 1) on return).  `px`: the lane's element of plane 0 in a buffer whose column 0 is step sbuf0.
template <int NRB, int PD>
__device__ __forceinline__ void split_ring_steps(SplitSlot (&ring)[PD], unsigned voff, const float4* __restrict__ wbase,
                                                 int s0, int s1, int stot, const unsigned short* px, int plane, int ld,
                                                 int sbuf0, sf32x4 (&acc)[NRB], int gap_at = 0, int gap_len = 0) {
    const int last = stot - 1;
    for (int s = s0; s < s1; s += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const unsigned short* q = px + (split_kstep(s + p, gap_at, gap_len) - sbuf0) * 32;
            split_rotate_prio(s + p);
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q), xm = *reinterpret_cast<const bf16x8*>(q + plane),
                         xl = *reinterpret_cast<const bf16x8*>(q + 2 * plane);
            // slot p was requested PD steps ago; 3 * (PD - 1) younger loads may stay in flight
            asm volatile("s_waitcnt vmcnt(%3)" : "+v"(ring[p].h), "+v"(ring[p].m), "+v"(ring[p].l) : "n"(3 * (PD - 1)) : "memory");
            split_step<NRB>(ring[p], xh, xm, xl, q, plane, ld, acc);
            __builtin_amdgcn_sched_barrier(0);
            const int sn = s + p + PD;
            const float4* sb = wbase + (size_t)split_kstep(sn < last ? sn : last, gap_at, gap_len) * 192;
            gload_b128(ring[p].h, voff, sb);
            gload_b128(ring[p].m, voff, sb + 64);
            gload_b128(ring[p].l, voff, sb + 128);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// The last stot - s1 (< PD) steps, already in slots 0.. when the main loop ended; drains the queue.
template <int NRB, int PD>
__device__ __forceinline__ void split_ring_tail(SplitSlot (&ring)[PD], int s1, int stot, const unsigned short* px,
                                                int plane, int ld, int sbuf0, sf32x4 (&acc)[NRB], int gap_at = 0,
                                                int gap_len = 0) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[0].h), "+v"(ring[0].m), "+v"(ring[0].l) :: "memory");
#ifndef DSAT_SPLIT_NO_ROTPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int p = 1; p < PD; ++p) asm volatile("" : "+v"(ring[p].h), "+v"(ring[p].m), "+v"(ring[p].l));
#pragma unroll
    for (int p = 0; p + 1 < PD; ++p) {
        if (s1 + p < stot) {
            const unsigned short* q = px + (split_kstep(s1 + p, gap_at, gap_len) - sbuf0) * 32;
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q), xm = *reinterpret_cast<const bf16x8*>(q + plane),
                         xl = *reinterpret_cast<const bf16x8*>(q + 2 * plane);
            split_step<NRB>(ring[p], xh, xm, xl, q, plane, ld, acc);
        }
    }
}

// One whole layer as ONE loop nest, ring included: prime, the caller's `prep()` (the last piece of input
// preparation, or the previous layer's epilogue), the barrier that publishes the input, the k-steps, the drain.
// A ring must not be shared between differently allocated loops or code paths: hipcc would copy its registers
// from one allocation to the other while their loads are still in flight (fused_common.h has the same rule).
// NRB == 0: a wave without a column tile in this layer only takes part in prep() and the barriers.
//
// split_layer_segments: the input arrives in three segments of `kseg` k-steps through two buffers (P2): steps
// [0, kseg) read pxA, [kseg, 2 kseg) pxB, then `mid()` (barrier - refill buffer A - barrier), then
// [2 kseg, 3 kseg) read pxA again.
template <int NRB, int PD, typename Prep, typename Mid>
__device__ __forceinline__ void split_layer_segments(unsigned voff, const float4* __restrict__ wbase, int kseg,
                                                     const unsigned short* pxA, const unsigned short* pxB, int plane,
                                                     int ld, Prep&& prep, Mid&& mid, sf32x4 (&acc)[NRB > 0 ? NRB : 1]) {
    if constexpr (NRB == 0) {
        prep();
        lds_barrier();
        mid();
    } else {
        const int stot = 3 * kseg, last = stot - 1;
        SplitSlot ring[PD];
        split_prime<PD>(ring, voff, wbase, stot);
        prep();
        lds_barrier();   // orders LDS only: the ring's loads stay in flight across it
        for (int s = 0; s < stot; s += PD) {
            if (s == 2 * kseg) mid();
            const unsigned short* px = s < kseg ? pxA + s * 32 : s < 2 * kseg ? pxB + (s - kseg) * 32 : pxA + (s - 2 * kseg) * 32;
#pragma unroll
            for (int p = 0; p < PD; ++p) {
                const unsigned short* q = px + p * 32;
                split_rotate_prio(s + p);
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q), xm = *reinterpret_cast<const bf16x8*>(q + plane),
                             xl = *reinterpret_cast<const bf16x8*>(q + 2 * plane);
                asm volatile("s_waitcnt vmcnt(%3)" : "+v"(ring[p].h), "+v"(ring[p].m), "+v"(ring[p].l) : "n"(3 * (PD - 1)) : "memory");
                split_step<NRB>(ring[p], xh, xm, xl, q, plane, ld, acc);
                __builtin_amdgcn_sched_barrier(0);
                const int sn = s + p + PD;
                const float4* sb = wbase + (size_t)(sn < last ? sn : last) * 192;
                gload_b128(ring[p].h, voff, sb);
                gload_b128(ring[p].m, voff, sb + 64);
                gload_b128(ring[p].l, voff, sb + 128);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[0].h), "+v"(ring[0].m), "+v"(ring[0].l) :: "memory");
#ifndef DSAT_SPLIT_NO_ROTPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int p = 1; p < PD; ++p) asm volatile("" : "+v"(ring[p].h), "+v"(ring[p].m), "+v"(ring[p].l));
    }
}

// BARRIER = false: the input is already published (prep() then only has to run somewhere before the k-steps).
template <int NRB, int PD, bool BARRIER = true, typename Prep>
__device__ __forceinline__ void split_layer(unsigned voff, const float4* __restrict__ wbase, int stot,
                                            const unsigned short* px, int plane, int ld, Prep&& prep,
                                            sf32x4 (&acc)[NRB > 0 ? NRB : 1]) {
    if constexpr (NRB == 0) {
        prep();
        if (BARRIER) lds_barrier();
    } else {
        SplitSlot ring[PD];
        split_prime<PD>(ring, voff, wbase, stot);
        prep();
        if (BARRIER) lds_barrier();
        const int main = stot / PD * PD;
        split_ring_steps<NRB, PD>(ring, voff, wbase, 0, main, stot, px, plane, ld, 0, acc);
        split_ring_tail<NRB, PD>(ring, main, stot, px, plane, ld, 0, acc);
    }
}

// ONE ring across the passes of a chain (role A of P1: five passes -- two column tiles in each of lit_mlp's 512-wide
// layers, one in the last -- that used to prime, run and drain a ring each).  The wave's fragments form one stream: a pass
// covers the virtual steps [B, B + ks), slot V % PD holds step V, and when step V has been consumed its slot asks for
// step V + PD, which may already belong to the NEXT pass (weights do not depend on activations): no refills clamped to
// the pass's last step (PD per pass: a quarter of the stream of a 16-row tile, which is bound by exactly that stream,
// 64 B/clk/CU: tools/l2_stream_probe.hip), no empty pipe when the next pass starts.  `wnext`: the next pass's fragments;
// for the chain's last pass its own last PD steps again, so that every wait has 3 * (PD - 1) younger loads behind the
// slot it releases.  The first turn of a pass may start inside the ring and the last may end inside it: the steps
// outside [B, B + ks) are skipped (a uniform branch).  tools/lit16w_probe.hip, mode 3: 57.4 k -> 52.6 k cycles.
template <int PD>
__device__ __forceinline__ void split_vprime(SplitSlot (&ring)[PD], unsigned voff, const float4* __restrict__ wbase) {
    static_assert(3 * PD <= 63, "vmcnt field is 6 bits");
#pragma unroll
    for (int p = 0; p < PD; ++p) {
        gload_b128(ring[p].h, voff, wbase + (size_t)p * 192);
        gload_b128(ring[p].m, voff, wbase + (size_t)p * 192 + 64);
        gload_b128(ring[p].l, voff, wbase + (size_t)p * 192 + 128);
    }
}
template <int NRB, int PD>
__device__ __forceinline__ void split_vpass(SplitSlot (&ring)[PD], unsigned voff, const float4* __restrict__ wb, int B, int ks,
                                            const float4* __restrict__ wnext, const unsigned short* px, int plane, int ld,
                                            sf32x4 (&acc)[NRB]) {
    const int E = B + ks;
    for (int c = B / PD * PD; c < E; c += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const int V = c + p;
            if (V >= B && V < E) {
                const unsigned short* q = px + (V - B) * 32;
                split_rotate_prio(V - B);
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q), xm = *reinterpret_cast<const bf16x8*>(q + plane),
                             xl = *reinterpret_cast<const bf16x8*>(q + 2 * plane);
                // slot p was requested PD steps ago; 3 * (PD - 1) younger loads may stay in flight
                asm volatile("s_waitcnt vmcnt(%3)" : "+v"(ring[p].h), "+v"(ring[p].m), "+v"(ring[p].l) : "n"(3 * (PD - 1)) : "memory");
                split_step<NRB>(ring[p], xh, xm, xl, q, plane, ld, acc);
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
// After the chain's last pass: the refills past its end land before their registers are given to anything else.
template <int PD>
__device__ __forceinline__ void split_vdrain(SplitSlot (&ring)[PD]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[0].h), "+v"(ring[0].m), "+v"(ring[0].l) :: "memory");
#pragma unroll
    for (int p = 1; p < PD; ++p) asm volatile("" : "+v"(ring[p].h), "+v"(ring[p].m), "+v"(ring[p].l));
#ifndef DSAT_SPLIT_NO_ROTPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// Two consecutive columns (8 bytes of fp32) -> the three planes (4-byte stores); `off` a multiple of 2.
__device__ __forceinline__ void split_st2(unsigned short* planes, int plane, int off, float2 v) {
    unsigned h, m, l;
    split_pair(v.x, v.y, h, m, l);
    *reinterpret_cast<unsigned*>(planes + off) = h;
    *reinterpret_cast<unsigned*>(planes + plane + off) = m;
    *reinterpret_cast<unsigned*>(planes + 2 * plane + off) = l;
}
__device__ __forceinline__ void split_zero4(unsigned short* planes, int plane, int off) {
    const uint2 z = make_uint2(0u, 0u);
    *reinterpret_cast<uint2*>(planes + off) = z;
    *reinterpret_cast<uint2*>(planes + plane + off) = z;
    *reinterpret_cast<uint2*>(planes + 2 * plane + off) = z;
}

// A 2- or 3-layer MLP on a tile of 16*NRB rows whose input already lies in LDS as slices: one column tile per
// wave and layer (ntiles <= waves), hidden activations (leaky ReLU 0.2) go to hA, then hB (row stride ldh, at
// least 32 * ksteps of the layer that reads them + 8), the last layer to final_store(row, col, float4).
// `prep()` is the caller's last piece of input preparation: it runs after the first weight fragments have been
// requested, and the chain takes the barrier that publishes the input.  Ends with a barrier.
__device__ __forceinline__ void pin_bias(float4 (&v)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
}
struct SplitLayer { const uint4* ws; const float* bias; int ksteps, ntiles; };
struct SplitChain {
    SplitLayer L[3];
    int nlayers;
    // layer 0 may start from partial sums formed elsewhere (`init` != nullptr: the caller passes this lane's values as
    // `start`): the products of k-steps [gap_at, gap_at + gap_len), which the layer then skips
    const float* init = nullptr;
    int gap_at = 0, gap_len = 0;
};

template <int NRB, int NTHREADS, int PD, typename Store, typename Prep>
__device__ __forceinline__ void split_chain(const SplitChain& m, const unsigned short* in, int ld_in,
                                            unsigned short* hA, unsigned short* hB, int ldh, int wv, int lane,
                                            int tid, Store&& final_store, Prep&& prep, const float4* start = nullptr) {
    // start: NRB float4 in the caller's registers -- m.init's values for this lane (row rb * 16 + (lane & 15), columns
    // wv * 16 + 4 * (lane >> 4) .. + 3), requested early enough that their latency is hidden
    constexpr int ROWS = 16 * NRB;
    const int col4 = 4 * (lane >> 4);
    bool on[3];
    float4 bv[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        on[l] = l < m.nlayers && wv < m.L[l].ntiles;
        bv[l] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on[l]) bv[l] = *reinterpret_cast<const float4*>(m.L[l].bias + wv * 16 + col4);
    }
    SplitSlot ring[PD];
    const unsigned voff = (unsigned)lane * 16u;
    const float4* wb[3];
#pragma unroll
    for (int l = 0; l < 3; ++l)
        wb[l] = reinterpret_cast<const float4*>(m.L[l < m.nlayers ? l : 0].ws) + ((size_t)(on[l] ? wv : 0) * m.L[l < m.nlayers ? l : 0].ksteps * 3) * 64;
    // layer 0's start values: already in registers (the caller loaded them early), or zeros
    float4 iv[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) iv[rb] = m.init != nullptr && start != nullptr ? start[rb] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int g_at = m.init != nullptr ? m.gap_at : 0, g_len = m.init != nullptr ? m.gap_len : 0;
    const int ks0 = m.L[0].ksteps - g_len;   // executed k-steps of layer 0
    pin_bias(bv);   // the biases are ordinary loads: complete them before the asm loads start
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) asm volatile("" : "+v"(iv[rb].x), "+v"(iv[rb].y), "+v"(iv[rb].z), "+v"(iv[rb].w));
    // the first layer's prime in two parts: the registers prep() still holds (a gather's sums) cap what fits before it
    constexpr int PRE = PD < 2 ? PD : 2;
    if (on[0]) split_prime<PD, 0, PRE>(ring, voff, wb[0], ks0, g_at, g_len);
    prep();
    if (on[0]) split_prime<PD, PRE, PD>(ring, voff, wb[0], ks0, g_at, g_len);
    lds_barrier();
    const unsigned short* src = in;
    int ld = ld_in;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        if (l >= m.nlayers) break;
        const bool last = l + 1 == m.nlayers;
        sf32x4 acc[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
            acc[rb] = l == 0 ? sf32x4{iv[rb].x, iv[rb].y, iv[rb].z, iv[rb].w} : sf32x4{0.f, 0.f, 0.f, 0.f};
        if (on[l]) {
            const int ks = l == 0 ? ks0 : m.L[l].ksteps, main = ks / PD * PD;
            const int ga = l == 0 ? g_at : 0, gl = l == 0 ? g_len : 0;
            const unsigned short* px = src + (lane & 15) * ld + 8 * (lane >> 4);
            split_ring_steps<NRB, PD>(ring, voff, wb[l], 0, main, ks, px, ROWS * ld, ld, 0, acc, ga, gl);
            split_ring_tail<NRB, PD>(ring, main, ks, px, ROWS * ld, ld, 0, acc, ga, gl);
        }
        if (!last) {
            if (on[l + 1]) split_prime<PD>(ring, voff, wb[l + 1], m.L[l + 1].ksteps);   // travel during the epilogue
            unsigned short* dst = l == 0 ? hA : hB;
            if (on[l]) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    float4 v = make_float4(acc[rb][0] + bv[l].x, acc[rb][1] + bv[l].y, acc[rb][2] + bv[l].z, acc[rb][3] + bv[l].w);
                    v.x = fmaxf(v.x, 0.2f * v.x); v.y = fmaxf(v.y, 0.2f * v.y);
                    v.z = fmaxf(v.z, 0.2f * v.z); v.w = fmaxf(v.w, 0.2f * v.w);
                    split_st4(dst, ROWS * ldh, (rb * 16 + (lane & 15)) * ldh + wv * 16 + col4, v);
                }
            }
            // columns between this layer's last tile and the end of the next layer's last k-step: zeros
            const int c0 = m.L[l].ntiles * 16, c1 = m.L[l + 1].ksteps * 32, nq = (c1 - c0) >> 2;
            for (int i = tid; i < ROWS * nq; i += NTHREADS) split_zero4(dst, ROWS * ldh, (i / nq) * ldh + c0 + 4 * (i % nq));
            lds_barrier();
            src = dst;
            ld = ldh;
        } else if (on[l]) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
                final_store(rb * 16 + (lane & 15), wv * 16 + col4,
                            make_float4(acc[rb][0] + bv[l].x, acc[rb][1] + bv[l].y, acc[rb][2] + bv[l].z, acc[rb][3] + bv[l].w));
        }
    }
    lds_barrier();
}

}  // namespace dsat
