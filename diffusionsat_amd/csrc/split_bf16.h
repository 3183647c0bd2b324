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
// Operand images: activations live in LDS as three planes of bf16 [rows][ld] (ld = K + 8 elements: a lane's
// ds_read_b128 of row l&15, k = 32s + 8(l>>4) .. +7 is conflict-free); weights come from HBM/L2 as
// [tile][k-step][plane][lane] 16-byte fragments (dsat_api.hip packs them), straight into registers.
// As in fused_common.h the weight fragment is the MFMA's A operand, so a lane ends up with four consecutive
// output columns of row l&15.
#pragma once
#include <hip/hip_runtime.h>

namespace dsat {

typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rne(float x) {   // finite inputs (state, messages and losses are)
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_up(unsigned h) { return __uint_as_float(h << 16); }

// Four consecutive columns of one row -> the three planes (8-byte stores); `off` = row * ld + col in elements,
// a multiple of 4; `plane` = elements per plane.
__device__ __forceinline__ void split_st4(unsigned short* planes, int plane, int off, float4 v) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = bf16_rne(x[i]);
        const float r1 = x[i] - bf16_up(h[i]);
        m[i] = bf16_rne(r1);
        const float r2 = r1 - bf16_up(m[i]);
        l[i] = bf16_rne(r2);
    }
    *reinterpret_cast<uint2*>(planes + off) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    *reinterpret_cast<uint2*>(planes + plane + off) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
    *reinterpret_cast<uint2*>(planes + 2 * plane + off) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}

// The three weight fragments of one (tile, k-step), one step ahead of their use.
struct SplitW { uint4 h, m, l; };
__device__ __forceinline__ SplitW split_wload(const uint4* __restrict__ wp, int step) {
    const uint4* p = wp + (size_t)step * 192;   // 3 planes x 64 lanes
    SplitW w;
    w.h = p[0]; w.m = p[64]; w.l = p[128];
    return w;
}

// k-steps [s0, s1) of one column tile for NRB row blocks.  `wp`: the lane's fragment of (tile, step 0, plane 0);
// `w`: fragments of step s0, already requested; on return those of step min(s1, stot - 1).  `px`: the lane's
// element (row l&15 of the first row block, k = 8(l>>4)) of plane 0 in a buffer whose column 0 is k-step `sbuf0`.
template <int NRB>
__device__ __forceinline__ void split_ksteps(const uint4* __restrict__ wp, int s0, int s1, int stot, SplitW& w,
                                             const unsigned short* px, int plane, int ld, int sbuf0,
                                             sf32x4 (&acc)[NRB]) {
    for (int s = s0; s < s1; ++s) {
        const SplitW nx = split_wload(wp, s + 1 < stot ? s + 1 : s);
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&w.h), wm = *reinterpret_cast<const bf16x8*>(&w.m),
                     wl = *reinterpret_cast<const bf16x8*>(&w.l);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const unsigned short* p = px + rb * 16 * ld + (s - sbuf0) * 32;
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(p);
            const bf16x8 xm = *reinterpret_cast<const bf16x8*>(p + plane);
            const bf16x8 xl = *reinterpret_cast<const bf16x8*>(p + 2 * plane);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[rb], 0, 0, 0);
        }
        w = nx;
    }
}

}  // namespace dsat
