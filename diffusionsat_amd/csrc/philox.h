// Philox4x32-10 counter-based RNG (Salmon et al., SC'11) for the device-side draws
// that replace TF's RNG at inference (model/query_sat.py:145, :239, :57).
// oracle/philox.py is the NumPy mirror used by the tests (bit-exact on the integers).
//
//   counter = (index, round, step, stream)   key = (seed_lo, seed_hi)
//   stream 0: normal[round][index][0..3]  (Box-Muller on the 4 outputs)
//   stream 1: labels[index]   = x0 & 1
//   stream 2: uniform[index]  = (x0 >> 8) * 2^-24
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsat {

struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)M0 * c.x;
        const uint64_t p1 = (uint64_t)M1 * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += W0;
        k1 += W1;
    }
    return c;
}

enum { STREAM_NORMAL = 0, STREAM_LABELS = 1, STREAM_UNIFORM = 2 };

__device__ inline void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 5.9604644775390625e-8f;  // (0,1]
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;           // [0,1)
    const float rad = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    z0 = rad * c;
    z1 = rad * s;
}

}  // namespace dsat
