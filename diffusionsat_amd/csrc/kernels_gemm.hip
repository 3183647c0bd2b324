// fp32 MFMA GEMM for the five QuerySAT MLPs (model/mlp.py:42-50, Dense = x@W+b,
// hidden activation leaky_relu(0.2)), gfx950.
//
//   out[M, npad] = act( [A0 | A1 | A2 | A3][M, kpad] @ W[kpad, npad] + bias )
//
// * concat-free: the reference concatenates inputs (query_sat.py:239,258,277); here
//   each concat member is an A block with its own base pointer and row stride, and
//   the packed weight has the members' rows back to back (each padded to 16).
// * W is stored transposed ([npad][kpad], K contiguous) so both operands are staged
//   as [64 rows][16 k] tiles with identical addressing.
// * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain in k order), 64
//   FLOP/clk/SIMD.  Workgroup = 4 waves = 64x64 output tile, each wave one 32x32
//   accumulator.  MFMA j of a 16-deep chunk multiplies k = {j, j+8}: lane half h
//   supplies k = 8h + j, so each lane's 8 operands are two ds_read_b128.
// * LDS rows padded to 20 floats: the ds_read_b128 lane groups of a wave then hit 16
//   distinct 16-B slots (conflict-free; see DESIGN.md "GEMM").
// * register-staged double buffering: chunk i+1 is loaded to VGPRs before chunk i's
//   MFMAs and written to the other LDS stage after them; one barrier per chunk.
#include "dsat_internal.h"

namespace dsat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 16, LDS_LD = 20;
constexpr int TILE_F = BM * LDS_LD;  // floats per staged tile

__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(GemmArgs a, const int* __restrict__ done) {
    if (done != nullptr && *done != 0) return;
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE_F];  // [stage][A|W][64][20]

    const int tid = threadIdx.x;
    const int lrow = tid >> 2;       // 0..63: tile row (A) / tile column (W)
    const int kq = (tid & 3) * 4;    // k offset of this thread's float4
    int m = blockIdx.x * BM + lrow;
    if (m >= a.M) m = a.M - 1;
    int n = blockIdx.y * BN + lrow;
    if (n >= a.npad) n = a.npad - 1;

    int total_chunks = 0;
    for (int b = 0; b < a.nblocks; ++b) total_chunks += a.kchunks[b];

    // running position of the loader
    int blk = 0, ch = 0;
    const float* ap = a.A[0] + (size_t)m * a.lda[0] + kq;
    const float* wp = a.Wt + (size_t)n * a.kpad + kq;

    auto load_next = [&](float4& ra, float4& rw) {
        ra = *reinterpret_cast<const float4*>(ap + ch * BK);
        rw = *reinterpret_cast<const float4*>(wp);
        wp += BK;
        if (++ch == a.kchunks[blk]) {
            ch = 0;
            ++blk;
            if (blk < a.nblocks) ap = a.A[blk] + (size_t)m * a.lda[blk] + kq;
        }
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const bool wave_active = (blockIdx.y * BN + wc * 32) < a.npad;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    float4 ra, rw;
    load_next(ra, rw);
    {
        float* sa = smem + lrow * LDS_LD + kq;
        *reinterpret_cast<float4*>(sa) = ra;
        *reinterpret_cast<float4*>(sa + TILE_F) = rw;
    }
    __syncthreads();

    for (int i = 0; i < total_chunks; ++i) {
        const int st = i & 1;
        const bool more = (i + 1) < total_chunks;
        if (more) load_next(ra, rw);
        if (wave_active) {
            const float* pa = smem + st * 2 * TILE_F + (wr * 32 + r) * LDS_LD + 8 * h;
            const float* pw = smem + st * 2 * TILE_F + TILE_F + (wc * 32 + r) * LDS_LD + 8 * h;
            const float4 a0 = *reinterpret_cast<const float4*>(pa);
            const float4 a1 = *reinterpret_cast<const float4*>(pa + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(pw);
            const float4 b1 = *reinterpret_cast<const float4*>(pw + 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
        }
        if (more) {
            float* sa = smem + (st ^ 1) * 2 * TILE_F + lrow * LDS_LD + kq;
            *reinterpret_cast<float4*>(sa) = ra;
            *reinterpret_cast<float4*>(sa + TILE_F) = rw;
        }
        __syncthreads();
    }

    if (!wave_active) return;
    const int col = blockIdx.y * BN + wc * 32 + r;
    const float bv = a.bias[col];
    float* obase;
    int ldo, ocol, nstore;
    if (col < a.split) {
        obase = a.out0; ldo = a.ldo0; ocol = col; nstore = a.nstore0;
    } else {
        obase = a.out1; ldo = a.ldo1; ocol = col - a.split; nstore = a.nstore1;
    }
    if (ocol >= nstore) return;
    const int row0 = blockIdx.x * BM + wr * 32 + 4 * h;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = row0 + (i & 3) + 8 * (i >> 2);
        if (row < a.M) {
            float v = acc[i] + bv;
            if (a.act) v = v > 0.f ? v : 0.2f * v;
            obase[(size_t)row * ldo + ocol] = v;
        }
    }
}

void launch_gemm(const GemmArgs& a, const int* done, hipStream_t st) {
    if (a.M <= 0) return;
    dim3 grid((a.M + BM - 1) / BM, (a.npad + BN - 1) / BN);
    hipLaunchKernelGGL(gemm_f32_mfma_kernel, grid, dim3(256), 0, st, a, done);
}

}  // namespace dsat
