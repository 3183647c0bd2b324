// Row-tile fused MLP chain: a workgroup owns 16*RB rows and runs every Dense layer of one
// model/mlp.py MLP back to back with the hidden activations in LDS (never in HBM); see
// fused_common.h for the per-layer scheme.  This stand-alone form backs dsat_op_mlp; the
// round kernels (kernels_round.hip) wrap the same layers in gather/PairNorm prologues.
#include "dsat_internal.h"
#include "fused_common.h"

namespace dsat {

// LDS is static: ROCm 7.2 rejects dynamic LDS above 64 KiB even with
// hipFuncAttributeMaxDynamicSharedMemorySize, while static declarations up to 160 KiB launch.
template <int RB, int NWAVES, int LDS_FLOATS>
__global__ __launch_bounds__(NWAVES * 64) void chain_kernel(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    constexpr int BM = 16 * RB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * BM;
    float* buf0 = smem;
    float* buf1 = smem + BM * a.ld_buf0;
    // X tile -> LDS (rows clamped: the padded rows are computed and discarded)
    {
        const int k4 = a.kpad0 / 4;
        for (int i = tid; i < BM * k4; i += NWAVES * 64) {
            const int row = i / k4, c4 = i % k4;
            int gr = r0 + row;
            if (gr >= a.rows) gr = a.rows - 1;
            *reinterpret_cast<float4*>(buf0 + row * a.ld_buf0 + c4 * 4) =
                *reinterpret_cast<const float4*>(a.x + (size_t)gr * a.kpad0 + c4 * 4);
        }
    }
    __syncthreads();
    float* in = buf0;
    float* out = buf1;
    int ld_in = a.ld_buf0, ld_out = a.ld_buf1;
    for (int l = 0; l < a.nlayers; ++l) {
        const ChainLayer& L = a.L[l];
        if (l + 1 < a.nlayers) {
            float* o = out;
            const int ld = ld_out;
            dense_layer<RB, NWAVES>(in, ld_in, L.kblocks, L.wp, L.bias, L.ntiles, L.tpw, 1, wave, lane,
                                    [=](int row, int col, float4 v) { *reinterpret_cast<float4*>(o + row * ld + col) = v; });
            __syncthreads();
            float* t = in; in = out; out = t;
            const int tl = ld_in; ld_in = ld_out; ld_out = tl;
        } else {
            float* y = a.y;
            const int ldy = a.ldy, ncols = a.ncols_y, rows = a.rows;
            dense_layer<RB, NWAVES>(in, ld_in, L.kblocks, L.wp, L.bias, L.ntiles, L.tpw, 0, wave, lane,
                                    [=](int row, int col, float4 v) {   // ncols / ldy are arbitrary: scalar stores
                                        if (r0 + row >= rows) return;
                                        float* yr = y + (size_t)(r0 + row) * ldy;
                                        if (col < ncols) yr[col] = v.x;
                                        if (col + 1 < ncols) yr[col + 1] = v.y;
                                        if (col + 2 < ncols) yr[col + 2] = v.z;
                                        if (col + 3 < ncols) yr[col + 3] = v.w;
                                    });
        }
    }
}

template <int RB, int NWAVES>
static hipError_t launch_chain_t(const ChainArgs& a, hipStream_t st) {
    constexpr int BM = 16 * RB;
    const size_t need = (size_t)BM * (a.ld_buf0 + a.ld_buf1);
    const dim3 grid((a.rows + BM - 1) / BM), block(NWAVES * 64);
    if (need <= 16384)
        hipLaunchKernelGGL((chain_kernel<RB, NWAVES, 16384>), grid, block, 0, st, a);
    else if (need <= 40960)
        hipLaunchKernelGGL((chain_kernel<RB, NWAVES, 40960>), grid, block, 0, st, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_chain(const ChainArgs& a, int rb, int nwaves, hipStream_t st) {
    if (a.rows <= 0) return hipSuccess;
    if (rb == 1 && nwaves == 4) return launch_chain_t<1, 4>(a, st);
    if (rb == 1 && nwaves == 8) return launch_chain_t<1, 8>(a, st);
    if (rb == 2 && nwaves == 4) return launch_chain_t<2, 4>(a, st);
    if (rb == 2 && nwaves == 8) return launch_chain_t<2, 8>(a, st);
    if (rb == 4 && nwaves == 8) return launch_chain_t<4, 8>(a, st);
    return hipErrorInvalidValue;
}

}  // namespace dsat
