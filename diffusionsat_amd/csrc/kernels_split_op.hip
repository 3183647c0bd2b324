// Stand-alone form of the split-product MLP chain (split_bf16.h: fp32 products formed from three bf16 slices per
// operand on the bf16 matrix pipe): a workgroup owns 16*NRB rows, slices its input tile into LDS planes and runs
// split_chain -- the very function the round kernels' P3 and role-B tiles call -- on it.  Backs
// dsat_op_mlp_ex(impl = 2): the parity tests feed it arbitrary inputs and weights (values near the fp32 range's
// ends, denormals, infinities), which the round kernels' own producers never emit.
#include "dsat_internal.h"
#include "fused_common.h"
#include "split_bf16.h"

namespace dsat {

template <int NRB>
__global__ __launch_bounds__(1024) void split_op_kernel(SplitOpArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[ROUND_LDS_FLOATS * 4];
    constexpr int ROWS = 16 * NRB, NT = 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned short* xs = reinterpret_cast<unsigned short*>(smem);
    unsigned short* hA = xs + 3 * ROWS * a.ld_in;
    unsigned short* hB = a.nlayers > 2 ? hA + 3 * ROWS * a.ldh : hA;   // a 2-layer chain has one hidden tile
    const int r0 = blockIdx.x * ROWS;
    const float* x = a.x;
    const int kpad0 = a.kpad0, rows = a.rows, ld_in = a.ld_in;
    const int c4n = a.ks[0] * 8;                      // float4 groups per row up to the end of the last k-step
    auto prep = [&] {
        for (int i = tid; i < ROWS * c4n; i += NT) {
            const int row = i / c4n, c4 = i - row * c4n;
            const int gr = min(r0 + row, rows - 1);    // rows past the end are computed and discarded
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 * 4 < kpad0) v = *reinterpret_cast<const float4*>(x + (size_t)gr * kpad0 + c4 * 4);
            split_st4(xs, ROWS * ld_in, row * ld_in + c4 * 4, v);
        }
    };
    float* y = a.y;
    const int ldy = a.ldy, ncols = a.ncols_y;
    auto final_store = [=](int row, int col, float4 v) {
        if (r0 + row >= rows) return;
        float* yr = y + (size_t)(r0 + row) * ldy;
        if (col < ncols) yr[col] = v.x;
        if (col + 1 < ncols) yr[col + 1] = v.y;
        if (col + 2 < ncols) yr[col + 2] = v.z;
        if (col + 3 < ncols) yr[col + 3] = v.w;
    };
    SplitChain sc;
    sc.nlayers = a.nlayers;
#pragma unroll
    for (int l = 0; l < 3; ++l) sc.L[l] = SplitLayer{a.ws[l], a.bias[l], a.ks[l], a.nt[l]};
    split_chain<NRB, NT, (NRB > 1 ? 2 : 3)>(sc, xs, ld_in, hA, hB, a.ldh, wvu, lane, tid, final_store, prep);
}

hipError_t launch_split_op(const SplitOpArgs& a, int rb, hipStream_t st) {
    if (a.rows <= 0) return hipSuccess;
    const int rows_per = 16 * rb;
    const size_t need = (size_t)3 * rows_per * (a.ld_in + (a.nlayers > 2 ? 2 : 1) * (size_t)a.ldh) * 2;
    if (need > (size_t)ROUND_LDS_FLOATS * 4) return hipErrorInvalidValue;
    const dim3 grid((a.rows + rows_per - 1) / rows_per), block(1024);
    if (rb == 1) hipLaunchKernelGGL(split_op_kernel<1>, grid, block, 0, st, a);
    else if (rb == 2) hipLaunchKernelGGL(split_op_kernel<2>, grid, block, 0, st, a);
    else if (rb == 4) hipLaunchKernelGGL(split_op_kernel<4>, grid, block, 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace dsat
