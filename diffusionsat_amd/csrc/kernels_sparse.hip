// The two sparse half-rounds of QuerySAT.loop over the CSR clause-literal graph, gfx950.
//
// Layout: every state tensor is row-major [rows][128] fp32 = 512-B rows.  A half-wave
// (32 lanes x float4) owns one row, so every gathered neighbour row is one fully
// coalesced 512-B read (two half-waves = one 1-KiB wave-instruction).
//
//  literal->clause (clause_gather):   one half-wave per clause, walks c_lit[]
//     R3  closs = exp(-sum softplus(+-q[var]))      loss/sat.py:125-137
//     R5  closs *= 4                                model/query_sat.py:248
//     R7  cmsg  = (sum lits[lit]) * rsqrt(len)      model/query_sat.py:255-256
//  clause->literal (literal_gather):  one wave per variable (half 0 = +v, half 1 = -v)
//     R4  grad  = (-sig(q) M+ + sig(-q) M-) * 4 rsqrt(deg)   query_sat.py:227-246 closed form
//     R11 vloss = (sum vla[clause]) * rsqrt(deg_lit)         query_sat.py:269-273
// Sums run in CSR order with plain fp32 adds: deterministic, no atomics.
#include "dsat_internal.h"

namespace dsat {

__device__ __forceinline__ float softplus_f(float x) {
    // log(1+exp(x)) = max(x,0) + log1p(exp(-|x|))
    return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

constexpr int GATHER_UNROLL = 4;

__global__ __launch_bounds__(256) void clause_gather_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ c_lit,
    const float* __restrict__ rev_w, const float* __restrict__ q, const float* __restrict__ lits,
    float* __restrict__ closs_out, float closs_scale, float* __restrict__ cmsg_out,
    int N, int C, const int* __restrict__ done) {
    if (done != nullptr && *done != 0) return;
    const int sub = threadIdx.x & 31;
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (c >= C) return;
    const int e0 = rowptr[c], e1 = rowptr[c + 1];
    float4 S = make_float4(0.f, 0.f, 0.f, 0.f), Mg = S;
    for (int e = e0; e < e1; e += GATHER_UNROLL) {
        int l[GATHER_UNROLL];
#pragma unroll
        for (int j = 0; j < GATHER_UNROLL; ++j) l[j] = (e + j < e1) ? c_lit[e + j] : -1;
        float4 qv[GATHER_UNROLL], lv[GATHER_UNROLL];
#pragma unroll
        for (int j = 0; j < GATHER_UNROLL; ++j) {
            if (l[j] >= 0) {
                const int var = l[j] < N ? l[j] : l[j] - N;
                qv[j] = ld4(q + (size_t)var * F + sub * 4);
                lv[j] = ld4(lits + (size_t)l[j] * F + sub * 4);
            }
        }
#pragma unroll
        for (int j = 0; j < GATHER_UNROLL; ++j) {
            if (l[j] >= 0) {
                const float sg = l[j] < N ? 1.f : -1.f;
                S.x += softplus_f(sg * qv[j].x); S.y += softplus_f(sg * qv[j].y);
                S.z += softplus_f(sg * qv[j].z); S.w += softplus_f(sg * qv[j].w);
                Mg.x += lv[j].x; Mg.y += lv[j].y; Mg.z += lv[j].z; Mg.w += lv[j].w;
            }
        }
    }
    const float rw = rev_w[c];
    st4(closs_out + (size_t)c * F + sub * 4,
        make_float4(closs_scale * expf(-S.x), closs_scale * expf(-S.y),
                    closs_scale * expf(-S.z), closs_scale * expf(-S.w)));
    st4(cmsg_out + (size_t)c * F + sub * 4, make_float4(Mg.x * rw, Mg.y * rw, Mg.z * rw, Mg.w * rw));
}

__global__ __launch_bounds__(256) void literal_gather_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ l_clause,
    const float* __restrict__ deg_w, const float* __restrict__ vdeg_w,
    const float* __restrict__ q, const float* __restrict__ closs, float closs_scale,
    const float* __restrict__ vla, float* __restrict__ grad_out, float* __restrict__ vloss_out,
    int N, const int* __restrict__ done) {
    if (done != nullptr && *done != 0) return;
    const int lane = threadIdx.x & 63;
    const int sub = lane & 31, half = lane >> 5;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= N) return;  // whole wave exits together
    const int lit = v + half * N;
    const int e0 = rowptr[lit], e1 = rowptr[lit + 1];
    float4 M = make_float4(0.f, 0.f, 0.f, 0.f), VL = M;
    for (int e = e0; e < e1; e += GATHER_UNROLL) {
        int cidx[GATHER_UNROLL];
#pragma unroll
        for (int j = 0; j < GATHER_UNROLL; ++j) cidx[j] = (e + j < e1) ? l_clause[e + j] : -1;
        float4 cv[GATHER_UNROLL], vv[GATHER_UNROLL];
#pragma unroll
        for (int j = 0; j < GATHER_UNROLL; ++j) {
            if (cidx[j] >= 0) {
                cv[j] = ld4(closs + (size_t)cidx[j] * F + sub * 4);
                vv[j] = ld4(vla + (size_t)cidx[j] * F + sub * 4);
            }
        }
#pragma unroll
        for (int j = 0; j < GATHER_UNROLL; ++j) {
            if (cidx[j] >= 0) {
                M.x += cv[j].x; M.y += cv[j].y; M.z += cv[j].z; M.w += cv[j].w;
                VL.x += vv[j].x; VL.y += vv[j].y; VL.z += vv[j].z; VL.w += vv[j].w;
            }
        }
    }
    const float dw = deg_w[lit];
    st4(vloss_out + (size_t)lit * F + sub * 4, make_float4(VL.x * dw, VL.y * dw, VL.z * dw, VL.w * dw));
    // combine the +v and -v halves for the gradient (closs was stored scaled by 1/closs_scale')
    float4 Mo;
    Mo.x = __shfl_xor(M.x, 32); Mo.y = __shfl_xor(M.y, 32);
    Mo.z = __shfl_xor(M.z, 32); Mo.w = __shfl_xor(M.w, 32);
    if (half == 0) {
        const float4 qv = ld4(q + (size_t)v * F + sub * 4);
        const float vw = vdeg_w[v];
        float4 g;
        g.x = (-sigmoid_f(qv.x) * (M.x * closs_scale) + sigmoid_f(-qv.x) * (Mo.x * closs_scale)) * vw;
        g.y = (-sigmoid_f(qv.y) * (M.y * closs_scale) + sigmoid_f(-qv.y) * (Mo.y * closs_scale)) * vw;
        g.z = (-sigmoid_f(qv.z) * (M.z * closs_scale) + sigmoid_f(-qv.z) * (Mo.z * closs_scale)) * vw;
        g.w = (-sigmoid_f(qv.w) * (M.w * closs_scale) + sigmoid_f(-qv.w) * (Mo.w * closs_scale)) * vw;
        st4(grad_out + (size_t)v * F + sub * 4, g);
    }
}

void launch_clause_gather(const dsat_graph* g, const float* q, const float* lits, float* closs_out,
                          float closs_scale, float* cmsg_out, const int* done, hipStream_t st) {
    if (g->C == 0) return;
    hipLaunchKernelGGL(clause_gather_kernel, dim3((g->C + 7) / 8), dim3(256), 0, st, g->c_rowptr,
                       g->c_lit, g->rev_w, q, lits, closs_out, closs_scale, cmsg_out, g->N, g->C, done);
}

void launch_literal_gather(const dsat_graph* g, const float* q, const float* closs, float closs_scale,
                           const float* vla, float* grad_out, float* vloss_out, const int* done,
                           hipStream_t st) {
    if (g->N == 0) return;
    hipLaunchKernelGGL(literal_gather_kernel, dim3((g->N + 3) / 4), dim3(256), 0, st, g->l_rowptr,
                       g->l_clause, g->deg_w, g->vdeg_w, q, closs, closs_scale, vla, grad_out,
                       vloss_out, g->N, done);
}

}  // namespace dsat
