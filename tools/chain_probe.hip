// Phase timing of a fused MLP chain (fused_common.h) with synthetic weights: per wave and
// layer, s_memtime stamps at layer entry (0), ring primed (4), main loop done (5), k-loop done
// (1), epilogue done (2), barrier passed (3).  Usage: chain_probe [rb] [blocks] dims...
#include <hip/hip_runtime.h>
#ifndef PNW
#define PNW 8
#endif
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
// stamps go to LDS (lgkmcnt), never to global memory: a global store would join the vmcnt
// queue that the weight ring counts by hand
extern __shared__ long long s_probe_dyn[];            // [PNW waves][PROBE_SLOTS records] (id, time) + counters
#define PROBE_SLOTS 32
#define DSAT_PROBE(id)                                                                     \
    do {                                                                                   \
        if ((threadIdx.x & 63) == 0) {                                                     \
            int* cnt = reinterpret_cast<int*>(s_probe_dyn + PNW * PROBE_SLOTS) + (threadIdx.x >> 6); \
            const int c = *cnt;                                                            \
            if (c < PROBE_SLOTS) s_probe_dyn[(threadIdx.x >> 6) * PROBE_SLOTS + c] = (clock64() << 3) | (id); \
            *cnt = c + 1;                                                                  \
        }                                                                                  \
    } while (0)
#include "../diffusionsat_amd/csrc/fused_common.h"
using namespace dsat;

struct Layer { const float4* wp; const float* bias; int kblocks, ntiles, tpw; };
struct Args { Layer L[4]; int nlayers; int ld0, ld1; float* y; long long* probe; int mix; };

template <int RB>
__global__ __launch_bounds__(PNW * 64) void kern(Args a) {
    __shared__ __attribute__((aligned(16))) float smem[39000];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int BM = 16 * RB;
    float* in = smem;
    float* out = smem + BM * a.ld0;
    if (tid < PNW) reinterpret_cast<int*>(s_probe_dyn + PNW * PROBE_SLOTS)[tid] = 0;
    for (int i = tid; i < BM * a.ld0; i += PNW * 64) in[i] = (float)(i % 13) * 0.01f;
    __syncthreads();
    int ld_in = a.ld0, ld_out = a.ld1;
#ifdef STATIC_CHAIN
    {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const StaticLayer S0{a.L[0].wp, a.L[0].bias, a.L[0].kblocks, a.L[0].ntiles};
        const StaticLayer S1{a.L[1].wp, a.L[1].bias, a.L[1].kblocks, a.L[1].ntiles};
        const StaticLayer S2{a.L[2].wp, a.L[2].bias, a.L[2].kblocks, a.L[2].ntiles};
        float* y = a.y;
        auto fin = [=](int row, int col, float4 v) { *reinterpret_cast<float4*>(y + ((size_t)blockIdx.x * BM + row) * 256 + (col & 255)) = v; };
        float* bufB = out + BM * a.ld1;
        if constexpr (RB == 4) {
            constexpr int T = PNW == 8 ? 2 : 1;
            if (a.mix) chain_static<4, PNW, T, T, 0, true>(S0, S1, S2, in, a.ld0, out, a.ld1, bufB, a.ld1, wv, lane, fin, false);
            else chain_static<4, PNW, T, T, 0>(S0, S1, S2, in, a.ld0, out, a.ld1, bufB, a.ld1, wv, lane, fin, false);
        } else if constexpr (RB == 1) {
            if constexpr (PNW == 8) {
                if (a.L[0].tpw == 4) chain_static<1, 8, 4, 4, 2>(S0, S1, S2, in, a.ld0, out, a.ld1, bufB, a.ld1, wv, lane, fin, false);
                else chain_static<1, 8, 2, 2, 1>(S0, S1, S2, in, a.ld0, out, a.ld1, bufB, a.ld1, wv, lane, fin, false);
            } else {
                if (a.L[0].tpw == 2) chain_static<1, 16, 2, 2, 1>(S0, S1, S2, in, a.ld0, out, a.ld1, bufB, a.ld1, wv, lane, fin, false);
                else chain_static<1, 16, 1, 1, 1>(S0, S1, S2, in, a.ld0, out, a.ld1, bufB, a.ld1, wv, lane, fin, false);
            }
        }
        DSAT_PROBE(2);
        __syncthreads();
        DSAT_PROBE(3);
    }
#else
    for (int l = 0; l < a.nlayers; ++l) {
        const Layer& L = a.L[l];
        float* o = out;
        const int ld = ld_out;
#ifdef SYNC_IN
        dense_layer<RB, PNW>(in, ld_in, L.kblocks, L.wp, L.bias, L.ntiles, L.tpw, 1, wave, lane,
                           [=](int row, int col, float4 v) { *reinterpret_cast<float4*>(o + row * ld + col) = v; }, l > 0);
        DSAT_PROBE(2);
#else
        dense_layer<RB, PNW>(in, ld_in, L.kblocks, L.wp, L.bias, L.ntiles, L.tpw, 1, wave, lane,
                           [=](int row, int col, float4 v) { *reinterpret_cast<float4*>(o + row * ld + col) = v; });
        DSAT_PROBE(2);
        __syncthreads();
#endif
        DSAT_PROBE(3);
        float* t = in; in = out; out = t;
        const int tl = ld_in; ld_in = ld_out; ld_out = tl;
    }
#endif
    __syncthreads();
    if (tid < 64 && blockIdx.x == 0) a.y[tid] += in[tid];
    __syncthreads();
    for (int i = tid; i < PNW * PROBE_SLOTS; i += PNW * 64) a.probe[(size_t)blockIdx.x * PNW * PROBE_SLOTS + i] = s_probe_dyn[i];
}

int main(int argc, char** argv) {
    const int rb = argc > 1 ? atoi(argv[1]) : 1;
    const int blocks = argc > 2 ? atoi(argv[2]) : 194;
    std::vector<int> dims;
    for (int i = 3; i < argc; ++i) dims.push_back(atoi(argv[i]));
    if (dims.size() < 2) dims = {137, 512, 512, 256};
    Args a{};
    a.nlayers = (int)dims.size() - 1;
    int maxw = 0;
#ifdef STATIC_CHAIN
    for (size_t i = 1; i + 1 < dims.size(); ++i) maxw = std::max(maxw, (dims[i] + 15) / 16 * 16);
#else
    for (size_t i = 1; i < dims.size(); ++i) maxw = std::max(maxw, (dims[i] + 15) / 16 * 16);
#endif
    a.ld0 = (dims[0] + 15) / 16 * 16 + 8;
    a.ld1 = maxw + 8;
    {
        const int bm = 16 * rb;
        const int need = bm * (a.ld0 + a.ld1 * (a.nlayers > 2 ? 2 : 1));
        if (need > 39000) { printf("LDS plan %d floats does not fit\n", need); return 1; }
    }
    for (int l = 0; l < a.nlayers; ++l) {
        const int kb = (dims[l] + 15) / 16, nt = (dims[l + 1] + 15) / 16;
        std::vector<float> w((size_t)nt * kb * 256), b(nt * 16, 0.01f);
        for (size_t i = 0; i < w.size(); ++i) w[i] = (float)((i * 2654435761u) >> 20) * 1e-5f - 0.02f;
        float4* wp; float* bias;
        (void)hipMalloc(&wp, w.size() * 4); (void)hipMalloc(&bias, b.size() * 4);
        (void)hipMemcpy(wp, w.data(), w.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(bias, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        a.L[l] = Layer{wp, bias, kb, nt, pick_tpw(nt, PNW, PNW == 16 ? 2 : max_tpw_for_rb(rb))};
    }
    (void)hipMalloc(&a.y, ((size_t)blocks * 16 * rb * 256 + 128) * 4);
    a.mix = getenv("MIX") != nullptr;
    long long* probe;
    const size_t np = (size_t)blocks * PNW * PROBE_SLOTS;
    (void)hipMalloc(&probe, np * 8);
    (void)hipMemset(probe, 0, np * 8);
    a.probe = probe;
    const size_t dyn = (PNW * PROBE_SLOTS + PNW) * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (rb == 1) hipLaunchKernelGGL(kern<1>, dim3(blocks), dim3(PNW * 64), dyn, 0, a);
        else if (rb == 2) hipLaunchKernelGGL(kern<2>, dim3(blocks), dim3(PNW * 64), dyn, 0, a);
        else hipLaunchKernelGGL(kern<4>, dim3(blocks), dim3(PNW * 64), dyn, 0, a);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> h(np);
    (void)hipMemcpy(h.data(), probe, np * 8, hipMemcpyDeviceToHost);
    printf("rb=%d blocks=%d kernel %.1f us (event)\n", rb, blocks, ms * 1e3);
    const char* nm[6] = {"prime ring", "main loop", "tail", "epilogue", "barrier wait", "layer total"};
    for (int l = 0; l < a.nlayers; ++l) {
        const int kb = a.L[l].kblocks, nt = a.L[l].ntiles, tpw = a.L[l].tpw;
        const int per_simd_tiles = ((nt + 8 * tpw - 1) / (8 * tpw)) * tpw * 2;   // busiest SIMD (2 waves)
        printf("layer %d: K=%d (%d kblocks) N=%d (%d tiles) tpw=%d  ideal MFMA cycles/SIMD ~%d\n", l, dims[l], kb,
               dims[l + 1], nt, tpw, kb * 4 * rb * std::min(per_simd_tiles, 2 * tpw * 1) * 32);
        std::vector<double> d[6];
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < PNW; ++w) {
                const long long* rec = &h[((size_t)b * PNW + w) * PROBE_SLOTS];
                // walk the records: a layer ends with id 3
                int li = 0;
                long long s[8] = {0};
                bool active = false;
                for (int i = 0; i < PROBE_SLOTS && rec[i]; ++i) {
                    const int id = (int)(rec[i] & 7);
                    s[id] = rec[i] >> 3;
                    if (id == 0) active = true;
                    if (id == 3) {
                        if (li == l && active) {
                            d[0].push_back((double)(s[4] - s[0])); d[1].push_back((double)(s[5] - s[4]));
                            d[2].push_back((double)(s[1] - s[5])); d[3].push_back((double)(s[2] - s[1]));
                            d[4].push_back((double)(s[3] - s[2])); d[5].push_back((double)(s[3] - s[0]));
                        }
                        ++li; active = false;
                    }
                }
            }
        for (int i = 0; i < 6; ++i) {
            if (d[i].empty()) continue;
            std::sort(d[i].begin(), d[i].end());
            printf("    %-13s median %7.0f  p90 %7.0f  max %7.0f\n", nm[i], d[i][d[i].size() / 2],
                   d[i][d[i].size() * 9 / 10], d[i].back());
        }
    }
    return 0;
}
