// Fused message-passing round of QuerySAT.loop (model/query_sat.py:225-348): three row-tile
// kernels per round plus one select/SAT kernel, instead of one launch per Dense layer /
// sparse product / normalisation.  Every kernel keeps its tile's activations in LDS and
// streams the weights through MFMA fragments (fused_common.h).
//
//   P1  variable tile (16 rows):  R13 (previous round's PairNorm, finished here from the
//       per-tile column sums P3 left), R14 (previous round's logits), then R1-R2 and R6:
//       query = variables_query(v1), literals = lit_mlp(v1); also softplus(+-query) for R3.
//   P2  clause tile (16*RB rows): R10 (previous round's clause PairNorm), R3+R5+R7 (CSR
//       gather of softplus / literal rows), R8-R9 clause_mlp, column sums of the new state.
//   P3  variable tile (16 rows):  R4+R11 (CSR gather of clause loss / clause messages),
//       R12 update_gate, column sums of the new state.
//   S   one workgroup per graph:  R15-R17 map selection + is_batch_sat / per-graph SAT.
//
// PairNorm's per-graph mean crosses tiles.  Each producer tile writes the column sums of its
// rows per graph it touches ("segments", ordered by row, so a graph's segments are contiguous);
// the consumer adds a graph's segments in index order: deterministic, no atomics.
#include "dsat_internal.h"
#include "fused_common.h"
#include "split_bf16.h"
#ifndef DSAT_SPLIT_PD4
#define DSAT_SPLIT_PD4 2   // ... in the 64-row split tile (four row blocks per fragment)
#endif
#ifndef DSAT_SPLIT_PD1
#define DSAT_SPLIT_PD1 3   // weight k-steps in flight per wave in the 16-row split chains
#endif
#ifndef DSAT_A1_ONE_RING
#define DSAT_A1_ONE_RING 1   // role A's lit_mlp passes share one weight ring (split_vpass); 0: a ring per pass (round 2)
#endif
#ifndef DSAT_A1_EARLY
#define DSAT_A1_EARLY 1      // role_a1s asks for its biases before the PairNorm and primes the ring before v1 is sliced
#endif
#ifndef DSAT_SPLIT_PD_A1
#define DSAT_SPLIT_PD_A1 2   // ... in role A's 16-row lit_mlp (one ring across the passes: 2 / 3 / 4 -> 376.5 / 375.1 / 375.0 steps/s)
#endif
#include "philox.h"

namespace dsat {

namespace {

constexpr int NT = ROUND_NWAVES * 64;

// Phase stamps for the diagnostic build of a session (never read by any kernel).
// -DDSAT_STAMP_ROTATE (diagnostic builds): workgroup b's stamps are taken by wave b mod 16 instead of wave 0, so a
// launch shows how the waves of a workgroup differ (tools/stamps_waves.py)
#ifdef DSAT_STAMP_ROTATE
#define DSAT_STAMP_THREAD ((blockIdx.x & 15u) * 64u)
#else
#define DSAT_STAMP_THREAD 0u
#endif
#define DSAT_STAMP(a, k)                                                                  \
    do {                                                                                  \
        if ((a).stamps != nullptr && threadIdx.x == DSAT_STAMP_THREAD && blockIdx.x < 4096) \
            (a).stamps[blockIdx.x * 8 + (k)] = clock64();                                   \
    } while (0)
// Slot 7 of a tile's stamps: the 100 MHz real-time counter's ticks between the tile's first and last
// stamp, so that tools/stamps.py can turn shader-clock cycles into time and report the clock the
// kernel actually ran at (MI355X_MICROARCH.md, DVFS give-back item 6).
#define DSAT_STAMP_RT_BEGIN(a) \
    const unsigned long long _rt0 = ((a).stamps != nullptr && threadIdx.x == DSAT_STAMP_THREAD) ? __builtin_amdgcn_s_memrealtime() : 0ull
#define DSAT_STAMP_RT_END(a)                                                              \
    do {                                                                                  \
        if ((a).stamps != nullptr && threadIdx.x == DSAT_STAMP_THREAD && blockIdx.x < 4096) \
            (a).stamps[blockIdx.x * 8 + 7] = (long long)(__builtin_amdgcn_s_memrealtime() - _rt0); \
    } while (0)
// Epilogue addresses are per-lane 64-bit values; computed from loop-invariant inputs, the compiler
// hoists them above the whole MLP chain, where they sit in VGPR pairs for tens of thousands of cycles
// and push a 1024-thread workgroup (128 VGPRs per lane) into scratch spills.  An empty asm on the row
// index pins the address arithmetic to the epilogue.
#define NO_HOIST(x) asm volatile("" : "+v"(x))
constexpr int LDS_FLOATS = ROUND_LDS_FLOATS;   // static (see kernels_chain.hip on dynamic LDS)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Store of a kernel OUTPUT row piece (read by the next kernel, never again by this one).
// DSAT_OUT_POLICY: 0 = plain store (the line stays dirty in this XCD's L2 until the end-of-kernel
// write-back), 1 = write-through `sc0 sc1` (nothing left to write back at the kernel boundary; the
// line does not stay in L2), 2 = `nt`.  Whole denoising step, config 2: 295.2 / 300.9 / 283.6 steps/s
// (every kernel boundary is ~1 us shorter with write-through; consumers sit on other CUs anyway).
#ifndef DSAT_OUT_POLICY
#define DSAT_OUT_POLICY 1
#endif
__device__ __forceinline__ void stg4(float* p, float4 v) {
#if DSAT_OUT_POLICY == 0
    *reinterpret_cast<float4*>(p) = v;
#else
    f32x4 x = {v.x, v.y, v.z, v.w};
#if DSAT_OUT_POLICY == 1
    // s_nop: a store wider than 64 bits still reads its data VGPRs in the cycle after issue; the
    // compiler pads that hazard for its own stores but cannot see inside an asm statement
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
#endif
#endif
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
// sigmoid(x) and sigmoid(-x) from one exponential: z = exp(-|x|) in (0,1], so nothing overflows
__device__ __forceinline__ void sigmoid_pair(float x, float& s, float& sn) {
    const float z = expf(-fabsf(x));
    const float big = __builtin_amdgcn_rcpf(1.f + z), small = z * big;
    s = x >= 0.f ? big : small;
    sn = x >= 0.f ? small : big;
}
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

// [normal(4) | noisy_num(2) | noise_scale | denoised(2) | 0 ...] = the non-state columns of v1
// (:214-219, :239).  aux_kernel writes them for every round once per denoising step (the
// Philox + Box-Muller draw costs ~10^3 instructions, too many for a tile prologue that a
// whole workgroup waits on); the tiles only copy their 64-byte rows.
__device__ __forceinline__ void load_aux_row(float* dst, const float* __restrict__ aux, int n, int N, int round,
                                             bool valid) {
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, v2 = v0, v3 = v0;
    if (valid) {
        const float* src = aux + ((size_t)round * N + n) * AUX_W;
        v0 = ld4(src); v1 = ld4(src + 4); v2 = ld4(src + 8); v3 = ld4(src + 12);
    }
    st4(dst, v0); st4(dst + 4, v1); st4(dst + 8, v2); st4(dst + 12, v3);
}

// Per-tile scalars precomputed on the host (one 64-B record, one load): removes the dependent
// gid -> seg -> gseg -> rowptr chains from the kernel prologues.
struct TileMetaDev {
    int g_first, g_last;      // graphs with rows in this tile
    int seg_base;             // index of this tile's first PairNorm segment
    int s0, s1, gs0, gs1;     // rows [s0,s1) and segments [gs0,gs1) of graph g_first
    int e_base0, e_end0;      // CSR entry range of the tile's rows (clause rows / +literal rows)
    int e_base1, e_end1;      // CSR entry range of the tile's -literal rows (variable tiles)
    int pad[5];
};
// Workgroups are dealt round-robin over the 8 XCDs (observed, speed only); this bijection gives
// each XCD one contiguous chunk of tiles.  Graphs are contiguous row ranges on both sides of the
// bipartite graph, so the variable tiles and the clause tiles of one formula then run on the same
// XCD in P1, P2 and P3, and the rows one kernel gathers were written into that XCD's L2 by the
// previous kernel (cdna_hip_programming.md T1).  Placement only affects speed.
__device__ __forceinline__ int xcd_tile(int b, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// What a block of P3's grid does: tile block `idx` (to be mapped by xcd_tile; >= p3_tiles: nothing), or role-C block
// `idx` (clause_norm_block; < 0: nothing).  The grid is rows of 8 blocks (one per XCD under round-robin placement), so
// a tile keeps the XCD of its index whether or not role-C rows sit between the tile rows.
struct P3Slot { bool is_c; int idx; };
__device__ __forceinline__ P3Slot p3_slot(const RoundArgs& a) {
    const int b = (int)blockIdx.x;
#ifdef DSAT_P3_NO_ROLEC   // experiment switch: P3's grid holds tiles only (what role C's dispatch costs the tiles)
    return P3Slot{false, b};
#endif
    if (a.cn_period <= 0) return b >= a.p3_tiles ? P3Slot{true, b - a.p3_tiles} : P3Slot{false, b};
    const int j = b >> 3, x = b & 7, cyc = a.cn_period + 1, c_rows = (a.cn_blocks + 7) >> 3;
    const int i = j / cyc;
    if (j - i * cyc == a.cn_period && i < c_rows) {
        const int c = i * 8 + x;
        return P3Slot{true, c < a.cn_blocks ? c : -1};
    }
    return P3Slot{false, (j - min(c_rows, i)) * 8 + x};
}

__device__ __forceinline__ TileMetaDev load_meta(const int32_t* meta, int tile) {
    TileMetaDev m;
    const int4* p = reinterpret_cast<const int4*>(meta + (size_t)tile * TILE_META_INTS);
    const int4 a = p[0], b = p[1], c = p[2];
    m.g_first = a.x; m.g_last = a.y; m.seg_base = a.z; m.s0 = a.w;
    m.s1 = b.x; m.gs0 = b.y; m.gs1 = b.z; m.e_base0 = b.w;
    m.e_end0 = c.x; m.e_base1 = c.y; m.e_end1 = c.z;
    return m;
}

// Keeps the compiler from sinking the tile-record load below the early-exit branch: both
// scalar loads are in flight when the flag is tested.
__device__ __forceinline__ int both_arrived(int done, const TileMetaDev& m) {
    int d = done, g = m.g_first;
    asm volatile("" : "+s"(d), "+s"(g));
    return d;
}

// state = 0.25*PairNorm(x) + 0.1*state for the rows of one tile (layers/normalization.py:43-71,
// query_sat.py:265-266 / :279-280), then the :347-348 mix for the copy that enters the next
// round.  sState[.][0..127] receives the carried state, sLogit (optional) the pre-mix value.
// Row r of the tile belongs to half-wave (r mod 16); its x / state rows are requested before
// the per-graph mean is known, so the kernel pays one memory latency, not one per level.
template <int BM>
__device__ __forceinline__ void pairnorm_tile(int tile_lo, int rows, const TileMetaDev& tm,
                                              const int32_t* __restrict__ seg, const int32_t* __restrict__ gseg,
                                              const float* __restrict__ psum, const int32_t* __restrict__ gseg2,
                                              const float* __restrict__ psum2, const float* __restrict__ x,
                                              const float* __restrict__ state_in, float* __restrict__ state_out,
                                              bool first, float* sState, int ldS,
                                              float* sLogit, int ldL, float* sMean, float* sPart, int ldP, int tid,
                                              float inv_f = 1.f / F, int fm = F, unsigned short* pLogit = nullptr,
                                              int ldPL = 0) {
    // pLogit (role B of P1 with variables_output as split products): the pre-mix value goes out as three bf16
    // planes [3][BM][ldPL] (split_bf16.h) instead of the fp32 tile sLogit
    // inv_f = 1 / feature_maps, fm = feature_maps: narrower states live zero-padded in the 128-wide rows
    // state_in: the carried state of the previous round; state_out: where this round's goes (nullptr:
    // another workgroup normalises the same rows and owns the write).  sMean: F floats; sPart: NT/F rows
    // of F floats, row stride ldP (scratch for the partial sums of the thread groups).
    constexpr int ROWS = (BM + NT / 32 - 1) / (NT / 32);
    const int tile_hi = min(tile_lo + BM, rows);
    const int sub = tid & 31, hw = tid >> 5;
    // padding rows of the last tile: finite zeros for the MFMAs
    for (int r = tile_hi - tile_lo + hw; r < BM; r += NT / 32) {
        st4(sState + r * ldS + sub * 4, make_float4(0.f, 0.f, 0.f, 0.f));
        if (sLogit != nullptr) st4(sLogit + r * ldL + sub * 4, make_float4(0.f, 0.f, 0.f, 0.f));
        if (pLogit != nullptr) split_zero4(pLogit, BM * ldPL, r * ldPL + sub * 4);
    }
    if (first) {   // query_sat.py:141, :148: both states start as ones
        const float4 one = make_float4(sub * 4 + 0 < fm ? 1.f : 0.f, sub * 4 + 1 < fm ? 1.f : 0.f,
                                       sub * 4 + 2 < fm ? 1.f : 0.f, sub * 4 + 3 < fm ? 1.f : 0.f);
        for (int r = tile_lo + hw; r < tile_hi; r += NT / 32) {
            st4(sState + (r - tile_lo) * ldS + sub * 4, one);
            if (state_out != nullptr) stg4(state_out + (size_t)r * F + sub * 4, one);
        }
        __syncthreads();
        return;
    }
    float4 xv[ROWS], ov[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int r = tile_lo + hw + k * (NT / 32);
        if (r < tile_hi) {
            xv[k] = ld4(x + (size_t)r * F + sub * 4);
            ov[k] = ld4(state_in + (size_t)r * F + sub * 4);
        }
    }
    for (int g = tm.g_first; g <= tm.g_last; ++g) {
        int s0, s1, gs0, gs1;
        if (g == tm.g_first) { s0 = tm.s0; s1 = tm.s1; gs0 = tm.gs0; gs1 = tm.gs1; }
        else { s0 = seg[g]; s1 = seg[g + 1]; gs0 = gseg[g]; gs1 = gseg[g + 1]; }
        const int lo = max(s0, tile_lo), hi = min(s1, tile_hi);
        if (lo >= hi) continue;   // wave-uniform
        {   // mean_g: the graph's per-tile column sums, NT/F of them requested at once (one memory round
            // trip for the usual handful of segments), combined in a fixed order
            constexpr int NG = NT / F;
            const float* ps = psum;
            if (gseg2 != nullptr) {   // many tiles: the segments were pre-summed in chunks (segsum2_kernel)
                const int b0 = gseg2[g], b1 = gseg2[g + 1];
                if (b1 > b0) { ps = psum2; gs0 = b0; gs1 = b1; }
            }
            const int c = tid & (F - 1), grp = tid / F;
            float m = 0.f;
            for (int sgi = gs0 + grp; sgi < gs1; sgi += NG) m += ps[(size_t)sgi * F + c];
            sPart[grp * ldP + c] = m;
            __syncthreads();
            if (tid < F) {
                float t = sPart[tid];
#pragma unroll
                for (int k = 1; k < NG; ++k) t += sPart[k * ldP + tid];
                sMean[tid] = t * (1.f / (float)(s1 - s0));
            }
        }
        __syncthreads();
        const float4 mean = ld4(sMean + sub * 4);
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const int r = tile_lo + hw + k * (NT / 32);
            if (r >= lo && r < hi) {
                float4 v = xv[k];
                v.x -= mean.x; v.y -= mean.y; v.z -= mean.z; v.w -= mean.w;
                float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
                const float rs = 1.f / sqrtf(ss * inv_f + 1e-6f);
                const float4 o4 = ov[k];
                float4 res;
                res.x = v.x * rs * 0.25f + 0.1f * o4.x; res.y = v.y * rs * 0.25f + 0.1f * o4.y;
                res.z = v.z * rs * 0.25f + 0.1f * o4.z; res.w = v.w * rs * 0.25f + 0.1f * o4.w;
                if (sLogit != nullptr) st4(sLogit + (r - tile_lo) * ldL + sub * 4, res);
                if (pLogit != nullptr) split_st4(pLogit, BM * ldPL, (r - tile_lo) * ldPL + sub * 4, res);
                res.x = __fadd_rn(__fmul_rn(res.x, 0.2f), __fmul_rn(res.x, 0.8f));
                res.y = __fadd_rn(__fmul_rn(res.y, 0.2f), __fmul_rn(res.y, 0.8f));
                res.z = __fadd_rn(__fmul_rn(res.z, 0.2f), __fmul_rn(res.z, 0.8f));
                res.w = __fadd_rn(__fmul_rn(res.w, 0.2f), __fmul_rn(res.w, 0.8f));
                st4(sState + (r - tile_lo) * ldS + sub * 4, res);
                if (state_out != nullptr) stg4(state_out + (size_t)r * F + sub * 4, res);
            }
        }
        __syncthreads();
    }
}

// Column sums of the tile's new rows (staged in LDS, ld = ldT), one segment per graph touched.
// Its barriers order LDS only: the tile's write-through output stores are still being acknowledged when the
// sums start (a __syncthreads() here would wait ~6 k cycles for them).
// Four thread groups each add a quarter of the segment's rows; the quarters are combined in
// a fixed order.  sPart: NT floats of LDS.
template <int BM>
__device__ __forceinline__ void tile_colsums(int tile_lo, int rows, const TileMetaDev& tm,
                                             const int32_t* __restrict__ seg, float* __restrict__ psum,
                                             const float* sT, int ldT, float* sPart, int tid, int fm = F) {
    // fm = feature_maps: the staged tile only has those columns (the chain has no column tile beyond them)
    const int tile_hi = min(tile_lo + BM, rows);
    if (tile_hi <= tile_lo) return;
    const int c = tid & (F - 1), grp = tid / F;
    constexpr int NG = NT / F;
    int ord = 0;
    for (int g = tm.g_first; g <= tm.g_last; ++g) {
        int s0, s1;
        if (g == tm.g_first) { s0 = tm.s0; s1 = tm.s1; } else { s0 = seg[g]; s1 = seg[g + 1]; }
        const int lo = max(s0, tile_lo), hi = min(s1, tile_hi);
        if (lo >= hi) continue;
        const int len = hi - lo, per = (len + NG - 1) / NG;
        const int a0 = lo + grp * per, a1 = min(a0 + per, hi);
        float sacc = 0.f;
        if (c < fm)
            for (int r = a0; r < a1; ++r) sacc += sT[(r - tile_lo) * ldT + c];
        sPart[tid] = sacc;
        lds_barrier();
        if (tid < F) {
            float t = sPart[tid];
#pragma unroll
            for (int k = 1; k < NG; ++k) t += sPart[k * F + tid];
            psum[(size_t)(tm.seg_base + ord) * F + tid] = t;
        }
        lds_barrier();
        ++ord;
    }
}

// Runs the layers of one MLP on the tile: input in LDS, hidden activations ping-pong between
// bufA/bufB, the last layer goes to `final_store(row, col, v)`.  Ends with a barrier.
// `prep()` is the caller's last piece of input-tile preparation; the chain runs it after asking
// for its first weights, then takes the barrier that publishes the tile (the caller must not).
template <int RB, typename Store, typename Prep>
__device__ __forceinline__ void run_chain(const ChainSpec& m, const float* sIn, int ldin, float* bufA, int ldA,
                                          float* bufB, int ldB, int wave, int lane, Store&& final_store,
                                          Prep&& prep) {
    if (m.static_id != CHAIN_GENERIC) {   // blocking known at compile time: cross-layer weight prefetch
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const StaticLayer S0{m.L[0].wp, m.L[0].bias, m.L[0].kblocks, m.L[0].ntiles};
        const StaticLayer S1{m.L[1].wp, m.L[1].bias, m.L[1].kblocks, m.L[1].ntiles};
        const StaticLayer S2{m.L[2].wp, m.L[2].bias, m.L[2].kblocks, m.L[2].ntiles};
        if constexpr (RB == 1 && ROUND_NWAVES == 8) {
            if (m.static_id == CHAIN_442)
                chain_static<1, 8, 4, 4, 2>(S0, S1, S2, sIn, ldin, bufA, ldA, bufB, ldB, wv, lane, final_store, true, prep);
            else
                chain_static<1, 8, 2, 2, 1>(S0, S1, S2, sIn, ldin, bufA, ldA, bufB, ldB, wv, lane, final_store, true, prep);
        } else if constexpr (RB == 1 && ROUND_NWAVES == 16) {
            if (m.static_id == CHAIN_221)
                chain_static<1, 16, 2, 2, 1>(S0, S1, S2, sIn, ldin, bufA, ldA, bufB, ldB, wv, lane, final_store, true, prep);
            else
                chain_static<1, 16, 1, 1, 1>(S0, S1, S2, sIn, ldin, bufA, ldA, bufB, ldB, wv, lane, final_store, true, prep);
        } else if constexpr (RB == 4 && ROUND_NWAVES == 8) {
            if (m.static_id == CHAIN_22M)
                chain_static<4, 8, 2, 2, 0, true>(S0, S1, S2, sIn, ldin, bufA, ldA, bufB, ldB, wv, lane, final_store, true, prep);
            else
                chain_static<4, 8, 2, 2, 0>(S0, S1, S2, sIn, ldin, bufA, ldA, bufB, ldB, wv, lane, final_store, true, prep);
        } else if constexpr (RB == 4 && ROUND_NWAVES == 16) {
            if (m.static_id == CHAIN_11M)
                chain_static<4, 16, 1, 1, 0, true>(S0, S1, S2, sIn, ldin, bufA, ldA, bufB, ldB, wv, lane, final_store, true, prep);
            else
                chain_static<4, 16, 1, 1, 0>(S0, S1, S2, sIn, ldin, bufA, ldA, bufB, ldB, wv, lane, final_store, true, prep);
        }
        lds_barrier();   // LDS only: the outputs' write-through stores need not be acknowledged here
        return;
    }
    prep();
    __syncthreads();
    const float* in = sIn;
    int ld_in = ldin;
    for (int l = 0; l < m.nlayers; ++l) {
        const ChainLayer& L = m.L[l];
        // the barrier that publishes layer l-1's output is taken inside layer l, after its
        // first weight fragments have been requested (fused_common.h, stream_gemm sync_in)
        const bool sync_in = l > 0;
        if (l + 1 < m.nlayers) {
            float* o = (l & 1) ? bufB : bufA;
            const int ldo = (l & 1) ? ldB : ldA;
            dense_layer<RB, ROUND_NWAVES, ROUND_TPW_CAP>(in, ld_in, L.kblocks, L.wp, L.bias, L.ntiles, L.tpw, 1, wave, lane,
                                          [=](int row, int col, float4 v) { st4(o + row * ldo + col, v); }, sync_in);
            in = o;
            ld_in = ldo;
        } else {
            dense_layer<RB, ROUND_NWAVES, ROUND_TPW_CAP>(in, ld_in, L.kblocks, L.wp, L.bias, L.ntiles, L.tpw, 0, wave, lane,
                                          final_store, sync_in);
        }
    }
    __syncthreads();
}

template <int RB, typename Store>
__device__ __forceinline__ void run_chain(const ChainSpec& m, const float* sIn, int ldin, float* bufA, int ldA,
                                          float* bufB, int ldB, int wave, int lane, Store&& final_store) {
    run_chain<RB>(m, sIn, ldin, bufA, ldA, bufB, ldB, wave, lane, final_store, [] {});
}

// Two independent 2-layer chains side by side: waves [0, NW/2) run chain A, the others chain B
// (each on its own hidden buffer), sharing the two layer barriers.  Small chains cannot keep
// eight waves busy (variables_output has 8 + 2 column tiles), so running them one after the
// other pays their fixed per-layer latency twice.
template <typename StoreA, typename StoreB>
__device__ __forceinline__ void run_chain_pair(const ChainSpec& ma, const float* sInA, int ldinA, float* bufA, int ldA,
                                               const ChainSpec& mb, const float* sInB, int ldinB, float* bufB, int ldB,
                                               int wave, int lane, StoreA&& storeA, StoreB&& storeB) {
    constexpr int H = ROUND_NWAVES / 2;
    const int wv = __builtin_amdgcn_readfirstlane(wave);   // wave-uniform for the compiler: the
    constexpr int WANT_A = ROUND_NWAVES == 8 ? CHAIN_H21 : CHAIN_H11;   // variables_output on H waves
    constexpr int WANT_B = ROUND_NWAVES == 8 ? CHAIN_H32 : CHAIN_H21;   // variables_query on H waves
    if (ma.static_half_id == WANT_A && mb.static_half_id == WANT_B) {
        const StaticLayer none{nullptr, nullptr, 0, 0};
        if (wv < H) {
            const StaticLayer S0{ma.L[0].wp, ma.L[0].bias, ma.L[0].kblocks, ma.L[0].ntiles};
            const StaticLayer S1{ma.L[1].wp, ma.L[1].bias, ma.L[1].kblocks, ma.L[1].ntiles};
            if constexpr (ROUND_NWAVES == 8)
                chain_static<1, H, 2, 1, 0>(S0, S1, none, sInA, ldinA, bufA, ldA, nullptr, 0, wv, lane, storeA, false);
            else
                chain_static<1, H, 1, 1, 0>(S0, S1, none, sInA, ldinA, bufA, ldA, nullptr, 0, wv, lane, storeA, false);
        } else {
            const StaticLayer S0{mb.L[0].wp, mb.L[0].bias, mb.L[0].kblocks, mb.L[0].ntiles};
            const StaticLayer S1{mb.L[1].wp, mb.L[1].bias, mb.L[1].kblocks, mb.L[1].ntiles};
            if constexpr (ROUND_NWAVES == 8)
                chain_static<1, H, 3, 2, 0>(S0, S1, none, sInB, ldinB, bufB, ldB, nullptr, 0, wv - H, lane, storeB, false);
            else
                chain_static<1, H, 2, 1, 0>(S0, S1, none, sInB, ldinB, bufB, ldB, nullptr, 0, wv - H, lane, storeB, false);
        }
        __syncthreads();
        return;
    }
    if (wv < H) {                                          // weight base must stay in SGPRs
        dense_layer<1, H, ROUND_TPW_CAP>(sInA, ldinA, ma.L[0].kblocks, ma.L[0].wp, ma.L[0].bias, ma.L[0].ntiles, ma.L[0].tpw_half, 1,
                          wv, lane, [=](int row, int col, float4 v) { st4(bufA + row * ldA + col, v); });
        dense_layer<1, H, ROUND_TPW_CAP>(bufA, ldA, ma.L[1].kblocks, ma.L[1].wp, ma.L[1].bias, ma.L[1].ntiles, ma.L[1].tpw_half, 0,
                          wv, lane, storeA, true);
        __syncthreads();
    } else {
        dense_layer<1, H, ROUND_TPW_CAP>(sInB, ldinB, mb.L[0].kblocks, mb.L[0].wp, mb.L[0].bias, mb.L[0].ntiles, mb.L[0].tpw_half, 1,
                          wv - H, lane, [=](int row, int col, float4 v) { st4(bufB + row * ldB + col, v); });
        dense_layer<1, H, ROUND_TPW_CAP>(bufB, ldB, mb.L[1].kblocks, mb.L[1].wp, mb.L[1].bias, mb.L[1].ntiles, mb.L[1].tpw_half, 0,
                          wv - H, lane, storeB, true);
        __syncthreads();
    }
}

// CSR entries of the tile's rows -> LDS (entry ranges of consecutive rows are contiguous):
// sPtr[i] = rowptr[row_lo + i] - e_base for i = 0..nrows, sIdx[k] = idx[e_base + k].
// Returns false (nothing staged) when the range does not fit `cap` entries.
__device__ __forceinline__ bool stage_csr(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx,
                                          int row_lo, int nrows, int e_base, int e_end, int* sPtr, int* sIdx,
                                          int cap, int tid) {
    if (e_end - e_base > cap) return false;
    if (tid <= nrows) sPtr[tid] = rowptr[row_lo + tid] - e_base;
    for (int k = tid; k < e_end - e_base; k += NT) sIdx[k] = idx[e_base + k];
    return true;
}

// A tile's list block (dsat_api.hip, build_list_blocks): fetched by the whole workgroup at an address that only
// depends on the tile index -- in the same memory round trip as the tile record -- and written to LDS once the
// workgroup knows it has work to do.
struct ListBlock { int r[LIST_BLOCK_LOADS]; };
__device__ __forceinline__ void list_block_fetch(ListBlock& lb, const int32_t* __restrict__ table, int stride, int tile, int tid) {
#pragma unroll
    for (int i = 0; i < LIST_BLOCK_LOADS; ++i) {
        const int k = tid + i * NT;
        lb.r[i] = (table != nullptr && k < stride) ? table[(size_t)tile * stride + k] : 0;
    }
}
__device__ __forceinline__ void list_block_store(const ListBlock& lb, int* dst, int stride, int tid) {
#pragma unroll
    for (int i = 0; i < LIST_BLOCK_LOADS; ++i) {
        const int k = tid + i * NT;
        if (k < stride) dst[k] = lb.r[i];
    }
}

// R15-R17 for one graph (forward declaration; defined with the stand-alone kernel below)
template <int NTH, bool UNSUP = false, bool SHARE = false>
__device__ __forceinline__ void select_sat_body(int blk, int nblocks, const int32_t* __restrict__ sel_tab,
                                const int32_t* __restrict__ vseg,
                                const int32_t* __restrict__ cseg, const int32_t* __restrict__ rowptr,
                                const int32_t* __restrict__ c_lit, const float* __restrict__ logits,
                                Ctl* __restrict__ ctl, float* __restrict__ pgl, int32_t* __restrict__ best,
                                float* __restrict__ out_logits, uint8_t* __restrict__ assign,
                                int32_t* __restrict__ graph_unsat, int N, int round, int set_done, float* sred,
                                int tid = (int)threadIdx.x, bool active = true,
                                unsigned long long* sel_exch = nullptr, int sel_ticks = 0);

// R10 for a contiguous range of clause rows (one of `nb` extra workgroups of P1): same arithmetic
// as pairnorm_tile, streamed; clause_state is updated in place for P2 to load.
__device__ __forceinline__ void clause_norm_block(const RoundArgs& a, int b, int nb, float* sMean, int tid) {
    const int C = a.cn_rows > 0 ? min(a.cn_rows, a.C) : a.C;   // its share of the rows role C covers in this launch
    const int per = (C + nb - 1) / nb;
    const int lo0 = b * per, hi0 = min(lo0 + per, C);
    if (lo0 >= hi0) return;
    const int sub = tid & 31, hw = tid >> 5;
    for (int g = a.cl_gid[lo0]; g <= a.cl_gid[hi0 - 1]; ++g) {
        const int s0 = a.cl_seg[g], s1 = a.cl_seg[g + 1];
        const int lo = max(s0, lo0), hi = min(s1, hi0);
        if (lo >= hi) continue;
        {
            constexpr int NG = NT / F;
            const float* ps = a.psum_c_rd;
            int gs0 = a.cgseg[g], gs1 = a.cgseg[g + 1];
            if (a.cgseg2 != nullptr) {
                const int b0 = a.cgseg2[g], b1 = a.cgseg2[g + 1];
                if (b1 > b0) { ps = a.psum2_c; gs0 = b0; gs1 = b1; }
            }
            const int c = tid & (F - 1), grp = tid / F;
            float m = 0.f;
            for (int sgi = gs0 + grp; sgi < gs1; sgi += NG) m += ps[(size_t)sgi * F + c];
            sMean[F + tid] = m;
            __syncthreads();
            if (tid < F) {
                float t = sMean[F + tid];
#pragma unroll
                for (int k = 1; k < NG; ++k) t += sMean[F + k * F + tid];
                sMean[tid] = t * (1.f / (float)(s1 - s0));
            }
        }
        __syncthreads();
        const float4 mean = ld4(sMean + sub * 4);
        // four rows per half-wave in flight: the loop is a chain of dependent round trips otherwise
        constexpr int UR = 4, STEP = NT / 32;
        for (int r0 = lo + hw; r0 < hi; r0 += UR * STEP) {
            float4 v[UR], o4[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int r = r0 + u * STEP;
                if (r < hi) {
                    v[u] = ld4(a.ncv + (size_t)r * F + sub * 4);
                    o4[u] = ld4(a.Cs + (size_t)r * F + sub * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int r = r0 + u * STEP;
                if (r < hi) {   // uniform over the half-wave
                    float4 x = v[u];
                    x.x -= mean.x; x.y -= mean.y; x.z -= mean.z; x.w -= mean.w;
                    float ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
#pragma unroll
                    for (int o = 16; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
                    const float rs = 1.f / sqrtf(ss * a.inv_f + 1e-6f);
                    float4 res;
                    res.x = x.x * rs * 0.25f + 0.1f * o4[u].x; res.y = x.y * rs * 0.25f + 0.1f * o4[u].y;
                    res.z = x.z * rs * 0.25f + 0.1f * o4[u].z; res.w = x.w * rs * 0.25f + 0.1f * o4[u].w;
                    res.x = __fadd_rn(__fmul_rn(res.x, 0.2f), __fmul_rn(res.x, 0.8f));
                    res.y = __fadd_rn(__fmul_rn(res.y, 0.2f), __fmul_rn(res.y, 0.8f));
                    res.z = __fadd_rn(__fmul_rn(res.z, 0.2f), __fmul_rn(res.z, 0.8f));
                    res.w = __fadd_rn(__fmul_rn(res.w, 0.2f), __fmul_rn(res.w, 0.8f));
                    stg4(a.Cs + (size_t)r * F + sub * 4, res);
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------- P1, role A on 32 rows
// lit_mlp (137 -> 512 -> 512 -> 256, query_sat.py:252-254) with fp32 products from bf16 slices (split_bf16.h) on
// two row blocks per weight fragment: a 16-row tile as slices is bound by its weight stream (role_a1s below, §4.7),
// a 32-row tile does twice the rows in about the same time.  Two 512-wide hidden tiles of 32 rows (100 KB each as
// three planes) do not fit LDS, so layer 1
// is computed in two column halves and layer 2 accumulates over the matching halves of its K range:
//   L0 (two passes of 16 column tiles) -> h1 | L1 cols 0-255 -> h2 | L2 k 0-255 | L1 cols 256-511 -> h2 | L2 k 256-511
// LDS: h1 [3][32][528] (first the fp32 PairNorm rows) + max(v1 planes [3][32][176], h2 half [3][32][272]) = 150 KB.
// 66 k cycles per 32 rows (tools/lit32_probe.hip) against 68 k per 16 rows on the fp32 MFMA.
#if DSAT_ROUND_NWAVES == 16
// `pair_here`: no role-B workgroup covers these rows (large grids: dsat_api.hip, plan_pair_blocks): the tile runs the
// output / query pair itself, as two split chains on its 32 rows, between the PairNorm and lit_mlp.  The pre-mix state
// leaves the PairNorm as planes behind the fp32 rows, the chains' hidden planes follow them -- all inside h1's space,
// which lit_mlp's first layer only overwrites afterwards; the query chain reads the v1 planes lit_mlp reads too.
__device__ __forceinline__ void role_a2(const RoundArgs& a, int tile, const TileMetaDev& tm, float* smem, int tid,
                                        bool pair_here) {
    constexpr int BM = 32, PD = DSAT_SPLIT_PD4;
    constexpr int KS0 = (F + AUX_W + 31) / 32, LDX = KS0 * 32 + 16, HID = 512, LD1 = HID + 16, LD2 = HID / 2 + 16;
    constexpr int KS1 = HID / 32, KS2 = HID / 32;
    const int wave = tid >> 6, lane = tid & 63;
    const int N = a.N, NL = a.NL;
    const int tile_lo = tile * BM;
    unsigned short* h1 = reinterpret_cast<unsigned short*>(smem);           // [3][32][LD1]
    unsigned short* r2 = h1 + 3 * BM * LD1;                                 // v1 planes, then the h2 half
    float* sMean = reinterpret_cast<float*>(r2 + 3 * BM * LD2);             // F floats
    float* sState = smem;                                                   // fp32 rows [32][ldS] inside h1's space
    constexpr int ldS = F + AUX_W + 8;
    float* sPart = sState + BM * ldS;                                       // NT / F rows of F floats
    DSAT_STAMP_RT_BEGIN(a);
    DSAT_STAMP(a, 0);
    if (tid < BM) {   // aux columns of v1 for this round
        const int n = tile_lo + tid;
        load_aux_row(sState + tid * ldS + F, a.aux, n, N, a.round, n < N && !a.tail);
    }
    // pair in the tile: the pre-mix state as planes [3][32][bo_ld] behind the fp32 rows and their partial sums
    unsigned short* xo = reinterpret_cast<unsigned short*>(sPart + (NT / F) * F);
    unsigned short* hp = xo + 3 * BM * a.bo_ld;                                         // hidden planes of either chain
    pairnorm_tile<BM>(tile_lo, N, tm, a.var_seg, a.vgseg, a.psum_v, a.vgseg2, a.psum2_v, a.newv, a.V_in, a.V_out,
                      a.round == 0, sState, ldS, nullptr, 0, sMean, sPart, F, tid, a.inv_f, a.fm,
                      pair_here ? xo : nullptr, a.bo_ld);
    DSAT_STAMP(a, 1);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // v1 = [state | aux | zeros up to the last k-step] as slices
    auto slice_v1 = [&]() {
        for (int i = tid; i < BM * (KS0 * 8); i += NT) {
            const int r = i / (KS0 * 8), c = 4 * (i % (KS0 * 8));
            const float4 v = c < F + AUX_W ? ld4(sState + r * ldS + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            split_st4(r2, BM * LDX, r * LDX + c, v);
        }
    };
    if (pair_here && a.round > 0) {   // R14: logits of the previous round from the pre-mix state
        float* lg = a.logits;
        float* tr = a.trace_logits != nullptr ? a.trace_logits + (size_t)(a.round - 1) * N * LOGIT_MAPS : nullptr;
        SplitChain so;
        so.nlayers = 2;
#pragma unroll
        for (int l = 0; l < 3; ++l) so.L[l] = SplitLayer{a.bo_ws[l < 2 ? l : 0], a.mout4.L[l < 2 ? l : 0].bias, a.bo_ks[l < 2 ? l : 0], a.bo_nt[l < 2 ? l : 0]};
        split_chain<2, NT, PD>(so, xo, a.bo_ld, hp, hp, a.bo_ld, wv, lane, tid, [=](int row, int col, float4 v) {
            int n = tile_lo + row;
            NO_HOIST(n);
            if (n < N && col < LOGIT_MAPS) {
                stg4(lg + (size_t)n * LOGIT_MAPS + col, v);
                if (tr != nullptr) st4(tr + (size_t)n * LOGIT_MAPS + col, v);
            }
        }, [] {});
    }
    DSAT_STAMP(a, 2);
    if (a.tail) {   // the tail launch only finishes the PairNorm (and makes the logits where no role B does)
        DSAT_STAMP(a, 3);
        DSAT_STAMP(a, 4);
        DSAT_STAMP_RT_END(a);
        return;
    }
    if (!pair_here) {
        slice_v1();
    } else {   // R2: query = variables_query(v1); softplus(+-query) feeds the clause gather (loss/sat.py:132-133); the v1
        // slices are made as the chain's prep(), after its first weights have been requested
        float* q = a.q;
        float* sp = a.sp;
        SplitChain sc;
        sc.nlayers = 2;
#pragma unroll
        for (int l = 0; l < 3; ++l) sc.L[l] = SplitLayer{a.sp_ws[l < 2 ? l : 0], a.mq4.L[l < 2 ? l : 0].bias, a.sp_ks[l < 2 ? l : 0], a.sp_nt[l < 2 ? l : 0]};
        split_chain<2, NT, PD>(sc, r2, LDX, hp, hp, a.s_ld_hid, wv, lane, tid, [=](int row, int col, float4 v) {
            int n = tile_lo + row;
            NO_HOIST(n);
            if (n < N && col < F) {
                stg4(q + (size_t)n * F + col, v);
                const float4 t = make_float4(log1pf(expf(-fabsf(v.x))), log1pf(expf(-fabsf(v.y))),
                                             log1pf(expf(-fabsf(v.z))), log1pf(expf(-fabsf(v.w))));
                stg4(sp + (size_t)n * F + col, make_float4(fmaxf(v.x, 0.f) + t.x, fmaxf(v.y, 0.f) + t.y,
                                                          fmaxf(v.z, 0.f) + t.z, fmaxf(v.w, 0.f) + t.w));
                stg4(sp + (size_t)(NL + n) * F + col, make_float4(fmaxf(-v.x, 0.f) + t.x, fmaxf(-v.y, 0.f) + t.y,
                                                                fmaxf(-v.z, 0.f) + t.z, fmaxf(-v.w, 0.f) + t.w));
            }
        }, slice_v1);
    }
    DSAT_STAMP(a, 3);
    const unsigned voff = (unsigned)lane * 16u;
    const int col4 = 4 * (lane >> 4), row = lane & 15;
    auto wb = [&](int l, int t, int ks) { return reinterpret_cast<const float4*>(a.a2_ws[l]) + ((size_t)t * ks * 3) * 64; };
    auto leaky = [](f32x4 acc, float4 b) { return bias_act(acc, b, 1); };
    float4 b0a = *reinterpret_cast<const float4*>(a.mlit.L[0].bias + wv * 16 + col4);
    float4 b0b = *reinterpret_cast<const float4*>(a.mlit.L[0].bias + (wv + 16) * 16 + col4);
    float4 b1a = *reinterpret_cast<const float4*>(a.mlit.L[1].bias + wv * 16 + col4);
    float4 b1b = *reinterpret_cast<const float4*>(a.mlit.L[1].bias + (wv + 16) * 16 + col4);
    float4 b2 = *reinterpret_cast<const float4*>(a.mlit.L[2].bias + wv * 16 + col4);
    asm volatile("" : "+v"(b0a.x), "+v"(b0a.y), "+v"(b0a.z), "+v"(b0a.w), "+v"(b0b.x), "+v"(b0b.y), "+v"(b0b.z), "+v"(b0b.w));
    asm volatile("" : "+v"(b1a.x), "+v"(b1a.y), "+v"(b1a.z), "+v"(b1a.w), "+v"(b1b.x), "+v"(b1b.y), "+v"(b1b.z), "+v"(b1b.w));
    asm volatile("" : "+v"(b2.x), "+v"(b2.y), "+v"(b2.z), "+v"(b2.w));
    const unsigned short* xin = r2 + row * LDX + 8 * (lane >> 4);
    const unsigned short* h1in = h1 + row * LD1 + 8 * (lane >> 4);
    const unsigned short* h2in = r2 + row * LD2 + 8 * (lane >> 4);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a0[2] = {z, z}, a0b[2] = {z, z}, a1[2] = {z, z}, a1b[2] = {z, z}, a2[2] = {z, z};
    auto ep0a = [&]() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) split_st4(h1, BM * LD1, (rb * 16 + row) * LD1 + wv * 16 + col4, leaky(a0[rb], b0a));
    };
    auto ep0b = [&]() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) split_st4(h1, BM * LD1, (rb * 16 + row) * LD1 + (wv + 16) * 16 + col4, leaky(a0b[rb], b0b));
    };
    auto ep1a = [&]() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) split_st4(r2, BM * LD2, (rb * 16 + row) * LD2 + wv * 16 + col4, leaky(a1[rb], b1a));
    };
    auto ep1b = [&]() {
        lds_barrier();   // every wave is done with the first half of h2
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) split_st4(r2, BM * LD2, (rb * 16 + row) * LD2 + wv * 16 + col4, leaky(a1b[rb], b1b));
    };
#if DSAT_A1_ONE_RING
    // one ring across the six passes (split_vpass, as role_a1s): L0 twice, L1 columns 0..255, L2 k 0..255, L1 columns
    // 256..511, L2 k 256..511; the first barrier also ends the reads of the fp32 rows h1 is about to overwrite
    static_assert(KS0 >= PD && KS1 >= PD && KS2 / 2 >= PD, "a pass is at least one turn of the ring");
    const float4* const pw[6] = {wb(0, wv, KS0), wb(0, wv + 16, KS0), wb(1, wv, KS1), wb(2, wv, KS2), wb(1, wv + 16, KS1),
                                 wb(2, wv, KS2) + (size_t)(KS2 / 2) * 192};
    constexpr int B1 = KS0, B2 = 2 * KS0, B3 = B2 + KS1, B4 = B3 + KS2 / 2, B5 = B4 + KS1;
    SplitSlot ring[PD];
    split_vprime<PD>(ring, voff, pw[0]);
    lds_barrier();
    split_vpass<2, PD>(ring, voff, pw[0], 0, KS0, pw[1], xin, BM * LDX, LDX, a0);
    ep0a();
    split_vpass<2, PD>(ring, voff, pw[1], B1, KS0, pw[2], xin, BM * LDX, LDX, a0b);
    ep0b();
    lds_barrier();
    split_vpass<2, PD>(ring, voff, pw[2], B2, KS1, pw[3], h1in, BM * LD1, LD1, a1);
    ep1a();
    lds_barrier();
    split_vpass<2, PD>(ring, voff, pw[3], B3, KS2 / 2, pw[4], h2in, BM * LD2, LD2, a2);
    split_vpass<2, PD>(ring, voff, pw[4], B4, KS1, pw[5], h1in, BM * LD1, LD1, a1b);
    ep1b();
    lds_barrier();
    split_vpass<2, PD>(ring, voff, pw[5], B5, KS2 / 2, pw[5] + (size_t)(KS2 / 2 - PD) * 192, h2in, BM * LD2, LD2, a2);
    split_vdrain<PD>(ring);
#else
    // L0, two passes; the first barrier also ends the reads of the fp32 rows h1 is about to overwrite
    split_layer<2, PD>(voff, wb(0, wv, KS0), KS0, xin, BM * LDX, LDX, [] {}, a0);
    split_layer<2, PD, false>(voff, wb(0, wv + 16, KS0), KS0, xin, BM * LDX, LDX, ep0a, a0b);
    split_layer<2, PD>(voff, wb(1, wv, KS1), KS1, h1in, BM * LD1, LD1, ep0b, a1);            // L1 columns 0..255
    split_layer<2, PD>(voff, wb(2, wv, KS2), KS2 / 2, h2in, BM * LD2, LD2, ep1a, a2);         // L2, k 0..255
    split_layer<2, PD, false>(voff, wb(1, wv + 16, KS1), KS1, h1in, BM * LD1, LD1, [] {}, a1b);   // L1 columns 256..511
    split_layer<2, PD>(voff, wb(2, wv, KS2) + (size_t)(KS2 / 2) * 192, KS2 / 2, h2in, BM * LD2, LD2, ep1b, a2);   // L2, k 256..511
#endif
    // R6: literals = [lit_mlp(v1)[:, :128] ; [:, 128:]]
    float* lits = a.lits;
    const int col = wv * 16 + col4;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        int n = tile_lo + rb * 16 + row;
        NO_HOIST(n);
        if (n < N) stg4(lits + ((size_t)(col < F ? n : NL + n)) * F + (col & (F - 1)), bias_act(a2[rb], b2, 0));
    }
    DSAT_STAMP(a, 4);
    DSAT_STAMP_RT_END(a);
}

// ------------------------------------------------------------------------------------- P1, role A on 16 rows, split
// The same lit_mlp as split products on a 16-ROW tile.  One row block per weight fragment streams 2.85 MB of planes
// per tile, but the tile still ends sooner than on the fp32 MFMA (tools/lit16w_probe.hip: 57 k cycles against 69 k
// at 194 workgroups; slicing fp32 weights in registers instead -- 4 B per weight, 36 VALU instructions per fragment
// -- took 66 k; two column tiles per wave sharing the activation fragments: 56 k in the probe, no change in the
// kernel).  Used when role B covers every row of the batch (the tile then is PairNorm + lit_mlp only) and
// role A is not on 32-row tiles.  LDS: h1 [3][16][528] (first the fp32 PairNorm rows) | h2 [3][16][528] | v1 planes
// [3][16][176] | mean: 116 KB.
__device__ __forceinline__ void role_a1s(const RoundArgs& a, int tile, const TileMetaDev& tm, float* smem, int tid) {
    constexpr int BM = 16, PD = DSAT_SPLIT_PD_A1;
    constexpr int KS0 = (F + AUX_W + 31) / 32, LDX = KS0 * 32 + 16, HID = 512, LD1 = HID + 16;
    constexpr int KS1 = HID / 32, KS2 = HID / 32;
    const int wave = tid >> 6, lane = tid & 63;
    const int N = a.N, NL = a.NL;
    const int tile_lo = tile * BM;
    unsigned short* h1 = reinterpret_cast<unsigned short*>(smem);           // [3][16][LD1]
    unsigned short* h2 = h1 + 3 * BM * LD1;                                 // [3][16][LD1]
    unsigned short* xs = h2 + 3 * BM * LD1;                                 // v1 planes [3][16][LDX]
    float* sMean = reinterpret_cast<float*>(xs + 3 * BM * LDX);             // F floats
    float* sState = smem;                                                   // fp32 rows [16][ldS] inside h1's space
    constexpr int ldS = F + AUX_W + 8;
    float* sPart = sState + BM * ldS;                                       // NT / F rows of F floats
    static_assert((BM * ldS + (NT / F) * F) * 4 <= 3 * BM * LD1 * 2, "PairNorm scratch fits h1");
#if DSAT_A1_EARLY
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)lane * 16u;
    const int col4 = 4 * (lane >> 4), row = lane & 15;
    auto wb = [&](int l, int t, int ks) { return reinterpret_cast<const float4*>(a.a2_ws[l]) + ((size_t)t * ks * 3) * 64; };
    // (the biases: requested before the PairNorm, whose latency hides theirs; pinned -- completed -- behind it)
    auto leaky = [](f32x4 acc, float4 b) { return bias_act(acc, b, 1); };
    float4 b0a = *reinterpret_cast<const float4*>(a.mlit.L[0].bias + wv * 16 + col4);
    float4 b0b = *reinterpret_cast<const float4*>(a.mlit.L[0].bias + (wv + 16) * 16 + col4);
    float4 b1a = *reinterpret_cast<const float4*>(a.mlit.L[1].bias + wv * 16 + col4);
    float4 b1b = *reinterpret_cast<const float4*>(a.mlit.L[1].bias + (wv + 16) * 16 + col4);
    float4 b2 = *reinterpret_cast<const float4*>(a.mlit.L[2].bias + wv * 16 + col4);
#endif
    DSAT_STAMP_RT_BEGIN(a);
    DSAT_STAMP(a, 0);
    if (tid < BM) {   // aux columns of v1 for this round
        const int n = tile_lo + tid;
        load_aux_row(sState + tid * ldS + F, a.aux, n, N, a.round, n < N);
    }
    pairnorm_tile<BM>(tile_lo, N, tm, a.var_seg, a.vgseg, a.psum_v, a.vgseg2, a.psum2_v, a.newv, a.V_in, a.V_out,
                      a.round == 0, sState, ldS, nullptr, 0, sMean, sPart, F, tid, a.inv_f, a.fm);
    DSAT_STAMP(a, 1);
    DSAT_STAMP(a, 2);
    DSAT_STAMP(a, 3);
#if DSAT_A1_EARLY
    asm volatile("" : "+v"(b0a.x), "+v"(b0a.y), "+v"(b0a.z), "+v"(b0a.w), "+v"(b0b.x), "+v"(b0b.y), "+v"(b0b.z), "+v"(b0b.w));
    asm volatile("" : "+v"(b1a.x), "+v"(b1a.y), "+v"(b1a.z), "+v"(b1a.w), "+v"(b1b.x), "+v"(b1b.y), "+v"(b1b.z), "+v"(b1b.w));
    asm volatile("" : "+v"(b2.x), "+v"(b2.y), "+v"(b2.z), "+v"(b2.w));
#if DSAT_A1_ONE_RING
    // the ring's first fragments travel while v1 is sliced (LDS work only: nothing else enters the vmcnt queue)
    const float4* const pw[5] = {wb(0, wv, KS0), wb(0, wv + 16, KS0), wb(1, wv, KS1), wb(1, wv + 16, KS1), wb(2, wv, KS2)};
    SplitSlot ring[PD];
    split_vprime<PD>(ring, voff, pw[0]);
#endif
#endif
    // v1 = [state | aux | zeros up to the last k-step] as slices
    for (int i = tid; i < BM * (KS0 * 8); i += NT) {
        const int r = i / (KS0 * 8), c = 4 * (i % (KS0 * 8));
        const float4 v = c < F + AUX_W ? ld4(sState + r * ldS + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        split_st4(xs, BM * LDX, r * LDX + c, v);
    }
#if !DSAT_A1_EARLY
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)lane * 16u;
    const int col4 = 4 * (lane >> 4), row = lane & 15;
    auto wb = [&](int l, int t, int ks) { return reinterpret_cast<const float4*>(a.a2_ws[l]) + ((size_t)t * ks * 3) * 64; };
    auto leaky = [](f32x4 acc, float4 b) { return bias_act(acc, b, 1); };
    float4 b0a = *reinterpret_cast<const float4*>(a.mlit.L[0].bias + wv * 16 + col4);
    float4 b0b = *reinterpret_cast<const float4*>(a.mlit.L[0].bias + (wv + 16) * 16 + col4);
    float4 b1a = *reinterpret_cast<const float4*>(a.mlit.L[1].bias + wv * 16 + col4);
    float4 b1b = *reinterpret_cast<const float4*>(a.mlit.L[1].bias + (wv + 16) * 16 + col4);
    float4 b2 = *reinterpret_cast<const float4*>(a.mlit.L[2].bias + wv * 16 + col4);
    asm volatile("" : "+v"(b0a.x), "+v"(b0a.y), "+v"(b0a.z), "+v"(b0a.w), "+v"(b0b.x), "+v"(b0b.y), "+v"(b0b.z), "+v"(b0b.w));
    asm volatile("" : "+v"(b1a.x), "+v"(b1a.y), "+v"(b1a.z), "+v"(b1a.w), "+v"(b1b.x), "+v"(b1b.y), "+v"(b1b.z), "+v"(b1b.w));
    asm volatile("" : "+v"(b2.x), "+v"(b2.y), "+v"(b2.z), "+v"(b2.w));
#endif
    const unsigned short* xin = xs + row * LDX + 8 * (lane >> 4);
    const unsigned short* h1in = h1 + row * LD1 + 8 * (lane >> 4);
    const unsigned short* h2in = h2 + row * LD1 + 8 * (lane >> 4);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a0a[1] = {z}, a0b[1] = {z}, a1a[1] = {z}, a1b[1] = {z}, a2[1] = {z};
    // L0 and L1: column tiles wv, then wv + 16; the first barrier also ends the reads of the fp32 rows h1 is about
    // to overwrite
#if DSAT_A1_ONE_RING
    // one ring across the five passes (split_vpass): the next pass's first fragments travel during this one's last steps
    static_assert(KS0 >= PD && KS1 >= PD && KS2 >= PD, "a pass is at least one turn of the ring");
#if !DSAT_A1_EARLY
    const float4* const pw[5] = {wb(0, wv, KS0), wb(0, wv + 16, KS0), wb(1, wv, KS1), wb(1, wv + 16, KS1), wb(2, wv, KS2)};
    SplitSlot ring[PD];
    split_vprime<PD>(ring, voff, pw[0]);
#endif
    lds_barrier();   // v1's planes are published; also ends the reads of the fp32 rows h1 is about to overwrite
    split_vpass<1, PD>(ring, voff, pw[0], 0, KS0, pw[1], xin, BM * LDX, LDX, a0a);
    split_vpass<1, PD>(ring, voff, pw[1], KS0, KS0, pw[2], xin, BM * LDX, LDX, a0b);
    split_st4(h1, BM * LD1, row * LD1 + wv * 16 + col4, leaky(a0a[0], b0a));
    split_st4(h1, BM * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(a0b[0], b0b));
    lds_barrier();
    split_vpass<1, PD>(ring, voff, pw[2], 2 * KS0, KS1, pw[3], h1in, BM * LD1, LD1, a1a);
    split_vpass<1, PD>(ring, voff, pw[3], 2 * KS0 + KS1, KS1, pw[4], h1in, BM * LD1, LD1, a1b);
    split_st4(h2, BM * LD1, row * LD1 + wv * 16 + col4, leaky(a1a[0], b1a));
    split_st4(h2, BM * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(a1b[0], b1b));
    lds_barrier();
    split_vpass<1, PD>(ring, voff, pw[4], 2 * KS0 + 2 * KS1, KS2, pw[4] + (size_t)(KS2 - PD) * 192, h2in, BM * LD1, LD1, a2);
    split_vdrain<PD>(ring);
#else
    split_layer<1, PD>(voff, wb(0, wv, KS0), KS0, xin, BM * LDX, LDX, [] {}, a0a);
    split_layer<1, PD, false>(voff, wb(0, wv + 16, KS0), KS0, xin, BM * LDX, LDX, [] {}, a0b);
    auto ep0 = [&]() {
        split_st4(h1, BM * LD1, row * LD1 + wv * 16 + col4, leaky(a0a[0], b0a));
        split_st4(h1, BM * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(a0b[0], b0b));
    };
    split_layer<1, PD>(voff, wb(1, wv, KS1), KS1, h1in, BM * LD1, LD1, ep0, a1a);
    split_layer<1, PD, false>(voff, wb(1, wv + 16, KS1), KS1, h1in, BM * LD1, LD1, [] {}, a1b);
    auto ep1 = [&]() {
        split_st4(h2, BM * LD1, row * LD1 + wv * 16 + col4, leaky(a1a[0], b1a));
        split_st4(h2, BM * LD1, row * LD1 + (wv + 16) * 16 + col4, leaky(a1b[0], b1b));
    };
    split_layer<1, PD>(voff, wb(2, wv, KS2), KS2, h2in, BM * LD1, LD1, ep1, a2);
#endif
    // R6: literals = [lit_mlp(v1)[:, :128] ; [:, 128:]]
    float* lits = a.lits;
    const int col = wv * 16 + col4;
    int n = tile_lo + row;
    NO_HOIST(n);
    if (n < N) stg4(lits + ((size_t)(col < F ? n : NL + n)) * F + (col & (F - 1)), bias_act(a2[0], b2, 0));
    DSAT_STAMP(a, 4);
    DSAT_STAMP_RT_END(a);
}
#endif

// ------------------------------------------------------------------------------------- P1
// Three roles in one grid:
//   A  blocks [0, vtiles): 16-row variable tiles -- PairNorm (R13, writes the carried state), lit_mlp (R6);
//      without role B also the output / query pair (R14, R2).
//   B  blocks [vtiles, vtiles + pair_blocks): 64-row tiles -- the same PairNorm again (registers and LDS
//      only; role A owns the state write), then variables_output and variables_query with four row
//      blocks per weight fragment.  At 20k nodes the 194 A tiles leave 62 CUs idle and a
//      kernel lasts one tile: moving the pair (a fifth of a tile's time, and a poor fit for 16-row
//      tiles: two small layers each) to those CUs shortens every A tile, and the 64-row form does the
//      pair's MFMA work with a quarter of the weight traffic and fixed per-layer cost per row.
//   C  blocks after that: finish the clause PairNorm of the previous round (R10; depends on P2(r-1)
//      only), off P2's critical path.
// The carried state is double-buffered by round (V_in -> V_out): A and B both read the old state.
// Two instantiations: A2 = role A on 32-row tiles (role_a2, which may run the pair itself) with roles B and C beside it;
// !A2 = role A on 16-row tiles (role_a1s / the fp32 chains) with roles B and C.  One kernel holding every form of role
// A is more than hipcc's register allocation digests once role_a2 carries its own pair ("illegal VGPR to SGPR copy").
template <bool A2>
__global__ __launch_bounds__(NT, ROUND_WGS_PER_CU * NT / 256) void round_p1_kernel(RoundArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    constexpr int BM = ROUND_BM_VAR;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int N = a.N, NL = a.NL;
    const int vtiles = a.a2_tiles > 0 ? a.a2_tiles : (N + BM - 1) / BM;   // role A: 16-row tiles, or 32-row ones (role_a2)
    const int b = blockIdx.x;
    const bool roleA = b < vtiles, roleB = !roleA && b < vtiles + a.pair_blocks;
    // role B covers the 64-row groups [pair_first, pair_first + pair_blocks): normally all of them; on grids of
    // several waves of workgroups only as many as make the block count fit the waves (dsat_api.hip,
    // plan_pair_blocks) -- the A tiles of the uncovered FIRST rows run the pair themselves and start first
    const int tile = roleA ? xcd_tile(b, vtiles) : (roleB ? a.pair_first + (b - vtiles) : 0);
    // the early-exit flag and the tile record are requested together: one memory latency, not two
    const int done = a.check_done ? a.ctl->done : 0;
    const TileMetaDev tm = load_meta(roleB ? a.vmeta4 : (a.a2_tiles > 0 ? a.vmeta2 : a.vmeta), tile);
    if (both_arrived(done, tm) != 0) return;
#if DSAT_ROUND_NWAVES == 16
    if constexpr (A2) {
        if (roleA) {
            role_a2(a, tile, tm, smem, tid, a.pair_blocks == 0 || tile * 32 < a.pair_first * 64);
            return;
        }
    } else {
        if (roleA && a.a1_split) {
            role_a1s(a, tile, tm, smem, tid);
            return;
        }
    }
#endif
    if (!roleA && !roleB) {
        DSAT_STAMP_RT_BEGIN(a);
        DSAT_STAMP(a, 0);
        clause_norm_block(a, b - vtiles - a.pair_blocks, a.cn_blocks, smem, tid);
        DSAT_STAMP(a, 4);
        DSAT_STAMP_RT_END(a);
        return;
    }
    const int ldx = a.ld_x;
    float* lg = a.logits;
    float* tr = a.trace_logits != nullptr && a.round > 0
                    ? a.trace_logits + (size_t)(a.round - 1) * N * LOGIT_MAPS : nullptr;
    float* q = a.q;
    float* sp = a.sp;
    if (roleB) {
        constexpr int BM4 = 64;
        const int tile_lo = tile * BM4;
        float* sX = smem + a.b_off_x;
        float* sVlog = smem + a.b_off_vlog;
        float* sA = smem + a.b_off_a;
        float* sB = smem + a.b_off_b;
        float* sMean = smem + a.b_off_mean;
        const int ldv = F + 8;
        DSAT_STAMP_RT_BEGIN(a);
        DSAT_STAMP(a, 0);
        if (tid < BM4) {
            const int n = tile_lo + tid;
            load_aux_row(sX + tid * ldx + F, a.aux, n, N, a.round, n < N && !a.tail);
        }
#if DSAT_ROUND_NWAVES == 16
        // variables_output as split products too (bo_ws): the pre-mix state leaves the PairNorm as planes, so the LDS
        // plan is [X | out-planes | out-hidden | mean] and the query chain's planes go over all of it afterwards
        const bool out_split = a.bo_ws[0] != nullptr;
        unsigned short* xo = reinterpret_cast<unsigned short*>(smem + a.bo_off_x);
        unsigned short* ho = reinterpret_cast<unsigned short*>(smem + a.bo_off_h);
        if (out_split) sMean = smem + a.bo_off_mean;
        pairnorm_tile<BM4>(tile_lo, N, tm, a.var_seg, a.vgseg, a.psum_v, a.vgseg2, a.psum2_v, a.newv, a.V_in, nullptr,
                           a.round == 0, sX, ldx, out_split ? nullptr : sVlog, ldv, sMean, sMean + F, F, tid, a.inv_f, a.fm,
                           out_split ? xo : nullptr, a.bo_ld);
#else
        constexpr bool out_split = false;
        pairnorm_tile<BM4>(tile_lo, N, tm, a.var_seg, a.vgseg, a.psum_v, a.vgseg2, a.psum2_v, a.newv, a.V_in, nullptr,
                           a.round == 0, sX, ldx, sVlog, ldv, sMean, sMean + F, F, tid, a.inv_f, a.fm);
#endif
        auto store_logits = [=](int row, int col, float4 v) {
            int n = tile_lo + row;
            NO_HOIST(n);
            if (n < N && col < LOGIT_MAPS) {
                stg4(lg + (size_t)n * LOGIT_MAPS + col, v);
                if (tr != nullptr) st4(tr + (size_t)n * LOGIT_MAPS + col, v);
            }
        };
        auto store_query = [=](int row, int col, float4 v) {
            int n = tile_lo + row;
            NO_HOIST(n);
            if (n < N && col < F) {
                stg4(q + (size_t)n * F + col, v);
                const float4 t = make_float4(log1pf(expf(-fabsf(v.x))), log1pf(expf(-fabsf(v.y))),
                                             log1pf(expf(-fabsf(v.z))), log1pf(expf(-fabsf(v.w))));
                stg4(sp + (size_t)n * F + col, make_float4(fmaxf(v.x, 0.f) + t.x, fmaxf(v.y, 0.f) + t.y,
                                                          fmaxf(v.z, 0.f) + t.z, fmaxf(v.w, 0.f) + t.w));
                stg4(sp + (size_t)(NL + n) * F + col, make_float4(fmaxf(-v.x, 0.f) + t.x, fmaxf(-v.y, 0.f) + t.y,
                                                                fmaxf(-v.z, 0.f) + t.z, fmaxf(-v.w, 0.f) + t.w));
            }
        };
        // the two small chains one after the other, each on all waves with four row blocks per weight
        // fragment (side by side on half the waves each, as the 16-row pair runs them, the query half
        // needs two column tiles per wave: more registers than a 1024-thread workgroup has)
        DSAT_STAMP(a, 1);
#if DSAT_ROUND_NWAVES == 16
        if (out_split) {
            if (a.round > 0) {
                SplitChain so;
                so.nlayers = 2;
#pragma unroll
                for (int l = 0; l < 3; ++l) so.L[l] = SplitLayer{a.bo_ws[l < 2 ? l : 0], a.mout4.L[l < 2 ? l : 0].bias, a.bo_ks[l < 2 ? l : 0], a.bo_nt[l < 2 ? l : 0]};
                const int wvo = __builtin_amdgcn_readfirstlane(wave);
                split_chain<4, NT, DSAT_SPLIT_PD4>(so, xo, a.bo_ld, ho, ho, a.bo_ld, wvo, lane, tid, store_logits, [] {});
            }
        } else
#endif
        if (a.round > 0) run_chain<4>(a.mout4, sVlog, ldv, sA, a.b_ld_a, sB, a.b_ld_b, wave, lane, store_logits);
        DSAT_STAMP(a, 2);
        // (the tail launch -- role B runs in it when role A is on 32-row tiles -- is the output half alone)
#if DSAT_ROUND_NWAVES == 16
        if (!a.tail && a.sp_ws[0] != nullptr) {
            // variables_query with split products (split_bf16.h): v1 as slices goes where the output chain's tiles were
            // (sVlog | sA, dead now), the hidden planes over the fp32 v1 rows and the unused second buffer (sX | sB)
            // (with the output chain on planes: v1 planes at the end of the region its planes occupied, hidden planes from
            // the start of the workgroup's LDS, over X once it has been sliced)
            unsigned short* xp = out_split ? reinterpret_cast<unsigned short*>(smem + a.bo_off_xq) : reinterpret_cast<unsigned short*>(sVlog);
            unsigned short* hp = reinterpret_cast<unsigned short*>(sX);
            const int ldp = a.s_ld_seg;
            for (int i = tid; i < BM4 * (ldp - 16) / 4; i += NT) {
                const int r = i / ((ldp - 16) / 4), c = 4 * (i % ((ldp - 16) / 4));
                const float4 v = c < F + AUX_W ? ld4(sX + r * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                split_st4(xp, BM4 * ldp, r * ldp + c, v);
            }
            SplitChain sc;
            sc.nlayers = 2;
#pragma unroll
            for (int l = 0; l < 3; ++l) sc.L[l] = SplitLayer{a.sp_ws[l < 2 ? l : 0], a.mq4.L[l < 2 ? l : 0].bias, a.sp_ks[l < 2 ? l : 0], a.sp_nt[l < 2 ? l : 0]};
            const int wvu = __builtin_amdgcn_readfirstlane(wave);
            split_chain<4, NT, DSAT_SPLIT_PD4>(sc, xp, ldp, hp, hp, a.s_ld_hid, wvu, lane, tid, store_query, [] {});
        } else
#endif
        if (!a.tail) run_chain<4>(a.mq4, sX, ldx, sB, a.b_ld_b, sA, a.b_ld_a, wave, lane, store_query);
        DSAT_STAMP(a, 3);
        DSAT_STAMP(a, 4);
        DSAT_STAMP_RT_END(a);
        return;
    }
    if constexpr (A2) return;   // (role A was role_a2 above)
    const int tile_lo = tile * BM;
    float* sX = smem + a.off_x;
    float* sVlog = smem + a.off_vlog;
    float* sA = smem + a.off_a;
    float* sB = smem + a.off_b;
    float* sMean = smem + a.off_mean;
    const bool pair_here = a.pair_blocks == 0 || tile_lo < a.pair_first * 64;   // no role-B workgroup covers these rows
    const int ldv = a.ld_v;

    DSAT_STAMP_RT_BEGIN(a);
    DSAT_STAMP(a, 0);
    if (tid < BM) {   // aux columns of v1 for this round (independent of everything else)
        const int n = tile_lo + tid;
        load_aux_row(sX + tid * ldx + F, a.aux, n, N, a.round, n < N && !a.tail);
    }
    pairnorm_tile<BM>(tile_lo, N, tm, a.var_seg, a.vgseg, a.psum_v, a.vgseg2, a.psum2_v, a.newv, a.V_in, a.V_out,
                      a.round == 0, sX, ldx, pair_here ? sVlog : nullptr, ldv, sMean, sMean + F, F, tid, a.inv_f, a.fm);
    DSAT_STAMP(a, 1);

    if (pair_here) {
        // R14: logits of the previous round from the pre-mix state
        auto store_logits = [=](int row, int col, float4 v) {
            int n = tile_lo + row;
            NO_HOIST(n);
            if (n < N && col < LOGIT_MAPS) {
                stg4(lg + (size_t)n * LOGIT_MAPS + col, v);
                if (tr != nullptr) st4(tr + (size_t)n * LOGIT_MAPS + col, v);
            }
        };
        // R2: query = variables_query(v1); softplus(+-query) feeds the clause gather (loss/sat.py:132-133)
        auto store_query = [=](int row, int col, float4 v) {
            int n = tile_lo + row;
            NO_HOIST(n);
            if (n < N && col < F) {
                stg4(q + (size_t)n * F + col, v);
                // softplus(+-x) = max(+-x, 0) + log1p(exp(-|x|)): one transcendental pair serves both signs
                const float4 t = make_float4(log1pf(expf(-fabsf(v.x))), log1pf(expf(-fabsf(v.y))),
                                             log1pf(expf(-fabsf(v.z))), log1pf(expf(-fabsf(v.w))));
                stg4(sp + (size_t)n * F + col, make_float4(fmaxf(v.x, 0.f) + t.x, fmaxf(v.y, 0.f) + t.y,
                                                          fmaxf(v.z, 0.f) + t.z, fmaxf(v.w, 0.f) + t.w));
                stg4(sp + (size_t)(NL + n) * F + col, make_float4(fmaxf(-v.x, 0.f) + t.x, fmaxf(-v.y, 0.f) + t.y,
                                                                fmaxf(-v.z, 0.f) + t.z, fmaxf(-v.w, 0.f) + t.w));
            }
        };
        if (a.round > 0 && !a.tail && a.mout.nlayers == 2 && a.mq.nlayers == 2) {
            run_chain_pair(a.mout, sVlog, ldv, sA, a.ld_a, a.mq, sX, ldx, sB, a.ld_b, wave, lane, store_logits,
                           store_query);
            DSAT_STAMP(a, 2);
        } else {
            if (a.round > 0) run_chain<1>(a.mout, sVlog, ldv, sA, a.ld_a, sB, a.ld_b, wave, lane, store_logits);
            DSAT_STAMP(a, 2);
            if (a.tail) return;
            run_chain<1>(a.mq, sX, ldx, sA, a.ld_a, sB, a.ld_b, wave, lane, store_query);
        }
    } else {
        DSAT_STAMP(a, 2);
    }
    DSAT_STAMP(a, 3);
    {   // R6: literals = [lit_mlp(v1)[:, :128] ; [:, 128:]]
        float* lits = a.lits;
        run_chain<1>(a.mlit, sX, ldx, sA, a.ld_a, sB, a.ld_b, wave, lane, [=](int row, int col, float4 v) {
            int n = tile_lo + row;
            NO_HOIST(n);
            if (n < N && col < 2 * F) stg4(lits + ((size_t)(col < F ? n : NL + n)) * F + (col & (F - 1)), v);
        });
    }
    DSAT_STAMP(a, 4);
    DSAT_STAMP_RT_END(a);
}

// ------------------------------------------------------------------------------------- P2
// Blocks [0, tiles) are clause tiles; blocks [tiles, tiles+G) (when s_enable) run the
// select/SAT step of the PREVIOUS round for one graph each, on CUs the tiles leave idle:
// it only feeds the early-exit flag and the step outputs, so it needs no launch of its own.
template <int RB>
__global__ __launch_bounds__(NT, ROUND_WGS_PER_CU * NT / 256) void round_p2_kernel(RoundArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    constexpr int BM = 16 * RB;
    constexpr int ROWS = (BM + NT / 32 - 1) / (NT / 32);   // clause rows per half-wave
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int C = a.C, NL = a.NL;
    const int tiles = (C + BM - 1) / BM;
    // select/SAT workgroups normally follow the tiles (idle CUs pick them up); when a large graph makes them
    // long (sel_lead > 0: its select is cut into parts that each take longer than a tile) they come FIRST,
    // so they run beside the first wave of tiles instead of after the last.  sel_lead is a multiple of 8:
    // the tiles keep their XCD mapping.
    const int lead = a.s_enable ? a.sel_lead : 0;
    const int tb = (int)blockIdx.x - lead;
    const int nsb = a.sel_pack > 1 ? (a.n_sel + a.sel_pack - 1) / a.sel_pack : a.n_sel;   // select workgroups
    if (tb < 0 && (int)blockIdx.x >= nsb) return;
    const bool extra = tb < 0 || tb >= tiles;
    const int sel_blk = tb < 0 ? (int)blockIdx.x : tb - tiles;
    const int tile = extra ? 0 : xcd_tile(tb, tiles);
    const int done = a.check_done ? a.ctl->done : 0;   // requested together with the tile record
    const TileMetaDev tm = load_meta(a.cmeta, tile);
    ListBlock lb;                                       // ... and with the tile's clause -> literal lists
    list_block_fetch(lb, extra ? nullptr : a.clist, a.clist_stride, tile, tid);
    if (both_arrived(done, tm) != 0) return;
    if (extra) {
        DSAT_STAMP_RT_BEGIN(a);
        DSAT_STAMP(a, 0);
        if (a.sel_pack > 1) {   // four graphs per workgroup, 256 threads each (large batches of small graphs: a select
            // workgroup per graph would be a whole extra wave of workgroups, 256 at config 3)
            const int sub = tid >> 8, e = sel_blk * 4 + sub;
            const bool act = e < a.n_sel;
            select_sat_body<256>(act ? e : a.n_sel - 1, a.n_sel, a.sel_tab, a.var_seg, a.cl_seg, a.c_rowptr, a.c_lit,
                                 a.logits, a.ctl, a.pgl, a.best, a.out_logits, a.assign, a.graph_unsat, NL, a.round - 1,
                                 a.check_done, smem + sub * 2560, tid & 255, act);
        } else
        select_sat_body<NT>(sel_blk, a.n_sel, a.sel_tab, a.var_seg, a.cl_seg, a.c_rowptr, a.c_lit, a.logits, a.ctl,
                            a.pgl, a.best, a.out_logits, a.assign, a.graph_unsat, NL, a.round - 1, a.check_done,
                            smem);
        DSAT_STAMP(a, 4);
        DSAT_STAMP_RT_END(a);
        return;
    }
    const int tile_lo = tile * BM;
    const int tile_hi = min(tile_lo + BM, C);
    float* sX = smem + a.off_x;
    float* sA = smem + a.off_a;
    float* sB = smem + a.off_b;
    float* sMean = smem + a.off_mean;
    const int ldx = a.ld_x;

    DSAT_STAMP_RT_BEGIN(a);
    DSAT_STAMP(a, 0);
    // the clause rows' literal lists -> LDS (sA is free until the first layer writes it)
    int* sPtr = reinterpret_cast<int*>(sA);
    int* sIdx = sPtr + BM + 8;
    bool staged;
    if (a.clist != nullptr) {   // the block is [ptr | list]
        list_block_store(lb, sPtr, a.clist_stride, tid);
        sIdx = sPtr + BM + 1;
        staged = tm.e_end0 - tm.e_base0 <= a.clist_max;
    } else {
        staged = stage_csr(a.cg_rowptr, a.cg_lit, tile_lo, tile_hi - tile_lo, tm.e_base0, tm.e_end0, sPtr, sIdx,
                           BM * a.ld_a - (BM + 8), tid);
    }
    if (a.cs_ready && a.round > 0 && (a.cs_ready_rows <= 0 || tile_hi <= a.cs_ready_rows)) {   // R10 was finished by a role C: just load the rows
        const int sub = tid & 31, hw = tid >> 5;
#pragma unroll
        for (int kq = 0; kq < ROWS; ++kq) {
            const int r = hw + kq * (NT / 32);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tile_lo + r < tile_hi) v = ld4(a.Cs + (size_t)(tile_lo + r) * F + sub * 4);
            st4(sX + r * ldx + sub * 4, v);
        }
        __syncthreads();
    } else {
        pairnorm_tile<BM>(tile_lo, C, tm, a.cl_seg, a.cgseg, a.psum_c_rd, a.cgseg2, a.psum2_c, a.ncv, a.Cs, a.Cs, a.round == 0, sX, ldx,
                          nullptr, 0, sMean, sX + F, ldx, tid, a.inv_f, a.fm);   // its barriers also publish sPtr / sIdx
        // (the partial sums borrow columns 128.. of the tile's first rows, which the gather fills later)
    }
    DSAT_STAMP(a, 1);
    // R3 + R5 + R7: half-wave hw owns rows hw, hw+16, ...; all its rows (ROWS <= 4) are in
    // flight together, four literals each, so a 3-SAT tile needs one round trip to memory
    {
        const int sub = tid & 31, hw = tid >> 5;
        constexpr int U = ROWS;
#ifndef DSAT_GJ2
#define DSAT_GJ2 4
#endif
        constexpr int GJ = DSAT_GJ2;   // literal rows in flight per clause (x U rows x 2 arrays)
        float4 S[U], M[U];
        float rwv[U];
        int e0[U], e1[U];
        int nmax = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            S[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            M[u] = S[u];
            const int r = hw + u * (NT / 32);
            e0[u] = 0; e1[u] = 0;
            rwv[u] = tile_lo + r < tile_hi ? a.rev_w[tile_lo + r] : 0.f;
            if (tile_lo + r < tile_hi) {
                if (staged) { e0[u] = sPtr[r]; e1[u] = sPtr[r + 1]; }
                else { e0[u] = a.cg_rowptr[tile_lo + r]; e1[u] = a.cg_rowptr[tile_lo + r + 1]; }
            }
            nmax = max(nmax, e1[u] - e0[u]);
        }
        const unsigned zrow = 2u * (unsigned)NL;   // the all-zero row of sp / lits
        for (int off = 0; off < nmax; off += GJ) {   // branch-free, see the P3 gather
            unsigned l[U][GJ];
            float4 sv[U][GJ], lv[U][GJ];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const int e = e0[u] + off + j;
                    const bool ok = e < e1[u];
                    int idx;
                    if (staged) idx = sIdx[e];
                    else idx = ok ? a.cg_lit[e] : 0;
                    l[u][j] = ok ? (unsigned)idx : zrow;
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const unsigned o = l[u][j] * (unsigned)F + (unsigned)sub * 4u;
                    sv[u][j] = ld4(a.sp + o);
                    lv[u][j] = ld4(a.lits + o);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    S[u].x += sv[u][j].x; S[u].y += sv[u][j].y; S[u].z += sv[u][j].z; S[u].w += sv[u][j].w;
                    M[u].x += lv[u][j].x; M[u].y += lv[u][j].y; M[u].z += lv[u][j].z; M[u].w += lv[u][j].w;
                }
        }
        if (a.c_hv != nullptr) {   // heavy clauses: their sums come in pieces from heavy_gather_kernel
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = tile_lo + hw + u * (NT / 32);
                if (c < tile_hi) {
                    const int2 hv = a.c_hv[c];
                    for (int p = 0; p < hv.y; ++p) {
                        const float* hp = a.chpart + ((size_t)(hv.x + p) * 2) * F + sub * 4;
                        const float4 s4 = ld4(hp), m4 = ld4(hp + F);
                        S[u].x += s4.x; S[u].y += s4.y; S[u].z += s4.z; S[u].w += s4.w;
                        M[u].x += m4.x; M[u].y += m4.y; M[u].z += m4.z; M[u].w += m4.w;
                    }
                }
            }
        }
        // the last part of the tile preparation runs inside the chain, after it has asked for
        // its first weight fragments (run_chain's `prep`), and the chain takes the barrier
        auto finish = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = hw + u * (NT / 32);
            const int c = tile_lo + r;
            float4 cl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) {
                cl = make_float4(4.f * expf(-S[u].x), 4.f * expf(-S[u].y), 4.f * expf(-S[u].z),
                                 4.f * expf(-S[u].w));
                stg4(a.closs4 + (size_t)c * F + sub * 4, cl);
            }
            const float rw = rwv[u];
            const float4 cm = make_float4(M[u].x * rw, M[u].y * rw, M[u].z * rw, M[u].w * rw);
            st4(sX + r * ldx + F + sub * 4, cm);
            st4(sX + r * ldx + 2 * F + sub * 4, cl);
            if (a.tap_cmsg != nullptr && c < C) st4(a.tap_cmsg + (size_t)c * F + sub * 4, cm);
        }
        DSAT_STAMP(a, 2);
        };
        // R8-R9: clause_data = clause_mlp([clause_state, clause_messages, clauses_loss])
        float* vla = a.vla;
        float* ncv = a.ncv;
        float* stage = sX;   // the input tile is dead once layer 0 has run
        const int lds = F + 8;
        run_chain<RB>(a.mcl, sX, ldx, sA, a.ld_a, sB, a.ld_b, wave, lane, [=](int row, int col, float4 v) {
            int c = tile_lo + row;
            NO_HOIST(c);
            if (col < F) {
                if (c < C) stg4(vla + (size_t)c * F + col, v);
            } else if (col < 2 * F) {
                st4(stage + row * lds + (col - F), v);
                if (c < C) stg4(ncv + (size_t)c * F + (col - F), v);
            }
        }, finish);
        DSAT_STAMP(a, 3);
        tile_colsums<BM>(tile_lo, C, tm, a.cl_seg, a.psum_c_wr, stage, lds, sMean, tid, a.fm);
        DSAT_STAMP(a, 4);
        DSAT_STAMP_RT_END(a);
    }
}

// ------------------------------------------------------------------------------------- P2, split products
// The same clause tile with clause_mlp's products formed from bf16 slices (split_bf16.h): 2.5x fewer MFMA cycles
// at fp32-level error.  64 rows, 16 waves.  The operands are 6 bytes per element, so the 64 x 384 input tile no
// longer fits LDS as a whole: layer 0 takes its three 128-column segments [clause_state | messages | loss]
// (query_sat.py:258) through two buffers,
//     buf1: state (k-steps 0-3)  ->  loss (k-steps 8-11)          52 KiB
//     buf2: messages (k-steps 4-7)  ->  layer 0's output planes    87 KiB (+ the staged CSR lists behind them)
// with two extra barriers; the fp32 stage of the column sums reuses buf1.  Column tiles of layer 0: one per wave;
// with 13 tiles (the released 204-wide layer) waves 12-15 take one row block of tile 12 each, as chain_static's
// MIX0 does.  Weight fragments are plain loads one k-step ahead (the compiler counts vmcnt here).
#if DSAT_ROUND_NWAVES == 16
// ------------------------------------------------------------------------------------- P2, role D
// update_gate's layer 0 (query_sat.py:277-278) is linear in unit = [grad | variables | aux | loss_pos | loss_neg],
// and the k-steps that read `variables` (columns 128..255: k-steps 4..7 of 17) depend on nothing P2 or P3 compute.
// A 16-row P3 tile is bound by its weight stream, so these workgroups -- on the CUs the clause
// tiles leave idle, 64 rows per weight fragment -- form  variables @ W0[128:256, :]  once per round, and P3 starts
// its accumulators from it and streams 13 k-steps instead of 17.  Same slices, same k order inside the partial sum.
__device__ __forceinline__ void role_d(const RoundArgs& a, int blk, float* smem, int tid) {
    constexpr int BM = 64, KS = F / 32, LD = F + 16, GAP_AT = F / 32;
    const int wave = tid >> 6, lane = tid & 63;
    const int N = a.N;
    unsigned short* xp = reinterpret_cast<unsigned short*>(smem);   // [3][64][LD]
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)lane * 16u;
    const int col4 = 4 * (lane >> 4), row = lane & 15;
    const unsigned short* xin = xp + row * LD + 8 * (lane >> 4);
    const float4* wb = reinterpret_cast<const float4*>(a.d_ws) + ((size_t)(wv < a.d_nt0 ? wv : 0) * a.d_ks0 + GAP_AT) * 3 * 64;
    // the 64-row groups are dealt round-robin to the d_workers workgroups: the d_blocks extra ones and, behind their
    // own work, the select/SAT workgroups (a third of a tile's time at 20k nodes)
    for (int lo = blk * BM; lo < N; lo += a.d_workers * BM) {
        for (int i = tid; i < BM * (F / 4); i += NT) {
            const int r = i / (F / 4), c = 4 * (i % (F / 4));
            const float4 v = lo + r < N ? ld4(a.V + (size_t)(lo + r) * F + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            split_st4(xp, BM * LD, r * LD + c, v);
        }
        f32x4 acc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wv < a.d_nt0) {
            split_layer<4, DSAT_SPLIT_PD4>(voff, wb, KS, xin, BM * LD, LD, [] {}, acc);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                int n = lo + rb * 16 + row;
                NO_HOIST(n);
                if (n < N) stg4(a.upd_part + (size_t)n * a.upd_ld + wv * 16 + col4, make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]));
            }
        } else {
            split_layer<0, DSAT_SPLIT_PD4>(voff, nullptr, 0, nullptr, 0, 0, [] {}, reinterpret_cast<f32x4(&)[1]>(acc[0]));
        }
        lds_barrier();   // every wave is done with the planes before the next group's rows overwrite them
    }
}

// SHARE: the select parts of a large graph exchange their partial KL sums (select_sat_body); its own instantiation, so
// that the kernel of the batches without such graphs keeps its code (and register allocation) as it was
template <bool SHARE>
__global__ __launch_bounds__(NT, ROUND_WGS_PER_CU * NT / 256) void round_p2s_kernel(RoundArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    constexpr int RB = 4, BM = 16 * RB;
    constexpr int ROWS = (BM + NT / 32 - 1) / (NT / 32);   // clause rows per half-wave
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int C = a.C, NL = a.NL;
    const int tiles = (C + BM - 1) / BM;
    const int lead = a.s_enable ? a.sel_lead : 0;
    const int tb = (int)blockIdx.x - lead;
    const int nsb = a.sel_pack > 1 ? (a.n_sel + a.sel_pack - 1) / a.sel_pack : a.n_sel;   // select workgroups
    if (tb < 0 && (int)blockIdx.x >= nsb) return;
    const bool extra = tb < 0 || tb >= tiles;
    const int sel_blk = tb < 0 ? (int)blockIdx.x : tb - tiles;
    const int tile = extra ? 0 : xcd_tile(tb, tiles);
    const int done = a.check_done ? a.ctl->done : 0;   // requested together with the tile record
    const TileMetaDev tm = load_meta(a.cmeta, tile);
    ListBlock lb;                                       // ... and with the tile's clause -> literal lists
    list_block_fetch(lb, extra ? nullptr : a.clist, a.clist_stride, tile, tid);
    const int tile_lo = tile * BM;
    const int tile_hi = min(tile_lo + BM, C);
    // ... and with the carried clause state of the tile's rows (finished by P1's extra blocks)
    const bool rows_ready = !extra && a.cs_ready && a.round > 0 && (a.cs_ready_rows <= 0 || tile_hi <= a.cs_ready_rows);
    float4 srow[ROWS];
#pragma unroll
    for (int kq = 0; kq < ROWS; ++kq) {
        const int r = (tid >> 5) + kq * (NT / 32);
        srow[kq] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rows_ready && tile_lo + r < tile_hi) srow[kq] = ld4(a.Cs + (size_t)(tile_lo + r) * F + (tid & 31) * 4);
    }
    if (both_arrived(done, tm) != 0) return;
    if (extra) {
        // role D: the last d_blocks blocks of the grid (they are `extra` blocks: behind the tiles)
        const int dblk = (int)blockIdx.x - ((int)gridDim.x - a.d_blocks);
        if (a.d_blocks > 0 && dblk >= 0) {
            role_d(a, a.d_workers - a.d_blocks + dblk, smem, tid);
            return;
        }
        DSAT_STAMP_RT_BEGIN(a);
        DSAT_STAMP(a, 0);
        if (a.sel_pack > 1) {   // four graphs per workgroup, 256 threads each (large batches of small graphs: a select
            // workgroup per graph would be a whole extra wave of workgroups, 256 at config 3)
            const int sub = tid >> 8, e = sel_blk * 4 + sub;
            const bool act = e < a.n_sel;
            select_sat_body<256>(act ? e : a.n_sel - 1, a.n_sel, a.sel_tab, a.var_seg, a.cl_seg, a.c_rowptr, a.c_lit,
                                 a.logits, a.ctl, a.pgl, a.best, a.out_logits, a.assign, a.graph_unsat, NL, a.round - 1,
                                 a.check_done, smem + sub * 2560, tid & 255, act);
        } else
        select_sat_body<NT, false, SHARE>(sel_blk, a.n_sel, a.sel_tab, a.var_seg, a.cl_seg, a.c_rowptr, a.c_lit, a.logits,
                                          a.ctl, a.pgl, a.best, a.out_logits, a.assign, a.graph_unsat, NL, a.round - 1,
                                          a.check_done, smem, tid, true, a.sel_exch, a.sel_ticks);
        DSAT_STAMP(a, 4);
        DSAT_STAMP_RT_END(a);
        if (a.d_sel != 0 && sel_blk < a.n_sel) {   // ... then this workgroup's share of role D
            __syncthreads();
            role_d(a, sel_blk, smem, tid);
        }
        return;
    }
    unsigned char* sbytes = reinterpret_cast<unsigned char*>(smem);
    unsigned short* buf1 = reinterpret_cast<unsigned short*>(sbytes + a.s_buf1);
    unsigned short* buf2 = reinterpret_cast<unsigned short*>(sbytes + a.s_buf2);
    int* sPtr = reinterpret_cast<int*>(sbytes + a.s_idx);
    int* sIdx = sPtr + BM + 8;
    bool staged;
    float* sMean = reinterpret_cast<float*>(sbytes + a.s_mean);
    const int ldg = a.s_ld_seg, ldh = a.s_ld_hid;
    const int pl_seg = BM * ldg, pl_hid = BM * ldh;   // elements per plane
    const int sub = tid & 31, hw = tid >> 5;

    DSAT_STAMP_RT_BEGIN(a);
    DSAT_STAMP(a, 0);
    if (a.clist != nullptr) {   // the block is [ptr | list]
        list_block_store(lb, sPtr, a.clist_stride, tid);
        sIdx = sPtr + BM + 1;
        staged = tm.e_end0 - tm.e_base0 <= a.clist_max;
    } else {
        staged = stage_csr(a.cg_rowptr, a.cg_lit, tile_lo, tile_hi - tile_lo, tm.e_base0, tm.e_end0, sPtr, sIdx,
                           a.s_idx_cap, tid);
    }
    if (rows_ready) {   // R10 was finished by P1's extra blocks: slice the rows loaded above
#pragma unroll
        for (int kq = 0; kq < ROWS; ++kq) split_st4(buf1, pl_seg, (hw + kq * (NT / 32)) * ldg + sub * 4, srow[kq]);
        __syncthreads();
    } else {   // PairNorm here, in fp32 (buf2 is free: rows, then the thread groups' partial sums), then the slices
        float* scratch = reinterpret_cast<float*>(buf2);
        constexpr int lds_ = F + 8;
        pairnorm_tile<BM>(tile_lo, C, tm, a.cl_seg, a.cgseg, a.psum_c_rd, a.cgseg2, a.psum2_c, a.ncv, a.Cs, a.Cs, a.round == 0,
                          scratch, lds_, nullptr, 0, sMean, scratch + BM * lds_, lds_, tid, a.inv_f, a.fm);
#pragma unroll
        for (int kq = 0; kq < ROWS; ++kq) {
            const int r = hw + kq * (NT / 32);
            split_st4(buf1, pl_seg, r * ldg + sub * 4, ld4(scratch + r * lds_ + sub * 4));
        }
        __syncthreads();
    }
    DSAT_STAMP(a, 1);
    // R3 + R5 + R7 (the gather of round_p2_kernel)
    constexpr int U = ROWS;
    constexpr int GJ = DSAT_GJ2;
    float4 S[U], M[U];
    float rwv[U];
    {
        int e0[U], e1[U];
        int nmax = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            S[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            M[u] = S[u];
            const int r = hw + u * (NT / 32);
            e0[u] = 0; e1[u] = 0;
            rwv[u] = tile_lo + r < tile_hi ? a.rev_w[tile_lo + r] : 0.f;
            if (tile_lo + r < tile_hi) {
                if (staged) { e0[u] = sPtr[r]; e1[u] = sPtr[r + 1]; }
                else { e0[u] = a.cg_rowptr[tile_lo + r]; e1[u] = a.cg_rowptr[tile_lo + r + 1]; }
            }
            nmax = max(nmax, e1[u] - e0[u]);
        }
        const unsigned zrow = 2u * (unsigned)NL;   // the all-zero row of sp / lits
        // literal rows in flight per clause: four -- or three when no clause of the batch is longer (3-SAT: a fourth
        // slot would load the zero row a quarter of the time; 355 -> 358 steps/s at config 2)
        auto gather = [&](auto gj) {
        constexpr int GJ = decltype(gj)::value;
        for (int off = 0; off < nmax; off += GJ) {
            unsigned l[U][GJ];
            float4 sv[U][GJ], lv[U][GJ];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const int e = e0[u] + off + j;
                    const bool ok = e < e1[u];
                    int idx;
                    if (staged) idx = sIdx[e];
                    else idx = ok ? a.cg_lit[e] : 0;
                    l[u][j] = ok ? (unsigned)idx : zrow;
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const unsigned o = l[u][j] * (unsigned)F + (unsigned)sub * 4u;
                    sv[u][j] = ld4(a.sp + o);
                    lv[u][j] = ld4(a.lits + o);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    S[u].x += sv[u][j].x; S[u].y += sv[u][j].y; S[u].z += sv[u][j].z; S[u].w += sv[u][j].w;
                    M[u].x += lv[u][j].x; M[u].y += lv[u][j].y; M[u].z += lv[u][j].z; M[u].w += lv[u][j].w;
                }
        }
        };
        if (a.gj2_three) gather(std::integral_constant<int, 3>{});
        else gather(std::integral_constant<int, DSAT_GJ2>{});
        if (a.c_hv != nullptr) {   // heavy clauses: their sums come in pieces from heavy_gather_kernel
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = tile_lo + hw + u * (NT / 32);
                if (c < tile_hi) {
                    const int2 hv = a.c_hv[c];
                    for (int p = 0; p < hv.y; ++p) {
                        const float* hp = a.chpart + ((size_t)(hv.x + p) * 2) * F + sub * 4;
                        const float4 s4 = ld4(hp), m4 = ld4(hp + F);
                        S[u].x += s4.x; S[u].y += s4.y; S[u].z += s4.z; S[u].w += s4.w;
                        M[u].x += m4.x; M[u].y += m4.y; M[u].z += m4.z; M[u].w += m4.w;
                    }
                }
            }
        }
    }
    // ---- layer 0: roles, first weight fragments, biases
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ks0 = a.sp_ks[0], ks1 = a.sp_ks[1], nt0 = a.sp_nt[0], nt1 = a.sp_nt[1];
    const bool mixed = nt0 == NT / 64 - 3;            // 13 tiles on 16 waves
    const bool full0 = mixed ? wv < NT / 64 - 4 : wv < nt0;
    const bool part0 = mixed && !full0;
    const int tile0 = full0 ? wv : nt0 - 1;
    const int prb = part0 ? wv - (NT / 64 - 4) : 0;
    const bool on0 = full0 || part0, on1 = wv < nt1;
    constexpr int PD = DSAT_SPLIT_PD4;
    const unsigned voff = (unsigned)lane * 16u;
    const float4* wb0 = reinterpret_cast<const float4*>(a.sp_ws[0]) + ((size_t)tile0 * ks0 * 3) * 64;
    const float4* wb1 = reinterpret_cast<const float4*>(a.sp_ws[1]) + ((size_t)(on1 ? wv : 0) * ks1 * 3) * 64;
    float4 bv0 = make_float4(0.f, 0.f, 0.f, 0.f), bv1 = bv0;
    if (on0) bv0 = *reinterpret_cast<const float4*>(a.mcl.L[0].bias + tile0 * 16 + 4 * (lane >> 4));
    if (on1) bv1 = *reinterpret_cast<const float4*>(a.mcl.L[1].bias + wv * 16 + 4 * (lane >> 4));
    asm volatile("" : "+v"(bv0.x), "+v"(bv0.y), "+v"(bv0.z), "+v"(bv0.w), "+v"(bv1.x), "+v"(bv1.y), "+v"(bv1.z), "+v"(bv1.w));
    // ---- layer 0: [state | messages | loss] through buf1 / buf2 / buf1; one loop nest per role
    float4 clv[U];
    auto finish = [&]() {   // R5 / R7 results: messages -> buf2 now, the loss rows wait for buf1
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = hw + u * (NT / 32);
            const int c = tile_lo + r;
            float4 cl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) cl = make_float4(4.f * expf(-S[u].x), 4.f * expf(-S[u].y), 4.f * expf(-S[u].z), 4.f * expf(-S[u].w));
            clv[u] = cl;   // (stored to closs4 at the end of the tile: a write-through store here would sit in front of
                           // the weight ring's counted waits until it is acknowledged)
            const float rw = rwv[u];
            const float4 cm = make_float4(M[u].x * rw, M[u].y * rw, M[u].z * rw, M[u].w * rw);
            split_st4(buf2, pl_seg, r * ldg + sub * 4, cm);
            if (a.tap_cmsg != nullptr && c < C) st4(a.tap_cmsg + (size_t)c * F + sub * 4, cm);
        }
        DSAT_STAMP(a, 2);
    };
    auto mid = [&]() {
        lds_barrier();   // every wave is done with buf1's state columns
#pragma unroll
        for (int u = 0; u < U; ++u) split_st4(buf1, pl_seg, (hw + u * (NT / 32)) * ldg + sub * 4, clv[u]);
        lds_barrier();
    };
    constexpr int KSEG = F / 32;   // k-steps per 128-column segment
    static_assert(KSEG % PD == 0, "a segment is a whole number of ring turns");
    const int xoff = (lane & 15) * ldg + 8 * (lane >> 4);
    f32x4 acc0[RB], accp[1];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc0[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    accp[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (full0)
        split_layer_segments<RB, PD>(voff, wb0, KSEG, buf1 + xoff, buf2 + xoff, pl_seg, ldg, finish, mid, acc0);
    else if (part0)
        split_layer_segments<1, PD>(voff, wb0, KSEG, buf1 + xoff + prb * 16 * ldg, buf2 + xoff + prb * 16 * ldg, pl_seg, ldg,
                                    finish, mid, accp);
    else
        split_layer_segments<0, PD>(voff, wb0, KSEG, buf1, buf2, pl_seg, ldg, finish, mid, accp);
    DSAT_STAMP(a, 5);
    // ---- layer 1 (one column tile per wave); its prep is layer 0's epilogue: slices of the hidden tile go to buf2
    // (nobody reads the messages any more)
    auto epilogue0 = [&]() {
        const int col = tile0 * 16 + 4 * (lane >> 4);
        if (full0) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                split_st4(buf2, pl_hid, (rb * 16 + (lane & 15)) * ldh + col, bias_act(acc0[rb], bv0, 1));
        } else if (part0) {
            split_st4(buf2, pl_hid, (prb * 16 + (lane & 15)) * ldh + col, bias_act(accp[0], bv0, 1));
        }
        // columns between the last tile and the end of layer 1's last k-step: zeros (finite operands)
        const int c0 = nt0 * 16, c1 = ks1 * 32, nq = (c1 - c0) >> 2;
        for (int i = tid; i < BM * nq; i += NT) split_zero4(buf2, pl_hid, (i / nq) * ldh + c0 + 4 * (i % nq));
    };
    f32x4 acc1[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc1[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* stage = reinterpret_cast<float*>(buf1);   // layer 0 is over: buf1 is free
    constexpr int lds = F + 8;
    if (on1) {
        split_layer<RB, PD>(voff, wb1, ks1, buf2 + (lane & 15) * ldh + 8 * (lane >> 4), pl_hid, ldh, epilogue0, acc1);
        DSAT_STAMP(a, 6);
        const int col = wv * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int row = rb * 16 + (lane & 15);
            int c = tile_lo + row;
            NO_HOIST(c);
            const float4 v = bias_act(acc1[rb], bv1, 0);
            if (col < F) {
                if (c < C) stg4(a.vla + (size_t)c * F + col, v);
            } else if (col < 2 * F) {
                st4(stage + row * lds + (col - F), v);
                if (c < C) stg4(a.ncv + (size_t)c * F + (col - F), v);
            }
        }
    } else {
        split_layer<0, PD>(voff, wb1, ks1, buf2, pl_hid, ldh, epilogue0, accp);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = tile_lo + hw + u * (NT / 32);
        if (c < C) stg4(a.closs4 + (size_t)c * F + sub * 4, clv[u]);
    }
    lds_barrier();   // the stage is complete; the global stores drain while the sums are taken
    DSAT_STAMP(a, 3);
    tile_colsums<BM>(tile_lo, C, tm, a.cl_seg, a.psum_c_wr, stage, lds, sMean, tid, a.fm);
    DSAT_STAMP(a, 4);
    DSAT_STAMP_RT_END(a);
}
#endif

// ------------------------------------------------------------------------------------- P3
__global__ __launch_bounds__(NT, ROUND_WGS_PER_CU * NT / 256) void round_p3_kernel(RoundArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    constexpr int BM = ROUND_BM_VAR;
    constexpr int PASSES = BM / ROUND_NWAVES;   // variables per wave
#ifndef DSAT_GJ3
#define DSAT_GJ3 4
#endif
    constexpr int GJ = DSAT_GJ3;                // clause rows in flight per literal (x PASSES x 2 arrays)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int N = a.N, NL = a.NL;
    const P3Slot slot = p3_slot(a);                    // a tile, or role C (see round_p3s_kernel)
    const bool roleC = slot.is_c || slot.idx >= a.p3_tiles;
    const int tile = roleC ? 0 : xcd_tile(slot.idx, a.p3_tiles);
    const int done = a.check_done ? a.ctl->done : 0;   // requested together with the tile record
    const TileMetaDev tm = load_meta(a.vmeta, tile);
    ListBlock lb;                                       // ... and with the tile's literal -> clause lists
    list_block_fetch(lb, roleC ? nullptr : a.vlist, a.vlist_stride, tile, tid);
    if (both_arrived(done, tm) != 0) return;
    if (roleC) {
        DSAT_STAMP_RT_BEGIN(a);
        DSAT_STAMP(a, 0);
        if (slot.is_c && slot.idx >= 0) clause_norm_block(a, slot.idx, a.cn_blocks, smem, tid);
        DSAT_STAMP(a, 4);
        DSAT_STAMP_RT_END(a);
        return;
    }
    if (a.sel_zero && blockIdx.x == 0)   // multi-part select adds its counts up: start the next one from zero
        for (int g = tid; g < a.G; g += NT) a.graph_unsat[g] = 0;
    const int tile_lo = tile * BM;
    const int tile_hi = min(tile_lo + BM, N);
    float* sX = smem + a.off_x;
    float* sA = smem + a.off_a;
    float* sB = smem + a.off_b;
    float* sPart = smem + a.off_mean;
    const int ldx = a.ld_x;
    // unit = [variables_grad | variables | aux | loss_pos | loss_neg]        (query_sat.py:277)
    constexpr int OFF_V = F, OFF_AUX = 2 * F, OFF_POS = 2 * F + AUX_W, OFF_NEG = 3 * F + AUX_W;
    DSAT_STAMP_RT_BEGIN(a);
    DSAT_STAMP(a, 0);
    DSAT_STAMP(a, 4);
    {   // R4 + R11: one wave per variable, half 0 = literal +v, half 1 = literal -v
        const int sub = lane & 31, half = lane >> 5;
        // literal -> clause lists of the tile's +v rows and -v rows -> LDS (sA is free until layer 0)
        int* sPtr0 = reinterpret_cast<int*>(sA);
        int* sPtr1 = sPtr0 + BM + 8;
        int* sIdx0 = sPtr1 + BM + 8;
        const int cap = (BM * a.ld_a - 2 * (BM + 8)) / 2;
        int* sIdx1 = sIdx0 + cap;
        const int nrows = tile_hi - tile_lo;
        bool staged;
        if (a.vlist != nullptr) {   // the block is [ptr0 | ptr1 (offset by list 0's length) | list 0 | list 1]
            list_block_store(lb, sPtr0, a.vlist_stride, tid);
            sPtr1 = sPtr0 + BM + 1;
            sIdx0 = sPtr1 + BM + 1;
            sIdx1 = sIdx0;
            staged = (tm.e_end0 - tm.e_base0) + (tm.e_end1 - tm.e_base1) <= a.vlist_max;
        } else {
            const bool st0 = stage_csr(a.l_rowptr, a.l_clause, tile_lo, nrows, tm.e_base0, tm.e_end0, sPtr0, sIdx0, cap, tid);
            const bool st1 = stage_csr(a.l_rowptr, a.l_clause, NL + tile_lo, nrows, tm.e_base1, tm.e_end1, sPtr1, sIdx1, cap, tid);
            staged = st0 && st1;
        }
        // rows that do not depend on the lists: q (half 0) and the carried state (half 1)
        float4 rowv[PASSES], qrow[PASSES];
        float wdeg[PASSES], wv[PASSES];
        int2 hv[PASSES];
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const int v = tile_lo + wave + u * ROUND_NWAVES;
            rowv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            qrow[u] = rowv[u];
            wdeg[u] = 0.f; wv[u] = 0.f;
            hv[u] = make_int2(-1, 0);
            if (v < N) {
                if (a.l_hv != nullptr) hv[u] = a.l_hv[v + half * NL];
                qrow[u] = ld4(a.q + (size_t)v * F + sub * 4);
                if (half == 1) rowv[u] = ld4(a.V + (size_t)v * F + sub * 4);
                wdeg[u] = a.deg_w[v + half * NL];
                wv[u] = a.vdeg_w[v];
            }
        }
        if (tid < BM) {
            const int n = tile_lo + tid;
            load_aux_row(sX + tid * ldx + OFF_AUX, a.aux, n, N, a.round, n < N);
        }
        __syncthreads();   // lists staged
        DSAT_STAMP(a, 5);
        float4 M[PASSES], VL[PASSES];
        int e0[PASSES], e1[PASSES];
        int nmax = 0;
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            M[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            VL[u] = M[u];
            const int r = wave + u * ROUND_NWAVES;
            e0[u] = 0; e1[u] = 0;
            if (tile_lo + r < N) {
                if (staged) {
                    const int* sp_ = half ? sPtr1 : sPtr0;
                    e0[u] = sp_[r]; e1[u] = sp_[r + 1];
                } else {
                    const int lit = tile_lo + r + half * NL;
                    e0[u] = a.l_rowptr[lit]; e1[u] = a.l_rowptr[lit + 1];
                }
            }
            nmax = max(nmax, e1[u] - e0[u]);
        }
        nmax = max(nmax, __shfl_xor(nmax, 32));   // both halves run the same trip count
        const int* sIdx = half ? sIdx1 : sIdx0;
        // Branch-free: a conditional load would make the compiler wait for memory at every
        // join, serialising the rows.  List slots past a row's end read the all-zero row C
        // (dsat_api.hip carve), so nothing is masked; offsets are 32-bit (SGPR base + VGPR).
        const unsigned zrow = (unsigned)a.CL;
        for (int off = 0; off < nmax; off += GJ) {
            unsigned ci[PASSES][GJ];
            float4 cv[PASSES][GJ], vv[PASSES][GJ];
#pragma unroll
            for (int u = 0; u < PASSES; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const int e = e0[u] + off + j;
                    const bool ok = e < e1[u];
                    int idx;
                    if (staged) idx = sIdx[e];              // LDS: reading past the list is harmless
                    else idx = ok ? a.l_clause[e] : 0;      // rare path (lists larger than the LDS budget)
                    ci[u][j] = ok ? (unsigned)idx : zrow;
                }
#pragma unroll
            for (int u = 0; u < PASSES; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const unsigned o = ci[u][j] * (unsigned)F + (unsigned)sub * 4u;
                    cv[u][j] = ld4(a.closs4 + o);
                    vv[u][j] = ld4(a.vla + o);
                }
#pragma unroll
            for (int u = 0; u < PASSES; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    M[u].x += cv[u][j].x; M[u].y += cv[u][j].y; M[u].z += cv[u][j].z; M[u].w += cv[u][j].w;
                    VL[u].x += vv[u][j].x; VL[u].y += vv[u][j].y; VL[u].z += vv[u][j].z; VL[u].w += vv[u][j].w;
                }
        }
        if (a.l_hv != nullptr) {   // heavy rows: their sums come in pieces from heavy_gather_kernel
#pragma unroll
            for (int u = 0; u < PASSES; ++u) {
                for (int p = 0; p < hv[u].y; ++p) {
                    const float* hp = a.hpart + ((size_t)(hv[u].x + p) * 2) * F + sub * 4;
                    const float4 m4 = ld4(hp), v4 = ld4(hp + F);
                    M[u].x += m4.x; M[u].y += m4.y; M[u].z += m4.z; M[u].w += m4.w;
                    VL[u].x += v4.x; VL[u].y += v4.y; VL[u].z += v4.z; VL[u].w += v4.w;
                }
            }
        }
        DSAT_STAMP(a, 6);
        // the last part of the tile preparation runs inside the chain, after it has asked for
        // its first weight fragments (run_chain's `prep`), and the chain takes the barrier
        auto finish = [&]() {
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const int r = wave + u * ROUND_NWAVES;
            const float dw = wdeg[u];
            const float4 vls = make_float4(VL[u].x * dw, VL[u].y * dw, VL[u].z * dw, VL[u].w * dw);
            st4(sX + r * ldx + (half ? OFF_NEG : OFF_POS) + sub * 4, vls);
            if (a.tap_vloss != nullptr && tile_lo + r < N)
                st4(a.tap_vloss + ((size_t)(tile_lo + r) + (size_t)half * NL) * F + sub * 4, vls);
            float4 Mo;
            Mo.x = __shfl_xor(M[u].x, 32); Mo.y = __shfl_xor(M[u].y, 32);
            Mo.z = __shfl_xor(M[u].z, 32); Mo.w = __shfl_xor(M[u].w, 32);
            {   // closed form of the inner tape (closs4 = 4*closs, 0.25 is exact); each half-wave
                // computes two of its lane's four columns: half 0 -> .x .y, half 1 -> .z .w
                const float4 qv = qrow[u];
                const float vw = wv[u];
                const float q0 = half ? qv.z : qv.x, q1 = half ? qv.w : qv.y;
                const float p0 = half ? Mo.z : M[u].x, p1 = half ? Mo.w : M[u].y;   // sums over +v clauses
                const float n0 = half ? M[u].z : Mo.x, n1 = half ? M[u].w : Mo.y;   // sums over -v clauses
                float s0, sn0, s1, sn1;
                sigmoid_pair(q0, s0, sn0);
                sigmoid_pair(q1, s1, sn1);
                float2 g;
                g.x = (-s0 * (p0 * 0.25f) + sn0 * (n0 * 0.25f)) * vw;
                g.y = (-s1 * (p1 * 0.25f) + sn1 * (n1 * 0.25f)) * vw;
                *reinterpret_cast<float2*>(sX + r * ldx + sub * 4 + 2 * half) = g;
                if (a.tap_grad != nullptr && tile_lo + r < N)
                    *reinterpret_cast<float2*>(a.tap_grad + (size_t)(tile_lo + r) * F + sub * 4 + 2 * half) = g;
            }
            if (half == 1) st4(sX + r * ldx + OFF_V + sub * 4, rowv[u]);
        }
        DSAT_STAMP(a, 1);
        };
        // R12: new_variables = update_gate(unit)
        float* newv = a.newv;
        float* stage = sX;
        const int lds = F + 8;
        run_chain<1>(a.mupd, sX, ldx, sA, a.ld_a, sB, a.ld_b, wave, lane, [=](int row, int col, float4 v) {
            int n = tile_lo + row;
            NO_HOIST(n);
            if (col < F) {
                st4(stage + row * lds + col, v);
                if (n < N) stg4(newv + (size_t)n * F + col, v);
            }
        }, finish);
        DSAT_STAMP(a, 2);
        tile_colsums<BM>(tile_lo, N, tm, a.var_seg, a.psum_v, stage, lds, sPart, tid, a.fm);
        DSAT_STAMP(a, 3);
        DSAT_STAMP_RT_END(a);
    }
}

// ------------------------------------------------------------------------------------- P3, split products
// round_p3_kernel with update_gate's products formed from bf16 slices (split_bf16.h).  The unit
// [grad | variables | aux | loss_pos | loss_neg] is written to LDS as three bf16 planes (16 x 544 + pad), the
// hidden tiles likewise; the CSR lists are staged in the hidden tiles' space, the fp32 stage of the column sums
// reuses the input planes.  With 16-row tiles the chain is bound by the weight stream (6 bytes per weight through
// the CU's 64 B/clk vector-memory path), not by the MFMA pipe.
#if DSAT_ROUND_NWAVES == 16
template <int BM, bool HOSTC>   // BM: 16 rows, or 32 (two row blocks per weight fragment: half the weight stream per row; the
                    // PairNorm segment tables of the session are then built for 32-row producer tiles).  HOSTC: the grid
                    // also holds role-C blocks (cn_blocks > 0); the tiles-only form pays nothing for their dispatch
                    // (0.45 % of a config-2 step when it was a runtime test)
__global__ __launch_bounds__(NT, ROUND_WGS_PER_CU * NT / 256) void round_p3s_kernel(RoundArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    constexpr int PASSES = BM / ROUND_NWAVES;   // variables per wave
#ifndef DSAT_GJ3
#define DSAT_GJ3 4
#endif
    constexpr int GJ = BM > 16 ? (DSAT_GJ3 + 1) / 2 : DSAT_GJ3;   // clause rows in flight per literal (x PASSES x 2 arrays)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int N = a.N, NL = a.NL;
    // role C: the blocks behind the tiles finish the clause PairNorm (R10) of THIS round for P2 of the next one, on the
    // CUs the last wave of tiles leaves idle (dsat_api.hip, cn3)
    const P3Slot slot = HOSTC ? p3_slot(a) : P3Slot{false, (int)blockIdx.x};
    const bool roleC = HOSTC && (slot.is_c || slot.idx >= a.p3_tiles);
    const int tile = roleC ? 0 : xcd_tile(slot.idx, HOSTC ? a.p3_tiles : (int)gridDim.x);
    const int done = a.check_done ? a.ctl->done : 0;   // requested together with the tile record
    const TileMetaDev tm = load_meta(a.vmeta, tile);
    ListBlock lb;                                       // ... and with the tile's literal -> clause lists
    list_block_fetch(lb, roleC ? nullptr : a.vlist, a.vlist_stride, tile, tid);
    const int tile_lo = tile * BM;
    const int tile_hi = roleC ? tile_lo : min(tile_lo + BM, N);   // (role C: none of the row prefetches below)
    // ... and with the rows that do not depend on the lists: q (half 0) and the carried state (half 1)
    float4 rowv[PASSES], qrow[PASSES];
    float wdeg[PASSES], wv[PASSES];
    int2 hv[PASSES];
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
        const int v = tile_lo + wave + u * ROUND_NWAVES, half_ = lane >> 5, sub_ = lane & 31;
        rowv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        qrow[u] = rowv[u];
        wdeg[u] = 0.f; wv[u] = 0.f;
        hv[u] = make_int2(-1, 0);
        if (v < tile_hi) {
            if (a.l_hv != nullptr) hv[u] = a.l_hv[v + half_ * NL];
            qrow[u] = ld4(a.q + (size_t)v * F + sub_ * 4);
            if (half_ == 1) rowv[u] = ld4(a.V + (size_t)v * F + sub_ * 4);
            wdeg[u] = a.deg_w[v + half_ * NL];
            wv[u] = a.vdeg_w[v];
        }
    }
    // ... and with the start values of update_gate's layer 0 (role D of P2 formed the `variables` k-steps): this lane's
    // four columns of its wave's column tile, row l & 15 of each row block
    float4 upd0[BM / 16];
#pragma unroll
    for (int rb = 0; rb < BM / 16; ++rb) {
        upd0[rb] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int n = tile_lo + rb * 16 + (lane & 15);
        if (a.upd_part != nullptr && wave < a.sp_nt[0] && n < tile_hi)
            upd0[rb] = ld4(a.upd_part + (size_t)n * a.upd_ld + wave * 16 + 4 * (lane >> 4));
    }
    if (both_arrived(done, tm) != 0) return;
    if (roleC) {
        DSAT_STAMP_RT_BEGIN(a);
        DSAT_STAMP(a, 0);
        if (slot.is_c && slot.idx >= 0) clause_norm_block(a, slot.idx, a.cn_blocks, smem, tid);
        DSAT_STAMP(a, 4);
        DSAT_STAMP_RT_END(a);
        return;
    }
    if (a.sel_zero && blockIdx.x == 0)   // multi-part select adds its counts up: start the next one from zero
        for (int g = tid; g < a.G; g += NT) a.graph_unsat[g] = 0;
    unsigned char* sbytes = reinterpret_cast<unsigned char*>(smem);
    unsigned short* xs = reinterpret_cast<unsigned short*>(sbytes + a.s_buf1);
    const int ldx = a.s_ld_seg, ldh = a.s_ld_hid;
    const int xpl = BM * ldx;                          // elements per input plane
    unsigned short* hA = reinterpret_cast<unsigned short*>(sbytes + a.s_buf2);
    unsigned short* hB = reinterpret_cast<unsigned short*>(sbytes + a.s_bufB);   // 32 rows: over the input planes
    float* sPart = reinterpret_cast<float*>(sbytes + a.s_mean);
    // unit = [variables_grad | variables | aux | loss_pos | loss_neg]        (query_sat.py:277)
    constexpr int OFF_V = F, OFF_AUX = 2 * F, OFF_POS = 2 * F + AUX_W, OFF_NEG = 3 * F + AUX_W;
    DSAT_STAMP_RT_BEGIN(a);
    DSAT_STAMP(a, 0);
    DSAT_STAMP(a, 4);
    {   // R4 + R11: one wave per variable, half 0 = literal +v, half 1 = literal -v
        const int sub = lane & 31, half = lane >> 5;
        // literal -> clause lists of the tile's +v rows and -v rows -> LDS (the hidden tiles are free until layer 0)
        int* sPtr0 = reinterpret_cast<int*>(sbytes + a.s_idx);
        int* sPtr1 = sPtr0 + BM + 8;
        int* sIdx0 = sPtr1 + BM + 8;
        const int cap = a.s_idx_cap;
        int* sIdx1 = sIdx0 + cap;
        const int nrows = tile_hi - tile_lo;
        bool staged;
        if (a.vlist != nullptr) {   // the block is [ptr0 | ptr1 (offset by list 0's length) | list 0 | list 1]
            list_block_store(lb, sPtr0, a.vlist_stride, tid);
            sPtr1 = sPtr0 + BM + 1;
            sIdx0 = sPtr1 + BM + 1;
            sIdx1 = sIdx0;
            staged = (tm.e_end0 - tm.e_base0) + (tm.e_end1 - tm.e_base1) <= a.vlist_max;
        } else {
            const bool st0 = stage_csr(a.l_rowptr, a.l_clause, tile_lo, nrows, tm.e_base0, tm.e_end0, sPtr0, sIdx0, cap, tid);
            const bool st1 = stage_csr(a.l_rowptr, a.l_clause, NL + tile_lo, nrows, tm.e_base1, tm.e_end1, sPtr1, sIdx1, cap, tid);
            staged = st0 && st1;
        }
        if (tid < 4 * BM) {   // aux columns of the rows (one float4 per thread), and the zero columns up to the end of
            const int r = tid >> 2, q4 = tid & 3;   // the last k-step
            const int n = tile_lo + r;
            static_assert(AUX_W == 16, "aux row");
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < N) v = ld4(a.aux + ((size_t)a.round * N + n) * AUX_W + 4 * q4);
            split_st4(xs, xpl, r * ldx + OFF_AUX + 4 * q4, v);
            for (int c = OFF_NEG + F + 4 * q4; c < a.sp_ks[0] * 32; c += 16) split_zero4(xs, xpl, r * ldx + c);
        }
        __syncthreads();   // lists staged
        DSAT_STAMP(a, 5);
        float4 M[PASSES], VL[PASSES];
        int e0[PASSES], e1[PASSES];
        int nmax = 0;
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            M[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            VL[u] = M[u];
            const int r = wave + u * ROUND_NWAVES;
            e0[u] = 0; e1[u] = 0;
            if (tile_lo + r < N) {
                if (staged) {
                    const int* sp_ = half ? sPtr1 : sPtr0;
                    e0[u] = sp_[r]; e1[u] = sp_[r + 1];
                } else {
                    const int lit = tile_lo + r + half * NL;
                    e0[u] = a.l_rowptr[lit]; e1[u] = a.l_rowptr[lit + 1];
                }
            }
            nmax = max(nmax, e1[u] - e0[u]);
        }
        nmax = max(nmax, __shfl_xor(nmax, 32));   // both halves run the same trip count
        const int* sIdx = half ? sIdx1 : sIdx0;
        // Branch-free: a conditional load would make the compiler wait for memory at every
        // join, serialising the rows.  List slots past a row's end read the all-zero row C
        // (dsat_api.hip carve), so nothing is masked; offsets are 32-bit (SGPR base + VGPR).
        const unsigned zrow = (unsigned)a.CL;
        for (int off = 0; off < nmax; off += GJ) {
            unsigned ci[PASSES][GJ];
            float4 cv[PASSES][GJ], vv[PASSES][GJ];
#pragma unroll
            for (int u = 0; u < PASSES; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const int e = e0[u] + off + j;
                    const bool ok = e < e1[u];
                    int idx;
                    if (staged) idx = sIdx[e];              // LDS: reading past the list is harmless
                    else idx = ok ? a.l_clause[e] : 0;      // rare path (lists larger than the LDS budget)
                    ci[u][j] = ok ? (unsigned)idx : zrow;
                }
#pragma unroll
            for (int u = 0; u < PASSES; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const unsigned o = ci[u][j] * (unsigned)F + (unsigned)sub * 4u;
                    cv[u][j] = ld4(a.closs4 + o);
                    vv[u][j] = ld4(a.vla + o);
                }
#pragma unroll
            for (int u = 0; u < PASSES; ++u)
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    M[u].x += cv[u][j].x; M[u].y += cv[u][j].y; M[u].z += cv[u][j].z; M[u].w += cv[u][j].w;
                    VL[u].x += vv[u][j].x; VL[u].y += vv[u][j].y; VL[u].z += vv[u][j].z; VL[u].w += vv[u][j].w;
                }
        }
        if (a.l_hv != nullptr) {   // heavy rows: their sums come in pieces from heavy_gather_kernel
#pragma unroll
            for (int u = 0; u < PASSES; ++u) {
                for (int p = 0; p < hv[u].y; ++p) {
                    const float* hp = a.hpart + ((size_t)(hv[u].x + p) * 2) * F + sub * 4;
                    const float4 m4 = ld4(hp), v4 = ld4(hp + F);
                    M[u].x += m4.x; M[u].y += m4.y; M[u].z += m4.z; M[u].w += m4.w;
                    VL[u].x += v4.x; VL[u].y += v4.y; VL[u].z += v4.z; VL[u].w += v4.w;
                }
            }
        }
        DSAT_STAMP(a, 6);
        // the last part of the tile preparation runs inside the chain, after it has asked for
        // its first weight fragments (run_chain's `prep`), and the chain takes the barrier
        auto finish = [&]() {
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const int r = wave + u * ROUND_NWAVES;
            const float dw = wdeg[u];
            const float4 vls = make_float4(VL[u].x * dw, VL[u].y * dw, VL[u].z * dw, VL[u].w * dw);
            split_st4(xs, xpl, r * ldx + (half ? OFF_NEG : OFF_POS) + sub * 4, vls);
            if (a.tap_vloss != nullptr && tile_lo + r < N)
                st4(a.tap_vloss + ((size_t)(tile_lo + r) + (size_t)half * NL) * F + sub * 4, vls);
            float4 Mo;
            Mo.x = __shfl_xor(M[u].x, 32); Mo.y = __shfl_xor(M[u].y, 32);
            Mo.z = __shfl_xor(M[u].z, 32); Mo.w = __shfl_xor(M[u].w, 32);
            {   // closed form of the inner tape (closs4 = 4*closs, 0.25 is exact); each half-wave
                // computes two of its lane's four columns: half 0 -> .x .y, half 1 -> .z .w
                const float4 qv = qrow[u];
                const float vw = wv[u];
                const float q0 = half ? qv.z : qv.x, q1 = half ? qv.w : qv.y;
                const float p0 = half ? Mo.z : M[u].x, p1 = half ? Mo.w : M[u].y;   // sums over +v clauses
                const float n0 = half ? M[u].z : Mo.x, n1 = half ? M[u].w : Mo.y;   // sums over -v clauses
                float s0, sn0, s1, sn1;
                sigmoid_pair(q0, s0, sn0);
                sigmoid_pair(q1, s1, sn1);
                float2 g;
                g.x = (-s0 * (p0 * 0.25f) + sn0 * (n0 * 0.25f)) * vw;
                g.y = (-s1 * (p1 * 0.25f) + sn1 * (n1 * 0.25f)) * vw;
                split_st2(xs, xpl, r * ldx + sub * 4 + 2 * half, g);
                if (a.tap_grad != nullptr && tile_lo + r < N)
                    *reinterpret_cast<float2*>(a.tap_grad + (size_t)(tile_lo + r) * F + sub * 4 + 2 * half) = g;
            }
            if (half == 1) split_st4(xs, xpl, r * ldx + OFF_V + sub * 4, rowv[u]);
        }
        DSAT_STAMP(a, 1);
        };
        // R12: new_variables = update_gate(unit)
        float* newv = a.newv;
        float* stage = reinterpret_cast<float*>(sbytes + a.s_stage);   // a region that is dead during the last layer
        const int lds = F + 8;
        const int wvu = __builtin_amdgcn_readfirstlane(wave);
        auto final_store = [=](int row, int col, float4 v) {
            int n = tile_lo + row;
            NO_HOIST(n);
            if (col < F) {
                st4(stage + row * lds + col, v);
                if (n < N) stg4(newv + (size_t)n * F + col, v);
            }
        };
        {
            SplitChain sc;
            sc.nlayers = a.sp_layers;
#pragma unroll
            for (int l = 0; l < 3; ++l) sc.L[l] = SplitLayer{a.sp_ws[l], a.mupd.L[l].bias, a.sp_ks[l], a.sp_nt[l]};
            if (a.upd_part != nullptr) {   // the `variables` k-steps of layer 0 were formed by P2's role D
                sc.init = a.upd_part;
                sc.gap_at = OFF_V / 32;
                sc.gap_len = F / 32;
            }
            split_chain<BM / 16, NT, (BM > 16 ? DSAT_SPLIT_PD4 : DSAT_SPLIT_PD1)>(sc, xs, ldx, hA, hB, ldh, wvu, lane, tid, final_store, finish, upd0);
        }
        DSAT_STAMP(a, 2);
        tile_colsums<BM>(tile_lo, N, tm, a.var_seg, a.psum_v, stage, lds, sPart, tid, a.fm);
        DSAT_STAMP(a, 3);
        DSAT_STAMP_RT_END(a);
    }
}
#endif

// ------------------------------------------------------------------------------------- PairNorm, level 2
// A graph spread over many tiles (one 10 000-variable formula = 625 variable tiles) would have every
// consumer tile add up hundreds of per-tile column sums.  For such graphs the sums are first combined
// in chunks of SEG2_CHUNK: one workgroup per chunk, eight row groups added in a fixed order.
__global__ __launch_bounds__(1024) void segsum2_kernel(const int2* __restrict__ chunks, const float* __restrict__ psum,
                                                       float* __restrict__ psum2, const int* __restrict__ done) {
    if (done != nullptr && *done != 0) return;
    __shared__ float red[8][F];
    const int c = threadIdx.x & (F - 1), grp = threadIdx.x >> 7;
    const int2 ch = chunks[blockIdx.x];
    float acc = 0.f;
    for (int r = ch.x + grp; r < ch.y; r += 8) acc += psum[(size_t)r * F + c];
    red[grp][c] = acc;
    __syncthreads();
    if (threadIdx.x < F) {
        float t = red[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][c];
        psum2[(size_t)blockIdx.x * F + c] = t;
    }
}

// ------------------------------------------------------------------------------------- heavy rows
// One workgroup (8 half-waves) per piece of a heavy literal row: half-wave h sums entries
// [e_lo + h*per, e_lo + (h+1)*per) of the piece (closs4 and vla rows of those clauses, in list
// order), the eight sums are combined in half-wave order: deterministic, no atomics.
// (The same kernel serves heavy CLAUSES before P2: idx = the clause's literal rows, the arrays sp / lits,
// the zero row 2*N.)
#ifndef DSAT_HEAVY_NHW
#define DSAT_HEAVY_NHW 8      // half-waves per piece
#endif
#ifndef DSAT_HEAVY_GJ
#define DSAT_HEAVY_GJ 4       // rows in flight per half-wave and array
#endif
// (16 half-waves with 8 rows in flight each -- a quarter of the dependent round trips per piece -- left the launch at
// 18.9 us at config 5: it moves ~100 MB of rows that no longer fit L2, and that is what it takes.)
__global__ __launch_bounds__(32 * DSAT_HEAVY_NHW) void heavy_gather_kernel(const int2* __restrict__ pieces,
                                                           const int32_t* __restrict__ l_clause,
                                                           const float* __restrict__ closs4,
                                                           const float* __restrict__ vla, int C,
                                                           float* __restrict__ hpart, const int* __restrict__ done) {
    if (done != nullptr && *done != 0) return;
    constexpr int NHW = DSAT_HEAVY_NHW, GJ = DSAT_HEAVY_GJ;
    __shared__ float4 red[NHW][2][32];
    const int sub = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int2 pc = pieces[blockIdx.x];
    const int len = pc.y - pc.x;
    const int per = ((len + NHW - 1) / NHW + GJ - 1) / GJ * GJ;
    const int e0 = pc.x + hw * per, e1 = min(e0 + per, pc.y);
    float4 M = make_float4(0.f, 0.f, 0.f, 0.f), VL = M;
    const unsigned zrow = (unsigned)C;   // the all-zero pad row of closs4 / vla
    for (int e = e0; e < e1; e += GJ) {
        unsigned ci[GJ];
        float4 cv[GJ], vv[GJ];
#pragma unroll
        for (int j = 0; j < GJ; ++j) ci[j] = e + j < e1 ? (unsigned)l_clause[e + j] : zrow;
#pragma unroll
        for (int j = 0; j < GJ; ++j) {
            const unsigned o = ci[j] * (unsigned)F + (unsigned)sub * 4u;
            cv[j] = ld4(closs4 + o);
            vv[j] = ld4(vla + o);
        }
#pragma unroll
        for (int j = 0; j < GJ; ++j) {
            M.x += cv[j].x; M.y += cv[j].y; M.z += cv[j].z; M.w += cv[j].w;
            VL.x += vv[j].x; VL.y += vv[j].y; VL.z += vv[j].z; VL.w += vv[j].w;
        }
    }
    red[hw][0][sub] = M;
    red[hw][1][sub] = VL;
    __syncthreads();
    if (threadIdx.x < 64) {   // the half-waves' sums in half-wave order: one fixed order, run to run
        const int arr = threadIdx.x >> 5;
        float4 t = red[0][arr][sub];
#pragma unroll
        for (int h = 1; h < NHW; ++h) {
            const float4 u = red[h][arr][sub];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        st4(hpart + ((size_t)blockIdx.x * 2 + arr) * F + sub * 4, t);
    }
}

// ------------------------------------------------------------------------------------- S
// R15-R17 per graph: Bernoulli-KL loss of every logit map against the (random) labels
// (query_sat.py:40-53, :289-292), argmin map (:317), selected logits (:328-329), rounded
// assignment, then the SAT check of the graph's clauses (utils/sat.py:118-124;
// metrics/sat_metrics.py:60-85).  The last workgroup publishes done / steps_taken.
__device__ __forceinline__ float bern_kl(float a, float b) {
    return a * (logf(a) - logf(b)) + (1.f - a) * (log1pf(-a) - log1pf(-b));
}

// One workgroup per entry of sel_tab = {graph, part, parts}.  A graph of ordinary size is one part.  A
// large graph (tens of thousands of clauses: its clause check would be one workgroup's serial chain
// of dependent loads) is cut into `parts` workgroups: every part recomputes the graph's eight loss
// sums in the same order (same argmin everywhere, no exchange), then writes the selected logits /
// bits of ITS slice of the variables and checks ITS slice of the clauses, taking literal values
// straight from the logits, so no part waits for another's bits.  Unsatisfied-clause counts meet in
// graph_unsat (integer atomics; zeroed by P3 before every select when parts > 1).
// UNSUP: also built for QuerySAT(supervised=False) (the clause-loss rule).  Only the stand-alone select_sat_kernel is: the
// copies that ride in P2's grid stay as they were -- the extra code in front of P2's tile loops cost config 2 0.5 % (its
// placement, not its execution: the branch is never taken there) -- and an unsupervised session launches the kernel
// between P1 and P2 instead (dsat_api.hip).
template <int NTH, bool UNSUP, bool SHARE>
__device__ __forceinline__ void select_sat_body(int blk, int nblocks, const int32_t* __restrict__ sel_tab,
                                const int32_t* __restrict__ vseg,
                                const int32_t* __restrict__ cseg, const int32_t* __restrict__ rowptr,
                                const int32_t* __restrict__ c_lit, const float* __restrict__ logits,
                                Ctl* __restrict__ ctl, float* __restrict__ pgl, int32_t* __restrict__ best,
                                float* __restrict__ out_logits, uint8_t* __restrict__ assign,
                                int32_t* __restrict__ graph_unsat, int N, int round, int set_done, float* sred,
                                int tid, bool active, unsigned long long* sel_exch, int sel_ticks) {
    // sred: NTH*(LOGIT_MAPS+1) + 2 floats of LDS.  tid / active: several entries may share a workgroup (P2's packed
    // select blocks: NTH threads each, tid = the thread's index within its entry, every barrier below is taken by all
    // of them together -- entries with parts == 1 only); an inactive sub-block mirrors a valid entry's control flow
    // and writes nothing.
    float (*red)[LOGIT_MAPS + 1] = reinterpret_cast<float (*)[LOGIT_MAPS + 1]>(sred);
    int* s_best = reinterpret_cast<int*>(sred + NTH * (LOGIT_MAPS + 1));
    int* s_unsat = s_best + 1;
    const int g = sel_tab[4 * blk], part = sel_tab[4 * blk + 1], parts = sel_tab[4 * blk + 2];
    const int v0 = vseg[g], v1 = vseg[g + 1];
    const float t = powf(ctl->noise_scale, 0.5f);
    const float tl = fminf(t + 0.01f, 1.f);
    const float norm = bern_kl(tl * 0.5f, 0.5f) + 1e-4f;
    float acc[LOGIT_MAPS];
#pragma unroll
    for (int k = 0; k < LOGIT_MAPS; ++k) acc[k] = 0.f;
    bool unsup = false;
    if constexpr (UNSUP) {
        unsup = ctl->unsupervised != 0;
        if (unsup)   // supervised=False: the graph's own clause loss per logit map (every part recomputes all of it)
            for (int c = cseg[g] + tid; c < cseg[g + 1]; c += NTH) unsup_clause_loss8(logits, rowptr, c_lit, c, N, acc);
    }
    // SHARE (a graph of several parts, the parts co-resident): each part sums the KL terms of its own slice of the
    // variables and the parts exchange their eight partial sums (below) instead of each summing all of them -- at
    // config 5 (10 000 variables, 15 parts) the redundant sums were the whole duration of P2 (r03 stamps).
    int kv0 = v0, kv1 = v1;
    bool share = false;
    if constexpr (SHARE) {
        share = sel_exch != nullptr && parts > 1 && !unsup;
        if (share) {
            const int per = (v1 - v0 + parts - 1) / parts;
            kv0 = min(v0 + part * per, v1);
            kv1 = min(kv0 + per, v1);
        }
    }
    auto kl_terms = [&](int lo, int hi) {
    for (int v = lo + tid; v < hi && !unsup; v += NTH) {
        int label;
        if (ctl->inj_labels != nullptr) {
            label = ctl->inj_labels[ctl->labels_step_stride * ctl->step + logical_var(ctl, v)];
        } else {
            u32x4 c = {(uint32_t)logical_var(ctl, v), 0u, ctl->step, (uint32_t)STREAM_LABELS};
            label = (int)(philox4x32_10(c, (uint32_t)ctl->seed, (uint32_t)(ctl->seed >> 32)).x & 1u);
        }
        const float a = (float)label * (1.f - tl) + tl * 0.5f;
        const float log_a = logf(a), log_1ma = log1pf(-a);     // bern_kl(a, b) with the b-free logs taken once
        const float4 l0 = ld4(logits + (size_t)v * LOGIT_MAPS);
        const float4 l1 = ld4(logits + (size_t)v * LOGIT_MAPS + 4);
        const float lg[LOGIT_MAPS] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
        for (int k = 0; k < LOGIT_MAPS; ++k) {
            const float b = sigmoid_f(lg[k]) * (1.f - t) + t * 0.5f;
            acc[k] += (a * (log_a - logf(b)) + (1.f - a) * (log_1ma - log1pf(-b))) / norm;
        }
    }
    };
    auto tree_sum = [&]() {   // red[0][k] = the sum over the workgroup's threads of acc[k]
#pragma unroll
    for (int k = 0; k < LOGIT_MAPS; ++k) red[tid][k] = acc[k];
    if (tid == 0) *s_unsat = 0;
    __syncthreads();
    for (int s = NTH / 2; s >= 1; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int k = 0; k < LOGIT_MAPS; ++k) red[tid][k] += red[tid + s][k];
        }
        __syncthreads();
    }
    };
    kl_terms(kv0, kv1);
    tree_sum();
    if constexpr (SHARE) {
        if (share) {
            // One 64-bit word per (part, map): {epoch = round + 1, the partial sum's bits}, stored and polled with
            // agent-scope atomics (write-through to the memory side, no fence: the word is its own flag).  The host
            // zeroes the words at the start of every denoising step, so an epoch is seen once.  Every part adds the
            // partials in part order: all of them pick the same map.  A part that is not served within the bound
            // (parts not co-resident: more parts than CUs) forms every slice's sum itself, slice by slice with the same
            // thread-to-variable assignment and the same tree -- bit for bit the exchanged result.
            const int base = blk - part;
            const unsigned epoch = (unsigned)(round + 1);
            if (tid < LOGIT_MAPS)
                __hip_atomic_store(sel_exch + (size_t)blk * LOGIT_MAPS + tid,
                                   ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(red[0][tid]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int ok = 1;
            if (tid == 0) *s_best = 1;   // "every word arrived", cleared by a thread whose word did not
            __syncthreads();
            const long long t0 = wall_clock64();
            for (int i = tid; i < parts * LOGIT_MAPS; i += NTH) {
                const unsigned long long* w = sel_exch + (size_t)base * LOGIT_MAPS + i;
                unsigned long long x;
                for (;;) {
                    x = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (sel_ticks < 0) { ok = 0; break; }   // tests: the unserved part's path
                    if ((unsigned)(x >> 32) == epoch) break;
                    if (wall_clock64() - t0 > sel_ticks) { ok = 0; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                red[1 + (i >> 3)][i & 7] = __uint_as_float((unsigned)x);
            }
            if (!ok) *s_best = 0;
            __syncthreads();
            ok = *s_best;
            if (ok) {
                if (tid < LOGIT_MAPS) {
                    float sum = 0.f;
                    for (int p = 0; p < parts; ++p) sum += red[1 + p][tid];
                    red[0][tid] = sum;
                }
                __syncthreads();
            } else {
                float tot = 0.f;   // threads 0 .. 7: one map each
                const int per = (v1 - v0 + parts - 1) / parts;
                for (int p = 0; p < parts; ++p) {
#pragma unroll
                    for (int k = 0; k < LOGIT_MAPS; ++k) acc[k] = 0.f;
                    const int lo = min(v0 + p * per, v1);
                    __syncthreads();   // red[0] of the slice before is read
                    kl_terms(lo, min(lo + per, v1));
                    tree_sum();
                    if (tid < LOGIT_MAPS) tot += red[0][tid];
                }
                __syncthreads();
                if (tid < LOGIT_MAPS) red[0][tid] = tot;
                __syncthreads();
            }
        }
    }
    if (tid == 0) {
        const float inv = 1.f / (float)(v1 - v0);
        int bi = 0;
        float bv = unsup ? unsup_graph_loss(red[0][0]) : red[0][0] * inv;
        if (part == 0 && active) pgl[g * LOGIT_MAPS] = bv;
        for (int k = 1; k < LOGIT_MAPS; ++k) {
            const float x = unsup ? unsup_graph_loss(red[0][k]) : red[0][k] * inv;
            if (part == 0 && active) pgl[g * LOGIT_MAPS + k] = x;
            if (x < bv) { bv = x; bi = k; }   // first minimum wins, as TF's CPU ArgMin
        }
        if (part == 0 && active) best[g] = bi;
        *s_best = bi;
    }
    __syncthreads();
    const int bi = *s_best;
    {   // this part's slice of the variables
        const int per = (v1 - v0 + parts - 1) / parts;
        const int a0 = v0 + part * per, a1 = active ? min(a0 + per, v1) : a0;
        for (int v = a0 + tid; v < a1; v += NTH) {
            const float x = logits[(size_t)v * LOGIT_MAPS + bi];
            out_logits[v] = x;
            assign[v] = (uint8_t)rintf(sigmoid_f(x));   // tf.round(tf.sigmoid(.)), half-to-even
        }
    }
    int unsat = 0;
    if (parts == 1) {
        __syncthreads();   // this workgroup's assign[] stores are visible to its own loads (same CU L1)
        for (int c = cseg[g] + tid; c < cseg[g + 1]; c += NTH) {
            bool sat = false;
            for (int e = rowptr[c]; e < rowptr[c + 1]; ++e) {
                const int l = c_lit[e];
                sat = sat || (l < N ? (assign[l] != 0) : (assign[l - N] == 0));
            }
            unsat += sat ? 0 : 1;
        }
    } else {   // this part's slice of the clauses; literal values recomputed from the logits
        const int c0 = cseg[g], c1 = cseg[g + 1];
        const int per = (c1 - c0 + parts - 1) / parts;
        const int a0 = c0 + part * per, a1 = min(a0 + per, c1);
        for (int c = a0 + tid; c < a1; c += NTH) {
            bool sat = false;
            for (int e = rowptr[c]; e < rowptr[c + 1]; ++e) {
                const int l = c_lit[e];
                const int v = l < N ? l : l - N;
                const bool val = rintf(sigmoid_f(logits[(size_t)v * LOGIT_MAPS + bi])) != 0.f;
                sat = sat || (l < N ? val : !val);
            }
            unsat += sat ? 0 : 1;
        }
    }
    if (unsat) atomicAdd(s_unsat, unsat);
    __syncthreads();
    if (tid == 0 && active) {
        if (parts == 1) graph_unsat[g] = *s_unsat;
        else if (*s_unsat) atomicAdd(&graph_unsat[g], *s_unsat);
        if (*s_unsat) atomicAdd(&ctl->unsat_total, *s_unsat);
        __threadfence();
        const int tk = atomicAdd(&ctl->ticket, 1);
        if (tk == nblocks - 1) {
            const int total = atomicAdd(&ctl->unsat_total, 0);
            ctl->steps_taken = round;
            if (set_done && total == 0) ctl->done = 1;
            ctl->unsat_total = 0;   // re-arm for the next check
            ctl->ticket = 0;
        }
    }
}

// The stand-alone form (the last round of a step; every round of a supervised=False session): the loops over a graph's
// variables and clauses are chains of dependent loads (record -> list -> logits), so the threads are what hides them:
// 1024 per graph or part, as the copies that ride in P2 (with 256: 14.2 -> 13.7 us at config 2, 76 -> 37 us at config 5)
#ifndef DSAT_SEL_NTH
#define DSAT_SEL_NTH 1024
#endif
constexpr int SEL_NTH = DSAT_SEL_NTH;
__global__ __launch_bounds__(SEL_NTH) void select_sat_kernel(
    const int32_t* __restrict__ sel_tab,
    const int32_t* __restrict__ vseg, const int32_t* __restrict__ cseg, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ c_lit, const float* __restrict__ logits, Ctl* __restrict__ ctl,
    float* __restrict__ pgl, int32_t* __restrict__ best, float* __restrict__ out_logits,
    uint8_t* __restrict__ assign, int32_t* __restrict__ graph_unsat, int N, int round, int set_done,
    unsigned long long* sel_exch, int sel_ticks) {
    if (set_done && ctl->done != 0) return;
    __shared__ float sred[SEL_NTH * (LOGIT_MAPS + 1) + 2];
    select_sat_body<SEL_NTH, true, true>(blockIdx.x, gridDim.x, sel_tab, vseg, cseg, rowptr, c_lit, logits, ctl, pgl, best,
                                     out_logits, assign, graph_unsat, N, round, set_done, sred, (int)threadIdx.x, true,
                                     sel_exch, sel_ticks);
}

}  // namespace

hipError_t launch_round_p1(const RoundArgs& a, hipStream_t st) {
    const int tiles = (a.a2_tiles > 0 ? a.a2_tiles : (a.N + ROUND_BM_VAR - 1) / ROUND_BM_VAR) + a.pair_blocks + a.cn_blocks;
#if DSAT_ROUND_NWAVES == 16
    if (a.a2_tiles > 0) hipLaunchKernelGGL(round_p1_kernel<true>, dim3(tiles), dim3(NT), 0, st, a);
    else
#endif
    hipLaunchKernelGGL(round_p1_kernel<false>, dim3(tiles), dim3(NT), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_round_p2(const RoundArgs& a, int rb, hipStream_t st) {
    const int bm = 16 * rb;
    const int nsb = a.sel_pack > 1 ? (a.n_sel + a.sel_pack - 1) / a.sel_pack : a.n_sel;
    const int tiles = (a.C + bm - 1) / bm + (a.s_enable ? (a.sel_lead > 0 ? a.sel_lead : nsb) : 0);
    if (tiles == 0) return hipSuccess;
#if DSAT_ROUND_NWAVES == 16
    if (rb == 4 && a.sp_ws[0] != nullptr) {
        if (a.sel_exch != nullptr && a.s_enable && a.sel_lead > 0)
            hipLaunchKernelGGL(round_p2s_kernel<true>, dim3(tiles + a.d_blocks), dim3(NT), 0, st, a);
        else
            hipLaunchKernelGGL(round_p2s_kernel<false>, dim3(tiles + a.d_blocks), dim3(NT), 0, st, a);
        return hipGetLastError();
    }
#endif
    if (rb == 4) hipLaunchKernelGGL(round_p2_kernel<4>, dim3(tiles), dim3(NT), 0, st, a);
    else if (rb == 2) hipLaunchKernelGGL(round_p2_kernel<2>, dim3(tiles), dim3(NT), 0, st, a);
    else if (rb == 1) hipLaunchKernelGGL(round_p2_kernel<1>, dim3(tiles), dim3(NT), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_round_p3(const RoundArgs& a0, hipStream_t st) {
    RoundArgs a = a0;
    const int bm = (a.sp_ws[0] != nullptr && a.p3_bm == 32) ? 32 : ROUND_BM_VAR;
    a.p3_tiles = (a.N + bm - 1) / bm;
    // tiles, then the role-C blocks; or, interleaved, rows of 8: cn_period tile rows, one role-C row, ...
    const int grid = a.cn_period > 0 ? 8 * ((a.p3_tiles + 7) / 8 + (a.cn_blocks + 7) / 8) : a.p3_tiles + a.cn_blocks;
#if DSAT_ROUND_NWAVES == 16
    if (a.sp_ws[0] != nullptr) {
        const bool hc = a.cn_blocks > 0;
        if (a.p3_bm == 32) {
            if (hc) hipLaunchKernelGGL((round_p3s_kernel<32, true>), dim3(grid), dim3(NT), 0, st, a);
            else hipLaunchKernelGGL((round_p3s_kernel<32, false>), dim3(grid), dim3(NT), 0, st, a);
        } else {
            if (hc) hipLaunchKernelGGL((round_p3s_kernel<16, true>), dim3(grid), dim3(NT), 0, st, a);
            else hipLaunchKernelGGL((round_p3s_kernel<16, false>), dim3(grid), dim3(NT), 0, st, a);
        }
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(round_p3_kernel, dim3(grid), dim3(NT), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_segsum2(const int2* chunks, int n_chunks, const float* psum, float* psum2, const int* done,
                          hipStream_t st) {
    if (n_chunks <= 0) return hipSuccess;
    hipLaunchKernelGGL(segsum2_kernel, dim3(n_chunks), dim3(1024), 0, st, chunks, psum, psum2, done);
    return hipGetLastError();
}
hipError_t launch_heavy_gather(const int2* pieces, int n_pieces, const int32_t* l_clause, const float* closs4,
                               const float* vla, int C, float* hpart, const int* done, hipStream_t st) {
    if (n_pieces <= 0) return hipSuccess;
    hipLaunchKernelGGL(heavy_gather_kernel, dim3(n_pieces), dim3(32 * DSAT_HEAVY_NHW), 0, st, pieces, l_clause, closs4, vla, C,
                       hpart, done);
    return hipGetLastError();
}
void launch_select_sat(const dsat_graph* g, const int32_t* sel_tab, int n_sel, const float* logits, Ctl* ctl,
                       float* pgl, int32_t* best, float* out_logits, uint8_t* assign, int32_t* graph_unsat,
                       int round, int set_done, unsigned long long* sel_exch, int sel_ticks, hipStream_t st) {
    hipLaunchKernelGGL(select_sat_kernel, dim3(n_sel), dim3(SEL_NTH), 0, st, sel_tab, g->var_seg, g->cl_seg, g->c_rowptr,
                       g->c_lit, logits, ctl, pgl, best, out_logits, assign, graph_unsat, g->N, round, set_done, sel_exch,
                       sel_ticks);
}

}  // namespace dsat
