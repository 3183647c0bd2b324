// Building block of the fused row-tile kernels: one Dense layer whose input tile lives in
// LDS and whose weights stream from L2/HBM straight into MFMA B fragments.
//
//   out[BM rows][N] = act( in[BM][K] @ W[K][N] + b ),  BM = 16*RB
//
// * v_mfma_f32_16x16x4_f32 (exact fp32).  Lane l of a wave holds, for a 16-deep k-block kb
//   and 16-wide column tile t: A fragment = in[row l&15][16kb + 4(l>>4) .. +3] (one
//   ds_read_b128) and B fragment = W[16kb + 4(l>>4) + j][16t + (l&15)], j=0..3 (one 16-B
//   global load from the fragment-ordered copy `wp`, 1 KiB contiguous per wave).  MFMA
//   step j multiplies k = {4g + j : g = 0..3}.
// * Waves split N: wave w owns TPW consecutive column tiles per pass and keeps one
//   accumulator per (row block, tile); the A fragment is reused across its tiles, so
//   LDS traffic is 1/TPW of the MFMA operand traffic and each weight byte is used BM times.
// * Weight loads are software-pipelined PD k-blocks ahead in registers (no LDS staging:
//   no two waves share a B fragment).
// * LDS rows are padded to ld = K + 8 (== 8 mod 16 floats): the ds_read_b128 lane groups
//   then fall on distinct bank slots (checked exhaustively).
#pragma once
#include <hip/hip_runtime.h>

namespace dsat {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Ring depth in k-blocks.  Measured on the whole denoising step (bench.py, config 2): depth 2-3
// beats the deeper rings the stand-alone k-loop benchmark favoured (8-12 for narrow layers) by
// 3-5 % -- an L2 hit returns within about one k-block of MFMA work, and every extra slot is a
// load to issue at layer start and registers the k-loop cannot use.  With 16 waves per workgroup
// (four per SIMD) depth 2 is best at two tiles per wave, depth 3 at one; depth 1 (no look-ahead)
// loses 10 %.  -DDSAT_PD{1,2,4}=n override the depth for 1, 2 and 3-4 tiles per wave.
#ifndef DSAT_PD1
#define DSAT_PD1 3
#endif
#ifndef DSAT_PD2
#define DSAT_PD2 2
#endif
#ifndef DSAT_PD4
#define DSAT_PD4 3
#endif
template <int TPW>
struct PrefetchDepth { static constexpr int value = TPW >= 6 ? 3 : TPW >= 3 ? DSAT_PD4 : TPW >= 2 ? DSAT_PD2 : DSAT_PD1; };

// Weight-fragment loads are inline asm: hipcc (ROCm 7.2) otherwise proves that a register
// prefetched PD iterations ahead equals a fresh load from the (read-only) weights and
// re-materialises it right before the MFMA, which serialises L2 latency with the matrix
// pipe.  An asm load is opaque, so the ring below stays a ring; completion is counted by
// hand (cdna_hip_programming.md 5.7 item 1, form ii): every wait names the registers it
// releases as "+v" operands, so no consumer can be scheduled above it.
// 16-B load: per-lane 32-bit byte offset (VGPR) + wave-uniform 64-bit base (SGPR pair), so
// the per-k-block address update is scalar arithmetic and costs no VALU issue slots.
#ifndef DSAT_WLOAD_POLICY
#define DSAT_WLOAD_POLICY ""   // cache-policy bits of the weight loads (" nt", " sc1", ...): experiment switch
#endif
__device__ __forceinline__ void gload_b128(f32x4& dst, unsigned voff, const float4* sbase) {
#ifdef DSAT_EXP_NOLOAD
    dst = f32x4{1.f, 2.f, 3.f, (float)voff};
#else
    asm volatile("global_load_dwordx4 %0, %1, %2" DSAT_WLOAD_POLICY : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
#endif
}
// tools/chain_probe.hip defines DSAT_PROBE to time-stamp the phases of a layer
#ifndef DSAT_PROBE
#define DSAT_PROBE(id)
#endif

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for
// vector-memory loads, so weight fragments requested before it stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N>
__device__ __forceinline__ void wait_vmcnt_release(f32x4& first) {
#ifdef DSAT_EXP_NOLOAD
    asm volatile("" : "+v"(first));
#else
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(first) : "n"(N) : "memory");
#endif
}
__device__ __forceinline__ void release(f32x4& r) { asm volatile("" : "+v"(r)); }
// Make the compiler complete (wait for) an ordinary load HERE: its own s_waitcnt counts only
// the loads it knows, so a wait placed after asm loads were issued would drain those as well.
template <int N>
__device__ __forceinline__ void pin_values(float4 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
}

// One k-block step of the ring: consume slot `p` (MFMAs in tile pairs), then refill it from
// `sb` when REFILL.  Tiles are processed in pairs: 4 k-steps x 2 tiles x RB row blocks of
// MFMAs, then the two refill loads of that pair -- issued only after the last MFMA that reads
// the old fragment, so a ring slot keeps its physical registers and no copy of an in-flight
// register can appear.  sched_barrier pins the interleave: one pair of 1-KiB loads per
// 8*RB MFMAs instead of a burst of TPW loads at the end of the k-block.
// The weight fragment is the MFMA's A operand and the activation fragment its B operand, i.e.
// the instruction computes the TRANSPOSED output tile: lane l ends up with output columns
// 16*t + 4*(l>>4) .. +3 of row (l & 15) -- four consecutive columns of one row, so the epilogue
// writes 16-byte vectors (LDS and HBM) instead of four strided scalars.
// PART: the last tile of the wave is a partial one -- only one of its RB row blocks belongs to
// this wave (A fragment `asp`, accumulator acc[0][TPW-1]); see chain_static's MIX0.
template <int RB, int TPW, bool REFILL, bool PART = false>
__device__ __forceinline__ void ring_step(const float4 (&a)[RB], const float4& asp, f32x4 (&slot)[TPW],
                                          const unsigned (&voff)[TPW], const float4* sb, f32x4 (&acc)[RB][TPW]) {
#pragma unroll
    for (int jp = 0; jp < TPW; jp += 2) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = jp; j < jp + 2 && j < TPW; ++j) {
                if (PART && j == TPW - 1) {
                    const float av = kk == 0 ? asp.x : kk == 1 ? asp.y : kk == 2 ? asp.z : asp.w;
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(slot[j][kk], av, acc[0][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const float av = kk == 0 ? a[rb].x : kk == 1 ? a[rb].y : kk == 2 ? a[rb].z : a[rb].w;
                        acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(slot[j][kk], av, acc[rb][j], 0, 0, 0);
                    }
                }
            }
        if (REFILL) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = jp; j < jp + 2 && j < TPW; ++j) gload_b128(slot[j], voff[j], sb);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Accumulate tiles t0 .. t0+TPW-1 (tile indices clamped to ntiles-1 for the loads; the
// caller discards clamped tiles) over all k-blocks.  acc must be zero-initialised.
// Ring of PD k-blocks in registers with static indices: the main loop is unrolled PD times
// and covers floor(kblocks/PD)*PD k-blocks, refilling each slot PD blocks ahead (index
// clamped to the last block, so the number of loads in flight -- hence every vmcnt count --
// is the same at every step); the remaining < PD blocks are already in the ring when the
// main loop ends and are consumed by a peeled tail.  The A fragment of step s+1 is read
// from LDS during step s.  The queue is drained before returning.
// `sync_in`: the input tile is still being written by other waves (previous layer's epilogue);
// the workgroup barrier that publishes it is taken here, AFTER the ring has been primed, so
// the first weight fragments travel while the waves wait for each other.
template <int TPW>
__device__ __forceinline__ void ring_prime_tiles(f32x4 (&breg)[PrefetchDepth<TPW>::value][TPW], unsigned (&voff)[TPW],
                                                 const float4* __restrict__ wp, int kblocks,
                                                 const int (&tiles)[TPW], int lane) {
    constexpr int PD = PrefetchDepth<TPW>::value;
    static_assert((PD - 1) * TPW <= 63, "vmcnt field is 6 bits");
#pragma unroll
    for (int j = 0; j < TPW; ++j) voff[j] = (unsigned)((tiles[j] * kblocks * 64 + lane) * 16);
    const int last = kblocks - 1;
#pragma unroll
    for (int p = 0; p < PD; ++p) {
        const int kb = p < last ? p : last;
#pragma unroll
        for (int j = 0; j < TPW; ++j) gload_b128(breg[p][j], voff[j], wp + (size_t)kb * 64);
    }
}

template <int TPW>
__device__ __forceinline__ void ring_prime(f32x4 (&breg)[PrefetchDepth<TPW>::value][TPW], unsigned (&voff)[TPW],
                                           const float4* __restrict__ wp, int kblocks, int t0, int ntiles,
                                           int lane) {
    int tiles[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) tiles[j] = (t0 + j < ntiles) ? t0 + j : ntiles - 1;
    ring_prime_tiles<TPW>(breg, voff, wp, kblocks, tiles, lane);
}

// The k-loop over a ring that ring_prime has filled.  Other vector-memory operations may be
// outstanding, older or younger than the ring's: they only make the counted waits stricter.
template <int RB, int TPW, bool PART = false>
__device__ __forceinline__ void gemm_primed(const float* __restrict__ sIn, int ldin, int kblocks,
                                            const float4* __restrict__ wp,
                                            f32x4 (&breg)[PrefetchDepth<TPW>::value][TPW], const unsigned (&voff)[TPW],
                                            int lane, f32x4 (&acc)[RB][TPW], int sp_rb = 0) {
    constexpr int PD = PrefetchDepth<TPW>::value;
    const float* pa = sIn + (lane & 15) * ldin + 4 * (lane >> 4);
    const float* pasp = pa + sp_rb * 16 * ldin;   // PART: the row block of the partial tile
    const int last = kblocks - 1;
    float4 asp_next = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PART) asp_next = *reinterpret_cast<const float4*>(pasp);
    float4 a_next[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#ifdef DSAT_EXP_NOLDS
        a_next[rb] = make_float4(1.f, 2.f, 3.f, 4.f);
#else
        a_next[rb] = *reinterpret_cast<const float4*>(pa + rb * 16 * ldin);
#endif
    }
    const int main_blocks = kblocks / PD * PD;
    int kb = 0;
    DSAT_PROBE(4);
    for (; kb < main_blocks; kb += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            float4 a[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[rb] = a_next[rb];
            const float4 asp = asp_next;
            {   // A fragment of the next step (address clamped to the last k-block)
                const int k1 = kb + p + 1;
                const int kc = k1 < last ? k1 : last;
                if (PART) asp_next = *reinterpret_cast<const float4*>(pasp + kc * 16);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
#ifdef DSAT_EXP_NOLDS
                    a_next[rb] = make_float4(1.f + kc, 2.f, 3.f, 4.f);
#else
                    a_next[rb] = *reinterpret_cast<const float4*>(pa + rb * 16 * ldin + kc * 16);
#endif
                }
            }
            // slot p was requested PD steps ago; (PD-1)*TPW younger loads may stay in flight
            wait_vmcnt_release<(PD - 1) * TPW>(breg[p][0]);
#pragma unroll
            for (int j = 1; j < TPW; ++j) release(breg[p][j]);
#ifdef DSAT_EXP_SAMEADDR
            const int kn = 0;
#else
            const int kn = kb + p + PD < last ? kb + p + PD : last;
#endif
            ring_step<RB, TPW, true, PART>(a, asp, breg[p], voff, wp + (size_t)kn * 64, acc);
        }
    }
    // tail: blocks main_blocks .. kblocks-1 sit in slots 0 .. rem-1; nothing more is requested
    DSAT_PROBE(5);
    wait_vmcnt_release<0>(breg[0][0]);
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
        for (int j = 0; j < TPW; ++j) release(breg[p][j]);
    const int rem = kblocks - main_blocks;
#pragma unroll
    for (int p = 0; p + 1 < PD; ++p) {
        if (p < rem) {
            float4 a[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[rb] = a_next[rb];
            const float4 asp = asp_next;
            const int k1 = main_blocks + p + 1;
            const int kc = k1 < last ? k1 : last;
            if (PART) asp_next = *reinterpret_cast<const float4*>(pasp + kc * 16);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
#ifdef DSAT_EXP_NOLDS
                a_next[rb] = make_float4(1.f + kc, 2.f, 3.f, 4.f);
#else
                a_next[rb] = *reinterpret_cast<const float4*>(pa + rb * 16 * ldin + kc * 16);
#endif
            }
            ring_step<RB, TPW, false, PART>(a, asp, breg[p], voff, wp, acc);
        }
    }
}

template <int RB, int TPW>
__device__ __forceinline__ void stream_gemm(const float* __restrict__ sIn, int ldin, int kblocks,
                                            const float4* __restrict__ wp, int t0, int ntiles, int lane,
                                            f32x4 (&acc)[RB][TPW], bool sync_in = false) {
    constexpr int PD = PrefetchDepth<TPW>::value;
    unsigned voff[TPW];
    f32x4 breg[PD][TPW];
    ring_prime<TPW>(breg, voff, wp, kblocks, t0, ntiles, lane);
    if (sync_in) lds_barrier();
    gemm_primed<RB, TPW>(sIn, ldin, kblocks, wp, breg, voff, lane, acc);
}

template <int TPW>
__device__ __forceinline__ void load_bias(float4 (&bv)[TPW], const float* __restrict__ bias, int t0, int ntiles,
                                          int lane) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = t0 + j < ntiles ? t0 + j : ntiles - 1;
        bv[j] = *reinterpret_cast<const float4*>(bias + t * 16 + 4 * (lane >> 4));
    }
}

__device__ __forceinline__ float4 bias_act(const f32x4& acc, const float4& b, int act) {
    float4 v = make_float4(acc[0] + b.x, acc[1] + b.y, acc[2] + b.z, acc[3] + b.w);
    if (act) {   // leaky_relu(0.2): max(v, 0.2 v)
        v.x = fmaxf(v.x, 0.2f * v.x); v.y = fmaxf(v.y, 0.2f * v.y);
        v.z = fmaxf(v.z, 0.2f * v.z); v.w = fmaxf(v.w, 0.2f * v.w);
    }
    return v;
}

// `store(row, col, v)` receives the four outputs (row, col .. col+3); col is a multiple of 4.
template <int RB, int TPW, typename Store>
__device__ __forceinline__ void layer_epilogue(const f32x4 (&acc)[RB][TPW], const float4 (&bv)[TPW], int t0,
                                               int ntiles, int act, int lane, Store&& store) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        if (t0 + j < ntiles) {
            const int col = (t0 + j) * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) store(rb * 16 + (lane & 15), col, bias_act(acc[rb][j], bv[j], act));
        }
    }
}

// One Dense layer for the whole workgroup.  `store(row, col, float4)` receives every output
// (row in [0,16*RB), columns col..col+3 in [0, 16*ntiles)) after bias and activation.
template <int RB, int TPW, int NWAVES, typename Store>
__device__ __forceinline__ void dense_layer_tpw(const float* __restrict__ sIn, int ldin, int kblocks,
                                                const float4* __restrict__ wp, const float* __restrict__ bias,
                                                int ntiles, int act, int wave, int lane, Store&& store,
                                                bool sync_in) {
    for (int base = 0; base < ntiles; base += NWAVES * TPW) {
        const int t0 = base + wave * TPW;
        const bool sync_now = sync_in && base == 0;   // every wave takes the barrier exactly once
        if (t0 >= ntiles) {
            if (sync_now) lds_barrier();
            continue;
        }
        f32x4 acc[RB][TPW];
        float4 bv[TPW];   // requested before the k-loop: a load in the epilogue would expose its latency
        load_bias<TPW>(bv, bias, t0, ntiles, lane);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        DSAT_PROBE(0);
        stream_gemm<RB, TPW>(sIn, ldin, kblocks, wp, t0, ntiles, lane, acc, sync_now);
        DSAT_PROBE(1);
        layer_epilogue<RB, TPW>(acc, bv, t0, ntiles, act, lane, store);
    }
}

// `tpw` (tiles per wave and pass, chosen on the host by pick_tpw) selects the register blocking.
template <int RB, int NWAVES, int CAP = 8, typename Store>
__device__ __forceinline__ void dense_layer(const float* __restrict__ sIn, int ldin, int kblocks,
                                            const float4* __restrict__ wp, const float* __restrict__ bias,
                                            int ntiles, int tpw, int act, int wave, int lane, Store&& store,
                                            bool sync_in = false) {
    // register budget: RB * TPW accumulators of 4 VGPRs + PD * TPW weight fragments in flight
    // (CAP: kernels launched with 16 waves have half the registers per wave)
    constexpr int MAXT0 = RB == 1 ? 8 : RB == 2 ? 4 : 2;
    constexpr int MAXT = MAXT0 < CAP ? MAXT0 : CAP;
    if constexpr (MAXT >= 8) {
        if (tpw == 8) { dense_layer_tpw<RB, 8, NWAVES>(sIn, ldin, kblocks, wp, bias, ntiles, act, wave, lane, store, sync_in); return; }
        if (tpw == 6) { dense_layer_tpw<RB, 6, NWAVES>(sIn, ldin, kblocks, wp, bias, ntiles, act, wave, lane, store, sync_in); return; }
    }
    if constexpr (MAXT >= 4) {
        if (tpw == 4) { dense_layer_tpw<RB, 4, NWAVES>(sIn, ldin, kblocks, wp, bias, ntiles, act, wave, lane, store, sync_in); return; }
        if (tpw == 3) { dense_layer_tpw<RB, 3, NWAVES>(sIn, ldin, kblocks, wp, bias, ntiles, act, wave, lane, store, sync_in); return; }
    }
    if (tpw == 2) { dense_layer_tpw<RB, 2, NWAVES>(sIn, ldin, kblocks, wp, bias, ntiles, act, wave, lane, store, sync_in); return; }
    dense_layer_tpw<RB, 1, NWAVES>(sIn, ldin, kblocks, wp, bias, ntiles, act, wave, lane, store, sync_in);
}

// A whole 2- or 3-layer chain with the register blocking of every layer fixed at compile time
// (T2 == 0: two layers).  Knowing the next layer's blocking lets the code request its first
// PD k-blocks of weights BEFORE the current layer's epilogue and barrier, so the per-layer
// fixed cost (L2 latency of the ring prime, ~1-1.5k cycles, plus epilogue and barrier skew)
// overlaps instead of adding up.  Requires a single pass per layer (ntiles <= NW * T).
// Layer l reads in_l (sIn, bufA, bufB) and hidden layers write bufA, then bufB.
struct StaticLayer { const float4* wp; const float* bias; int kblocks, ntiles; };

// MIX0 (RB == 4; layer 0 has (NW-4)*T0 + 4*(T0-1) + 1 column tiles, e.g. 13 for NW = 8, T0 = 2 and
// for NW = 16, T0 = 1): waves [0, NW-4) own T0 whole tiles; wave NW-4+i owns T0-1 whole tiles and
// row block i of the last tile.  Waves w, w+4, ... share a SIMD and NW-4 is a multiple of 4, so
// every SIMD gets 3*RB + 1 = 13 (tile, row block) units instead of the 16 that whole tiles per
// wave put on the busiest one.
// `between()` runs after layer 0's weight ring has been requested and before the barrier that
// publishes the input tile: the caller's last piece of tile preparation (the tail of a gather)
// goes there, so the ring's L2 latency is hidden behind it.  With `sync_in` the chain takes
// that barrier itself (the caller must not).
template <int RB, int NW, int T0, int T1, int T2, bool MIX0 = false, typename Store, typename Between>
__device__ __forceinline__ void chain_static(const StaticLayer& L0, const StaticLayer& L1, const StaticLayer& L2,
                                             const float* sIn, int ldin, float* bufA, int ldA, float* bufB,
                                             int ldB, int wv /* wave-uniform (SGPR) */, int lane,
                                             Store&& final_store, bool sync_in, Between&& between) {
    static_assert(!MIX0 || (RB == 4 && NW % 4 == 0 && NW > 4), "MIX0 layout");
    constexpr int TL = T2 > 0 ? T2 : 1;
    constexpr int NFULL = NW - 4;      // MIX0: waves that own whole tiles only
    const int t0 = wv * T0, t1 = wv * T1, t2 = wv * TL;
    const bool mixed = MIX0 && wv >= NFULL;
    const bool on0 = MIX0 || t0 < L0.ntiles, on1 = t1 < L1.ntiles, on2 = T2 > 0 && t2 < L2.ntiles;
    int tiles0[T0];
#pragma unroll
    for (int j = 0; j < T0; ++j) tiles0[j] = t0 + j < L0.ntiles ? t0 + j : L0.ntiles - 1;
    if (mixed) {
#pragma unroll
        for (int j = 0; j + 1 < T0; ++j) tiles0[j] = NFULL * T0 + (wv - NFULL) * (T0 - 1) + j;
        tiles0[T0 - 1] = L0.ntiles - 1;
    }
    const int sp_rb = mixed ? wv - NFULL : 0;
    float4 bv0[T0], bv1[T1], bv2[TL];
    if (on0) {
#pragma unroll
        for (int j = 0; j < T0; ++j)
            bv0[j] = *reinterpret_cast<const float4*>(L0.bias + tiles0[j] * 16 + 4 * (lane >> 4));
    }
    if (on1) load_bias<T1>(bv1, L1.bias, t1, L1.ntiles, lane);
    if (on2) load_bias<TL>(bv2, L2.bias, t2, L2.ntiles, lane);
    // ---- layer 0 (the two layouts are separate code paths from the ring prime on: a ring
    // shared between two differently allocated loops would be copied while still in flight)
    DSAT_PROBE(0);
    f32x4 acc0[RB][T0];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int j = 0; j < T0; ++j) acc0[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MIX0 && mixed) {
        unsigned vo0[T0];
        f32x4 r0[PrefetchDepth<T0>::value][T0];
        ring_prime_tiles<T0>(r0, vo0, L0.wp, L0.kblocks, tiles0, lane);
        between();
        if (sync_in) lds_barrier();
        gemm_primed<RB, T0, MIX0>(sIn, ldin, L0.kblocks, L0.wp, r0, vo0, lane, acc0, sp_rb);
    } else if (on0) {
        unsigned vo0[T0];
        f32x4 r0[PrefetchDepth<T0>::value][T0];
        ring_prime_tiles<T0>(r0, vo0, L0.wp, L0.kblocks, tiles0, lane);
        between();
        if (sync_in) lds_barrier();
        gemm_primed<RB, T0>(sIn, ldin, L0.kblocks, L0.wp, r0, vo0, lane, acc0);
    } else {
        between();
        if (sync_in) lds_barrier();
    }
    DSAT_PROBE(1);
    // biases were requested before layer 0; the queue is empty here (the k-loop drained it)
    if (on0) pin_values(bv0);
    if (on1) pin_values(bv1);
    if (on2) pin_values(bv2);
    // ---- layer 1: ring requested before layer 0's epilogue and the barrier
    f32x4 acc1[RB][T1];
    unsigned vo1[T1];
    f32x4 r1[PrefetchDepth<T1>::value][T1];
    if (on1) ring_prime<T1>(r1, vo1, L1.wp, L1.kblocks, t1, L1.ntiles, lane);
    if (on0) {
        auto store0 = [=](int row, int col, float4 v) { *reinterpret_cast<float4*>(bufA + row * ldA + col) = v; };
        if (mixed) {   // tile 0 whole, last tile: row block sp_rb only (its accumulator is acc0[0][T0-1])
#pragma unroll
            for (int j = 0; j < T0; ++j) {
                const int col = tiles0[j] * 16 + 4 * (lane >> 4);
                if (j == T0 - 1) {
                    store0(sp_rb * 16 + (lane & 15), col, bias_act(acc0[0][j], bv0[j], 1));
                } else {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) store0(rb * 16 + (lane & 15), col, bias_act(acc0[rb][j], bv0[j], 1));
                }
            }
        } else {
            layer_epilogue<RB, T0>(acc0, bv0, t0, L0.ntiles, 1, lane, store0);
        }
    }
    DSAT_PROBE(2);
    lds_barrier();
    DSAT_PROBE(3);
    DSAT_PROBE(0);
    if (on1) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < T1; ++j) acc1[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_primed<RB, T1>(bufA, ldA, L1.kblocks, L1.wp, r1, vo1, lane, acc1);
    }
    DSAT_PROBE(1);
    if constexpr (T2 > 0) {
        f32x4 acc2[RB][TL];
        unsigned vo2[TL];
        f32x4 r2[PrefetchDepth<TL>::value][TL];
        if (on2) ring_prime<TL>(r2, vo2, L2.wp, L2.kblocks, t2, L2.ntiles, lane);
        if (on1) layer_epilogue<RB, T1>(acc1, bv1, t1, L1.ntiles, 1, lane,
                                        [=](int row, int col, float4 v) { *reinterpret_cast<float4*>(bufB + row * ldB + col) = v; });
        DSAT_PROBE(2);
        lds_barrier();
        DSAT_PROBE(3);
        DSAT_PROBE(0);
        if (on2) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int j = 0; j < TL; ++j) acc2[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_primed<RB, TL>(bufB, ldB, L2.kblocks, L2.wp, r2, vo2, lane, acc2);
            DSAT_PROBE(1);
            layer_epilogue<RB, TL>(acc2, bv2, t2, L2.ntiles, 0, lane, final_store);
        }
    } else {
        if (on1) layer_epilogue<RB, T1>(acc1, bv1, t1, L1.ntiles, 0, lane, final_store);
    }
}

template <int RB, int NW, int T0, int T1, int T2, bool MIX0 = false, typename Store>
__device__ __forceinline__ void chain_static(const StaticLayer& L0, const StaticLayer& L1, const StaticLayer& L2,
                                             const float* sIn, int ldin, float* bufA, int ldA, float* bufB,
                                             int ldB, int wv, int lane, Store&& final_store, bool sync_in) {
    chain_static<RB, NW, T0, T1, T2, MIX0>(L0, L1, L2, sIn, ldin, bufA, ldA, bufB, ldB, wv, lane, final_store,
                                           sync_in, [] {});
}

inline int max_tpw_for_rb(int rb) { return rb == 1 ? 8 : rb == 2 ? 4 : 2; }

// Host: tiles per wave per pass minimising (passes x tpw), i.e. the busiest wave's work.
inline int pick_tpw(int ntiles, int nwaves, int max_tpw) {
    const int cands[6] = {8, 6, 4, 3, 2, 1};
    int best = 1, best_cost = 1 << 30;
    for (int c : cands) {
        if (c > max_tpw) continue;
        const int passes = (ntiles + nwaves * c - 1) / (nwaves * c);
        const int cost = passes * c * 16 + passes;   // tile-times, then fewer passes
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

}  // namespace dsat
