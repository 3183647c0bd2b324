// Stage executor of the fused round kernels: runs the Dense layers of several MLP chains on one
// row tile as a sequence of STAGES separated by workgroup barriers.  A stage is a list of JOBS
// (TPW adjacent 16-column tiles of one layer, over all of that layer's k-blocks); wave w
// runs jobs w, w+NW, ... of every stage.  Layers of different chains that do not depend on each
// other share a stage (P1: variables_output L0 + variables_query L0 + lit_mlp L0, ...), so a
// round costs 3+2+3 stages instead of 12 layer passes.
//
// The weight stream is ONE continuous software pipeline per wave: a ring of STAGE_PD k-blocks in
// registers (fused_common.h) whose prefetch cursor runs STAGE_PD k-blocks ahead of the MFMAs and
// walks straight across job, stage and barrier boundaries -- weights do not depend on the
// activations, so a barrier only waits for the LDS tile, never for the L2 latency of the next
// layer's first weights.
//
// Ring slots need static register indices AND must never be copied while a load is in flight
// (the compiler cannot know).  A `switch (phase)` over the slots makes hipcc insert phi copies of
// in-flight registers (seen in the ISA: NaNs), so the item loop is instead unrolled STAGE_PD
// times with slot p handled by step p, and everything dynamic -- end of job (epilogue), end of
// stage (barrier), end of work -- is a wave-uniform branch INSIDE each step.
//
// Rules that keep the hand-counted vmcnt valid (cdna_hip_programming.md 5.7):
//   * every consumed k-block issues exactly TPW refill loads (dummy loads of the slab base
//     once the cursor ran out), so (PD-1)*TPW younger loads are in flight at every wait;
//   * nothing else in the executor waits on vmcnt: job descriptors are SCALAR loads (indices
//     through readfirstlane), biases are staged in LDS, barriers are raw s_barrier after
//     lgkmcnt(0) (a __syncthreads() would drain the ring whenever an epilogue stored to HBM);
//   * stores issued by the final-store functor are younger queue entries: they can only make a
//     wait conservative, never wrong (vector memory operations retire in issue order).
#pragma once
#include "fused_common.h"
#include "stage_exec_types.h"

namespace dsat {

__device__ __forceinline__ StageJob load_job_scalar(const StagePlan& P, int stage, int k) {
    const int idx = __builtin_amdgcn_readfirstlane(P.first[stage] + k);
    return P.jobs[idx];
}

template <int NW, int TPW>
struct PrefetchCursor {
    int stage, k, kb, kblocks;
    unsigned voff[TPW];
    bool valid;

    __device__ __forceinline__ void seek(const StagePlan& P, int wave, int lane) {
        while (stage < P.nstages && k >= P.first[stage + 1] - P.first[stage]) { ++stage; k = wave; }
        valid = stage < P.nstages;
        kb = 0;
        if (valid) {
            const StageJob jb = load_job_scalar(P, stage, k);
            kblocks = jb.kblocks;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int tj = j < jb.ntiles ? j : jb.ntiles - 1;
                voff[j] = (jb.w_off + (unsigned)tj * jb.tile_stride + (unsigned)lane) * 16u;
            }
        } else {
            kblocks = 1 << 30;
#pragma unroll
            for (int j = 0; j < TPW; ++j) voff[j] = (unsigned)lane * 16u;
        }
    }
    __device__ __forceinline__ void advance(const StagePlan& P, int wave, int lane) {
        if (!valid) return;
        if (++kb == kblocks) { k += NW; seek(P, wave, lane); }
    }
    __device__ __forceinline__ const float4* base(const StagePlan& P) const { return P.wslab + (size_t)kb * 64; }
};

__device__ __forceinline__ void stage_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes are done
    __builtin_amdgcn_s_barrier();
}

// FinalStore: void(int kind, int row, int col, float v) -- row within the tile, col within the layer.
template <int RB, int TPW, int NW, typename FinalStore>
__device__ __forceinline__ void run_stages(const StagePlan& P, float* smem, int wave_in, int lane, FinalStore&& fs) {
    constexpr int PD = STAGE_PD;
    const int wave = __builtin_amdgcn_readfirstlane(wave_in);
    const float* sBias = smem + P.off_bias;

    f32x4 ring[PD][TPW];
    PrefetchCursor<NW, TPW> pf;
    pf.stage = 0;
    pf.k = wave;
    pf.seek(P, wave, lane);
#pragma unroll
    for (int p = 0; p < PD; ++p) {
        const float4* sb = pf.base(P);
#pragma unroll
        for (int j = 0; j < TPW; ++j) gload_b128(ring[p][j], pf.voff[j], sb);
        pf.advance(P, wave, lane);
    }

    // consume cursor.  `enter()` positions it on this wave's next job, passing (and joining) the
    // barrier of every stage it leaves; returns false when all stages are done.
    int stage = 0, k = wave, kb = 0;
    StageJob jb;
    const float* pa = nullptr;
    int ldin = 0;
    f32x4 acc[RB][TPW];
    float4 a_next[RB];
    auto enter = [&]() -> bool {
        while (k >= P.first[stage + 1] - P.first[stage]) {
            stage_barrier();
            ++stage;
            k = wave;
            if (stage >= P.nstages) return false;
        }
        jb = load_job_scalar(P, stage, k);
        ldin = P.buf_ld[jb.in_buf];
        pa = smem + P.buf_off[jb.in_buf] + (lane & 15) * ldin + 4 * (lane >> 4);
        kb = 0;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            a_next[rb] = *reinterpret_cast<const float4*>(pa + rb * 16 * ldin);
        }
        return true;
    };
    bool alive = enter();

    while (alive) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            if (alive) {
                const int last = jb.kblocks - 1;
                float4 a[RB];
                const int kc = kb < last ? kb + 1 : last;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    a[rb] = a_next[rb];
                    a_next[rb] = *reinterpret_cast<const float4*>(pa + rb * 16 * ldin + kc * 16);
                }
                const float4* sb = pf.base(P);
                wait_vmcnt_release<(PD - 1) * TPW>(ring[p][0]);
#pragma unroll
                for (int j = 1; j < TPW; ++j) release(ring[p][j]);
                ring_step<RB, TPW, true>(a, ring[p], pf.voff, sb, acc);
                pf.advance(P, wave, lane);
                if (kb == last) {
                    // epilogue of the job: bias (+ leaky_relu and LDS for hidden layers)
#pragma unroll
                    for (int j = 0; j < TPW; ++j) {
                        if (j < jb.ntiles) {
                            const int col = jb.col0 + j * 16 + (lane & 15);
                            const float bv = sBias[jb.bias_off + j * 16 + (lane & 15)];
                            if (jb.out_kind == 0) {
                                float* o = smem + P.buf_off[jb.out_buf];
                                const int ldo = P.buf_ld[jb.out_buf];
#pragma unroll
                                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                                    for (int i = 0; i < 4; ++i) {
                                        float v = acc[rb][j][i] + bv;
                                        v = v > 0.f ? v : 0.2f * v;
                                        o[(rb * 16 + 4 * (lane >> 4) + i) * ldo + col] = v;
                                    }
                            } else {
#pragma unroll
                                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                                    for (int i = 0; i < 4; ++i)
                                        fs((int)jb.out_kind, rb * 16 + 4 * (lane >> 4) + i, col, acc[rb][j][i] + bv);
                            }
                        }
                    }
                    k += NW;
                    alive = enter();
                } else {
                    ++kb;
                }
            }
        }
    }
    // drain: the compiler does not know the ring registers still have writes in flight
    wait_vmcnt_release<0>(ring[0][0]);
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
        for (int j = 0; j < TPW; ++j) release(ring[p][j]);
}

}  // namespace dsat
