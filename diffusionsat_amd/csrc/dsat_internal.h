// Internal structures shared by the kernels and the C-ABI layer (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/dsat.h"

namespace dsat {

constexpr int F = 128;        // feature_maps == query_maps (row = 512 B = 32 lanes x float4)
constexpr int LOGIT_MAPS = 8;
constexpr int AUX_W = 16;     // [normal(4) | noisy_num(2) | noise_scale(1) | denoised(2) | 0 x7]
constexpr int AUX_USED = 9;   // 4 + 5 = the non-state columns of v1 (model/query_sat.py:239)

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define DSAT_HIP(call)                                                        \
    do {                                                                      \
        hipError_t _e = (call);                                               \
        if (_e != hipSuccess) return ::dsat::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define DSAT_REQUIRE(cond, status, msg)            \
    do {                                           \
        if (!(cond)) {                             \
            ::dsat::set_error(msg);                \
            return (status);                       \
        }                                          \
    } while (0)

// ---- packed dense layer -------------------------------------------------------
struct PackedLayer {
    int in_dim = 0, out_dim = 0;  // logical Keras dims
    int out_ext = 0;              // extent of the output columns in the 128-wide padded layout (== out_dim unless the
                                  // layer's output is a concatenation of parts narrower than 128: see dsat_api.hip)
    int kpad = 0;                 // padded K (sum of padded input blocks)
    int npad = 0;                 // padded N (multiple of 32)
    float* wt = nullptr;          // dev [npad][kpad] = kernel^T, zero padded
    float* bias = nullptr;        // dev [npad]
    // MFMA-fragment order for direct-to-register streaming (kernels_fused.hip):
    // [npad/16 tiles][kpad/16 k-blocks][64 lanes][4]; lane l, element j holds
    // W[k = 16*kb + 4*(l>>4) + j][n = 16*t + (l&15)]
    float* wp = nullptr;
    // the same weights as three bf16 slices (w = h + m + l to 24 bits) for v_mfma_f32_16x16x32_bf16:
    // [npad/16 tiles][ksteps of 32][3 planes][64 lanes][8]; lane l, element j holds
    // slice(W[k = 32*s + 8*(l>>4) + j][n = 16*t + (l&15)]), zero beyond kpad (split_bf16.h)
    uint16_t* ws = nullptr;
    int ksteps() const { return (kpad + 31) / 32; }
    int ntiles() const { return npad / 16; }
    int kblocks() const { return kpad / 16; }
};

// Input block of a concat-free GEMM: `k` logical columns at A (row stride lda),
// occupying `kp` (multiple of 16) columns of the packed K axis.
struct BlockSpec { int k; int kp; };

}  // namespace dsat

struct dsat_weights {
    int device = -1;   // HIP device the slab lives on (the current device at dsat_weights_load)
    dsat_layout layout;
    // [mlp][layer]
    std::vector<std::vector<dsat::PackedLayer>> mlp;
    // fused first layer of variables_query and lit_mlp (same input v1): N = q_h_pad + lit_h
    dsat::PackedLayer ql1;
    int q_hpad = 0, lit_h = 0;
    float* slab = nullptr;  // one allocation behind every wt/bias pointer
    size_t slab_bytes = 0;
};

struct dsat_arena;

struct dsat_graph {
    int device = -1;   // HIP device the arrays live on (the current device at dsat_graph_create)
    dsat_arena* arena = nullptr;   // non-null: the device arrays are carved from this arena (not freed one by one)
    int32_t N = 0, C = 0, G = 0;
    int64_t E = 0;
    int32_t max_clause_len = 0, max_lit_deg = 0;
    // device arrays
    int32_t* c_rowptr = nullptr;  // [C+1]
    int32_t* c_lit = nullptr;     // [E] literal row in [0,2N)
    int32_t* l_rowptr = nullptr;  // [2N+1]
    int32_t* l_clause = nullptr;  // [E]
    int32_t* var_gid = nullptr;   // [N]
    int32_t* cl_gid = nullptr;    // [C]
    int32_t* var_seg = nullptr;   // [G+1]
    int32_t* cl_seg = nullptr;    // [G+1]
    float* deg_w = nullptr;       // [2N]  rsqrt(max(deg_lit,1))           (query_sat.py:194)
    float* vdeg_w = nullptr;      // [N]   4*rsqrt(max(deg_pos+deg_neg,1)) (:195)
    float* rev_w = nullptr;       // [C]   rsqrt(max(clause_len,1))        (:197)
    std::vector<int32_t> h_var_seg, h_cl_seg;
    std::vector<int32_t> h_c_rowptr, h_l_rowptr;   // host copies for the per-tile records
    std::vector<int32_t> h_l_clause;               // host copy of the literal->clause lists (heavy-row split)
    std::vector<int32_t> h_c_lit;                  // host copy of the clause->literal lists (heavy-clause split)
    bool identical_copies = false;                 // every graph is the same formula (replicate batch)
};

namespace dsat {

// ---- GEMM ----------------------------------------------------------------------
constexpr int GEMM_MAX_BLOCKS = 6;
struct GemmArgs {
    const float* A[GEMM_MAX_BLOCKS];
    int lda[GEMM_MAX_BLOCKS];
    int kchunks[GEMM_MAX_BLOCKS];   // 16-wide chunks per block
    int nblocks;
    const float* Wt;  // [npad][kpad]
    const float* bias;
    int kpad;
    int npad;
    float* out0;
    float* out1;
    int ldo0, ldo1;
    int split;        // columns >= split go to out1 at (col - split)
    int nstore0;      // store col < nstore0 in part 0
    int nstore1;      // store (col-split) < nstore1 in part 1
    int M;
    int act;          // 1 = leaky_relu(0.2)
};

// Device-side control block of one session.  Everything that changes from one
// denoising step to the next lives here, so that a captured hipGraph of a step can
// be replayed unchanged (kernel arguments are frozen at capture).
struct Ctl {
    int done;          // set when is_batch_sat fired (early exit)
    int steps_taken;   // last executed round
    int unsat_total;   // scratch of the SAT check
    int ticket;        // last-workgroup ticket of the SAT check
    float noise_scale;
    uint32_t step;     // denoising step index t (Philox counter word; injected-array row)
    int32_t n_steps;   // T
    int32_t rounds;
    uint64_t seed;
    const float* inj_normal;     // injected normal draws or nullptr (device Philox)
    const int32_t* inj_labels;
    const float* inj_uniform;
    int64_t normal_step_stride;  // elements per step in the injected arrays (0 = no step axis)
    int64_t labels_step_stride;
    int64_t uniform_step_stride;
    float* pred_steps_out;       // optional [T,N] record of every step's prediction
    const float* denoised;       // optional [N]: self-conditioning input of diffusion_step_self (query_sat.py:217-218)
    float* denoised_wr;          // sampler with self_supervised: sigmoid(prediction) of this step (DiffusionSampler.py:110-113)
    uint32_t denoised_from_step; // the input is used from this step on (the sampler's first step has none)
    // Early retirement of solved copies (dsat_session_set_retire): the batch is a prefix of `slots`;
    // slot g holds the copy whose ORIGINAL position is lid[g].  Random draws, injected noise and the
    // per-copy results are indexed by the original position, so a copy's trajectory does not depend
    // on which slot it sits in.  lid == nullptr: slot == original position.
    const int32_t* lid;
    int32_t n_per;               // variables per copy
    // QuerySAT(supervised=False): the logit map of a graph is chosen by the unsupervised clause loss of its own
    // clauses (model/query_sat.py:293-309 with use_linear_loss = False) instead of the KL to the labels (:289-292)
    int32_t unsupervised;
};

// softplus_mixed_loss_adj (loss/sat.py:92-100) of clause c for each of the eight logit maps, added to acc:
// val = exp(-sum_lit softplus(+-logit)); loss = val * -(log(1 - val + 1e-8) - log(1 + 1e-8)).  In fp32 1 + 1e-8 is 1
// (as in the reference's float32 graph), so the second logarithm is zero.
__device__ __forceinline__ void unsup_clause_loss8(const float* __restrict__ logits, const int32_t* __restrict__ rowptr,
                                                   const int32_t* __restrict__ c_lit, int c, int N,
                                                   float (&acc)[LOGIT_MAPS]) {
    float S[LOGIT_MAPS];
#pragma unroll
    for (int k = 0; k < LOGIT_MAPS; ++k) S[k] = 0.f;
    for (int e = rowptr[c]; e < rowptr[c + 1]; ++e) {
        const int l = c_lit[e];
        const int v = l < N ? l : l - N;
        const float sgn = l < N ? 1.f : -1.f;
        const float4 l0 = *reinterpret_cast<const float4*>(logits + (size_t)v * LOGIT_MAPS);
        const float4 l1 = *reinterpret_cast<const float4*>(logits + (size_t)v * LOGIT_MAPS + 4);
        const float lg[LOGIT_MAPS] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
        for (int k = 0; k < LOGIT_MAPS; ++k) {
            const float x = sgn * lg[k];
            S[k] += fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));       // tf.nn.softplus
        }
    }
#pragma unroll
    for (int k = 0; k < LOGIT_MAPS; ++k) {
        const float val = expf(-S[k]);
        acc[k] += val * -(logf((1.f - val) + 1e-8f) - logf(1.f + 1e-8f));
    }
}
// per_graph_loss of the unsupervised branch from the graph's sum (:309)
__device__ __forceinline__ float unsup_graph_loss(float sum) { return sqrtf(sum + 1e-6f) - sqrtf(1e-6f); }

// original (logical) index of variable v of the active prefix
__device__ __forceinline__ int logical_var(const Ctl* ctl, int v) {
    const int32_t* lid = ctl->lid;
    if (lid == nullptr) return v;
    const int np = ctl->n_per, g = v / np;
    return lid[g] * np + (v - g * np);
}

void launch_gemm(const GemmArgs& a, const int* done, hipStream_t st);

// ---- fused row-tile chain (kernels_chain.hip, fused_common.h) -------------------------
struct ChainLayer {
    const float4* wp;   // fragment-ordered weights
    const float* bias;
    int kblocks, ntiles, tpw;
    int tpw_half;       // tiles per wave when only half of the workgroup's waves run this layer
};
struct ChainArgs {
    ChainLayer L[4];
    int nlayers;
    const float* x;     // [rows][kpad0] packed input (blocks padded)
    int kpad0;
    int ld_buf0, ld_buf1;  // LDS row strides (floats, == 8 mod 16) of the two ping-pong activation tiles
    int rows;
    float* y;
    int ldy, ncols_y;
};
hipError_t launch_chain(const ChainArgs& a, int rb, int nwaves, hipStream_t st);

// ---- stand-alone split-product chain (kernels_split_op.hip; dsat_op_mlp_ex impl 2) -------------------
struct SplitOpArgs {
    const uint4* ws[3];     // per layer: weights as three bf16 planes in fragment order (PackedLayer::ws)
    const float* bias[3];
    int ks[3], nt[3];       // k-steps of 32 and 16-column tiles per layer (nt <= 16: one tile per wave)
    int nlayers;            // 2 or 3
    const float* x;         // [rows][kpad0] packed input (blocks padded)
    int kpad0, rows;
    int ld_in, ldh;         // row strides of the plane images in LDS, bf16 elements
    float* y;
    int ldy, ncols_y;
};
hipError_t launch_split_op(const SplitOpArgs& a, int rb, hipStream_t st);

// ---- fused round kernels (kernels_round.hip) ---------------------------------------------
// One message-passing round = three row-tile kernels (+ the select/SAT kernel):
//   P1 var tile   : finish PairNorm of the previous round (R13), output MLP + logits of the
//                   previous round (R14), then variables_query and lit_mlp chains (R2, R6)
//   P2 clause tile: finish clause PairNorm of the previous round (R10), literal->clause
//                   gather (R3, R5, R7), clause_mlp chain (R8, R9), column partial sums
//   P3 var tile   : clause->literal gather (R4, R11), update_gate chain (R12), column sums
// static_id / static_half_id: which compile-time blocking of fused_common.h's chain_static the
// chain matches when all NW (resp. NW/2) waves run it; 0 = none (generic per-layer path).
enum { CHAIN_GENERIC = 0, CHAIN_442 = 1, CHAIN_221 = 2, CHAIN_22 = 3, CHAIN_H21 = 4, CHAIN_H32 = 5, CHAIN_22M = 6,
       CHAIN_111 = 7, CHAIN_11 = 8, CHAIN_11M = 9, CHAIN_H11 = 10 };
struct ChainSpec { ChainLayer L[4]; int nlayers; int static_id, static_half_id; };
struct RoundArgs {
    // graph
    const int32_t *c_rowptr, *c_lit, *l_rowptr, *l_clause, *var_gid, *cl_gid, *var_seg, *cl_seg;
    const float *deg_w, *vdeg_w, *rev_w;
    const int32_t *vgseg, *cgseg;     // PairNorm partial-sum segments of each graph: [gseg[g], gseg[g+1])
    const int32_t *vmeta, *cmeta;     // per-tile records, TILE_META_INTS int32 each (see TileMetaDev)
    int N, C, G;           // ACTIVE sizes (bounds, grids); with retirement a prefix of the batch
    int NL, CL;            // layout sizes of the graph: literal rows are v / NL + v, the zero pad rows 2*NL / CL
    // state and intermediates (HBM)
    float *V, *Cs, *q, *sp, *lits, *closs4, *vla, *ncv, *newv, *logits, *psum_v;
    // clause column sums are read (previous round) and written (this round) by the same kernel,
    // P2, whose workgroups are not all co-resident on large batches: two buffers, by round parity
    const float* psum_c_rd;
    float* psum_c_wr;
    const float* noisy;    // [N,2] noisy_num of this denoising step
    const float* aux;      // [rounds,N,AUX_W] non-state columns of v1 for every round (aux_kernel, once per step)
    Ctl* ctl;
    float* trace_logits;   // optional [rounds,N,8]
    // select/SAT outputs (the S blocks appended to P2 and the stand-alone S kernel)
    float *pgl, *out_logits;
    int32_t *best, *graph_unsat;
    uint8_t* assign;
    // P1: the carried variable state is double-buffered by round: read V_in (previous round), write V_out;
    // P3 and the taps read `V` (= this round's V_out)
    const float* V_in;
    float* V_out;
    // P1 role B (kernels_round.hip): 64-row tiles running the output / query pair on otherwise idle CUs
    const int32_t* vmeta4;   // per 64-row tile records
    int pair_blocks;         // 0 = the 16-row tiles run the pair themselves
    int pair_first;          // first 64-row group role B covers (rows below it: the pair runs in the 16-row tile)
    int b_off_x, b_off_vlog, b_off_a, b_off_b, b_off_mean, b_ld_a, b_ld_b;   // LDS plan of a role-B workgroup
    int s_enable;          // P2 only: append G blocks running select/SAT of round-1
    int cn_blocks;         // P1: extra blocks that finish the clause PairNorm of round-1 (R10)
                           //     on CUs the variable tiles leave idle; 0 = P2 does it itself.
                           // P3: the same for THIS round's clause state (round 3), behind the tiles (cn_period = 0) or,
                           //     on grids of several waves of workgroups, as whole rows of 8 blocks after every
                           //     cn_period rows of 8 tiles, so that their row streaming runs beside the tiles' MFMA work
    int cs_ready;          // P2 only: clause_state was already normalised by P1's extra blocks
    int round;             // message-passing round this launch belongs to
    int check_done;        // early exit armed: skip when ctl->done
    int tail;              // P1 only: run the PairNorm + output-MLP part and stop
    long long* stamps;     // diagnostics (DSAT_STAMPS=1): [tile][8] shader-clock stamps, else nullptr
    // Heavy literal rows (more than HEAVY_ROW_T clauses; irregular-degree graphs): P3 walks the "light"
    // CSR in l_rowptr / l_clause, in which those rows are empty, and adds the row's partial sums that
    // heavy_gather_kernel left in hpart ([piece][closs4 | vla][128]) in piece order.
    const int2* l_hv;      // [2N] {first piece, pieces} or {-1, 0}; nullptr when the batch has no heavy row
    const float* hpart;
    // Heavy clauses (more than HEAVY_ROW_T literals), the same way on the clause side: P2's gather walks
    // cg_rowptr / cg_lit (the clause->literal CSR with those rows emptied; = c_rowptr / c_lit when there are
    // none) and adds the pieces in chpart ([piece][softplus sum | literal sum][128]); select/SAT keeps the
    // full CSR in c_rowptr / c_lit.
    const int32_t *cg_rowptr, *cg_lit;
    const int2* c_hv;      // [C] {first piece, pieces}; nullptr when the batch has no heavy clause
    const float* chpart;
    // Graphs spread over more than SEG2_MIN tiles: their per-tile column sums are pre-summed in chunks
    // (segsum2_kernel); [gseg2[g], gseg2[g+1]) is the graph's range in psum2 (empty for other graphs).
    const int32_t *vgseg2, *cgseg2;   // nullptr when the batch has no such graph
    const float *psum2_v, *psum2_c;
    // select/SAT workgroups: {graph, part, parts, 0} each (large graphs are cut into several parts)
    const int32_t* sel_tab;
    int n_sel;
    int sel_zero;          // P3: zero graph_unsat (some graph has several select parts)
    int sel_lead;          // P2: > 0 = the select/SAT workgroups occupy the first sel_lead blocks (multiple of 8)
    // debug taps (dsat_session_set_tap; nullptr in every production launch): values that only exist in
    // LDS on the fused path, written out by the tile that owns them
    float* tap_cmsg;       // P2: clause_messages [C,128]                       (query_sat.py:255-256)
    float* tap_grad;       // P3: variables_grad [N,128]                        (:242-246)
    float* tap_vloss;      // P3: variables_loss [2N,128] (pos rows, then neg)  (:269-273)
    // weights
    ChainSpec mq, mlit, mcl, mupd, mout;
    ChainSpec mq4, mout4;  // variables_query / variables_output blocked for 64-row tiles (P1 role B)
    // P1 role A on 32-row tiles with lit_mlp's products formed from bf16 slices (kernels_round.hip, role_a2):
    // a2_tiles > 0 replaces the 16-row A tiles; role B then covers every 64-row group
    const int32_t* vmeta2;  // 32-row records (as vmeta4)
    // per-tile list blocks at a fixed stride (dsat_api.hip, build_list_blocks): P3's tiles (two lists), P2's (one);
    // nullptr = none, *_max = the longest block that was copied (longer tiles read the global CSR)
    const int32_t *vlist, *clist;
    int vlist_stride, vlist_max, clist_stride, clist_max;
    int gj2_three;          // P2: no clause of the batch has more than three literals
    float inv_f;            // 1 / feature_maps (the per-row variance of PairNorm is over the real features)
    int fm;                 // feature_maps: state columns [fm, 128) are padding and stay zero
    int a2_tiles;
    int a1_split;          // role A's 16-row tiles run lit_mlp as split products (role_a1s): role B covers every row
    // update_gate's layer 0 is linear in its inputs and 4 of its 17 k-steps read only the carried state: extra
    // workgroups of P2 (role D, on CUs its tiles leave idle) form that partial product on 64-row tiles; P3 starts its
    // accumulators from it and skips those k-steps.  P2: d_blocks > 0, d_ws = update_gate layer 0 as planes,
    // upd_part = out; P3: upd_part != nullptr = in.
    int d_blocks, d_workers, d_sel, d_ks0, d_nt0, upd_ld;   // d_sel: the select workgroups are workers 0 .. n_sel-1
    const uint4* d_ws;
    float* upd_part;
    const uint4* a2_ws[3];  // lit_mlp's three layers as bf16 slices
    // P2 / P3 with fp32 products formed from bf16 slices (round_p2s_kernel, round_p3s_kernel, split_bf16.h): per
    // layer the weights as three bf16 planes in fragment order, k-steps of 32 and column tiles; LDS plan in
    // BYTES / bf16 elements (P3: s_buf1 = the input planes, s_buf2 = hidden A then hidden B)
    const uint4* sp_ws[3];
    int sp_layers;
    int sp_ks[3], sp_nt[3];
    int s_buf1, s_buf2, s_idx, s_idx_cap, s_mean;   // byte offsets (s_idx_cap: ints)
    int s_bufB, s_stage;                            // P3: hidden B and the fp32 stage of the column sums (byte offsets)
    int p3_bm;                                      // P3: rows per tile of the split-product form (16, or 32)
    int s_ld_seg, s_ld_hid;                         // row strides of the plane images, bf16 elements
    // LDS plan (float offsets / row strides), filled by plan_round_lds
    int ld_x, ld_a, ld_b, ld_v, off_x, off_a, off_b, off_vlog, off_mean;   // ld_v: row stride of the pre-mix state tile
    // ---- round 3 (appended: the offsets of everything above are those of the round-2 kernels, whose register
    // allocation -- P2's in particular, 76 scalar spills -- shifts with them: 1 % of a config-2 step, measured)
    // P1 role B: variables_output (128 -> 128 -> 8) as split products as well: the PairNorm of the 64-row tile emits the
    // pre-mix state as planes.  bo_ws[0] == nullptr: the fp32-MFMA chain (mout4).  LDS plan in float offsets: input
    // planes, hidden planes, the query chain's v1 planes afterwards, the PairNorm means; bo_ld: row stride, bf16 elements
    const uint4* bo_ws[2];
    int bo_ks[2], bo_nt[2];
    int bo_off_x, bo_off_h, bo_off_xq, bo_off_mean, bo_ld;
    int sel_pack;          // P2: graphs per select/SAT workgroup (1, or 4 x 256 threads on batches of many small graphs)
    int cn_period;         // P3: see cn_blocks
    int cn_rows;           // role C covers clause rows [0, cn_rows) (a multiple of P2's tile height unless it is all of them)
    int cs_ready_rows;     // P2: clause rows below this were normalised by a role C before this launch (replaces the
                           //     all-or-nothing cs_ready for P3-hosted role C on large batches)
    int p3_tiles;          // P3: variable tiles of the launch (the grid also holds the role-C blocks)
    // P2's select parts of a large graph (and the stand-alone select kernel): n_sel x LOGIT_MAPS words
    // {epoch, partial KL sum}, zeroed by the host at the start of every step; nullptr: every part sums everything
    unsigned long long* sel_exch;
    int sel_ticks;         // bound on a part's wait for the others' words, 100 MHz ticks (< 0: never served -- tests)
};
constexpr int TILE_META_INTS = 16;
constexpr int LIST_BLOCK_LOADS = 4;    // ints per thread with which a workgroup fetches a tile's list block
constexpr int ROUND_BM_VAR = 16;    // rows per variable-side tile
// Waves per workgroup of the round kernels (8 or 16; one workgroup per CU either way).  16 = four
// waves per SIMD at <= 128 VGPRs each: with the shallow weight ring that fits, and the tile
// preparation (gathers, PairNorm) and the layer boundaries hide twice as much latency: +5 % on the
// denoising step against 8 waves at <= 256 VGPRs (bench.py, config 2; build with
// -DDSAT_ROUND_NWAVES=8 to compare).
#ifndef DSAT_ROUND_NWAVES
#define DSAT_ROUND_NWAVES 16
#endif
constexpr int ROUND_NWAVES = DSAT_ROUND_NWAVES;
static_assert(ROUND_NWAVES == 8 || ROUND_NWAVES == 16, "round kernels are written for 8 or 16 waves");
constexpr int LIST_BLOCK_CAP = (LIST_BLOCK_LOADS * ROUND_NWAVES * 64 - 136 < 3840) ? LIST_BLOCK_LOADS * ROUND_NWAVES * 64 - 136 : 3840;   // entries of a tile's list block
// LDS of one round-kernel workgroup, in floats: all 160 KiB of the CU (one workgroup per CU), or, for the
// two-workgroups-per-CU experiment (-DDSAT_ROUND_NWAVES=8 -DDSAT_ROUND_LDS_FLOATS=20480), half of it
#ifndef DSAT_ROUND_LDS_FLOATS
#define DSAT_ROUND_LDS_FLOATS 40960
#endif
constexpr int ROUND_LDS_FLOATS = DSAT_ROUND_LDS_FLOATS;
constexpr int ROUND_WGS_PER_CU = 40960 / ROUND_LDS_FLOATS;
// tiles per wave the round kernels are built for (register budget: 128 VGPRs per lane at four waves per SIMD)
constexpr int ROUND_TPW_CAP = ROUND_NWAVES >= 16 ? 2 : (ROUND_WGS_PER_CU > 1 ? 4 : 8);
hipError_t launch_round_p1(const RoundArgs& a, hipStream_t st);
hipError_t launch_round_p2(const RoundArgs& a, int rb, hipStream_t st);
hipError_t launch_round_p3(const RoundArgs& a, hipStream_t st);
// Rows of the literal->clause CSR longer than this are "heavy": split into pieces of at most
// HEAVY_PIECE entries, each summed by one workgroup of heavy_gather_kernel (spread over the whole
// chip) instead of serially by the one half-wave that owns the row in its P3 tile.
constexpr int HEAVY_ROW_T = 32;
constexpr int HEAVY_ROW_MIN = 16;   // ... and once a graph has such a row, every row above this takes the heavy path (dsat_api.hip)
constexpr int HEAVY_PIECE = 512;
hipError_t launch_heavy_gather(const int2* pieces, int n_pieces, const int32_t* l_clause, const float* closs4,
                               const float* vla, int C, float* hpart, const int* done, hipStream_t st);
void launch_select_sat(const dsat_graph* g, const int32_t* sel_tab, int n_sel, const float* logits, Ctl* ctl,
                       float* pgl, int32_t* best, float* out_logits, uint8_t* assign, int32_t* graph_unsat,
                       int round, int set_done, unsigned long long* sel_exch, int sel_ticks,
                       hipStream_t st);
constexpr int SEG2_MIN = 32;      // graphs with more PairNorm segments than this get level-2 sums
constexpr int SEG2_CHUNK = 64;    // level-1 segments per level-2 sum
constexpr int SEL_PART_CLAUSES = 4096;   // clauses per select/SAT workgroup of a large graph
constexpr int SEL_EXCH_TICKS = 20000;   // bound on a select part's wait for the other parts' sums: 200 us of the 100 MHz clock
hipError_t launch_segsum2(const int2* chunks, int n_chunks, const float* psum, float* psum2, const int* done,
                          hipStream_t st);

// closs_out = closs_scale * exp(-S)
void launch_clause_gather(const dsat_graph* g, const float* q, const float* lits,
                          float* closs_out, float closs_scale, float* cmsg_out,
                          const int* done, hipStream_t st);
// `closs` holds closs/closs_scale' ; the kernel multiplies its sums by closs_scale
void launch_literal_gather(const dsat_graph* g, const float* q, const float* closs,
                           float closs_scale, const float* vla, float* grad_out,
                           float* vloss_out, const int* done, hipStream_t st);
int colsum_chunks(int G);
void launch_colsum(const int32_t* seg, int G, int nch, const float* x, float* partial,
                   const int* done, hipStream_t st);
// v = scale*PairNorm(x) + keep*old ; y_logit (optional) = v ; y_carry = carry ? v*0.2+v*0.8 : v
void launch_pairnorm_apply(float inv_f, const int32_t* gid, const int32_t* seg, int rows, int nch,
                           const float* x, const float* partial, const float* old,
                           float scale, float keep, int carry, float* y_carry, float* y_logit,
                           const int* done, hipStream_t st);
void launch_fill(float* p, float v, int64_t n, hipStream_t st);
void launch_fill_cols(float* p, float v, int64_t rows, int cols, hipStream_t st);   // [rows][128]: columns < cols, zero beyond
void launch_ctl_set(Ctl* dst, const Ctl& val, hipStream_t st);
void launch_aux(float* aux, int N, int rounds, const float* noisy_num, const Ctl* ctl, hipStream_t st);
void launch_labels(int32_t* labels, int N, const Ctl* ctl, hipStream_t st);
void launch_rng(float* normal, int32_t* labels, float* uniform, int N, int rounds, uint64_t seed,
                uint32_t step, hipStream_t st);
void launch_select(const dsat_graph* g, const float* logits, const int32_t* labels, Ctl* ctl,
                   float* pgl, int32_t* best, float* out_logits, uint8_t* assign,
                   int32_t* graph_unsat, hipStream_t st);
void launch_sat_check(const dsat_graph* g, const uint8_t* assign, int32_t* graph_unsat, Ctl* ctl,
                      int round, int set_done, hipStream_t st);
void launch_copy_if(const float* src, float* dst, int64_t n, const int* done, hipStream_t st);
void launch_sample_init(float* x0, float* x1, int N, int32_t* fixed_step, uint8_t* cum_acc, int G,
                        Ctl* ctl, hipStream_t st);
void launch_sample_pre(float* x0, float* x1, float* noisy, int N, Ctl* ctl, int rounds, hipStream_t st);
void launch_sample_post(float* x0, float* x1, const float* pred, const dsat_graph* g,
                        const int32_t* graph_unsat, int32_t* fixed_step, float* x_fixed,
                        uint8_t* cum_acc, Ctl* ctl, hipStream_t st);
void launch_sample_pre_retire(float* x0, float* x1, float* noisy, int N, Ctl* ctl, int rounds, hipStream_t st);
void launch_sample_post_retire(float* x0, float* x1, const float* pred, const int32_t* graph_unsat,
                               const int32_t* fixed_step, float* x_fixed, int N, int64_t n_full, const Ctl* ctl,
                               hipStream_t st);
void launch_retire_plan(int32_t* fixed_step, uint8_t* cum_acc, const int32_t* graph_unsat, int32_t* lid,
                        int32_t* move_src, int n_act, Ctl* ctl, int32_t* n_act_out, int reorder, hipStream_t st);
void launch_retire_move(const float* x0_in, const float* x1_in, float* x0_out, float* x1_out, const int32_t* move_src,
                        const int32_t* n_new_p, int n_act_old, int n_per, hipStream_t st);
void launch_sample_final_retire(const float* pred, const int32_t* lid, int n_act, int n_per,
                                const int32_t* fixed_step, const float* x_fixed, uint8_t* assignment, int n_full,
                                const uint8_t* cum_acc, uint8_t* solved, int G, hipStream_t st);
void launch_sample_final(const float* pred, const dsat_graph* g, const int32_t* fixed_step,
                         const float* x_fixed, uint8_t* assignment, const uint8_t* cum_acc,
                         uint8_t* solved, hipStream_t st);

}  // namespace dsat
