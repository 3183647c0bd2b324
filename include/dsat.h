/*
 * dsat.h -- C-ABI of the MI355X-native DiffusionSAT denoising-sampling hot path.
 *
 * This is the drop-in boundary (DESIGN.md "Boundary").  The reference has no FFI:
 * its hot path is a Python method contract on TensorFlow tensors.  Each entry
 * point below names the reference interface it replaces (paths relative to the
 * reference repository).  Plain pointers and sizes only; no torch / HIP types.
 *
 * Conventions
 *   - every function returns 0 (DSAT_OK) or a negative dsat_status; the message
 *     for the last failure on the calling thread is dsat_last_error().
 *   - "dev" pointers are device (HBM) addresses owned by the caller (e.g. a torch
 *     tensor's data_ptr()); "host" pointers are ordinary host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *     stream-ordered and do not synchronise unless they return a host scalar
 *     (documented per function).
 *   - one session per (graph, weights) pair per GPU; a session is not re-entrant.
 *   - objects live on the HIP device that was current when they were created; a session needs its
 *     graph and weights on one device, and every call on an object must be made with that device
 *     current (hipSetDevice / torch.cuda.device), else it returns DSAT_E_STATE and launches nothing.
 *   - feature_maps and query_maps from 1 to 128 (the reference's defaults are 128 / 128,
 *     model/query_sat.py:87-89): narrower states, queries, messages and losses live zero-padded in the
 *     kernels' 128-wide rows, so every device buffer of this interface stays [rows][128] and the real
 *     columns come first.  Wider ones and logit_maps != 8 return DSAT_E_UNSUPPORTED, and the
 *     single-operator entry points (dsat_op_mlp*, dsat_op_pair_norm) take the released widths only.
 */
#ifndef DSAT_H
#define DSAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dsat_status {
    DSAT_OK = 0,
    DSAT_E_INVALID = -1,      /* bad argument / malformed graph */
    DSAT_E_UNSUPPORTED = -2,  /* layout the kernels are not built for */
    DSAT_E_HIP = -3,          /* HIP runtime error (no device, OOM, launch failure) */
    DSAT_E_STATE = -4         /* call order / session misuse */
} dsat_status;

typedef struct dsat_graph dsat_graph;
typedef struct dsat_weights dsat_weights;
typedef struct dsat_session dsat_session;

/* Order of the five MLPs everywhere in this API. */
enum { DSAT_MLP_VARIABLES_QUERY = 0, DSAT_MLP_LIT = 1, DSAT_MLP_CLAUSE = 2,
       DSAT_MLP_UPDATE_GATE = 3, DSAT_MLP_VARIABLES_OUTPUT = 4, DSAT_N_MLP = 5 };

/* Shape of the model, i.e. the arithmetic of QuerySAT.__init__
 * (model/query_sat.py:86-126).  dsat_default_layout() fills the reference defaults:
 * feature_maps=query_maps=128, logit_maps=8, layers {2,3,2,3,2},
 * hidden {153,512,204,230,128}. */
typedef struct dsat_layout {
    int32_t feature_maps;
    int32_t query_maps;
    int32_t logit_maps;
    int32_t n_layers[DSAT_N_MLP];
    int32_t hidden[DSAT_N_MLP];
} dsat_layout;

const char* dsat_last_error(void);
int32_t dsat_version(void);
int dsat_default_layout(dsat_layout* out);
/* Number of floats in a weight blob of this layout (see dsat_weights_load). */
int64_t dsat_layout_blob_floats(const dsat_layout* layout);
/* in/out widths of layer `layer` of MLP `mlp` (Keras Dense kernel is [in,out]). */
int dsat_layout_layer_dims(const dsat_layout* layout, int32_t mlp, int32_t layer,
                           int32_t* in_dim, int32_t* out_dim);

/* ---- graph ---------------------------------------------------------------------
 * Replaces SatSpecifics.create_adj_matrices (data/SatSpecifics.py:21-69) +
 * compute_adj_indices (data/dimac.py:14-18): builds, on the device, the clause->
 * literal CSR (A^T, `cl_adj_matrix`) and the literal->clause CSR (A, `adj_matrix`)
 * of the batched formula, the degree weights of QuerySAT.loop's prologue
 * (model/query_sat.py:193-197) and the per-graph segments that stand for
 * `variables_graph` / `clauses_graph` (:206-211).
 *
 *   clause_rowptr [C+1]  host, int32, rowptr[0]=0, non-decreasing
 *   clause_lits   [E]    host, int32, signed DIMACS literals in BATCH numbering:
 *                        +v / -v with 1 <= v <= N (duplicates allowed; they add up,
 *                        as duplicate COO entries do in tf.sparse.sparse_dense_matmul)
 *                        (the order of the literals inside a clause does not matter: the graph keeps them by
 *                        ascending literal row, so a clause's sums have one fixed order)
 *   var_graph_id  [N]    host, int32 in [0,G), non-decreasing, every graph non-empty
 *   clause_graph_id [C]  host, int32 in [0,G), non-decreasing
 * Every literal of a clause must belong to the clause's graph (block-diagonal batch).
 * The host arrays are only read during the call. */
int dsat_graph_create(const int32_t* clause_rowptr, const int32_t* clause_lits,
                      const int32_t* var_graph_id, const int32_t* clause_graph_id,
                      int32_t n_vars, int32_t n_clauses, int32_t n_graphs, dsat_graph** out);
/* The same graph from the three sparse tensors QuerySAT.diffusion_step is called with
 * (model/query_sat.py:467-481: adj_matrix, clauses_graph, variables_graph), exactly as
 * SatSpecifics.create_adj_matrices builds them (data/SatSpecifics.py:21-69), so that the reference's own
 * step_data can be passed through unchanged.  All arrays host memory, read during the call only.
 *   adj_indices [n_entries][2] int64   SparseTensor.indices of adj_matrix: (literal row, clause); rows
 *                                      0..N-1 are the positive literals, N..2N-1 the negative ones (:25-35);
 *                                      any order, duplicates kept (they add up)
 *   adj_values  [n_entries] or NULL    SparseTensor.values; the reference writes 1.0 (:76-80); whole numbers
 *                                      k >= 1 count as k parallel entries, anything else DSAT_E_UNSUPPORTED
 *   n_literal_rows, n_clauses          SparseTensor.dense_shape of adj_matrix = [2N, C]
 *   clauses_graph_indices [C][2]       indices of clauses_graph [G x C]: (graph, clause), one per clause (:45-50)
 *   variables_graph_indices [N][2]     indices of variables_graph [G x N]: (graph, variable) (:52-57)
 * Both routes order a clause's literals the same way (ascending literal row), so the entry order of the
 * tensor does not matter and this graph is, bit for bit, the one dsat_graph_create builds from the same
 * clauses.  Graphs must be contiguous index ranges, as the reference's batches are; otherwise DSAT_E_INVALID. */
int dsat_graph_create_coo(const int64_t* adj_indices, const float* adj_values, int64_t n_entries,
                          int64_t n_literal_rows, int64_t n_clauses,
                          const int64_t* clauses_graph_indices, int64_t n_clauses_graph_entries,
                          const int64_t* variables_graph_indices, int64_t n_variables_graph_entries,
                          int64_t n_graphs, dsat_graph** out);
void dsat_graph_destroy(dsat_graph* g);
int dsat_graph_info(const dsat_graph* g, int32_t* n_vars, int32_t* n_clauses,
                    int64_t* n_edges, int32_t* n_graphs);

/* ---- weights -------------------------------------------------------------------
 * Replaces the weight state of the five model/mlp.py MLPs that
 * tf.train.Checkpoint.restore fills (satuniformity/DiffusionSampler.py:215-227).
 * `blob` (host or device memory) is, for MLP 0..4 in the DSAT_MLP_* order and for
 * each Dense layer in order: kernel[in,out] row-major (Keras layout) then bias[out].
 * Kernels are re-packed on load (transposed, K/N padded with zeros) for the MFMA
 * GEMM kernels. */
int dsat_weights_load(const float* blob, int64_t n_floats, const dsat_layout* layout,
                      dsat_weights** out);
void dsat_weights_destroy(dsat_weights* w);

/* ---- session -------------------------------------------------------------------
 * Device workspace for one (graph, weights) pair: the state tensors of
 * QuerySAT.call/loop (model/query_sat.py:141,148) and all intermediates.
 * `max_rounds` bounds `rounds` of later calls (test_rounds, :88). */
int dsat_session_create(const dsat_graph* g, const dsat_weights* w, int32_t max_rounds,
                        dsat_session** out);
void dsat_session_destroy(dsat_session* s);
/* QuerySAT(..., supervised=...) (model/query_sat.py:86-91).  1 (default, what the released sampler constructs,
 * satuniformity/DiffusionSampler.py:199-201): a graph's logit map is the one with the smallest normalised
 * Bernoulli-KL to the step's labels (:289-292, :40-53).  0: the one with the smallest unsupervised clause loss of the
 * graph's own clauses, sqrt(sum_c softplus_mixed_loss_adj + 1e-6) - sqrt(1e-6) (:293-309 with use_linear_loss = False;
 * loss/sat.py:92-100); the labels are then unused.  Takes effect from the next step. */
int dsat_session_set_supervised(dsat_session* s, int32_t supervised);
/* 1 (default) = replay each denoising step from a captured hipGraph; 0 = eager. */
int dsat_session_set_graph_replay(dsat_session* s, int32_t enable);

/* ---- pooled objects (streams of batches) ----------------------------------------------------------
 * A stream of node-budget batches (data/dimac.py:267-293 feeding QuerySAT batch after batch) creates a
 * graph and a session per batch.  An arena is one device allocation plus one pinned host staging
 * area, made once: graphs and sessions created "in" it take their device memory from it (no
 * hipMalloc / hipFree per batch -- hipFree synchronises the whole device and would drain the other
 * batches in flight) and upload through the staging area with asynchronous copies on `stream`.
 * Use the same stream for the calls that follow (or synchronise): the uploads are stream-ordered.
 * dsat_arena_reset gives all of it back at once; every object created in the arena must have been
 * destroyed and its work finished.  A request that does not fit returns DSAT_E_INVALID; the sizes
 * it would have needed are reported by dsat_arena_usage (device_needed / pinned_needed). */
typedef struct dsat_arena dsat_arena;
int dsat_arena_create(int64_t device_bytes, int64_t pinned_bytes, dsat_arena** out);
int dsat_arena_reset(dsat_arena* a);
int dsat_arena_usage(const dsat_arena* a, int64_t* device_used, int64_t* pinned_used,
                     int64_t* device_needed, int64_t* pinned_needed);
/* Frees the arena.  While objects created in it are alive this does nothing but set dsat_last_error():
 * their device arrays live in the arena and their destroy calls count themselves out of it. */
void dsat_arena_destroy(dsat_arena* a);
int dsat_graph_create_in(dsat_arena* arena, void* stream, const int32_t* clause_rowptr,
                         const int32_t* clause_lits, const int32_t* var_graph_id,
                         const int32_t* clause_graph_id, int32_t n_vars, int32_t n_clauses,
                         int32_t n_graphs, dsat_graph** out);
int dsat_session_create_in(dsat_arena* arena, void* stream, const dsat_graph* g, const dsat_weights* w,
                           int32_t max_rounds, dsat_session** out);
int dsat_graph_create_coo_in(dsat_arena* arena, void* stream, const int64_t* adj_indices,
                             const float* adj_values, int64_t n_entries, int64_t n_literal_rows,
                             int64_t n_clauses, const int64_t* clauses_graph_indices,
                             int64_t n_clauses_graph_entries, const int64_t* variables_graph_indices,
                             int64_t n_variables_graph_entries, int64_t n_graphs, dsat_graph** out);

/* One denoising step = QuerySAT.diffusion_step (model/query_sat.py:467-481) ->
 * call (:133-184) -> loop (:186-373) at inference, for `rounds` message-passing
 * rounds.
 *   noise_scale            :476 argument
 *   noisy_num   dev [N,2]  :476 argument (x_noisy of DiffusionSampler.py:107)
 *   normal      dev [rounds,N,4] or NULL.  The tf.random.normal([n_vars,4]) of
 *               every round (:239), injected.  NULL = drawn on the device from
 *               Philox4x32-10 keyed by (seed, step_index) (oracle/philox.py mirrors it).
 *   labels      dev int32 [N] in {0,1} or NULL.  The random labels of :145
 *               (labels=None at inference); NULL = device Philox.
 *   early_exit  1 = stop at the first round whose selected assignment satisfies
 *               every clause of the batch (is_batch_sat, :330-338; utils/sat.py:118-124);
 *               0 = always run `rounds` rounds (fixed work, for throughput runs).
 * Outputs (all optional except prediction_out):
 *   logits_out        dev [N,8]         logits of the last executed round (:283)
 *   logits_rounds_out dev [rounds,N,8]  logits of every executed round (parity tests)
 *   prediction_out    dev [N]           "prediction" (:371-373, :480)
 *   best_map_out      dev int32 [G]     argmin logit map per graph (:317)
 *   steps_taken_out   dev int32 [1]     index of the last executed round (:478)
 * Stream-ordered; does not synchronise. */
int dsat_diffusion_step(dsat_session* s, float noise_scale, const float* noisy_num,
                        const float* normal, const int32_t* labels,
                        uint64_t seed, uint32_t step_index,
                        int32_t rounds, int32_t early_exit,
                        float* logits_out, float* logits_rounds_out, float* prediction_out,
                        int32_t* best_map_out, int32_t* steps_taken_out, void* stream);

/* QuerySAT.diffusion_step_self (model/query_sat.py:483-498): the same step with the
 * self-conditioning input `denoised_num` dev [N] (the previous step's sigmoid(prediction));
 * the model sees [denoised, 1 - denoised] in place of the two zero columns (:217-218). */
int dsat_diffusion_step_self(dsat_session* s, float noise_scale, const float* noisy_num,
                             const float* denoised_num, const float* normal, const int32_t* labels,
                             uint64_t seed, uint32_t step_index, int32_t rounds, int32_t early_exit,
                             float* logits_out, float* logits_rounds_out, float* prediction_out,
                             int32_t* best_map_out, int32_t* steps_taken_out, void* stream);

/* T denoising steps around dsat_diffusion_step = diffusion()
 * (satuniformity/DiffusionSampler.py:78-191) with use_baseline_sampling=True,
 * self_supervised=False (:17-21), entirely on the device: randomized rounding
 * (model/query_sat.py:55-60), the step, SATAccuracyTF.accuracy per graph
 * (metrics/sat_metrics.py:60-85), reverse_distribution_step_theoretic (:29-37) and
 * freeze-on-first-solution per graph (:154-170, :181-184).
 *   uniform dev [T,N]        or NULL   noise of randomized_rounding_tf, injected
 *   normal  dev [T,rounds,N,4] or NULL
 *   labels  dev int32 [T,N]  or NULL   (NULL = device Philox keyed by seed)
 * Outputs:
 *   assignment_out dev uint8 [N]  final predictions (:181-184), 1 = True
 *   solved_out     dev uint8 [G]  cum_accuracy per graph (:130)
 *   prediction_steps_out dev [T,N] or NULL: "prediction" logits of every step
 *   solved_fraction_out  host float* or NULL: mean(cum_accuracy) (:191); when not
 *                        NULL the call synchronises the stream before returning. */
int dsat_sample(dsat_session* s, int32_t n_steps, int32_t rounds, uint64_t seed,
                int32_t early_exit, const float* uniform, const float* normal,
                const int32_t* labels, uint8_t* assignment_out, uint8_t* solved_out,
                float* prediction_steps_out, float* solved_fraction_out, void* stream);

/* Early retirement for batches of identical copies of ONE formula (the sampler's batches,
 * data/diffusion_sat_instances.py:80-94): with enable = 1, dsat_sample / dsat_sample_self drop a copy
 * from the batch once it has been solved -- the reference freezes its bits at that step and never
 * uses its later predictions (satuniformity/DiffusionSampler.py:154-170, :181-184) -- so the remaining
 * denoising steps run on ever fewer copies.  A copy's random draws are keyed by its original position:
 * its outputs are those of the non-retiring run, except that the batch-global early exit
 * (is_batch_sat, model/query_sat.py:330-338) then refers to the copies still in the batch.  Costs one
 * host synchronisation per denoising step.  DSAT_E_UNSUPPORTED when the graphs of the batch are not
 * identical copies, or the formula is large enough to have its select / PairNorm work split; DSAT_E_STATE
 * while profiling is on (dsat_session_set_profile times the non-retiring launch sequence; the two exclude
 * each other, whichever is asked for second is refused). */
int dsat_session_set_retire(dsat_session* s, int32_t enable);

/* diffusion() with the reference's `self_supervised` switch on
 * (satuniformity/DiffusionSampler.py:21, :110-113): step t >= 1 runs diffusion_step_self with
 * sigmoid(prediction) of step t-1 as denoised_num; step 0 has no previous prediction (:93).
 * Arguments as dsat_sample. */
int dsat_sample_self(dsat_session* s, int32_t n_steps, int32_t rounds, uint64_t seed, int32_t early_exit,
                     const float* uniform, const float* normal, const int32_t* labels,
                     uint8_t* assignment_out, uint8_t* solved_out, float* prediction_steps_out,
                     float* solved_fraction_out, void* stream);

/* ---- single operators (parity tests and per-kernel roofline measurements) --------
 * The layers the kernels sit behind, exposed one at a time.  All pointers dev. */

/* MLP.call (model/mlp.py:42-50) of MLP `mlp` on x[rows, in_dim] -> y[rows, out_dim]. */
int dsat_op_mlp(dsat_session* s, int32_t mlp, const float* x, int32_t rows, float* y,
                void* stream);
/* Same operator with the implementation chosen explicitly: impl 0 = one GEMM launch per Dense
 * layer (kernels_gemm.hip), impl 1 = row-tile fused chain (kernels_chain.hip) with 16*rb rows
 * per workgroup and `nwaves` wavefronts (rb in {1,2,4}, nwaves in {4,8}), impl 2 = the same chain with every
 * fp32 product formed from bf16 slices on the bf16 matrix pipe (kernels_split_op.hip: the split_chain the round
 * kernels run, 16 wavefronts, rb in {1,2,4}; MLPs of 2-3 layers of at most 256 columns, i.e. all but lit_mlp;
 * DSAT_E_UNSUPPORTED otherwise). */
int dsat_op_mlp_ex(dsat_session* s, int32_t mlp, const float* x, int32_t rows, float* y,
                   int32_t impl, int32_t rb, int32_t nwaves, void* stream);
/* Measurement helper: average device time in microseconds of `iters` back-to-back runs of
 * MLP `mlp` on `rows` rows (HIP events on `stream`; synchronises). */
int dsat_bench_mlp(dsat_session* s, int32_t mlp, int32_t rows, int32_t impl, int32_t rb,
                   int32_t nwaves, int32_t iters, double* avg_us, void* stream);
/* softplus_loss_adj (loss/sat.py:125-137) fused with the literal->clause message
 * product (model/query_sat.py:255-256):
 *   closs_out[C,128] = exp(-(A^T softplus([q;-q])));  cmsg_out[C,128] = (A^T lits)*rsqrt(len)
 *   q dev [N,128], lits dev [2N,128]. */
int dsat_op_clause_gather(dsat_session* s, const float* q, const float* lits,
                          float* closs_out, float* cmsg_out, void* stream);
/* Inner-tape gradient in closed form (model/query_sat.py:227-246) fused with the
 * clause->literal message product (:269-273):
 *   grad_out[N,128] = (-sig(q)*(A closs)[:N] + sig(-q)*(A closs)[N:]) * 4*rsqrt(deg_var)
 *   vloss_out[2N,128] = (A vla) * rsqrt(deg_lit) */
int dsat_op_literal_gather(dsat_session* s, const float* q, const float* closs,
                           const float* vla, float* grad_out, float* vloss_out, void* stream);
/* PairNorm.call with subtract_mean=True (layers/normalization.py:43-71) per graph:
 * side 0 = variables [N,128], 1 = clauses [C,128]. y = PairNorm(x). */
int dsat_op_pair_norm(dsat_session* s, int32_t side, const float* x, float* y, void* stream);
/* is_batch_sat (utils/sat.py:118-124) of prediction logits[N]; result (0/1) to
 * host `is_sat_out`; synchronises. */
int dsat_op_is_batch_sat(dsat_session* s, const float* logits, int32_t* is_sat_out, void* stream);
/* is_graph_sat (utils/sat.py:165-180) of prediction logits[N]: sat_out dev uint8 [G], 1 = every clause
 * of the graph is satisfied by round(sigmoid(logits)) (the per-graph form predict_step uses when
 * prediction_tries > 1, model/query_sat.py:423-443).  Stream-ordered. */
int dsat_op_is_graph_sat(dsat_session* s, const float* logits, uint8_t* sat_out, void* stream);
/* The int encoding of a sampled assignment (utils/VariableAssignment.py:63-69: bit i <-> x_{i+1}) packed
 * on the device, so that samples() (satuniformity/DiffusionSampler.py:283-307) copies G * ceil(n/8)
 * bytes instead of one byte per variable: graph g's bits, little-endian, occupy bytes
 * [offsets[g], offsets[g+1]) of packed_out.  dsat_session_packed_bytes returns the total size and, when
 * offsets_out (host int32 [G+1]) is not NULL, the offsets.  assignment dev uint8 [N] as dsat_sample
 * returns it; packed_out dev uint8 [total].  Stream-ordered. */
int64_t dsat_session_packed_bytes(const dsat_session* s, int32_t* offsets_out);
int dsat_op_pack_assignment(dsat_session* s, const uint8_t* assignment, uint8_t* packed_out, void* stream);
/* Philox draws as the device path makes them (normal [rounds,N,4]; labels [N];
 * uniform [N]) for a given (seed, step_index); any output may be NULL. */
int dsat_op_rng(dsat_session* s, uint64_t seed, uint32_t step_index, int32_t rounds,
                float* normal_out, int32_t* labels_out, float* uniform_out, void* stream);

/* ---- layers (free-standing objects with the reference's own argument lists) -----------------------
 * MLP(layer_count, hidden_nmap, out_nmap, activation=tf.nn.leaky_relu)(inputs) of model/mlp.py:13-50 in the
 * configuration QuerySAT builds its five MLPs in (do_layer_norm=False, dropout_rate=0, out_activation=None;
 * model/query_sat.py:117-122): layer_count-1 Dense(hidden) with leaky_relu(0.2), then a linear Dense(out).  Any
 * widths; not bound to a session or to the five roles of dsat_layout.
 *   dims  host int32 [n_layers+1]: input width, then every layer's output width
 *   blob  host or dev: per layer kernel[in,out] row-major (Keras Dense), then bias[out]
 * dsat_mlp_forward: x dev [rows, dims[0]] -> y dev [rows, dims[n_layers]]; workspace dev, at least
 * dsat_mlp_workspace_floats(m, rows) floats, the caller's (a torch tensor).  Stream-ordered. */
typedef struct dsat_mlp dsat_mlp;
int dsat_mlp_create(int32_t n_layers, const int32_t* dims, const float* blob, int64_t n_floats,
                    dsat_mlp** out);
int64_t dsat_mlp_workspace_floats(const dsat_mlp* m, int32_t rows);
int dsat_mlp_forward(const dsat_mlp* m, const float* x, int32_t rows, float* y, float* workspace,
                     void* stream);
void dsat_mlp_destroy(dsat_mlp* m);
/* PairNorm(epsilon=1e-6, subtract_mean=False)(inputs, graph=None) of layers/normalization.py:31-71 on
 * x dev [rows, width], any width:
 *   subtract_mean != 0: inputs -= gather(graph @ inputs, graph.indices[:,0]) (:52-54), or the column mean over
 *                       all rows when there is no graph (n_graphs = 0, :55-57); then, in both cases,
 *   y = inputs * rsqrt(mean(inputs^2 over the features) + epsilon)              (:69-71)
 *   seg        host int32 [n_graphs+1]: graph g owns rows [seg[g], seg[g+1]) -- graph.indices[:,0] of the
 *              reference's membership matrices is non-decreasing (data/SatSpecifics.py:40-57)
 *   row_weight dev [rows] or NULL: graph.values per row (QuerySAT passes 1/|graph|, model/query_sat.py:206-211);
 *              NULL = 1/|graph|
 *   workspace  dev, dsat_pair_norm_workspace_floats(rows, width, n_graphs) floats; may be NULL when
 *              subtract_mean = 0
 * Stream-ordered, except that seg is copied to the device before the call returns (one stream wait). */
int64_t dsat_pair_norm_workspace_floats(int32_t rows, int32_t width, int32_t n_graphs);
int dsat_pair_norm(const float* x, int32_t rows, int32_t width, const int32_t* seg, int32_t n_graphs,
                   const float* row_weight, float epsilon, int32_t subtract_mean, float* y,
                   float* workspace, void* stream);

/* Average device time in microseconds of the kernels of the last
 * dsat_diffusion_step/dsat_sample call when profiling was enabled with
 * dsat_session_set_profile(s, 1) (HIP events on the launch stream, eager mode).
 * names/us_out have room for `cap` entries; returns the number of kernel classes. */
int dsat_session_set_profile(dsat_session* s, int32_t enable);
/* Diagnostics: when the session was created with the environment variable DSAT_STAMPS=1 the
 * fused round kernels record shader-clock stamps at their phase boundaries; this copies
 * the last launch's stamps ([3 kernels][4096 tiles][8 phases] int64) to host `out`. */
int dsat_session_read_stamps(dsat_session* s, int64_t* out, int64_t n);
int dsat_session_profile(dsat_session* s, int32_t cap, const char** names, double* us_out,
                         int64_t* launches_out);
/* Which arithmetic the round kernels of this session run their MLPs (model/mlp.py:42-50) in: bit 0 set = the
 * clause tile (P2), bit 1 set = the variable-update tile (P3), bit 2 set = lit_mlp of P1 on 32-row tiles, bit 5
 * set = lit_mlp of P1 on 16-row tiles (bit 3: role B's variables_query, bit 4: P3 on 32-row tiles; bit 6: the
 * `variables` part of update_gate's first layer is formed ahead of P3 by extra workgroups of P2 when CUs are idle; bit 7:
 * role B's variables_output) form
 * every fp32 product from three bf16 slices
 * per operand on the bf16 matrix pipe (six slice products, fp32 accumulation: fp32-level error, see
 * diffusionsat_amd/csrc/split_bf16.h); clear = v_mfma_f32_16x16x4_f32.  The environment variables
 * DSAT_P2_SPLIT=0 / DSAT_P3_SPLIT=0 / DSAT_P1_ASPLIT=0 / DSAT_P1_BSPLIT=0 (/ DSAT_P1_BOSPLIT=0 for role B's output
 * chain alone) at session creation select the fp32-MFMA form. */
int32_t dsat_session_split_mask(const dsat_session* s);
/* Several batches in flight on one GPU (a stream of batches, satuniformity/DiffusionSampler.py:283-307 with n > one
 * batch; data/dimac.py:267-293): prefer CU-cycles over the latency of one batch.  Today this runs lit_mlp
 * (model/query_sat.py:252-254) on 32-row tiles with split products also for batches small enough to give every
 * 16-row tile a CU of its own (bit 2 of dsat_session_split_mask; batches of more than one wave of workgroups get
 * it anyway; DSAT_P1_A2=0/1 in the environment at session creation overrides both).  Results are the same to fp32
 * rounding (not bitwise: another summation order).  Takes effect from the next step. */
int dsat_session_set_throughput(dsat_session* s, int32_t enable);

/* Debug tap (parity tests of the kernels that are actually hot): the intermediates of message-passing
 * round `round` of the NEXT dsat_diffusion_step call, copied out of the running kernels' buffers (or,
 * for values the fused kernels keep in LDS only, written by the tile that owns them).  The oracle's
 * trace carries the same tensors (oracle/querysat_oracle.py, reference lines per field).  All
 * pointers dev, each may be NULL.  While a tap is set the step runs eagerly (no graph replay);
 * dsat_session_set_tap(s, NULL) clears it.  Works on the fused path and on the per-layer path. */
typedef struct dsat_tap {
    int32_t round;
    float* query;               /* [N,128]   variables_query(v1)                      model/query_sat.py:240 */
    float* literals;            /* [2N,128]  lit_mlp(v1) split into +/- literal rows  :252-254 */
    float* clauses_loss4;       /* [C,128]   4*softplus_loss_adj(query)               :241, :248 */
    float* clause_messages;     /* [C,128]   (A^T literals)*rsqrt(len)                :255-256 */
    float* variables_loss_all;  /* [C,128]   clause_data[:, :128]                     :263 */
    float* new_clause_value;    /* [C,128]   clause_data[:, 128:] before PairNorm     :264 */
    float* variables_grad;      /* [N,128]   inner tape gradient * var_degree_weight  :242-246 */
    float* variables_loss;      /* [2N,128]  (A variables_loss_all)*degree_weight     :269-273 */
    float* new_variables;       /* [N,128]   update_gate(unit) before PairNorm        :278 */
    float* variables;           /* [N,128]   state after the round, carried copy      :279-280, :347 */
    float* clause_state;        /* [C,128]   state after the round, carried copy      :265-266, :348;
                                   produced while round+1 runs: needs round < rounds-1 */
} dsat_tap;
int dsat_session_set_tap(dsat_session* s, const dsat_tap* tap);

/* ---- instance preparation (host only, no GPU needed) ------------------------------------------
 * SAT/UNSAT (+ one model) of a CNF: the filter the reference's generators run through PySAT
 * (data/CNFGen.py:44-58 keeps drawing threshold 3-SAT until satisfiable; data/k_sat.py:44-60 adds
 * clauses until UNSAT).  clause_lits are signed DIMACS literals, clause c = [rowptr[c], rowptr[c+1]).
 * *result_out: 1 satisfiable (model_out[v] = value of x_{v+1}, if non-NULL), 0 unsatisfiable,
 * -1 conflict_limit (> 0) reached. */
int dsat_host_solve(int32_t n_vars, int32_t n_clauses, const int32_t* clause_rowptr,
                    const int32_t* clause_lits, int64_t conflict_limit, int32_t* result_out,
                    uint8_t* model_out);

/* One draw of the reference's NeuroSAT-style SR(n) generator (data/k_sat.py:36-93), natively: clauses
 * of width k = (1 w.p. p_k_2 else 2) + Geometric(p_geo) over distinct variables are added while the
 * formula stays satisfiable; the clause that makes it unsatisfiable gets its first literal negated;
 * duplicate and subsumed clauses are removed (utils/DimacsFile.py:110-128); for
 * solutions_multiplier > 1 some clauses are dropped (:66-84; the reference's default is 10).
 * n = uniform in [min_vars, max_vars].  Deterministic in `seed` (this library's own random stream,
 * not Python's).  Outputs: *n_vars_out, *n_clauses_out, clause CSR with signed DIMACS literals
 * (rowptr needs n_clauses+1 entries); DSAT_E_INVALID when a buffer is too small. */
int dsat_host_gen_sr(uint64_t seed, int32_t min_vars, int32_t max_vars, double p_k_2, double p_geo,
                     double solutions_multiplier, int32_t* n_vars_out, int32_t* n_clauses_out,
                     int32_t* clause_rowptr_out, int32_t rowptr_cap, int32_t* clause_lits_out,
                     int32_t lits_cap);

#ifdef __cplusplus
}
#endif
#endif /* DSAT_H */
