// C-ABI layer (include/dsat.h): graph / weights / session objects and the launch
// sequence of one denoising step.  Host-side orchestration only; the kernels are in
// kernels_*.hip.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <tuple>
#include <vector>

#include "dsat_internal.h"
#include "fused_common.h"

namespace dsat {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d in %s", (int)e, hipGetErrorString(e), file,
             line, what);
    g_last_error = buf;
    return DSAT_E_HIP;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Objects are bound to the HIP device that was current when they were created.  A call made with
// another device current would launch on the wrong GPU with pointers of this one (an illegal
// access at best), so every entry point checks first and launches nothing on a mismatch.
static int current_device(int* dev) {
    DSAT_HIP(hipGetDevice(dev));
    return DSAT_OK;
}
static int require_device(int want, const char* who) {
    int cur = -1;
    int rc = current_device(&cur);
    if (rc != DSAT_OK) return rc;
    if (cur != want) {
        set_error(std::string(who) + ": object lives on HIP device " + std::to_string(want) +
                  " but the calling thread's current device is " + std::to_string(cur));
        return DSAT_E_STATE;
    }
    return DSAT_OK;
}

}  // namespace dsat

// Device + pinned-host bump arena (include/dsat.h "pooled objects"): graphs and sessions created in
// an arena cost no hipMalloc / hipFree (hipFree synchronises the whole device, which would drain the
// other batches in flight) and upload through the arena's pinned staging area with asynchronous
// copies on the caller's stream.
struct dsat_arena {
    int device = -1;
    char* dev = nullptr;
    size_t dev_cap = 0, dev_off = 0;
    char* pin = nullptr;
    size_t pin_cap = 0, pin_off = 0;
    int live = 0;            // objects created in the arena and not yet destroyed
    size_t dev_need = 0, pin_need = 0;   // high-water marks of requests (also of failed ones)
    void* take_dev(size_t bytes) {
        dev_off = (dev_off + 255) & ~size_t(255);
        dev_need = std::max(dev_need, dev_off + bytes);
        if (dev_off + bytes > dev_cap) { dev_off += bytes; return nullptr; }
        void* p = dev + dev_off;
        dev_off += bytes;
        return p;
    }
    void* take_pin(size_t bytes) {
        pin_off = (pin_off + 63) & ~size_t(63);
        pin_need = std::max(pin_need, pin_off + bytes);
        if (pin_off + bytes > pin_cap) { pin_off += bytes; return nullptr; }
        void* p = pin + pin_off;
        pin_off += bytes;
        return p;
    }
};

namespace dsat {

// Upload `n` elements to a fresh device array: own allocation + synchronous copy, or (arena) a carve
// of the arena + an asynchronous copy on `st` out of the arena's pinned staging area.
template <typename T>
static hipError_t up(dsat_arena* a, hipStream_t st, T** dst, const T* src, size_t n) {
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (a == nullptr) {
        hipError_t e = hipMalloc((void**)dst, bytes);
        if (e != hipSuccess) return e;
        if (n) e = hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
    *dst = static_cast<T*>(a->take_dev(bytes));
    void* stage = n ? a->take_pin(n * sizeof(T)) : nullptr;
    if (*dst == nullptr || (n && stage == nullptr)) return hipErrorOutOfMemory;
    if (n == 0) return hipSuccess;
    memcpy(stage, src, n * sizeof(T));
    return hipMemcpyAsync(*dst, stage, n * sizeof(T), hipMemcpyHostToDevice, st);
}
static int arena_full(dsat_arena* a, const char* who) {
    set_error(std::string(who) + ": arena too small (needs " + std::to_string(a->dev_need) + " device bytes and " +
              std::to_string(a->pin_need) + " pinned bytes; has " + std::to_string(a->dev_cap) + " / " +
              std::to_string(a->pin_cap) + ")");
    return DSAT_E_INVALID;
}

// ---- bump allocator over one hipMalloc slab -------------------------------------
struct Slab {
    char* base = nullptr;
    size_t size = 0, off = 0;
    bool dry = true;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};

// ---- layout ----------------------------------------------------------------------
// Input blocks of the first layer of each MLP (concat members of query_sat.py:239,
// :258, :277) as (logical k, padded k).
// Concatenation members of an MLP's input: `k` logical columns each, in a slot of `kp` columns of the packed K axis.
// States, queries, messages and losses are [rows][128] arrays whatever feature_maps / query_maps (<= 128) are:
// narrower ones sit zero-padded in their 128-column slot, and the matching weight rows are zero.
static std::vector<BlockSpec> first_layer_blocks(int mlp, int fm = F, int qm = F) {
    switch (mlp) {
        case DSAT_MLP_VARIABLES_QUERY:
        case DSAT_MLP_LIT:
            return {{fm, F}, {AUX_USED, AUX_W}};
        case DSAT_MLP_CLAUSE:
            return {{fm, F}, {qm, F}, {qm, F}};
        case DSAT_MLP_UPDATE_GATE:
            return {{qm, F}, {fm, F}, {AUX_USED, AUX_W}, {qm, F}, {qm, F}};
        default:
            return {{fm, F}};
    }
}
// Output columns of an MLP's last layer in the padded layout: {first logical column, count, first padded column}.
// lit_mlp's [positive | negative] halves and clause_mlp's [variables_loss_all | new_clause_value] are stored to two
// 128-wide arrays, so the second part starts at padded column 128 (query_sat.py:253, :263-264).
struct OutPart { int src, n, dst; };
static std::vector<OutPart> last_layer_parts(int mlp, int fm, int qm, int out_dim) {
    if (mlp == DSAT_MLP_LIT) return {{0, qm, 0}, {qm, qm, F}};
    if (mlp == DSAT_MLP_CLAUSE) return {{0, qm, 0}, {qm, fm, F}};
    return {{0, out_dim, 0}};
}

static int layer_dims(const dsat_layout* L, int mlp, int layer, int* in_dim, int* out_dim) {
    if (mlp < 0 || mlp >= DSAT_N_MLP || layer < 0 || layer >= L->n_layers[mlp]) return DSAT_E_INVALID;
    const int v1 = L->feature_maps + AUX_USED;
    const int in0[DSAT_N_MLP] = {v1, v1, L->feature_maps + 2 * L->query_maps,
                                 L->query_maps + v1 + 2 * L->query_maps, L->feature_maps};
    const int outl[DSAT_N_MLP] = {L->query_maps, 2 * L->query_maps, L->feature_maps + L->query_maps,
                                  L->feature_maps, L->logit_maps};
    const int last = L->n_layers[mlp] - 1;
    *in_dim = layer == 0 ? in0[mlp] : L->hidden[mlp];
    *out_dim = layer == last ? outl[mlp] : L->hidden[mlp];
    return DSAT_OK;
}

}  // namespace dsat

using namespace dsat;

#define DSAT_ON_DEVICE(sess, who)                                   \
    do {                                                            \
        int _rc = require_device((sess)->device, who);              \
        if (_rc != DSAT_OK) return _rc;                             \
    } while (0)

// ============================================================================ misc
extern "C" const char* dsat_last_error(void) { return g_last_error.c_str(); }
extern "C" int32_t dsat_version(void) { return 100; }

extern "C" int dsat_default_layout(dsat_layout* out) {
    DSAT_REQUIRE(out != nullptr, DSAT_E_INVALID, "dsat_default_layout: null out");
    out->feature_maps = 128;
    out->query_maps = 128;
    out->logit_maps = 8;
    const int nl[DSAT_N_MLP] = {2, 3, 2, 3, 2};
    // int(128*1.2), 128*4, int(128*1.6), int(128*1.8), int(128*1)   (query_sat.py:109-122)
    const int hd[DSAT_N_MLP] = {153, 512, 204, 230, 128};
    for (int i = 0; i < DSAT_N_MLP; ++i) { out->n_layers[i] = nl[i]; out->hidden[i] = hd[i]; }
    return DSAT_OK;
}

extern "C" int dsat_layout_layer_dims(const dsat_layout* layout, int32_t mlp, int32_t layer,
                                       int32_t* in_dim, int32_t* out_dim) {
    DSAT_REQUIRE(layout && in_dim && out_dim, DSAT_E_INVALID, "dsat_layout_layer_dims: null argument");
    int i, o;
    if (layer_dims(layout, mlp, layer, &i, &o) != DSAT_OK) {
        set_error("dsat_layout_layer_dims: mlp/layer out of range");
        return DSAT_E_INVALID;
    }
    *in_dim = i;
    *out_dim = o;
    return DSAT_OK;
}

extern "C" int64_t dsat_layout_blob_floats(const dsat_layout* layout) {
    if (!layout) return -1;
    int64_t n = 0;
    for (int m = 0; m < DSAT_N_MLP; ++m)
        for (int l = 0; l < layout->n_layers[m]; ++l) {
            int i, o;
            if (layer_dims(layout, m, l, &i, &o) != DSAT_OK) return -1;
            n += (int64_t)i * o + o;
        }
    return n;
}

static int check_layout(const dsat_layout* L) {
    DSAT_REQUIRE(L != nullptr, DSAT_E_INVALID, "null layout");
    DSAT_REQUIRE(L->feature_maps >= 1 && L->feature_maps <= F && L->query_maps >= 1 && L->query_maps <= F, DSAT_E_UNSUPPORTED,
                 "feature_maps and query_maps must be in 1..128 (narrower states are zero-padded to the kernels' 128-wide rows)");
    DSAT_REQUIRE(L->logit_maps == LOGIT_MAPS, DSAT_E_UNSUPPORTED, "kernels are built for logit_maps = 8 only");
    for (int m = 0; m < DSAT_N_MLP; ++m) {
        DSAT_REQUIRE(L->n_layers[m] >= 2 && L->n_layers[m] <= 4, DSAT_E_UNSUPPORTED,
                     "each MLP needs 2..4 layers (query_sat.py:101-104)");
        DSAT_REQUIRE(L->hidden[m] >= 1 && L->hidden[m] <= 1024, DSAT_E_UNSUPPORTED,
                     "hidden width out of range (1..1024)");
    }
    return DSAT_OK;
}

// ============================================================================ graph
static int graph_create_impl(dsat_arena* arena, hipStream_t st, const int32_t* clause_rowptr,
                             const int32_t* clause_lits, const int32_t* var_graph_id,
                             const int32_t* clause_graph_id, int32_t N, int32_t C, int32_t G, dsat_graph** out) {
    DSAT_REQUIRE(out != nullptr, DSAT_E_INVALID, "dsat_graph_create: null out");
    *out = nullptr;
    DSAT_REQUIRE(N > 0 && C >= 0 && G > 0, DSAT_E_INVALID, "dsat_graph_create: need n_vars>0, n_clauses>=0, n_graphs>0");
    DSAT_REQUIRE(clause_rowptr && var_graph_id && (C == 0 || clause_graph_id), DSAT_E_INVALID,
                 "dsat_graph_create: null input array");
    DSAT_REQUIRE(clause_rowptr[0] == 0, DSAT_E_INVALID, "dsat_graph_create: clause_rowptr[0] must be 0");
    for (int c = 0; c < C; ++c)
        DSAT_REQUIRE(clause_rowptr[c + 1] >= clause_rowptr[c], DSAT_E_INVALID,
                     "dsat_graph_create: clause_rowptr must be non-decreasing");
    const int64_t E = clause_rowptr[C];
    DSAT_REQUIRE(E == 0 || clause_lits != nullptr, DSAT_E_INVALID, "dsat_graph_create: null clause_lits");
    DSAT_REQUIRE((int64_t)2 * N < (int64_t)1 << 30 && E < ((int64_t)1 << 31) - 1, DSAT_E_INVALID,
                 "dsat_graph_create: graph too large for int32 indexing");
    // the fused gathers address rows with 32-bit float offsets: (row * 128 + col) over 2N+1 literal rows
    // (sp, lits) and C+1 clause rows (closs4, vla), the +1 being the all-zero pad row
    DSAT_REQUIRE(((int64_t)2 * N + 1) * F < ((int64_t)1 << 32) && ((int64_t)C + 1) * F < ((int64_t)1 << 32),
                 DSAT_E_INVALID,
                 "dsat_graph_create: batch too large for the kernels' 32-bit row offsets "
                 "((2*n_vars+1)*128 and (n_clauses+1)*128 must stay below 2^32); split the batch");

    std::vector<int32_t> vseg(G + 1, 0), cseg(G + 1, 0);
    for (int v = 0; v < N; ++v) {
        const int g = var_graph_id[v];
        DSAT_REQUIRE(g >= 0 && g < G, DSAT_E_INVALID, "dsat_graph_create: var_graph_id out of range");
        DSAT_REQUIRE(v == 0 || g >= var_graph_id[v - 1], DSAT_E_INVALID,
                     "dsat_graph_create: var_graph_id must be non-decreasing (graphs are contiguous)");
        vseg[g + 1]++;
    }
    for (int c = 0; c < C; ++c) {
        const int g = clause_graph_id[c];
        DSAT_REQUIRE(g >= 0 && g < G, DSAT_E_INVALID, "dsat_graph_create: clause_graph_id out of range");
        DSAT_REQUIRE(c == 0 || g >= clause_graph_id[c - 1], DSAT_E_INVALID,
                     "dsat_graph_create: clause_graph_id must be non-decreasing (graphs are contiguous)");
        cseg[g + 1]++;
    }
    for (int g = 0; g < G; ++g) {
        DSAT_REQUIRE(vseg[g + 1] > 0, DSAT_E_INVALID, "dsat_graph_create: a graph has no variables");
        vseg[g + 1] += vseg[g];
        cseg[g + 1] += cseg[g];
    }

    // Literal order inside a clause: ascending literal row (positive literals by variable, then negative ones by
    // variable), whatever order the caller listed them in.  The reference's tensor lists a clause's positives before
    // its negatives too (adj_matrix_lit = concat([adj_indices_pos, adj_indices_neg]), data/SatSpecifics.py:33), and
    // a fixed order makes the sums of a clause's terms independent of how the formula was written down: the graph
    // built from the COO tensors (dsat_graph_create_coo, in any entry order) and the one built from the clause
    // lists are the same graph, bit for bit, and shuffling literals inside clauses changes nothing at all.
    std::vector<int32_t> canon(clause_lits, clause_lits + E);
    {
        auto row_of = [N](int32_t l) { return l > 0 ? (int64_t)l - 1 : (int64_t)N - 1 - (int64_t)l; };
        for (int c = 0; c < C; ++c)
            std::stable_sort(canon.begin() + clause_rowptr[c], canon.begin() + clause_rowptr[c + 1],
                             [&](int32_t a, int32_t b) { return row_of(a) < row_of(b); });
    }
    clause_lits = canon.data();

    // clause -> literal rows, literal degrees
    std::vector<int32_t> c_lit(E), l_rowptr(2 * (size_t)N + 1, 0);
    int32_t max_len = 0;
    for (int c = 0; c < C; ++c) {
        const int g = clause_graph_id[c];
        max_len = std::max(max_len, clause_rowptr[c + 1] - clause_rowptr[c]);
        for (int e = clause_rowptr[c]; e < clause_rowptr[c + 1]; ++e) {
            const int32_t l = clause_lits[e];
            const int32_t v = (l > 0 ? l : -l) - 1;
            DSAT_REQUIRE(l != 0 && v < N, DSAT_E_INVALID, "dsat_graph_create: literal out of range");
            DSAT_REQUIRE(var_graph_id[v] == g, DSAT_E_INVALID,
                         "dsat_graph_create: a clause references a variable of another graph");
            const int32_t row = l > 0 ? v : v + N;   // SatSpecifics.py:25-35
            c_lit[e] = row;
            l_rowptr[row + 1]++;
        }
    }
    int32_t max_deg = 0;
    for (size_t i = 0; i < 2 * (size_t)N; ++i) {
        max_deg = std::max(max_deg, l_rowptr[i + 1]);
        l_rowptr[i + 1] += l_rowptr[i];
    }
    // literal -> clause rows (counting sort: clauses ascending within a literal)
    std::vector<int32_t> l_clause(E), cursor(l_rowptr.begin(), l_rowptr.end() - 1);
    for (int c = 0; c < C; ++c)
        for (int e = clause_rowptr[c]; e < clause_rowptr[c + 1]; ++e) l_clause[cursor[c_lit[e]]++] = c;

    // degree weights, query_sat.py:193-197
    std::vector<float> deg_w(2 * (size_t)N), vdeg_w(N), rev_w(std::max(C, 1));
    for (size_t i = 0; i < 2 * (size_t)N; ++i) {
        const float d = (float)(l_rowptr[i + 1] - l_rowptr[i]);
        deg_w[i] = 1.0f / std::sqrt(std::max(d, 1.0f));
    }
    for (int v = 0; v < N; ++v) {
        const float d = (float)(l_rowptr[v + 1] - l_rowptr[v]) + (float)(l_rowptr[N + v + 1] - l_rowptr[N + v]);
        vdeg_w[v] = 4.0f * (1.0f / std::sqrt(std::max(d, 1.0f)));
    }
    for (int c = 0; c < C; ++c) {
        const float d = (float)(clause_rowptr[c + 1] - clause_rowptr[c]);
        rev_w[c] = 1.0f / std::sqrt(std::max(d, 1.0f));
    }

    std::unique_ptr<dsat_graph> g(new dsat_graph());
    { int rc = current_device(&g->device); if (rc != DSAT_OK) return rc; }
    if (arena != nullptr) {
        int rc = require_device(arena->device, "dsat_graph_create_in");
        if (rc != DSAT_OK) return rc;
        g->arena = arena;
    }
    g->N = N; g->C = C; g->G = G; g->E = E;
    g->max_clause_len = max_len; g->max_lit_deg = max_deg;
    g->h_var_seg = vseg; g->h_cl_seg = cseg;
    g->h_c_rowptr.assign(clause_rowptr, clause_rowptr + C + 1);
    g->h_l_rowptr = l_rowptr;
    g->h_l_clause = l_clause;
    g->h_c_lit = c_lit;
    {   // replicate batch? every graph the same formula up to the variable offset (early retirement needs it)
        bool same = G >= 1 && N % G == 0 && C % G == 0;
        const int np = same ? N / G : 0, mp = same ? C / G : 0;
        for (int gi = 0; same && gi < G; ++gi) same = vseg[gi] == gi * np && cseg[gi] == gi * mp;
        for (int c = mp; same && c < C; ++c) {
            const int c0 = c % mp, gi = c / mp;
            const int len = clause_rowptr[c + 1] - clause_rowptr[c];
            same = len == clause_rowptr[c0 + 1] - clause_rowptr[c0];
            for (int k = 0; same && k < len; ++k) {
                const int32_t a = clause_lits[clause_rowptr[c] + k], b = clause_lits[clause_rowptr[c0] + k];
                same = (a > 0) == (b > 0) && std::abs(a) - gi * np == std::abs(b);
            }
        }
        g->identical_copies = same;
    }
    hipError_t e = hipSuccess;
    if ((e = up(arena, st, &g->c_rowptr, clause_rowptr, (size_t)C + 1)) != hipSuccess ||
        (e = up(arena, st, &g->c_lit, c_lit.data(), (size_t)E)) != hipSuccess ||
        (e = up(arena, st, &g->l_rowptr, l_rowptr.data(), 2 * (size_t)N + 1)) != hipSuccess ||
        (e = up(arena, st, &g->l_clause, l_clause.data(), (size_t)E)) != hipSuccess ||
        (e = up(arena, st, &g->var_gid, var_graph_id, (size_t)N)) != hipSuccess ||
        (e = up(arena, st, &g->cl_gid, clause_graph_id, (size_t)C)) != hipSuccess ||
        (e = up(arena, st, &g->var_seg, vseg.data(), (size_t)G + 1)) != hipSuccess ||
        (e = up(arena, st, &g->cl_seg, cseg.data(), (size_t)G + 1)) != hipSuccess ||
        (e = up(arena, st, &g->deg_w, deg_w.data(), 2 * (size_t)N)) != hipSuccess ||
        (e = up(arena, st, &g->vdeg_w, vdeg_w.data(), (size_t)N)) != hipSuccess ||
        (e = up(arena, st, &g->rev_w, rev_w.data(), (size_t)C)) != hipSuccess) {
        g->arena = nullptr;            // nothing counted yet
        if (arena == nullptr) dsat_graph_destroy(g.release());
        if (arena != nullptr && e == hipErrorOutOfMemory) return arena_full(arena, "dsat_graph_create_in");
        return hip_fail(e, "graph upload", __FILE__, __LINE__);
    }
    if (arena != nullptr) arena->live++;
    *out = g.release();
    return DSAT_OK;
}

extern "C" int dsat_graph_create(const int32_t* clause_rowptr, const int32_t* clause_lits,
                                  const int32_t* var_graph_id, const int32_t* clause_graph_id,
                                  int32_t N, int32_t C, int32_t G, dsat_graph** out) {
    return graph_create_impl(nullptr, nullptr, clause_rowptr, clause_lits, var_graph_id, clause_graph_id, N, C, G, out);
}

extern "C" int dsat_graph_create_in(dsat_arena* arena, void* stream, const int32_t* clause_rowptr,
                                     const int32_t* clause_lits, const int32_t* var_graph_id,
                                     const int32_t* clause_graph_id, int32_t N, int32_t C, int32_t G,
                                     dsat_graph** out) {
    DSAT_REQUIRE(arena != nullptr, DSAT_E_INVALID, "dsat_graph_create_in: null arena");
    return graph_create_impl(arena, (hipStream_t)stream, clause_rowptr, clause_lits, var_graph_id, clause_graph_id,
                             N, C, G, out);
}

// The same graph from the tensors QuerySAT.diffusion_step receives (model/query_sat.py:467-481), exactly as
// SatSpecifics.create_adj_matrices (data/SatSpecifics.py:21-69) builds them.  Host-side conversion to the clause
// lists graph_create_impl takes (entries bucketed by clause; graph_create_impl orders a clause's literals itself).
static int graph_create_coo_impl(dsat_arena* arena, hipStream_t st, const int64_t* adj_indices,
                                 const float* adj_values, int64_t n_entries, int64_t n_literal_rows,
                                 int64_t n_clauses, const int64_t* cg_indices, int64_t n_cg,
                                 const int64_t* vg_indices, int64_t n_vg, int64_t n_graphs, dsat_graph** out) {
    DSAT_REQUIRE(out != nullptr, DSAT_E_INVALID, "dsat_graph_create_coo: null out");
    *out = nullptr;
    DSAT_REQUIRE(n_entries >= 0 && n_clauses >= 0 && n_graphs > 0 && n_literal_rows > 0, DSAT_E_INVALID,
                 "dsat_graph_create_coo: need n_entries>=0, n_clauses>=0, n_graphs>0, n_literal_rows>0");
    DSAT_REQUIRE(n_literal_rows % 2 == 0, DSAT_E_INVALID,
                 "dsat_graph_create_coo: adj_matrix must have 2*n_vars rows (positive literals, then negative ones)");
    DSAT_REQUIRE(n_literal_rows < ((int64_t)1 << 31) && n_clauses < ((int64_t)1 << 31) && n_graphs < ((int64_t)1 << 31),
                 DSAT_E_INVALID, "dsat_graph_create_coo: graph too large for int32 indexing");
    const int64_t N = n_literal_rows / 2, C = n_clauses, G = n_graphs;
    DSAT_REQUIRE((n_entries == 0 || adj_indices) && (C == 0 || cg_indices) && vg_indices, DSAT_E_INVALID,
                 "dsat_graph_create_coo: null index array");
    DSAT_REQUIRE(n_vg == N, DSAT_E_INVALID,
                 "dsat_graph_create_coo: variables_graph must hold one entry per variable (data/SatSpecifics.py:52-57)");
    DSAT_REQUIRE(n_cg == C, DSAT_E_INVALID,
                 "dsat_graph_create_coo: clauses_graph must hold one entry per clause (data/SatSpecifics.py:45-50)");
    // membership matrices: one (graph, index) pair per column, in any order
    std::vector<int32_t> vgid((size_t)N, -1), cgid((size_t)C, -1);
    for (int64_t i = 0; i < N; ++i) {
        const int64_t g = vg_indices[2 * i], v = vg_indices[2 * i + 1];
        DSAT_REQUIRE(g >= 0 && g < G && v >= 0 && v < N, DSAT_E_INVALID, "dsat_graph_create_coo: variables_graph index out of range");
        DSAT_REQUIRE(vgid[v] < 0, DSAT_E_INVALID, "dsat_graph_create_coo: a variable belongs to two graphs");
        vgid[v] = (int32_t)g;
    }
    for (int64_t i = 0; i < C; ++i) {
        const int64_t g = cg_indices[2 * i], c = cg_indices[2 * i + 1];
        DSAT_REQUIRE(g >= 0 && g < G && c >= 0 && c < C, DSAT_E_INVALID, "dsat_graph_create_coo: clauses_graph index out of range");
        DSAT_REQUIRE(cgid[c] < 0, DSAT_E_INVALID, "dsat_graph_create_coo: a clause belongs to two graphs");
        cgid[c] = (int32_t)g;
    }
    // adjacency: value k (a whole number >= 1; the reference writes 1.0) = k parallel entries
    std::vector<int32_t> rowptr((size_t)C + 1, 0);
    std::vector<int32_t> mult((size_t)n_entries, 1);
    int64_t E = 0;
    for (int64_t e = 0; e < n_entries; ++e) {
        const int64_t l = adj_indices[2 * e], c = adj_indices[2 * e + 1];
        DSAT_REQUIRE(l >= 0 && l < n_literal_rows && c >= 0 && c < C, DSAT_E_INVALID,
                     "dsat_graph_create_coo: adj_matrix index out of range");
        if (adj_values != nullptr) {
            const float v = adj_values[e];
            DSAT_REQUIRE(v >= 1.f && v <= 1024.f && v == std::floor(v), DSAT_E_UNSUPPORTED,
                         "dsat_graph_create_coo: adj_matrix values must be whole multiplicities >= 1 (the reference "
                         "writes 1.0, data/SatSpecifics.py:76-80)");
            mult[e] = (int32_t)v;
        }
        rowptr[c + 1] += mult[e];
        E += mult[e];
        DSAT_REQUIRE(E < ((int64_t)1 << 31) - 1, DSAT_E_INVALID, "dsat_graph_create_coo: graph too large for int32 indexing");
    }
    for (int64_t c = 0; c < C; ++c) rowptr[c + 1] += rowptr[c];
    std::vector<int32_t> lits((size_t)E), cursor(rowptr.begin(), rowptr.end() - 1);
    for (int64_t e = 0; e < n_entries; ++e) {
        const int64_t l = adj_indices[2 * e], c = adj_indices[2 * e + 1];
        const int32_t lit = l < N ? (int32_t)(l + 1) : -(int32_t)(l - N + 1);   // data/SatSpecifics.py:25-35
        for (int k = 0; k < mult[e]; ++k) lits[cursor[c]++] = lit;
    }
    return graph_create_impl(arena, st, rowptr.data(), lits.data(), vgid.data(), cgid.data(), (int32_t)N, (int32_t)C,
                             (int32_t)G, out);
}

extern "C" int dsat_graph_create_coo(const int64_t* adj_indices, const float* adj_values, int64_t n_entries,
                                      int64_t n_literal_rows, int64_t n_clauses,
                                      const int64_t* clauses_graph_indices, int64_t n_clauses_graph_entries,
                                      const int64_t* variables_graph_indices, int64_t n_variables_graph_entries,
                                      int64_t n_graphs, dsat_graph** out) {
    return graph_create_coo_impl(nullptr, nullptr, adj_indices, adj_values, n_entries, n_literal_rows, n_clauses,
                                 clauses_graph_indices, n_clauses_graph_entries, variables_graph_indices,
                                 n_variables_graph_entries, n_graphs, out);
}

extern "C" int dsat_graph_create_coo_in(dsat_arena* arena, void* stream, const int64_t* adj_indices,
                                         const float* adj_values, int64_t n_entries, int64_t n_literal_rows,
                                         int64_t n_clauses, const int64_t* clauses_graph_indices,
                                         int64_t n_clauses_graph_entries, const int64_t* variables_graph_indices,
                                         int64_t n_variables_graph_entries, int64_t n_graphs, dsat_graph** out) {
    DSAT_REQUIRE(arena != nullptr, DSAT_E_INVALID, "dsat_graph_create_coo_in: null arena");
    return graph_create_coo_impl(arena, (hipStream_t)stream, adj_indices, adj_values, n_entries, n_literal_rows,
                                 n_clauses, clauses_graph_indices, n_clauses_graph_entries, variables_graph_indices,
                                 n_variables_graph_entries, n_graphs, out);
}

extern "C" void dsat_graph_destroy(dsat_graph* g) {
    if (!g) return;
    if (g->arena != nullptr) {
        g->arena->live--;              // the arrays go back with dsat_arena_reset
    } else {
        void* ptrs[] = {g->c_rowptr, g->c_lit, g->l_rowptr, g->l_clause, g->var_gid, g->cl_gid,
                        g->var_seg, g->cl_seg, g->deg_w, g->vdeg_w, g->rev_w};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    }
    delete g;
}

// ============================================================================ arena
extern "C" int dsat_arena_create(int64_t device_bytes, int64_t pinned_bytes, dsat_arena** out) {
    DSAT_REQUIRE(out != nullptr, DSAT_E_INVALID, "dsat_arena_create: null out");
    *out = nullptr;
    DSAT_REQUIRE(device_bytes > 0 && pinned_bytes > 0, DSAT_E_INVALID, "dsat_arena_create: sizes must be positive");
    std::unique_ptr<dsat_arena> a(new dsat_arena());
    { int rc = current_device(&a->device); if (rc != DSAT_OK) return rc; }
    DSAT_HIP(hipMalloc((void**)&a->dev, (size_t)device_bytes));
    hipError_t e = hipHostMalloc((void**)&a->pin, (size_t)pinned_bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipFree(a->dev);
        return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
    }
    a->dev_cap = (size_t)device_bytes;
    a->pin_cap = (size_t)pinned_bytes;
    *out = a.release();
    return DSAT_OK;
}

extern "C" int dsat_arena_reset(dsat_arena* a) {
    DSAT_REQUIRE(a != nullptr, DSAT_E_INVALID, "dsat_arena_reset: null arena");
    DSAT_REQUIRE(a->live == 0, DSAT_E_STATE,
                 "dsat_arena_reset: " + std::to_string(a->live) + " object(s) created in the arena are still alive");
    a->dev_off = a->pin_off = 0;
    return DSAT_OK;
}

extern "C" int dsat_arena_usage(const dsat_arena* a, int64_t* device_used, int64_t* pinned_used,
                                 int64_t* device_needed, int64_t* pinned_needed) {
    DSAT_REQUIRE(a != nullptr, DSAT_E_INVALID, "dsat_arena_usage: null arena");
    if (device_used) *device_used = (int64_t)a->dev_off;
    if (pinned_used) *pinned_used = (int64_t)a->pin_off;
    if (device_needed) *device_needed = (int64_t)a->dev_need;
    if (pinned_needed) *pinned_needed = (int64_t)a->pin_need;
    return DSAT_OK;
}

extern "C" void dsat_arena_destroy(dsat_arena* a) {
    if (!a) return;
    if (a->live != 0) {   // graphs / sessions carved from it still point into it (and count themselves out at destroy)
        set_error("dsat_arena_destroy: " + std::to_string(a->live) + " object(s) created in the arena are still alive; "
                  "destroy them first (the arena was NOT freed)");
        return;
    }
    if (a->dev) (void)hipFree(a->dev);
    if (a->pin) (void)hipHostFree(a->pin);
    delete a;
}

extern "C" int dsat_graph_info(const dsat_graph* g, int32_t* n_vars, int32_t* n_clauses, int64_t* n_edges,
                                int32_t* n_graphs) {
    DSAT_REQUIRE(g != nullptr, DSAT_E_INVALID, "dsat_graph_info: null graph");
    if (n_vars) *n_vars = g->N;
    if (n_clauses) *n_clauses = g->C;
    if (n_edges) *n_edges = g->E;
    if (n_graphs) *n_graphs = g->G;
    return DSAT_OK;
}

// ============================================================================ weights
// x = h + m + l with three bf16 values (round to nearest even at every step; the residuals are exact in
// fp32): the slices the split-product kernels multiply (split_bf16.h has the device twin).
static inline uint16_t bf16_rne_host(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u);
    // a finite weight within half a bf16 ulp of the format's end would round to infinity: truncate it instead (the
    // residual slices then carry the rest, still exactly -- weights are finite, dsat_weights_load checks)
    if ((r & 0x7F800000u) == 0x7F800000u && (u & 0x7F800000u) != 0x7F800000u) return (uint16_t)(u >> 16);
    return (uint16_t)(r >> 16);
}
static inline float bf16_to_float_host(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline void split3_host(float x, uint16_t (&o)[3]) {
    o[0] = bf16_rne_host(x);
    const float r1 = x - bf16_to_float_host(o[0]);
    o[1] = bf16_rne_host(r1);
    const float r2 = r1 - bf16_to_float_host(o[1]);
    o[2] = bf16_rne_host(r2);
}

// Pack one Dense layer: wt[npad][kpad] = kernel^T with each input block padded to a
// multiple of 16 along K and N padded to a multiple of 32, zeros elsewhere.
static void pack_layer(const float* kernel, const float* bias, int in_dim, int out_dim,
                       const std::vector<BlockSpec>& blocks, int npad, int n_off, int kpad, float* wt,
                       float* bpad, const std::vector<OutPart>* parts = nullptr) {
    const std::vector<OutPart> whole{{0, out_dim, 0}};
    const std::vector<OutPart>& pp = parts != nullptr ? *parts : whole;
    int src = 0, dst = 0;
    for (const BlockSpec& b : blocks) {
        for (int j = 0; j < b.k; ++j)
            for (const OutPart& q : pp)
                for (int n = 0; n < q.n; ++n)
                    wt[(size_t)(n_off + q.dst + n) * kpad + dst + j] = kernel[(size_t)(src + j) * out_dim + q.src + n];
        src += b.k;
        dst += b.kp;
    }
    (void)in_dim; (void)npad;
    for (const OutPart& q : pp)
        for (int n = 0; n < q.n; ++n) bpad[n_off + q.dst + n] = bias[q.src + n];
}

extern "C" int dsat_weights_load(const float* blob, int64_t n_floats, const dsat_layout* layout,
                                  dsat_weights** out) {
    DSAT_REQUIRE(out != nullptr, DSAT_E_INVALID, "dsat_weights_load: null out");
    *out = nullptr;
    int rc = check_layout(layout);
    if (rc != DSAT_OK) return rc;
    DSAT_REQUIRE(blob != nullptr, DSAT_E_INVALID, "dsat_weights_load: null blob");
    const int64_t need = dsat_layout_blob_floats(layout);
    DSAT_REQUIRE(n_floats == need, DSAT_E_INVALID,
                 "dsat_weights_load: blob has " + std::to_string(n_floats) + " floats, layout needs " +
                     std::to_string(need));
    std::vector<float> host((size_t)need);
    DSAT_HIP(hipMemcpy(host.data(), blob, (size_t)need * sizeof(float), hipMemcpyDefault));
    for (int64_t i = 0; i < need; ++i)
        DSAT_REQUIRE(std::isfinite(host[i]), DSAT_E_INVALID, "dsat_weights_load: blob holds a non-finite value");

    std::unique_ptr<dsat_weights> w(new dsat_weights());
    { int rc = current_device(&w->device); if (rc != DSAT_OK) return rc; }
    w->layout = *layout;
    w->mlp.resize(DSAT_N_MLP);

    struct Staged { PackedLayer* dst; std::vector<float> wt, bias; };
    std::vector<Staged> staged;
    const float* p = host.data();
    const float* l0_kernel[2] = {nullptr, nullptr};
    const float* l0_bias[2] = {nullptr, nullptr};
    for (int m = 0; m < DSAT_N_MLP; ++m) {
        w->mlp[m].resize(layout->n_layers[m]);
        for (int l = 0; l < layout->n_layers[m]; ++l) {
            int in_dim, out_dim;
            layer_dims(layout, m, l, &in_dim, &out_dim);
            const float* kernel = p;
            const float* bias = p + (size_t)in_dim * out_dim;
            p = bias + out_dim;
            std::vector<BlockSpec> blocks = l == 0 ? first_layer_blocks(m, layout->feature_maps, layout->query_maps)
                                                    : std::vector<BlockSpec>{{in_dim, round_up(in_dim, 16)}};
            const bool last = l == layout->n_layers[m] - 1;
            const std::vector<OutPart> parts = last ? last_layer_parts(m, layout->feature_maps, layout->query_maps, out_dim)
                                                    : std::vector<OutPart>{{0, out_dim, 0}};
            PackedLayer& pl = w->mlp[m][l];
            pl.in_dim = in_dim;
            pl.out_dim = out_dim;
            pl.out_ext = parts.back().dst + parts.back().n;
            pl.kpad = 0;
            for (auto& b : blocks) pl.kpad += b.kp;
            pl.npad = round_up(pl.out_ext, 32);
            Staged s;
            s.dst = &pl;
            s.wt.assign((size_t)pl.npad * pl.kpad, 0.f);
            s.bias.assign(pl.npad, 0.f);
            pack_layer(kernel, bias, in_dim, out_dim, blocks, pl.npad, 0, pl.kpad, s.wt.data(), s.bias.data(), &parts);
            staged.push_back(std::move(s));
            if (l == 0 && m <= DSAT_MLP_LIT) { l0_kernel[m] = kernel; l0_bias[m] = bias; }
        }
    }
    // fused first layer of variables_query and lit_mlp: both consume v1 (query_sat.py:240, :252)
    {
        const int qh = layout->hidden[DSAT_MLP_VARIABLES_QUERY], lh = layout->hidden[DSAT_MLP_LIT];
        w->q_hpad = round_up(qh, 32);
        w->lit_h = lh;
        PackedLayer& pl = w->ql1;
        auto blocks = first_layer_blocks(DSAT_MLP_VARIABLES_QUERY, layout->feature_maps, layout->query_maps);
        pl.in_dim = layout->feature_maps + AUX_USED;
        pl.out_dim = w->q_hpad + lh;
        pl.out_ext = pl.out_dim;
        pl.kpad = F + AUX_W;
        pl.npad = w->q_hpad + round_up(lh, 32);
        Staged s;
        s.dst = &pl;
        s.wt.assign((size_t)pl.npad * pl.kpad, 0.f);
        s.bias.assign(pl.npad, 0.f);
        pack_layer(l0_kernel[0], l0_bias[0], pl.in_dim, qh, blocks, pl.npad, 0, pl.kpad, s.wt.data(), s.bias.data());
        pack_layer(l0_kernel[1], l0_bias[1], pl.in_dim, lh, blocks, pl.npad, w->q_hpad, pl.kpad, s.wt.data(), s.bias.data());
        staged.push_back(std::move(s));
    }
    size_t total = 0;
    // per layer: wt, its fragment-ordered copy, bias, and the three bf16 slices (6 bytes per weight of the
    // k-step-padded matrix, counted in floats)
    auto split_floats = [](const PackedLayer& pl) {
        return (((size_t)pl.ntiles() * pl.ksteps() * 3 * 64 * 8 / 2) + 63) & ~size_t(63);
    };
    for (auto& s : staged)
        total += 2 * ((s.wt.size() + 63) & ~size_t(63)) + ((s.bias.size() + 63) & ~size_t(63)) + split_floats(*s.dst);
    w->slab_bytes = total * sizeof(float);
    DSAT_HIP(hipMalloc((void**)&w->slab, w->slab_bytes));
    size_t off = 0;
    for (auto& s : staged) {
        s.dst->wt = w->slab + off;
        hipError_t e = hipMemcpy(s.dst->wt, s.wt.data(), s.wt.size() * sizeof(float), hipMemcpyHostToDevice);
        off += (s.wt.size() + 63) & ~size_t(63);
        s.dst->bias = w->slab + off;
        if (e == hipSuccess)
            e = hipMemcpy(s.dst->bias, s.bias.data(), s.bias.size() * sizeof(float), hipMemcpyHostToDevice);
        off += (s.bias.size() + 63) & ~size_t(63);
        {   // fragment-ordered copy for the weight-streaming kernels
            const PackedLayer& pl = *s.dst;
            std::vector<float> wp(s.wt.size());
            const int kb_n = pl.kpad / 16, nt = pl.npad / 16;
            for (int t = 0; t < nt; ++t)
                for (int kb = 0; kb < kb_n; ++kb)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 4; ++j)
                            wp[(((size_t)t * kb_n + kb) * 64 + l) * 4 + j] =
                                s.wt[(size_t)(t * 16 + (l & 15)) * pl.kpad + kb * 16 + 4 * (l >> 4) + j];
            s.dst->wp = w->slab + off;
            if (e == hipSuccess)
                e = hipMemcpy(s.dst->wp, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice);
            off += (wp.size() + 63) & ~size_t(63);
        }
        {   // bf16 slices of the same weights (split_bf16.h)
            const PackedLayer& pl = *s.dst;
            const int ks = pl.ksteps(), nt = pl.ntiles();
            std::vector<uint16_t> ws((size_t)nt * ks * 3 * 64 * 8, 0);
            for (int t = 0; t < nt; ++t)
                for (int st = 0; st < ks; ++st)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int k = 32 * st + 8 * (l >> 4) + j;
                            const float v = k < pl.kpad ? s.wt[(size_t)(t * 16 + (l & 15)) * pl.kpad + k] : 0.f;
                            uint16_t h3[3];
                            split3_host(v, h3);
                            const size_t base = (((size_t)t * ks + st) * 3) * 64 * 8;
                            for (int pnum = 0; pnum < 3; ++pnum) ws[base + ((size_t)pnum * 64 + l) * 8 + j] = h3[pnum];
                        }
            s.dst->ws = reinterpret_cast<uint16_t*>(w->slab + off);
            if (e == hipSuccess)
                e = hipMemcpy(s.dst->ws, ws.data(), ws.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
            off += split_floats(pl);
        }
        if (e != hipSuccess) {
            dsat_weights_destroy(w.release());
            return hip_fail(e, "weights upload", __FILE__, __LINE__);
        }
    }
    *out = w.release();
    return DSAT_OK;
}

extern "C" void dsat_weights_destroy(dsat_weights* w) {
    if (!w) return;
    if (w->slab) (void)hipFree(w->slab);
    delete w;
}

// ============================================================================ session
enum KernelClass { KC_GEMM_QL1 = 0, KC_GEMM_Q, KC_GEMM_LIT, KC_GEMM_CLAUSE, KC_GEMM_UPDATE, KC_GEMM_OUT,
                   KC_CLAUSE_GATHER, KC_LITERAL_GATHER, KC_COLSUM, KC_PAIRNORM, KC_SELECT, KC_SATCHECK,
                   KC_ROUND_P1, KC_ROUND_P2, KC_ROUND_P3, KC_SELECT_SAT, KC_MISC, KC_HEAVY, KC_SEGSUM2, KC_COUNT };
static const char* kKernelClassNames[KC_COUNT] = {
    "gemm_query_lit_l1", "gemm_query", "gemm_lit", "gemm_clause", "gemm_update", "gemm_output",
    "clause_gather", "literal_gather", "pairnorm_colsum", "pairnorm_apply", "select", "sat_check",
    "round_p1_var_query_lit", "round_p2_clause", "round_p3_var_update", "select_sat", "misc", "heavy_row_gather", "pairnorm_segsum2"};

struct dsat_session {
    int device = -1;
    int n_cus = 256;               // hipDeviceAttributeMultiprocessorCount of `device` (launch planning: one workgroup per CU)
    dsat_arena* arena = nullptr;   // non-null: the workspace slab is carved from this arena
    const dsat_graph* g = nullptr;
    const dsat_weights* w = nullptr;
    int max_rounds = 0;
    bool tap_on = false;
    dsat_tap tap;                  // dsat_session_set_tap (parity tests)
    // early retirement of solved copies (dsat_session_set_retire)
    int retire = 0;
    int unsupervised = 0;          // dsat_session_set_supervised(s, 0): QuerySAT(supervised=False) selection
    int32_t *lid = nullptr, *move_src = nullptr, *n_act_dev = nullptr;
    float *x0b = nullptr, *x1b = nullptr;   // second x buffers (the survivors move between steps)
    int32_t* n_act_host = nullptr;          // pinned
    int no_cn = 0;                 // DSAT_NO_CN (ablation): P2 finishes the clause PairNorm itself
    int cn_p1_tail = -1;           // role C in the last wave of a multi-wave P1 instead of P3's: -1 by rule, DSAT_CN_P1_TAIL=0/1
    int sel_pack_on = 1;           // DSAT_SEL_PACK=0: one select/SAT workgroup per graph always
    int cn_in_p3 = 1;              // role C rides in P3's grid when its last wave of tiles leaves CUs idle (DSAT_CN_P3=0: only in P1's)
    double cn_frac = 1.0;          // ... on grids of several waves: this share of the clause rows (DSAT_CN_P3_FRAC; 0.5 / 0.75 measured: less gain)
    char* slab = nullptr;
    size_t slab_bytes = 0;
    // state + intermediates (see DESIGN.md "HBM layout")
    float *V, *Vlog, *Cs, *q, *hq, *lits, *grad, *vl, *newv, *ho, *logits, *aux;
    float *cmsg, *closs4, *vla, *ncv;
    float* hid[DSAT_N_MLP][2];   // ping-pong hidden activations per MLP
    int hid_ld[DSAT_N_MLP];
    float *pv, *pc, *pgl, *out_logits;
    int32_t *best, *graph_unsat, *labels, *in_labels, *fixed_step;
    uint8_t *assign, *cum_acc;
    float *in_noisy, *in_normal, *in_denoised, *x0, *x1, *x_fixed, *scratch;
    Ctl* ctl;
    int nch_v = 1, nch_c = 1;
    // fused round kernels (kernels_round.hip)
    int use_fused = 1;
    int p2_rb = 4;
    int p2_split = 0;   // P2 runs round_p2s_kernel (fp32 products from bf16 slices)
    int p3_split = 0;   // P3 runs round_p3s_kernel
    int p3_bm = 16;     // ... on tiles of this many rows (the PairNorm segment tables follow it)
    float *sp = nullptr, *psum_v = nullptr, *psum_c = nullptr;
    int32_t *vmeta = nullptr, *vgseg = nullptr, *cmeta = nullptr, *cgseg = nullptr;
    std::vector<int32_t> h_vtile_seg, h_vgseg, h_ctile_seg, h_cgseg, h_vmeta, h_cmeta;
    // heavy literal rows (kernels_round.hip, heavy_gather_kernel): empty unless the batch has any
    int heavy_t = HEAVY_ROW_T;
    int n_pieces = 0;
    std::vector<int32_t> h_lrp_light, h_lcl_light, h_lhv, h_pieces;   // lhv / pieces: int2 pairs, flattened
    int32_t *lrp_light = nullptr, *lcl_light = nullptr, *lhv = nullptr, *pieces = nullptr;
    float* hpart = nullptr;
    // heavy clauses, the same on the clause side
    int n_cpieces = 0;
    std::vector<int32_t> h_crp_light, h_ccl_light, h_chv, h_cpieces;
    int32_t *crp_light = nullptr, *ccl_light = nullptr, *chv = nullptr, *cpieces = nullptr;
    float* chpart = nullptr;
    // level-2 PairNorm sums of graphs spread over many tiles (segsum2_kernel); empty unless needed
    std::vector<int32_t> h_vgseg2, h_vchunks, h_cgseg2, h_cchunks;    // chunks: int2 pairs, flattened
    int32_t *vgseg2 = nullptr, *vchunks = nullptr, *cgseg2 = nullptr, *cchunks = nullptr;
    float *psum2_v = nullptr, *psum2_c = nullptr;
    int n_vchunks = 0, n_cchunks = 0;
    // P1 role B: 64-row tile records; 0 blocks = the 16-row tiles run the output / query pair themselves
    std::vector<int32_t> h_vmeta4;
    int32_t* vmeta4 = nullptr;
    size_t vmeta2_at = 0;   // offset (int32) of the 32-row records inside vmeta4
    size_t vlist_at = 0, clist_at = 0;   // ... and of the per-tile list blocks (build_list_blocks)
    int vlist_stride = 0, vlist_max = 0, clist_stride = 0, clist_max = 0;
    int p1_a2 = 0;          // P1 role A runs on 32-row tiles with split products (role_a2)
    int p1_a2_ok = 0;       // ... is possible for this session (shapes, LDS, role B present)
    int prefer_tp = 0;      // dsat_session_set_throughput: several batches in flight, CU-cycles count
    int p3_part = 0;        // P2's role D forms update_gate's `variables` partial product for P3 (when CUs are free)
    int p1_a1s = 0;         // role A's 16-row tiles run lit_mlp as split products when role B covers every row (role_a1s)
    int p1_a2_env = -1;     // DSAT_P1_A2: 0 never, 1 always, -1 (unset) by batch size / dsat_session_set_throughput
    // 32-row A tiles can run the output / query pair themselves on large grids (role_a2, pair_here; DSAT_P1_A2_PAIR=1).  Off by
    // default: measured at config 3 (tools/stamps_a2.py) the pair costs a 32-row tile 28 k cycles (out chain 8.7 k, query
    // chain 19 k, PairNorm planes 1.2 k) -- two tiles' worth is what the role-B workgroup they replace costs (58 k), so no
    // CU-cycles are saved, and the list schedule's picks were slower (P1 184 -> 207 us) or equal.
    int a2_pair_here = 0;
    double a2_cost[3] = {102.0, 74.0, 58.0};   // plan_pair_blocks: A32 tile with the pair, without, a role-B workgroup (k cycles, config 3)
    int pair_blocks = 0;
    // select/SAT workgroups {graph, part, parts, 0}
    std::vector<int32_t> h_sel_tab;
    int32_t* sel_tab = nullptr;
    // packed assignments: graph g's bits (bit i <-> its variable i, little-endian) start at byte h_pack_off[g]
    std::vector<int32_t> h_pack_off;
    int32_t* pack_off = nullptr;
    int n_sel = 0, sel_zero = 0;
    // a large graph's select parts split its KL sums and exchange the partials (select_sat_body, SHARE): only while
    // the parts are certainly co-resident (one workgroup per CU in P2); DSAT_SEL_SHARE=0: every part sums everything
    int sel_share = 0;
    unsigned long long* sel_exch = nullptr;
    int sel_ticks = SEL_EXCH_TICKS;   // DSAT_SEL_EXCH_TICKS (-1: the parts never see each other's words -- tests)
    long long* stamps = nullptr;   // [3][4096][8] when DSAT_STAMPS=1
    RoundArgs ra1, ra2, ra3;   // templates: per-launch fields (round, tail, ...) are patched in
    // graph replay
    int use_graph = 1;
    hipStream_t cap_stream = nullptr;
    std::map<std::tuple<int, int, int, int>, hipGraphExec_t> graphs;
    // profiling
    int profile = 0;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, int>> ev_used;  // (class, pool index of start)
    double prof_overhead_us = 0.0;
    double prof_us[KC_COUNT];
    int64_t prof_n[KC_COUNT];
};

static void carve(dsat_session* s, Slab& sl) {
    const dsat_graph* g = s->g;
    const dsat_layout& L = s->w->layout;
    const size_t N = g->N, C = std::max(g->C, 1), G = g->G, R = s->max_rounds;
    s->V = sl.take<float>(N * F); s->Vlog = sl.take<float>(N * F); s->Cs = sl.take<float>(C * F);
    s->q = sl.take<float>(N * F); s->hq = nullptr;
    // lits / sp / closs4 / vla carry one extra all-zero row (index 2N resp. C) that the fused
    // gathers read for list slots past a row's end (the slab is zeroed once, nothing writes there)
    s->lits = sl.take<float>((2 * N + 1) * F); s->grad = sl.take<float>(N * F); s->vl = sl.take<float>(2 * N * F);
    s->newv = sl.take<float>(N * F); s->ho = nullptr;
    s->logits = sl.take<float>(N * LOGIT_MAPS); s->aux = sl.take<float>(R * N * AUX_W);
    s->cmsg = sl.take<float>(C * F); s->closs4 = sl.take<float>((C + 1) * F);
    s->vla = sl.take<float>((C + 1) * F); s->ncv = sl.take<float>(C * F);
    for (int m = 0; m < DSAT_N_MLP; ++m) {
        const size_t rows = (m == DSAT_MLP_CLAUSE) ? C : N;
        s->hid_ld[m] = round_up(L.hidden[m], 32);
        s->hid[m][0] = sl.take<float>(rows * s->hid_ld[m]);
        s->hid[m][1] = sl.take<float>(rows * s->hid_ld[m]);
    }
    s->nch_v = colsum_chunks(g->G);
    s->nch_c = colsum_chunks(g->G);
    s->pv = sl.take<float>(G * s->nch_v * F); s->pc = sl.take<float>(G * s->nch_c * F);
    s->pgl = sl.take<float>(G * LOGIT_MAPS); s->out_logits = sl.take<float>(N);
    s->best = sl.take<int32_t>(G); s->graph_unsat = sl.take<int32_t>(G);
    s->labels = sl.take<int32_t>(N); s->in_labels = sl.take<int32_t>(N); s->fixed_step = sl.take<int32_t>(G);
    s->assign = sl.take<uint8_t>(N); s->cum_acc = sl.take<uint8_t>(G);
    s->in_noisy = sl.take<float>(N * 2); s->in_normal = sl.take<float>(R * N * 4);
    s->in_denoised = sl.take<float>(N);
    s->x0 = sl.take<float>(N); s->x1 = sl.take<float>(N); s->x_fixed = sl.take<float>(N);
    // scratch for the single-operator entry points: packed MLP input, widest case
    size_t max_k = 0;
    for (int m = 0; m < DSAT_N_MLP; ++m) max_k = std::max<size_t>(max_k, s->w->mlp[m][0].kpad);
    s->scratch = sl.take<float>(std::max(N, C) * max_k);
    s->ctl = sl.take<Ctl>(1);
    s->sp = sl.take<float>((2 * N + 1) * F);
    s->psum_v = sl.take<float>(std::max<size_t>(s->h_vtile_seg.empty() ? 1 : s->h_vtile_seg.back(), 1) * F);
    s->psum_c = sl.take<float>(2 * std::max<size_t>(s->h_ctile_seg.empty() ? 1 : s->h_ctile_seg.back(), 1) * F);
    if (s->n_pieces > 0) {
        s->lrp_light = sl.take<int32_t>(s->h_lrp_light.size());
        s->lcl_light = sl.take<int32_t>(std::max<size_t>(s->h_lcl_light.size(), 1));
        s->lhv = sl.take<int32_t>(s->h_lhv.size());
        s->pieces = sl.take<int32_t>(s->h_pieces.size());
        s->hpart = sl.take<float>((size_t)s->n_pieces * 2 * F);
    }
    if (s->n_cpieces > 0) {
        s->crp_light = sl.take<int32_t>(s->h_crp_light.size());
        s->ccl_light = sl.take<int32_t>(std::max<size_t>(s->h_ccl_light.size(), 1));
        s->chv = sl.take<int32_t>(s->h_chv.size());
        s->cpieces = sl.take<int32_t>(s->h_cpieces.size());
        s->chpart = sl.take<float>((size_t)s->n_cpieces * 2 * F);
    }
    if (s->n_vchunks > 0) {
        s->vgseg2 = sl.take<int32_t>(s->h_vgseg2.size());
        s->vchunks = sl.take<int32_t>(s->h_vchunks.size());
        s->psum2_v = sl.take<float>((size_t)s->n_vchunks * F);
    }
    if (s->n_cchunks > 0) {
        s->cgseg2 = sl.take<int32_t>(s->h_cgseg2.size());
        s->cchunks = sl.take<int32_t>(s->h_cchunks.size());
        s->psum2_c = sl.take<float>((size_t)s->n_cchunks * F);
    }
    s->sel_tab = sl.take<int32_t>(s->h_sel_tab.size());
    if (s->sel_share) s->sel_exch = sl.take<unsigned long long>((size_t)s->n_sel * LOGIT_MAPS);
    s->vmeta4 = sl.take<int32_t>(s->h_vmeta4.size());
    s->pack_off = sl.take<int32_t>(s->h_pack_off.size());
    s->lid = sl.take<int32_t>(G); s->move_src = sl.take<int32_t>(G); s->n_act_dev = sl.take<int32_t>(4);
    s->x0b = sl.take<float>(N); s->x1b = sl.take<float>(N);
    s->vmeta = sl.take<int32_t>(s->h_vmeta.size());
    s->vgseg = sl.take<int32_t>(s->h_vgseg.size());
    s->cmeta = sl.take<int32_t>(s->h_cmeta.size());
    s->cgseg = sl.take<int32_t>(s->h_cgseg.size());
}

// PairNorm partial-sum segments: one per (tile, graph with rows in that tile), in row order.
static void build_segments(const std::vector<int32_t>& seg, int rows, int bm, std::vector<int32_t>& tile_seg,
                           std::vector<int32_t>& gseg) {
    const int G = (int)seg.size() - 1;
    const int ntiles = (rows + bm - 1) / bm;
    std::vector<int32_t> per_graph(G, 0);
    tile_seg.assign(ntiles + 1, 0);
    int g = 0, count = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int lo = t * bm, hi = std::min(rows, lo + bm);
        tile_seg[t] = count;
        while (g < G && seg[g + 1] <= lo) ++g;
        for (int gg = g; gg < G && seg[gg] < hi; ++gg)
            if (std::max(seg[gg], lo) < std::min(seg[gg + 1], hi)) { per_graph[gg]++; count++; }
    }
    tile_seg[ntiles] = count;
    gseg.assign(G + 1, 0);
    for (int i = 0; i < G; ++i) gseg[i + 1] = gseg[i] + per_graph[i];
}

// Per-tile records (TileMetaDev in kernels_round.hip): graphs touched, first segment, the first
// graph's row/segment ranges and the CSR entry range of the tile's rows.  `neg_off` >= 0 adds
// the entry range of the rows shifted by neg_off (the -literal rows of a variable tile).
static void build_tile_meta(const std::vector<int32_t>& seg, int rows, int bm, const std::vector<int32_t>& tile_seg,
                            const std::vector<int32_t>& gseg, const int32_t* rowptr, int neg_off,
                            std::vector<int32_t>& meta) {
    const int G = (int)seg.size() - 1;
    const int ntiles = (rows + bm - 1) / bm;
    meta.assign((size_t)std::max(ntiles, 1) * TILE_META_INTS, 0);
    int g = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int lo = t * bm, hi = std::min(rows, lo + bm);
        while (g < G && seg[g + 1] <= lo) ++g;
        int gl = g;
        while (gl + 1 < G && seg[gl + 1] < hi) ++gl;
        int32_t* m = &meta[(size_t)t * TILE_META_INTS];
        m[0] = g; m[1] = gl; m[2] = tile_seg[t];
        m[3] = seg[g]; m[4] = seg[g + 1]; m[5] = gseg[g]; m[6] = gseg[g + 1];
        m[7] = rowptr[lo]; m[8] = rowptr[hi];
        if (neg_off >= 0) { m[9] = rowptr[neg_off + lo]; m[10] = rowptr[neg_off + hi]; }
    }
}

// Per-tile list blocks: the CSR lists of a tile's rows copied out at a FIXED stride, so that a tile's workgroup knows
// the address from its index alone and fetches the lists in the same memory round trip as its tile record (whose
// e_base / e_end it would otherwise have to wait for first).  Block = [ptr0 (bm+1, relative) | ptr1 (bm+1, offset by
// the length of list 0; only with neg_off >= 0) | entries of list 0 | entries of list 1].  Tiles whose lists exceed
// `cap` entries keep an empty block (the kernels then read the global CSR; `*max_n` tells them which).
static int build_list_blocks(const int32_t* rowptr, const int32_t* idx, int rows, int bm, int neg_off, int cap,
                             std::vector<int32_t>& out, size_t* at, int* max_n) {
    const int ntiles = (rows + bm - 1) / bm, nl = neg_off >= 0 ? 2 : 1, hdr = nl * (bm + 1);
    int mx = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int lo = t * bm, hi = std::min(rows, lo + bm);
        int n = rowptr[hi] - rowptr[lo];
        if (nl == 2) n += rowptr[neg_off + hi] - rowptr[neg_off + lo];
        if (n <= cap) mx = std::max(mx, n);
    }
    const int stride = round_up(hdr + mx, 4);
    *at = out.size();
    *max_n = mx;
    out.resize(*at + (size_t)std::max(ntiles, 1) * stride, 0);
    for (int t = 0; t < ntiles; ++t) {
        const int lo = t * bm, hi = std::min(rows, lo + bm);
        const int n0 = rowptr[hi] - rowptr[lo];
        const int n1 = nl == 2 ? rowptr[neg_off + hi] - rowptr[neg_off + lo] : 0;
        if (n0 + n1 > mx) continue;
        int32_t* b = &out[*at + (size_t)t * stride];
        for (int i = 0; i <= bm; ++i) {
            b[i] = rowptr[std::min(lo + i, hi)] - rowptr[lo];
            if (nl == 2) b[bm + 1 + i] = n0 + rowptr[neg_off + std::min(lo + i, hi)] - rowptr[neg_off + lo];
        }
        std::copy(idx + rowptr[lo], idx + rowptr[hi], b + hdr);
        if (nl == 2) std::copy(idx + rowptr[neg_off + lo], idx + rowptr[neg_off + hi], b + hdr + n0);
    }
    return stride;
}

static ChainSpec make_chain(const std::vector<PackedLayer>& layers, int rb) {
    ChainSpec c;
    memset(&c, 0, sizeof(c));
    c.nlayers = (int)layers.size();
    for (int l = 0; l < c.nlayers; ++l) {
        c.L[l].wp = reinterpret_cast<const float4*>(layers[l].wp);
        c.L[l].bias = layers[l].bias;
        c.L[l].kblocks = layers[l].kblocks();
        // column tiles that hold real outputs (npad is rounded to 32 for the per-layer GEMM path;
        // the fragment-ordered weights are tile-major, so trailing all-padding tiles are just not visited)
        c.L[l].ntiles = (layers[l].out_ext + 15) / 16;
        const int cap = std::min(max_tpw_for_rb(rb), ROUND_TPW_CAP);
        c.L[l].tpw = pick_tpw(c.L[l].ntiles, ROUND_NWAVES, cap);
        c.L[l].tpw_half = pick_tpw(c.L[l].ntiles, ROUND_NWAVES / 2, cap);
    }
    // compile-time blockings (single pass per layer): see chain_static in fused_common.h
    auto matches = [&](bool half, std::initializer_list<int> tp) {
        if ((int)tp.size() != c.nlayers) return false;
        const int nw = half ? ROUND_NWAVES / 2 : ROUND_NWAVES;
        int l = 0;
        for (int t : tp) {
            const int have = half ? c.L[l].tpw_half : c.L[l].tpw;
            if (have != t || c.L[l].ntiles > nw * t) return false;
            ++l;
        }
        return true;
    };
    if (getenv("DSAT_NO_STATIC_CHAIN") == nullptr) {
        const bool mix_ok = getenv("DSAT_NO_MIX") == nullptr;
        if (rb == 1 && ROUND_TPW_CAP >= 4 && matches(false, {4, 4, 2})) c.static_id = CHAIN_442;
        if (rb == 1 && matches(false, {2, 2, 1})) c.static_id = CHAIN_221;
        if (rb == 1 && ROUND_NWAVES == 16 && matches(false, {1, 1, 1})) c.static_id = CHAIN_111;
        // RB 4: 13 column tiles get the balanced (tile, row block) split, see chain_static MIX0
        if (rb == 4 && ROUND_NWAVES == 8 && matches(false, {2, 2}))
            c.static_id = (mix_ok && c.L[0].ntiles == (ROUND_NWAVES - 4) * 2 + 4 + 1) ? CHAIN_22M : CHAIN_22;
        if (rb == 4 && ROUND_NWAVES == 16 && matches(false, {1, 1}))
            c.static_id = (mix_ok && c.L[0].ntiles == ROUND_NWAVES - 4 + 1) ? CHAIN_11M : CHAIN_11;
        if (rb == 1 && matches(true, {2, 1})) c.static_half_id = CHAIN_H21;
        if (rb == 1 && matches(true, {3, 2})) c.static_half_id = CHAIN_H32;
        if (rb == 1 && matches(true, {1, 1})) c.static_half_id = CHAIN_H11;
    }
    if (getenv("DSAT_DEBUG_PLAN") != nullptr) {
        fprintf(stderr, "[dsat] chain rb=%d layers=%d static=%d half=%d :", rb, c.nlayers, c.static_id, c.static_half_id);
        for (int l = 0; l < c.nlayers; ++l)
            fprintf(stderr, " (kb=%d nt=%d tpw=%d/%d)", c.L[l].kblocks, c.L[l].ntiles, c.L[l].tpw, c.L[l].tpw_half);
        fprintf(stderr, "\n");
    }
    return c;
}
static int max_hidden_pad(const std::vector<PackedLayer>& layers) {
    int m = 16;
    for (size_t l = 0; l + 1 < layers.size(); ++l) m = std::max(m, layers[l].npad);
    return m;
}

// Fill the three RoundArgs templates (graph, buffers, weights, LDS plan).  Returns false when
// a tile does not fit the 160 KiB of LDS (then the per-layer path is used).
static bool plan_round(dsat_session* s) {
    const dsat_graph* g = s->g;
    const dsat_weights* w = s->w;
    RoundArgs a;
    memset(&a, 0, sizeof(a));
    a.c_rowptr = g->c_rowptr; a.c_lit = g->c_lit; a.l_rowptr = g->l_rowptr; a.l_clause = g->l_clause;
    a.var_gid = g->var_gid; a.cl_gid = g->cl_gid; a.var_seg = g->var_seg; a.cl_seg = g->cl_seg;
    a.deg_w = g->deg_w; a.vdeg_w = g->vdeg_w; a.rev_w = g->rev_w;
    a.vmeta = s->vmeta; a.vgseg = s->vgseg; a.cmeta = s->cmeta; a.cgseg = s->cgseg;
    a.pgl = s->pgl; a.out_logits = s->out_logits; a.best = s->best; a.graph_unsat = s->graph_unsat;
    a.assign = s->assign;
    a.N = g->N; a.C = g->C; a.G = g->G;
    a.NL = g->N; a.CL = g->C;
    a.cg_rowptr = s->n_cpieces > 0 ? s->crp_light : g->c_rowptr;
    a.cg_lit = s->n_cpieces > 0 ? s->ccl_light : g->c_lit;
    a.c_hv = s->n_cpieces > 0 ? reinterpret_cast<const int2*>(s->chv) : nullptr;
    a.chpart = s->chpart;
    a.vgseg2 = s->n_vchunks > 0 ? s->vgseg2 : nullptr; a.psum2_v = s->psum2_v;
    a.cgseg2 = s->n_cchunks > 0 ? s->cgseg2 : nullptr; a.psum2_c = s->psum2_c;
    a.sel_tab = s->sel_tab; a.n_sel = s->n_sel; a.sel_zero = s->sel_zero;
    a.sel_exch = s->sel_exch; a.sel_ticks = s->sel_ticks;
    a.inv_f = 1.f / (float)w->layout.feature_maps; a.fm = w->layout.feature_maps;
    a.gj2_three = g->max_clause_len <= 3 && getenv("DSAT_GJ2_FOUR") == nullptr;
    a.vlist = s->vlist_stride > 0 ? s->vmeta4 + s->vlist_at : nullptr; a.vlist_stride = s->vlist_stride; a.vlist_max = s->vlist_max;
    a.clist = s->clist_stride > 0 ? s->vmeta4 + s->clist_at : nullptr; a.clist_stride = s->clist_stride; a.clist_max = s->clist_max;
    a.sel_lead = s->sel_zero ? round_up(s->n_sel, 8) : 0;   // long select parts (a large graph) go first
    a.V = s->V; a.Cs = s->Cs; a.q = s->q; a.sp = s->sp; a.lits = s->lits; a.closs4 = s->closs4;
    a.vla = s->vla; a.ncv = s->ncv; a.newv = s->newv; a.logits = s->logits;
    a.psum_v = s->psum_v; a.psum_c_rd = s->psum_c; a.psum_c_wr = s->psum_c; a.noisy = s->in_noisy; a.ctl = s->ctl;
    a.aux = s->aux;
    const int LDS = ROUND_LDS_FLOATS;
    bool ok = true;
    {   // P1: [X 16x152 | Vlog 16x136 | A | B | mean]
        RoundArgs& r = s->ra1;
        r = a;
        r.mq = make_chain(w->mlp[DSAT_MLP_VARIABLES_QUERY], 1);
        r.mlit = make_chain(w->mlp[DSAT_MLP_LIT], 1);
        r.mout = make_chain(w->mlp[DSAT_MLP_VARIABLES_OUTPUT], 1);
        const int hid = std::max({max_hidden_pad(w->mlp[0]), max_hidden_pad(w->mlp[1]), max_hidden_pad(w->mlp[4])});
        r.ld_x = F + AUX_W + 8;
        r.ld_a = r.ld_b = hid + 8;
        r.off_x = 0;
        // the pre-mix state tile (input of variables_output) lives only while the output / query pair runs,
        // when the hidden buffers hold at most 160 columns: it borrows columns 256.. of buffer A's rows
        const bool vlog_in_a = max_hidden_pad(w->mlp[DSAT_MLP_VARIABLES_OUTPUT]) + 8 <= 256 && r.ld_a >= 256 + F + 8;
        r.off_vlog = ROUND_BM_VAR * r.ld_x;
        r.off_a = vlog_in_a ? r.off_vlog : r.off_vlog + ROUND_BM_VAR * (F + 8);
        r.ld_v = vlog_in_a ? r.ld_a : F + 8;
        if (vlog_in_a) r.off_vlog = r.off_a + 256;
        r.off_b = r.off_a + ROUND_BM_VAR * r.ld_a;
        r.off_mean = r.off_b + ROUND_BM_VAR * r.ld_b;
        const int mean_floats = (ROUND_NWAVES * 64 / F + 1) * F;   // per-graph mean + one partial per thread group
        ok = ok && (r.off_mean + mean_floats <= LDS);
        // role B: 64-row tiles for the output / query pair (kernels_round.hip, round_p1_kernel)
        r.mq4 = make_chain(w->mlp[DSAT_MLP_VARIABLES_QUERY], 4);
        r.mout4 = make_chain(w->mlp[DSAT_MLP_VARIABLES_OUTPUT], 4);
        r.vmeta4 = s->vmeta4;
        r.b_ld_a = max_hidden_pad(w->mlp[DSAT_MLP_VARIABLES_OUTPUT]) + 8;
        r.b_ld_b = max_hidden_pad(w->mlp[DSAT_MLP_VARIABLES_QUERY]) + 8;
        // [X | B | Vlog | A | mean]: X and B adjacent, Vlog and A adjacent -- the split-product query chain reuses each
        // pair as one region (kernels_round.hip, role B)
        r.b_off_x = 0;
        r.b_off_b = 64 * r.ld_x;
        r.b_off_vlog = r.b_off_b + 64 * r.b_ld_b;
        r.b_off_a = r.b_off_vlog + 64 * (F + 8);
        r.b_off_mean = r.b_off_a + 64 * r.b_ld_a;
        const bool fits = r.b_off_mean + mean_floats <= LDS;
        const bool shapes = ROUND_NWAVES == 16 && r.mq4.static_id == CHAIN_11 && r.mout4.static_id == CHAIN_11;
        s->pair_blocks = (fits && shapes && getenv("DSAT_NO_PAIRB") == nullptr) ? (g->N + 63) / 64 : 0;
        {   // variables_query of role B with split products: v1 planes [3][64][K + 16] in Vlog | A, hidden planes in X | B
            const auto& ql = w->mlp[DSAT_MLP_VARIABLES_QUERY];
            const char* swb = getenv("DSAT_P1_BSPLIT");
            if (s->pair_blocks != 0 && ql.size() == 2 && ql[0].kpad == F + AUX_W && r.mq4.L[0].ntiles <= ROUND_NWAVES &&
                r.mq4.L[1].ntiles <= ROUND_NWAVES && !(swb != nullptr && swb[0] == '0')) {
                const int ldp = ql[0].ksteps() * 32 + 16, ldq = ql[1].ksteps() * 32 + 16;
                const int xp_bytes = 3 * 64 * ldp * 2, hp_bytes = 3 * 64 * ldq * 2;
                if (xp_bytes <= 64 * ((F + 8) + r.b_ld_a) * 4 && hp_bytes <= 64 * (r.ld_x + r.b_ld_b) * 4 &&
                    r.mq4.L[0].ntiles * 16 <= ql[1].ksteps() * 32) {
                    r.s_ld_seg = ldp;
                    r.s_ld_hid = ldq;
                    for (int l = 0; l < 2; ++l) {
                        r.sp_ws[l] = reinterpret_cast<const uint4*>(ql[l].ws);
                        r.sp_ks[l] = ql[l].ksteps();
                        r.sp_nt[l] = r.mq4.L[l].ntiles;
                    }
                    r.sp_layers = 2;
                }
            }
            if (getenv("DSAT_DEBUG_PLAN") != nullptr)
                fprintf(stderr, "[dsat] P1 role B query chain as split products: %d\n", r.sp_ws[0] != nullptr);
            // ... and variables_output (round 2 left it on the fp32 MFMA: 18.5 k of the role-B workgroup's 62.6 k cycles).
            // Needs the query chain as split products too (the LDS plan below is shared): [X fp32 | output planes |
            // output hidden planes | mean]; the query chain's v1 planes then end where the hidden planes ended and its
            // hidden planes start at 0, over X (sliced by then).  DSAT_P1_BOSPLIT=0 keeps the fp32-MFMA output chain.
            const auto& ol = w->mlp[DSAT_MLP_VARIABLES_OUTPUT];
            const char* swo = getenv("DSAT_P1_BOSPLIT");
            if (r.sp_ws[0] != nullptr && ol.size() == 2 && ol[0].kpad == F && r.mout4.L[0].ntiles <= ROUND_NWAVES &&
                r.mout4.L[1].ntiles <= ROUND_NWAVES && r.mout4.L[0].ntiles * 16 <= ol[1].ksteps() * 32 &&
                !(swo != nullptr && swo[0] == '0')) {
                const int ldo = std::max(ol[0].ksteps(), ol[1].ksteps()) * 32 + 16;
                const int xo_f = 3 * 64 * ldo * 2 / 4, x_f = 64 * r.ld_x;                 // floats
                const int xq_f = 3 * 64 * r.s_ld_seg * 2 / 4, hq_f = 3 * 64 * r.s_ld_hid * 2 / 4;
                const int mean_at = LDS - mean_floats;
                const int end_o = x_f + 2 * xo_f;                                           // end of the output chain's planes
                const int xq_at = std::max(end_o, x_f + xq_f) - xq_f;                      // v1 planes: behind X, ending with them
                if (end_o <= mean_at && xq_at + xq_f <= mean_at && hq_f <= xq_at && xq_at >= x_f) {
                    for (int l = 0; l < 2; ++l) {
                        r.bo_ws[l] = reinterpret_cast<const uint4*>(ol[l].ws);
                        r.bo_ks[l] = ol[l].ksteps();
                        r.bo_nt[l] = r.mout4.L[l].ntiles;
                    }
                    r.bo_ld = ldo;
                    r.bo_off_x = x_f; r.bo_off_h = x_f + xo_f; r.bo_off_xq = xq_at; r.bo_off_mean = mean_at;
                }
            }
            if (getenv("DSAT_DEBUG_PLAN") != nullptr)
                fprintf(stderr, "[dsat] P1 role B output chain as split products: %d\n", r.bo_ws[0] != nullptr);
        }
        // role A on 32-row tiles with lit_mlp as split products (kernels_round.hip, role_a2): the released
        // 137 -> 512 -> 512 -> 256 shape, role B present (it then covers every 64-row group)
        const auto& lit = w->mlp[DSAT_MLP_LIT];
        const char* sw2 = getenv("DSAT_P1_A2");
        s->p1_a2_ok = ROUND_NWAVES == 16 && s->pair_blocks != 0 && lit.size() == 3 && lit[0].kpad == F + AUX_W &&
                      r.mlit.L[0].ntiles == 32 && r.mlit.L[1].ntiles == 32 && r.mlit.L[2].ntiles == 16 &&
                      lit[1].kpad == 512 && lit[2].kpad == 512 && LDS * 4 >= 154112;
        s->p1_a2_env = sw2 == nullptr ? -1 : (sw2[0] == '0' ? 0 : 1);
        // Twice the rows per CU-cycle, but no shorter for a batch whose 16-row tiles all have a CU of their own
        // (config 2: 194 + 49 workgroups; and the bf16 pipe's power costs the rest of the round 3 % of its clock):
        // by default only for batches of more than one wave of workgroups, or when the caller asks for throughput
        const bool large = (g->N + ROUND_BM_VAR - 1) / ROUND_BM_VAR + s->pair_blocks > s->n_cus;
        s->p1_a2 = s->p1_a2_ok && (s->p1_a2_env == 1 || (s->p1_a2_env == -1 && large));
        // 16-row tiles as split products: the same shapes; 57 k cycles per tile against 69 k on the fp32 MFMA
        // (tools/lit16w_probe.hip).  DSAT_P1_ASPLIT=0 keeps the fp32-MFMA chain.
        const char* sw3 = getenv("DSAT_P1_ASPLIT");
        s->p1_a1s = s->p1_a2_ok && !(sw3 != nullptr && sw3[0] == '0');
        r.vmeta2 = s->vmeta4 + s->vmeta2_at;
        for (int l = 0; l < 3 && l < (int)lit.size(); ++l) r.a2_ws[l] = reinterpret_cast<const uint4*>(lit[l].ws);
        if (getenv("DSAT_DEBUG_PLAN") != nullptr) fprintf(stderr, "[dsat] P1 role A on 32-row split tiles: %d\n", s->p1_a2);
    }
    {   // P2: [X BMx392 | A | (B) | mean]
        RoundArgs& r = s->ra2;
        const auto& layers = w->mlp[DSAT_MLP_CLAUSE];
        const int hid = max_hidden_pad(layers);
        int rb = 4;
        for (; rb >= 1; rb >>= 1) {
            const int bm = 16 * rb;
            const int need = bm * (3 * F + 8) + bm * (hid + 8) * (layers.size() > 2 ? 2 : 1) + ROUND_NWAVES * 64;
            if (need <= LDS) break;
        }
        ok = ok && rb >= 1;
        if (rb < 1) rb = 1;
        s->p2_rb = rb;
        r = a;
        r.mcl = make_chain(layers, rb);
        const int bm = 16 * rb;
        r.ld_x = 3 * F + 8;
        r.ld_a = r.ld_b = hid + 8;
        r.off_x = 0;
        r.off_a = bm * r.ld_x;
        r.off_b = layers.size() > 2 ? r.off_a + bm * r.ld_a : r.off_a;
        r.off_mean = r.off_b + bm * r.ld_b;
        ok = ok && (r.off_mean + ROUND_NWAVES * 64 <= LDS);
        // the split-product form of the tile (round_p2s_kernel): two layers, 64 rows, 16 waves, one column
        // tile per wave; LDS = [buf1: one 128-column segment as three bf16 planes | buf2: the hidden tile's
        // planes (first a segment, and fp32 scratch of the in-kernel PairNorm) + the staged CSR lists | sums]
        s->p2_split = 0;
        const char* sw = getenv("DSAT_P2_SPLIT");
        if (rb == 4 && ROUND_NWAVES == 16 && layers.size() == 2 && layers[0].kpad == 3 * F &&
            r.mcl.L[0].ntiles <= ROUND_NWAVES && r.mcl.L[1].ntiles <= ROUND_NWAVES &&
            r.mcl.L[0].ntiles * 16 <= layers[1].ksteps() * 32 && !(sw != nullptr && sw[0] == '0')) {
            const int ks1 = layers[1].ksteps();
            r.s_ld_seg = F + 16;          // row stride == 32 bytes mod 64: conflict-free fragment reads (split_bf16.h)
            r.s_ld_hid = ks1 * 32 + 16;
            const int seg_bytes = 3 * bm * r.s_ld_seg * 2, hid_bytes = 3 * bm * r.s_ld_hid * 2;
            const int scratch_bytes = (bm + ROUND_NWAVES * 64 / F) * (F + 8) * 4;   // PairNorm rows + partial sums
            const int lists_at = std::max(seg_bytes, scratch_bytes);
            const int buf2_bytes = std::max(hid_bytes, lists_at + 16 * 1024);
            r.s_buf1 = 0;
            r.s_buf2 = seg_bytes;
            r.s_idx = r.s_buf2 + lists_at;
            r.s_idx_cap = (buf2_bytes - lists_at) / 4 - (bm + 8);
            r.s_mean = r.s_buf2 + buf2_bytes;
            if (r.s_mean + ROUND_NWAVES * 64 * 4 <= LDS * 4 && (bm * (F + 8) + ROUND_NWAVES * 64) * 4 <= seg_bytes) {
                r.sp_ws[0] = reinterpret_cast<const uint4*>(layers[0].ws);
                r.sp_ws[1] = reinterpret_cast<const uint4*>(layers[1].ws);
                r.sp_ks[0] = layers[0].ksteps(); r.sp_ks[1] = ks1;
                r.sp_nt[0] = r.mcl.L[0].ntiles; r.sp_nt[1] = r.mcl.L[1].ntiles;
                s->p2_split = 1;
            }
        }
        // the tile's list block must fit where the kernel stages it (and P2's tile size must be the blocks')
        if (r.clist != nullptr && (rb != s->p2_rb || r.clist_stride > (s->p2_split ? r.s_idx_cap + bm + 8 : bm * r.ld_a)))
            r.clist = nullptr;
        if (getenv("DSAT_DEBUG_PLAN") != nullptr)
            fprintf(stderr, "[dsat] P2 rb=%d split=%d (buf2 at %d, lists at %d cap %d, sums at %d of %d bytes)\n", rb,
                    s->p2_split, r.s_buf2, r.s_idx, r.s_idx_cap, r.s_mean, LDS * 4);
    }
    {   // P3: [X 16x536 | A | B]
        RoundArgs& r = s->ra3;
        const auto& layers = w->mlp[DSAT_MLP_UPDATE_GATE];
        const int hid = max_hidden_pad(layers);
        r = a;
        if (s->n_pieces > 0) {
            r.l_rowptr = s->lrp_light;
            r.l_clause = s->lcl_light;
            r.l_hv = reinterpret_cast<const int2*>(s->lhv);
            r.hpart = s->hpart;
        }
        r.mupd = make_chain(layers, 1);
        r.ld_x = 4 * F + AUX_W + 8;
        r.ld_a = r.ld_b = hid + 8;
        r.off_x = 0;
        r.off_a = ROUND_BM_VAR * r.ld_x;
        r.off_b = r.off_a + ROUND_BM_VAR * r.ld_a;
        r.off_mean = r.off_b + ROUND_BM_VAR * r.ld_b;
        ok = ok && (r.off_mean + ROUND_NWAVES * 64 <= LDS);
        // the split-product form (round_p3s_kernel): [input planes | hidden A | hidden B | sums]; the staged CSR
        // lists borrow the hidden tiles
        s->p3_split = 0;
        const char* sw = getenv("DSAT_P3_SPLIT");
        bool shapes = ROUND_NWAVES == 16 && (layers.size() == 2 || layers.size() == 3) &&
                      layers[0].kpad == 4 * F + AUX_W && !(sw != nullptr && sw[0] == '0');
        for (int l = 0; shapes && l < r.mupd.nlayers; ++l) shapes = r.mupd.L[l].ntiles <= ROUND_NWAVES;
        if (shapes) {
            const int bm = s->p3_bm;
            int ldh = 48;
            for (size_t l = 1; l < layers.size(); ++l) ldh = std::max(ldh, layers[l].ksteps() * 32 + 16);
            r.s_ld_seg = layers[0].ksteps() * 32 + 16;
            r.s_ld_hid = ldh;
            const int x_bytes = 3 * bm * r.s_ld_seg * 2, h_bytes = 3 * bm * ldh * 2;
            // 16 rows: [X | hA | hB | sums], the stage of the column sums over X, the CSR lists over hA and hB;
            // 32 rows: [X = hB | hA | sums] (X is dead once layer 0 has run), the stage and the lists over hA
            const bool wide = bm == 32;
            r.p3_bm = bm;
            r.s_buf1 = 0;
            r.s_buf2 = x_bytes;
            r.s_bufB = wide ? 0 : x_bytes + h_bytes;
            // the stage must be dead during the LAST layer: that layer reads hB (= X when wide) in a 3-layer chain and
            // hA in a 2-layer one
            r.s_stage = wide && r.mupd.nlayers == 3 ? x_bytes : 0;
            r.s_idx = r.s_buf2;
            r.s_idx_cap = ((wide ? 1 : 2) * h_bytes / 4 - 2 * (bm + 8)) / 2;
            r.s_mean = r.s_buf2 + (wide ? 1 : 2) * h_bytes;
            if (wide) r.vmeta = s->vmeta4 + s->vmeta2_at;
            if (r.s_mean + ROUND_NWAVES * 64 * 4 <= LDS * 4 && bm * (F + 8) * 4 <= (r.s_stage != 0 ? h_bytes : x_bytes) &&
                r.s_idx_cap >= 256) {
                r.sp_layers = r.mupd.nlayers;
                for (int l = 0; l < r.mupd.nlayers; ++l) {
                    r.sp_ws[l] = reinterpret_cast<const uint4*>(layers[l].ws);
                    r.sp_ks[l] = layers[l].ksteps();
                    r.sp_nt[l] = r.mupd.L[l].ntiles;
                }
                s->p3_split = 1;
                // role D of P2 (kernels_round.hip): needs the split-product P2 tile's kernel, whole k-steps for the
                // `variables` columns, and the per-layer path's hidden buffer of update_gate as the partial sums' home
                const char* ep = getenv("DSAT_P3_PART");
                s->p3_part = s->p2_split && !(ep != nullptr && ep[0] == '0') && F % 32 == 0 &&
                             s->hid_ld[DSAT_MLP_UPDATE_GATE] >= 16 * r.sp_nt[0] && r.sp_nt[0] <= ROUND_NWAVES &&
                             3 * 64 * (F + 16) * 2 <= LDS * 4;
                // 2: also in the last wave of a multi-wave P2 (role_d_blocks only lets it when three groups per idle CU
                // cover the batch -- config 5: 157 groups on 70 idle CUs; config 3 with its 16 idle CUs: no); 1: single wave only
                if (s->p3_part) s->p3_part = (ep != nullptr && ep[0] == '1') ? 1 : 2;
            }
        }
        if (s->p3_bm == 32 && !s->p3_split) ok = false;   // the tables were built for 32-row producer tiles
        if (r.vlist != nullptr && r.vlist_stride > (s->p3_split ? 2 * r.s_idx_cap + 2 * (s->p3_bm + 8) : ROUND_BM_VAR * r.ld_a))
            r.vlist = nullptr;
        if (getenv("DSAT_DEBUG_PLAN") != nullptr)
            fprintf(stderr, "[dsat] P3 split=%d rows %d (ldx %d ldh %d, hidden at %d, lists cap %d, sums at %d of %d bytes)\n",
                    s->p3_split, s->p3_bm, r.s_ld_seg, r.s_ld_hid, r.s_buf2, r.s_idx_cap, r.s_mean, LDS * 4);
    }
    return ok;
}

static int session_create_impl(dsat_arena* arena, hipStream_t st, const dsat_graph* g, const dsat_weights* w,
                               int32_t max_rounds, dsat_session** out) {
    DSAT_REQUIRE(out != nullptr, DSAT_E_INVALID, "dsat_session_create: null out");
    *out = nullptr;
    DSAT_REQUIRE(g && w, DSAT_E_INVALID, "dsat_session_create: null graph or weights");
    DSAT_REQUIRE(max_rounds >= 1 && max_rounds <= 4096, DSAT_E_INVALID, "dsat_session_create: max_rounds out of range");
    DSAT_REQUIRE(g->device == w->device, DSAT_E_STATE,
                 "dsat_session_create: graph lives on HIP device " + std::to_string(g->device) +
                     " but the weights on device " + std::to_string(w->device));
    { int rc = require_device(g->device, "dsat_session_create"); if (rc != DSAT_OK) return rc; }
    std::unique_ptr<dsat_session> s(new dsat_session());
    s->device = g->device;
    {   // launch planning counts CUs (one 1024-thread workgroup per CU): the device's own number, not a literal 256
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, g->device) == hipSuccess && cus > 0)
            s->n_cus = cus;
        if (const char* e = getenv("DSAT_N_CUS")) { const int v = atoi(e); if (v >= 8) s->n_cus = v; }   // planning experiments
    }
    s->g = g; s->w = w; s->max_rounds = max_rounds;
    s->no_cn = getenv("DSAT_NO_CN") != nullptr;
    { const char* e = getenv("DSAT_CN_P3"); s->cn_in_p3 = e == nullptr ? 1 : atoi(e); }
    { const char* e = getenv("DSAT_SEL_PACK"); s->sel_pack_on = !(e != nullptr && e[0] == '0'); }
    { const char* e = getenv("DSAT_CN_P1_TAIL"); if (e != nullptr) s->cn_p1_tail = e[0] == '1' ? 1 : 0; }
    { const char* e = getenv("DSAT_P1_A2_PAIR"); s->a2_pair_here = e != nullptr && e[0] == '1'; }
    { const char* e = getenv("DSAT_P1_A2_COST");     // "with,without,b" in k cycles (planning experiments)
      if (e != nullptr) sscanf(e, "%lf,%lf,%lf", &s->a2_cost[0], &s->a2_cost[1], &s->a2_cost[2]); }
    { const char* e = getenv("DSAT_CN_P3_FRAC"); if (e != nullptr) s->cn_frac = std::min(1.0, std::max(0.0, atof(e))); }
    memset(&s->tap, 0, sizeof(s->tap));
    {   // tile sizes are fixed before the workspace is carved: P1/P3 16 rows, P2 16*p2_rb rows
        const auto& cl = w->mlp[DSAT_MLP_CLAUSE];
        const int hid = max_hidden_pad(cl);
        int rb = 4;
        for (; rb > 1; rb >>= 1)
            if (16 * rb * (3 * F + 8) + 16 * rb * (hid + 8) * (cl.size() > 2 ? 2 : 1) + ROUND_NWAVES * 64 <= ROUND_LDS_FLOATS) break;
        s->p2_rb = rb;
        {   // P3 as split products on 32-row tiles (half the weight stream per row, kernels_round.hip): decided here
            // because the PairNorm segment tables follow the producer's tile size.  Batches of more than one wave of
            // 16-row tiles only (a smaller batch gives every tile a CU of its own: nothing to save); DSAT_P3_BM forces.
            const auto& up = w->mlp[DSAT_MLP_UPDATE_GATE];
            const char* e32 = getenv("DSAT_P3_BM");
            const char* esp = getenv("DSAT_P3_SPLIT");
            bool shapes = ROUND_NWAVES == 16 && (up.size() == 2 || up.size() == 3) && up[0].kpad == 4 * F + AUX_W &&
                          !(esp != nullptr && esp[0] == '0');
            int ldh = 48;
            for (size_t l = 0; shapes && l < up.size(); ++l) {
                shapes = (up[l].out_dim + 15) / 16 <= ROUND_NWAVES;
                if (l > 0) ldh = std::max(ldh, up[l].ksteps() * 32 + 16);
            }
            const int ldx = up[0].ksteps() * 32 + 16;
            const bool fits = 3 * 32 * ldx * 2 + 3 * 32 * ldh * 2 + ROUND_NWAVES * 64 * 4 <= ROUND_LDS_FLOATS * 4 && ldh <= ldx &&
                              32 * (F + 8) * 4 <= 3 * 32 * ldh * 2;
            const bool large = (g->N + ROUND_BM_VAR - 1) / ROUND_BM_VAR > s->n_cus;
            const int want = e32 != nullptr ? atoi(e32) : (large ? 32 : 16);
            s->p3_bm = (shapes && fits && want == 32) ? 32 : ROUND_BM_VAR;
        }
        build_segments(g->h_var_seg, g->N, s->p3_bm, s->h_vtile_seg, s->h_vgseg);
        build_segments(g->h_cl_seg, g->C, 16 * rb, s->h_ctile_seg, s->h_cgseg);
        // level-2 PairNorm sums for graphs spread over many tiles
        auto level2 = [](const std::vector<int32_t>& gseg, std::vector<int32_t>& gseg2, std::vector<int32_t>& chunks) {
            const int G = (int)gseg.size() - 1;
            gseg2.assign(G + 1, 0);
            for (int g = 0; g < G; ++g) {
                const int n = gseg[g + 1] - gseg[g];
                int made = 0;
                if (n > SEG2_MIN)
                    for (int c = gseg[g]; c < gseg[g + 1]; c += SEG2_CHUNK, ++made) {
                        chunks.push_back(c);
                        chunks.push_back(std::min(c + SEG2_CHUNK, gseg[g + 1]));
                    }
                gseg2[g + 1] = gseg2[g] + made;
            }
            return (int)(chunks.size() / 2);
        };
        s->n_vchunks = level2(s->h_vgseg, s->h_vgseg2, s->h_vchunks);
        s->n_cchunks = level2(s->h_cgseg, s->h_cgseg2, s->h_cchunks);
        // select/SAT workgroups: one per graph, several for a graph with tens of thousands of clauses
        for (int gi = 0; gi < g->G; ++gi) {
            const int cg = g->h_cl_seg[gi + 1] - g->h_cl_seg[gi];
            const int parts = std::max(1, std::min(64, (cg + SEL_PART_CLAUSES - 1) / SEL_PART_CLAUSES));
            if (parts > 1) s->sel_zero = 1;
            for (int p = 0; p < parts; ++p) {
                const int32_t e[4] = {gi, p, parts, 0};
                s->h_sel_tab.insert(s->h_sel_tab.end(), e, e + 4);
            }
        }
        s->n_sel = (int)(s->h_sel_tab.size() / 4);
        s->sel_share = s->sel_zero && round_up(s->n_sel, 8) <= s->n_cus;
        if (const char* ev = getenv("DSAT_SEL_SHARE")) s->sel_share = s->sel_share && atoi(ev) != 0;
        if (const char* ev = getenv("DSAT_SEL_EXCH_TICKS")) s->sel_ticks = atoi(ev);
        s->h_pack_off.assign(g->G + 1, 0);
        for (int gi = 0; gi < g->G; ++gi)
            s->h_pack_off[gi + 1] = s->h_pack_off[gi] + (g->h_var_seg[gi + 1] - g->h_var_seg[gi] + 7) / 8;
        // heavy literal rows: a "light" copy of the literal->clause CSR with those rows emptied, the
        // pieces heavy_gather_kernel sums, and per literal row where its pieces are
        // Two thresholds: a graph with a literal row above heavy_t gets the heavy path at all (a launch per round);
        // once it has it, every row above heavy_row goes through it -- a tile's walk over a row's clauses is a chain of
        // dependent loads, the streaming kernel is not: config 5, rows above 32 / 16 / 8: 108.6 / 110.0 / 110.0 steps/s
        // (P3 64.9 / 58.9 / 53.0 us, the heavy launch growing by as much below 16).  DSAT_HEAVY_T sets both.
        int heavy_row = std::min(s->heavy_t, HEAVY_ROW_MIN);
        if (const char* ev = getenv("DSAT_HEAVY_T")) s->heavy_t = heavy_row = std::max(1, atoi(ev));
        if (const char* ev = getenv("DSAT_HEAVY_ROW")) heavy_row = std::max(1, std::min(s->heavy_t, atoi(ev)));
        const int32_t* lrp = g->h_l_rowptr.data();
        if (g->max_lit_deg > s->heavy_t) {
            const size_t rows = 2 * (size_t)g->N;
            s->h_lrp_light.assign(rows + 1, 0);
            s->h_lhv.assign(2 * rows, 0);
            for (size_t i = 0; i < rows; ++i) {
                const int deg = lrp[i + 1] - lrp[i];
                if (deg > heavy_row) {
                    const int np = (deg + HEAVY_PIECE - 1) / HEAVY_PIECE;
                    s->h_lhv[2 * i] = (int32_t)(s->h_pieces.size() / 2);
                    s->h_lhv[2 * i + 1] = np;
                    const int per = (deg + np - 1) / np;
                    for (int p = 0; p < np; ++p) {
                        s->h_pieces.push_back(lrp[i] + p * per);
                        s->h_pieces.push_back(std::min(lrp[i] + (p + 1) * per, lrp[i + 1]));
                    }
                    s->h_lrp_light[i + 1] = s->h_lrp_light[i];
                } else {
                    s->h_lhv[2 * i] = -1;
                    s->h_lcl_light.insert(s->h_lcl_light.end(), g->h_l_clause.begin() + lrp[i],
                                          g->h_l_clause.begin() + lrp[i + 1]);
                    s->h_lrp_light[i + 1] = s->h_lrp_light[i] + deg;
                }
            }
            s->n_pieces = (int)(s->h_pieces.size() / 2);
            lrp = s->h_lrp_light.data();
        }
        std::vector<int32_t> zero_rp(2 * (size_t)g->N + 1, 0);
        if (s->p3_bm == ROUND_BM_VAR) {
            build_tile_meta(g->h_var_seg, g->N, ROUND_BM_VAR, s->h_vtile_seg, s->h_vgseg, lrp, g->N, s->h_vmeta);
        } else {   // P3 produces on 32-row tiles: the 16-row records only serve P1's fp32 role A as a consumer
            std::vector<int32_t> none16((size_t)(g->N + ROUND_BM_VAR - 1) / ROUND_BM_VAR + 1, 0);
            build_tile_meta(g->h_var_seg, g->N, ROUND_BM_VAR, none16, s->h_vgseg, zero_rp.data(), -1, s->h_vmeta);
        }
        {   // 64-row records for P1's role B (graphs, first graph's rows / segments; no CSR ranges, no segment base)
            std::vector<int32_t> none((size_t)(g->N + 63) / 64 + 1, 0);
            build_tile_meta(g->h_var_seg, g->N, 64, none, s->h_vgseg, zero_rp.data(), -1, s->h_vmeta4);
            // ... and 32-row records behind them in the same table: role A on 32-row tiles (consumer), and P3 on
            // 32-row tiles (producer: segment base and the CSR ranges of its rows)
            std::vector<int32_t> none2((size_t)(g->N + 31) / 32 + 1, 0), meta2;
            if (s->p3_bm == 32) build_tile_meta(g->h_var_seg, g->N, 32, s->h_vtile_seg, s->h_vgseg, lrp, g->N, meta2);
            else build_tile_meta(g->h_var_seg, g->N, 32, none2, s->h_vgseg, zero_rp.data(), -1, meta2);
            s->vmeta2_at = s->h_vmeta4.size();
            s->h_vmeta4.insert(s->h_vmeta4.end(), meta2.begin(), meta2.end());
        }
        if (getenv("DSAT_NO_LISTBLOCKS") == nullptr)   // the literal -> clause lists of P3's tiles at a fixed stride
            s->vlist_stride = build_list_blocks(lrp, s->n_pieces > 0 ? s->h_lcl_light.data() : g->h_l_clause.data(), g->N,
                                                s->p3_bm, g->N, LIST_BLOCK_CAP, s->h_vmeta4, &s->vlist_at, &s->vlist_max);
        const int32_t* crp = g->h_c_rowptr.data();
        if (g->max_clause_len > s->heavy_t) {   // heavy clauses: the mirror image of the heavy literal rows
            s->h_crp_light.assign((size_t)g->C + 1, 0);
            s->h_chv.assign(2 * (size_t)g->C, 0);
            for (int c = 0; c < g->C; ++c) {
                const int len = crp[c + 1] - crp[c];
                if (len > s->heavy_t) {
                    const int np = (len + HEAVY_PIECE - 1) / HEAVY_PIECE;
                    s->h_chv[2 * (size_t)c] = (int32_t)(s->h_cpieces.size() / 2);
                    s->h_chv[2 * (size_t)c + 1] = np;
                    const int per = (len + np - 1) / np;
                    for (int p = 0; p < np; ++p) {
                        s->h_cpieces.push_back(crp[c] + p * per);
                        s->h_cpieces.push_back(std::min(crp[c] + (p + 1) * per, crp[c + 1]));
                    }
                    s->h_crp_light[c + 1] = s->h_crp_light[c];
                } else {
                    s->h_chv[2 * (size_t)c] = -1;
                    s->h_ccl_light.insert(s->h_ccl_light.end(), g->h_c_lit.begin() + crp[c], g->h_c_lit.begin() + crp[c + 1]);
                    s->h_crp_light[c + 1] = s->h_crp_light[c] + len;
                }
            }
            s->n_cpieces = (int)(s->h_cpieces.size() / 2);
            crp = s->h_crp_light.data();
        }
        build_tile_meta(g->h_cl_seg, g->C, 16 * rb, s->h_ctile_seg, s->h_cgseg, crp, -1, s->h_cmeta);
        if (getenv("DSAT_NO_LISTBLOCKS") == nullptr)   // the clause -> literal lists of P2's tiles likewise
            s->clist_stride = build_list_blocks(crp, s->n_cpieces > 0 ? s->h_ccl_light.data() : g->h_c_lit.data(), g->C,
                                                16 * rb, -1, LIST_BLOCK_CAP, s->h_vmeta4, &s->clist_at, &s->clist_max);
    }
    Slab dry;
    carve(s.get(), dry);
    s->slab_bytes = dry.off + 256;
    hipError_t e = hipSuccess;
    if (arena != nullptr) {
        int rc = require_device(arena->device, "dsat_session_create_in");
        if (rc != DSAT_OK) return rc;
        s->slab = static_cast<char*>(arena->take_dev(s->slab_bytes));
        // staging for the four tile / segment tables
        const size_t meta_bytes = (s->h_vmeta.size() + s->h_vgseg.size() + s->h_cmeta.size() + s->h_cgseg.size() +
                                   s->h_lrp_light.size() + s->h_lcl_light.size() + s->h_lhv.size() +
                                   s->h_pieces.size() + s->h_vgseg2.size() + s->h_vchunks.size() +
                                   s->h_cgseg2.size() + s->h_cchunks.size() + s->h_sel_tab.size() + s->h_vmeta4.size() +
                                   s->h_pack_off.size() + s->h_crp_light.size() + s->h_ccl_light.size() +
                                   s->h_chv.size() + s->h_cpieces.size()) * sizeof(int32_t) + 32 * 64;
        char* stage = static_cast<char*>(arena->take_pin(meta_bytes));
        if (s->slab == nullptr || stage == nullptr) return arena_full(arena, "dsat_session_create_in");
        e = hipMemsetAsync(s->slab, 0, s->slab_bytes, st);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return hip_fail(e, "session init", __FILE__, __LINE__);
        Slab real;
        real.base = s->slab; real.size = s->slab_bytes; real.dry = false;
        carve(s.get(), real);
        auto cp = [&](int32_t* d, const std::vector<int32_t>& h) {
            if (h.empty()) return hipSuccess;
            memcpy(stage, h.data(), h.size() * sizeof(int32_t));
            const hipError_t r = hipMemcpyAsync(d, stage, h.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
            stage += (h.size() * sizeof(int32_t) + 63) & ~size_t(63);
            return r;
        };
        hipError_t e2 = cp(s->vmeta, s->h_vmeta);
        if (e2 == hipSuccess) e2 = cp(s->vgseg, s->h_vgseg);
        if (e2 == hipSuccess) e2 = cp(s->cmeta, s->h_cmeta);
        if (e2 == hipSuccess) e2 = cp(s->cgseg, s->h_cgseg);
        if (e2 == hipSuccess && s->n_pieces > 0) {
            e2 = cp(s->lrp_light, s->h_lrp_light);
            if (e2 == hipSuccess) e2 = cp(s->lcl_light, s->h_lcl_light);
            if (e2 == hipSuccess) e2 = cp(s->lhv, s->h_lhv);
            if (e2 == hipSuccess) e2 = cp(s->pieces, s->h_pieces);
        }
        if (e2 == hipSuccess && s->n_cpieces > 0) {
            e2 = cp(s->crp_light, s->h_crp_light);
            if (e2 == hipSuccess) e2 = cp(s->ccl_light, s->h_ccl_light);
            if (e2 == hipSuccess) e2 = cp(s->chv, s->h_chv);
            if (e2 == hipSuccess) e2 = cp(s->cpieces, s->h_cpieces);
        }
        if (e2 == hipSuccess && s->n_vchunks > 0) {
            e2 = cp(s->vgseg2, s->h_vgseg2);
            if (e2 == hipSuccess) e2 = cp(s->vchunks, s->h_vchunks);
        }
        if (e2 == hipSuccess && s->n_cchunks > 0) {
            e2 = cp(s->cgseg2, s->h_cgseg2);
            if (e2 == hipSuccess) e2 = cp(s->cchunks, s->h_cchunks);
        }
        if (e2 == hipSuccess) e2 = cp(s->sel_tab, s->h_sel_tab);
        if (e2 == hipSuccess) e2 = cp(s->vmeta4, s->h_vmeta4);
        if (e2 == hipSuccess) e2 = cp(s->pack_off, s->h_pack_off);
        if (e2 != hipSuccess) {
            (void)hipStreamDestroy(s->cap_stream);
            return hip_fail(e2, "segment upload", __FILE__, __LINE__);
        }
    } else {
    DSAT_HIP(hipMalloc((void**)&s->slab, s->slab_bytes));
    e = hipMemset(s->slab, 0, s->slab_bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        (void)hipFree(s->slab);
        return hip_fail(e, "session init", __FILE__, __LINE__);
    }
    Slab real;
    real.base = s->slab; real.size = s->slab_bytes; real.dry = false;
    carve(s.get(), real);
    {
        auto cp = [](int32_t* d, const std::vector<int32_t>& h) {
            return h.empty() ? hipSuccess : hipMemcpy(d, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        };
        hipError_t e2 = cp(s->vmeta, s->h_vmeta);
        if (e2 == hipSuccess) e2 = cp(s->vgseg, s->h_vgseg);
        if (e2 == hipSuccess) e2 = cp(s->cmeta, s->h_cmeta);
        if (e2 == hipSuccess) e2 = cp(s->cgseg, s->h_cgseg);
        if (e2 == hipSuccess && s->n_pieces > 0) {
            e2 = cp(s->lrp_light, s->h_lrp_light);
            if (e2 == hipSuccess) e2 = cp(s->lcl_light, s->h_lcl_light);
            if (e2 == hipSuccess) e2 = cp(s->lhv, s->h_lhv);
            if (e2 == hipSuccess) e2 = cp(s->pieces, s->h_pieces);
        }
        if (e2 == hipSuccess && s->n_cpieces > 0) {
            e2 = cp(s->crp_light, s->h_crp_light);
            if (e2 == hipSuccess) e2 = cp(s->ccl_light, s->h_ccl_light);
            if (e2 == hipSuccess) e2 = cp(s->chv, s->h_chv);
            if (e2 == hipSuccess) e2 = cp(s->cpieces, s->h_cpieces);
        }
        if (e2 == hipSuccess && s->n_vchunks > 0) {
            e2 = cp(s->vgseg2, s->h_vgseg2);
            if (e2 == hipSuccess) e2 = cp(s->vchunks, s->h_vchunks);
        }
        if (e2 == hipSuccess && s->n_cchunks > 0) {
            e2 = cp(s->cgseg2, s->h_cgseg2);
            if (e2 == hipSuccess) e2 = cp(s->cchunks, s->h_cchunks);
        }
        if (e2 == hipSuccess) e2 = cp(s->sel_tab, s->h_sel_tab);
        if (e2 == hipSuccess) e2 = cp(s->vmeta4, s->h_vmeta4);
        if (e2 == hipSuccess) e2 = cp(s->pack_off, s->h_pack_off);
        if (e2 != hipSuccess) {
            (void)hipFree(s->slab);
            return hip_fail(e2, "segment upload", __FILE__, __LINE__);
        }
    }
    }
    if (const char* ev = getenv("DSAT_STAMPS")) {
        if (atoi(ev) != 0) {
            DSAT_HIP(hipMalloc((void**)&s->stamps, 3 * 4096 * 8 * sizeof(long long)));
            DSAT_HIP(hipMemset(s->stamps, 0, 3 * 4096 * 8 * sizeof(long long)));
        }
    }
    if (!plan_round(s.get())) s->use_fused = 0;
    if (s->stamps) { s->ra1.stamps = s->stamps; s->ra2.stamps = s->stamps + 4096 * 8; s->ra3.stamps = s->stamps + 2 * 4096 * 8; }
    if (const char* ev = getenv("DSAT_FUSED")) s->use_fused = s->use_fused && atoi(ev) != 0;
    for (int i = 0; i < KC_COUNT; ++i) { s->prof_us[i] = 0; s->prof_n[i] = 0; }
    if (arena != nullptr) { s->arena = arena; arena->live++; }
    *out = s.release();
    return DSAT_OK;
}

extern "C" int dsat_session_create(const dsat_graph* g, const dsat_weights* w, int32_t max_rounds,
                                    dsat_session** out) {
    return session_create_impl(nullptr, nullptr, g, w, max_rounds, out);
}

extern "C" int dsat_session_create_in(dsat_arena* arena, void* stream, const dsat_graph* g, const dsat_weights* w,
                                       int32_t max_rounds, dsat_session** out) {
    DSAT_REQUIRE(arena != nullptr, DSAT_E_INVALID, "dsat_session_create_in: null arena");
    return session_create_impl(arena, (hipStream_t)stream, g, w, max_rounds, out);
}

extern "C" void dsat_session_destroy(dsat_session* s) {
    if (!s) return;
    for (auto& kv : s->graphs) (void)hipGraphExecDestroy(kv.second);
    for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
    if (s->cap_stream) (void)hipStreamDestroy(s->cap_stream);
    if (s->n_act_host) (void)hipHostFree(s->n_act_host);
    if (s->arena != nullptr) s->arena->live--;
    else if (s->slab) (void)hipFree(s->slab);
    if (s->stamps) (void)hipFree(s->stamps);
    delete s;
}

extern "C" int dsat_session_read_stamps(dsat_session* s, int64_t* out, int64_t n) {
    DSAT_REQUIRE(s && out, DSAT_E_INVALID, "dsat_session_read_stamps: null argument");
    DSAT_REQUIRE(s->stamps != nullptr, DSAT_E_STATE, "dsat_session_read_stamps: session was not created with DSAT_STAMPS=1");
    DSAT_REQUIRE(n <= 3 * 4096 * 8, DSAT_E_INVALID, "dsat_session_read_stamps: n too large");
    DSAT_HIP(hipDeviceSynchronize());
    DSAT_HIP(hipMemcpy(out, s->stamps, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
    return DSAT_OK;
}

extern "C" int dsat_session_set_graph_replay(dsat_session* s, int32_t enable) {
    DSAT_REQUIRE(s != nullptr, DSAT_E_INVALID, "dsat_session_set_graph_replay: null session");
    s->use_graph = enable ? 1 : 0;
    return DSAT_OK;
}

extern "C" int dsat_session_set_tap(dsat_session* s, const dsat_tap* tap) {
    DSAT_REQUIRE(s != nullptr, DSAT_E_INVALID, "dsat_session_set_tap: null session");
    if (tap == nullptr) {
        s->tap_on = false;
        memset(&s->tap, 0, sizeof(s->tap));
        return DSAT_OK;
    }
    DSAT_REQUIRE(tap->round >= 0 && tap->round < s->max_rounds, DSAT_E_INVALID,
                 "dsat_session_set_tap: round must be in [0, max_rounds of the session)");
    s->tap = *tap;
    s->tap_on = true;
    return DSAT_OK;
}

extern "C" int dsat_session_set_retire(dsat_session* s, int32_t enable) {
    DSAT_REQUIRE(s != nullptr, DSAT_E_INVALID, "dsat_session_set_retire: null session");
    if (!enable) { s->retire = 0; return DSAT_OK; }
    DSAT_REQUIRE(!s->profile, DSAT_E_STATE,
                 "dsat_session_set_retire: profiling is on (dsat_session_set_profile); it times the non-retiring sequence");
    DSAT_REQUIRE(s->g->identical_copies, DSAT_E_UNSUPPORTED,
                 "dsat_session_set_retire: the batch is not a set of identical copies of one formula");
    DSAT_REQUIRE(s->use_fused, DSAT_E_UNSUPPORTED, "dsat_session_set_retire: needs the fused round kernels");
    DSAT_REQUIRE(s->n_sel == s->g->G && s->n_vchunks == 0 && s->n_cchunks == 0, DSAT_E_UNSUPPORTED,
                 "dsat_session_set_retire: formula too large (its select / PairNorm work is split over workgroups)");
    if (s->n_act_host == nullptr) DSAT_HIP(hipHostMalloc((void**)&s->n_act_host, 4 * sizeof(int32_t), hipHostMallocDefault));
    s->retire = 1;
    return DSAT_OK;
}

extern "C" int dsat_session_set_supervised(dsat_session* s, int32_t supervised) {
    DSAT_REQUIRE(s != nullptr, DSAT_E_INVALID, "dsat_session_set_supervised: null session");
    s->unsupervised = supervised ? 0 : 1;   // (its own captured graph: select/SAT is then a launch of its own per round)
    return DSAT_OK;
}

extern "C" int dsat_session_set_profile(dsat_session* s, int32_t enable) {
    DSAT_REQUIRE(s != nullptr, DSAT_E_INVALID, "dsat_session_set_profile: null session");
    DSAT_REQUIRE(!(enable && s->retire), DSAT_E_STATE,
                 "dsat_session_set_profile: the session retires solved copies (dsat_session_set_retire); the profiling "
                 "pass times the non-retiring launch sequence, so the two cannot be on together");
    s->profile = enable ? 1 : 0;
    return DSAT_OK;
}

// ---- profiled launch helper ------------------------------------------------------
struct Launcher {
    dsat_session* s;
    hipStream_t st;
    bool prof;
    template <typename Fn>
    void run(int cls, Fn&& fn) {
        if (!prof) { fn(); return; }
        const size_t i = s->ev_used.size() * 2;
        while (s->ev_pool.size() < i + 2) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            s->ev_pool.push_back(e);
        }
        (void)hipEventRecord(s->ev_pool[i], st);
        fn();
        (void)hipEventRecord(s->ev_pool[i + 1], st);
        s->ev_used.push_back({cls, (int)i});
    }
};

static int collect_profile(dsat_session* s, hipStream_t st) {
    DSAT_HIP(hipStreamSynchronize(st));
    // An event pair costs time even with nothing between the two records (the markers are
    // packets of their own).  Calibrate that on this stream and take it off every bracket, so the
    // per-kernel averages agree with rocprofv3's kernel durations.
    double overhead_us = 0.0;
    {
        constexpr int NCAL = 16;
        hipEvent_t ev[2 * NCAL];
        for (auto& e : ev) DSAT_HIP(hipEventCreate(&e));
        for (int i = 0; i < NCAL; ++i) {
            DSAT_HIP(hipEventRecord(ev[2 * i], st));
            DSAT_HIP(hipEventRecord(ev[2 * i + 1], st));
        }
        DSAT_HIP(hipStreamSynchronize(st));
        std::vector<double> v;
        for (int i = 0; i < NCAL; ++i) {
            float ms = 0.f;
            DSAT_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            v.push_back((double)ms * 1e3);
        }
        std::sort(v.begin(), v.end());
        overhead_us = v[NCAL / 2];
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    s->prof_overhead_us = overhead_us;
    for (int i = 0; i < KC_COUNT; ++i) { s->prof_us[i] = 0; s->prof_n[i] = 0; }
    for (auto& u : s->ev_used) {
        float ms = 0.f;
        DSAT_HIP(hipEventElapsedTime(&ms, s->ev_pool[u.second], s->ev_pool[u.second + 1]));
        s->prof_us[u.first] += std::max(0.0, (double)ms * 1e3 - overhead_us);
        s->prof_n[u.first] += 1;
    }
    s->ev_used.clear();
    return DSAT_OK;
}

extern "C" int dsat_session_profile(dsat_session* s, int32_t cap, const char** names, double* us_out,
                                     int64_t* launches_out) {
    DSAT_REQUIRE(s != nullptr, DSAT_E_INVALID, "dsat_session_profile: null session");
    for (int i = 0; i < KC_COUNT && i < cap; ++i) {
        if (names) names[i] = kKernelClassNames[i];
        if (us_out) us_out[i] = s->prof_n[i] ? s->prof_us[i] / (double)s->prof_n[i] : 0.0;
        if (launches_out) launches_out[i] = s->prof_n[i];
    }
    return KC_COUNT;
}

// ---- MLP chains ---------------------------------------------------------------------
struct OutSpec {
    float* out0; float* out1; int ldo0, ldo1, split, nstore0, nstore1;
};
static OutSpec single_out(float* p, int ld, int nstore) { return {p, p, ld, ld, 1 << 30, nstore, 0}; }

static GemmArgs make_gemm(const PackedLayer& pl, int M, int act, const OutSpec& o) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.Wt = pl.wt; a.bias = pl.bias; a.kpad = pl.kpad; a.npad = pl.npad;
    a.out0 = o.out0; a.out1 = o.out1; a.ldo0 = o.ldo0; a.ldo1 = o.ldo1;
    a.split = o.split; a.nstore0 = o.nstore0; a.nstore1 = o.nstore1;
    a.M = M; a.act = act;
    return a;
}
static void add_block(GemmArgs& a, const float* p, int lda, int kp) {
    a.A[a.nblocks] = p; a.lda[a.nblocks] = lda; a.kchunks[a.nblocks] = kp / 16;
    a.nblocks++;
}

// layers [from, n_layers) of MLP m; `in`/`in_ld` feed layer `from` as one block
static void run_mlp_tail(dsat_session* s, Launcher& L, int cls, int m, int from, const float* in, int in_ld,
                         int rows, const OutSpec& final_out, const int* done) {
    const auto& layers = s->w->mlp[m];
    const int n = (int)layers.size();
    const float* cur = in;
    int cur_ld = in_ld;
    for (int l = from; l < n; ++l) {
        const bool last = l == n - 1;
        float* dst = s->hid[m][l & 1];
        OutSpec o = last ? final_out : single_out(dst, s->hid_ld[m], s->hid_ld[m]);
        GemmArgs a = make_gemm(layers[l], rows, last ? 0 : 1, o);
        add_block(a, cur, cur_ld, layers[l].kpad);
        L.run(cls, [&] { launch_gemm(a, done, L.st); });
        cur = dst;
        cur_ld = s->hid_ld[m];
    }
}

// ---- one denoising step: the launch sequence ------------------------------------------
// SURVEY.md §2.3 R1-R18; reference lines cited at each launch.
static void enqueue_step(dsat_session* s, hipStream_t st, int rounds, int early_exit, float* trace_logits,
                         bool prof) {
    const dsat_graph* g = s->g;
    const dsat_weights* w = s->w;
    const int N = g->N, C = g->C;
    Launcher L{s, st, prof};
    const int* done = early_exit ? &s->ctl->done : nullptr;
    const dsat_tap* tap = s->tap_on ? &s->tap : nullptr;
    const size_t nf = (size_t)N * F, cf = (size_t)C * F;
    auto tap_copy = [&](float* dst, const float* src, size_t n) {
        if (dst != nullptr && n > 0) (void)hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    };

    L.run(KC_MISC, [&] { launch_aux(s->aux, N, rounds, s->in_noisy, s->ctl, st); });     // :214-219, :239
    L.run(KC_MISC, [&] { launch_labels(s->labels, N, s->ctl, st); });                    // :145
    L.run(KC_MISC, [&] { launch_fill_cols(s->V, 1.f, N, w->layout.feature_maps, st); });   // :148
    L.run(KC_MISC, [&] { launch_fill_cols(s->Cs, 1.f, C, w->layout.feature_maps, st); });  // :141

    for (int r = 0; r < rounds; ++r) {
        const float* auxr = s->aux + (size_t)r * N * AUX_W;
        const bool tapped = tap != nullptr && r == tap->round;
        // R2 + R6 first layers: variables_query.L0 and lit_mlp.L0 share the input v1 (:240, :252)
        {
            float* hq = s->hid[DSAT_MLP_VARIABLES_QUERY][0];
            float* hl = s->hid[DSAT_MLP_LIT][0];
            OutSpec o{hq, hl, s->hid_ld[0], s->hid_ld[1], w->q_hpad, w->q_hpad, s->hid_ld[1]};
            GemmArgs a = make_gemm(w->ql1, N, 1, o);
            add_block(a, s->V, F, F);
            add_block(a, auxr, AUX_W, AUX_W);
            L.run(KC_GEMM_QL1, [&] { launch_gemm(a, done, st); });
        }
        // R2 query = variables_query(v1)                                              (:240)
        run_mlp_tail(s, L, KC_GEMM_Q, DSAT_MLP_VARIABLES_QUERY, 1, s->hid[0][0], s->hid_ld[0], N,
                     single_out(s->q, F, F), done);
        // R6 literals = [lit_mlp(v1)[:, :128]; [:, 128:]]                              (:252-254)
        run_mlp_tail(s, L, KC_GEMM_LIT, DSAT_MLP_LIT, 1, s->hid[1][0], s->hid_ld[1], N,
                     OutSpec{s->lits, s->lits + (size_t)N * F, F, F, F, F, F}, done);
        // R3 + R5 + R7                                                                 (:241, :248, :255-256)
        L.run(KC_CLAUSE_GATHER, [&] { launch_clause_gather(g, s->q, s->lits, s->closs4, 4.f, s->cmsg, done, st); });
        if (tapped) {
            tap_copy(tap->query, s->q, nf); tap_copy(tap->literals, s->lits, 2 * nf);
            tap_copy(tap->clauses_loss4, s->closs4, cf); tap_copy(tap->clause_messages, s->cmsg, cf);
        }
        // R8 clause_data = clause_mlp([clause_state, clause_messages, clauses_loss])   (:258-264)
        {
            const auto& layers = w->mlp[DSAT_MLP_CLAUSE];
            const bool single = layers.size() == 1;
            (void)single;
            float* h0 = s->hid[DSAT_MLP_CLAUSE][0];
            GemmArgs a = make_gemm(layers[0], C, 1, single_out(h0, s->hid_ld[2], s->hid_ld[2]));
            add_block(a, s->Cs, F, F);
            add_block(a, s->cmsg, F, F);
            add_block(a, s->closs4, F, F);
            L.run(KC_GEMM_CLAUSE, [&] { launch_gemm(a, done, st); });
            run_mlp_tail(s, L, KC_GEMM_CLAUSE, DSAT_MLP_CLAUSE, 1, h0, s->hid_ld[2], C,
                         OutSpec{s->vla, s->ncv, F, F, F, F, F}, done);
            if (tapped) { tap_copy(tap->variables_loss_all, s->vla, cf); tap_copy(tap->new_clause_value, s->ncv, cf); }
        }
        // R10 clause_state = 0.25*PairNorm(new_clause_value) + 0.1*clause_state (+ R18) (:265-266, :348)
        L.run(KC_COLSUM, [&] { launch_colsum(g->cl_seg, g->G, s->nch_c, s->ncv, s->pc, done, st); });
        L.run(KC_PAIRNORM, [&] {
            launch_pairnorm_apply(1.f / (float)w->layout.feature_maps, g->cl_gid, g->cl_seg, C, s->nch_c, s->ncv, s->pc, s->Cs, 0.25f, 0.1f, 1,
                                  s->Cs, nullptr, done, st);
        });
        if (tapped) tap_copy(tap->clause_state, s->Cs, cf);
        // R4 + R11 (closs4 holds 4*closs; 0.25 restores the tape's operand exactly)     (:242-246, :269-273)
        L.run(KC_LITERAL_GATHER, [&] {
            launch_literal_gather(g, s->q, s->closs4, 0.25f, s->vla, s->grad, s->vl, done, st);
        });
        if (tapped) { tap_copy(tap->variables_grad, s->grad, nf); tap_copy(tap->variables_loss, s->vl, 2 * nf); }
        // R12 new_variables = update_gate([grad, v1, loss_pos, loss_neg])               (:277-278)
        {
            const auto& layers = w->mlp[DSAT_MLP_UPDATE_GATE];
            float* h0 = s->hid[DSAT_MLP_UPDATE_GATE][0];
            GemmArgs a = make_gemm(layers[0], N, 1, single_out(h0, s->hid_ld[3], s->hid_ld[3]));
            add_block(a, s->grad, F, F);
            add_block(a, s->V, F, F);
            add_block(a, auxr, AUX_W, AUX_W);
            add_block(a, s->vl, F, F);
            add_block(a, s->vl + (size_t)N * F, F, F);
            L.run(KC_GEMM_UPDATE, [&] { launch_gemm(a, done, st); });
            run_mlp_tail(s, L, KC_GEMM_UPDATE, DSAT_MLP_UPDATE_GATE, 1, h0, s->hid_ld[3], N,
                         single_out(s->newv, F, F), done);
            if (tapped) tap_copy(tap->new_variables, s->newv, nf);
        }
        // R13 variables = 0.25*PairNorm(new_variables) + 0.1*variables (+ R18)          (:279-280, :347)
        L.run(KC_COLSUM, [&] { launch_colsum(g->var_seg, g->G, s->nch_v, s->newv, s->pv, done, st); });
        L.run(KC_PAIRNORM, [&] {
            launch_pairnorm_apply(1.f / (float)w->layout.feature_maps, g->var_gid, g->var_seg, N, s->nch_v, s->newv, s->pv, s->V, 0.25f, 0.1f, 1,
                                  s->V, s->Vlog, done, st);
        });
        if (tapped) tap_copy(tap->variables, s->V, nf);
        const bool last = r == rounds - 1;
        if (early_exit || last || trace_logits) {
            // R14 logits = variables_output(variables)                                 (:283)
            const auto& layers = w->mlp[DSAT_MLP_VARIABLES_OUTPUT];
            float* h0 = s->hid[DSAT_MLP_VARIABLES_OUTPUT][0];
            GemmArgs a = make_gemm(layers[0], N, 1, single_out(h0, s->hid_ld[4], s->hid_ld[4]));
            add_block(a, s->Vlog, F, F);
            L.run(KC_GEMM_OUT, [&] { launch_gemm(a, done, st); });
            run_mlp_tail(s, L, KC_GEMM_OUT, DSAT_MLP_VARIABLES_OUTPUT, 1, h0, s->hid_ld[4], N,
                         single_out(s->logits, LOGIT_MAPS, LOGIT_MAPS), done);
            if (trace_logits)
                L.run(KC_MISC, [&] {
                    launch_copy_if(s->logits, trace_logits + (size_t)r * N * LOGIT_MAPS, (int64_t)N * LOGIT_MAPS, done, st);
                });
        }
        if (early_exit || last) {
            // R15-R17: map selection and is_batch_sat                                    (:289-292, :317-338)
            L.run(KC_SELECT, [&] {
                launch_select(g, s->logits, s->labels, s->ctl, s->pgl, s->best, s->out_logits, s->assign,
                              s->graph_unsat, st);
            });
            L.run(KC_SATCHECK, [&] { launch_sat_check(g, s->assign, s->graph_unsat, s->ctl, r, early_exit, st); });
        }
    }
}

// How many 64-row groups P1's role B should cover.  One workgroup per CU, n_cus CUs, workgroups start in block
// order: with a single wave of workgroups (vtiles + groups <= n_cus) everything is co-resident and role B takes
// all groups.  On larger grids a few workgroups too many cost a whole extra tile time at the end (625 + 157
// workgroups = 3.05 waves), so the candidates are compared by a list-scheduling estimate with the measured
// relative durations (profiles/r02_stamps_*: A tile without the pair 74, with it 95, role-B workgroup 72).
static void plan_pair_blocks(int n_cus, int n_active, int* first, int* blocks, int bm = ROUND_BM_VAR,
                             double cost_with = 95.0, double cost_without = 74.0, double cost_b = 72.0) {
    // bm / costs: the A tile's height and the relative durations of an A tile that runs the pair itself, of one that
    // does not, and of a role-B workgroup (16-row tiles on the fp32 MFMA: profiles/r02_stamps_*; 32-row split tiles,
    // role_a2: profiles/r03_stamps_cfg3.txt)
    const int vtiles = (n_active + bm - 1) / bm, groups = (n_active + 63) / 64;
    *first = 0;
    *blocks = groups;
    if (vtiles + groups <= n_cus) return;
    auto makespan = [&](int nb) {
        const int f = groups - nb;                    // rows below 64*f: pair in the A tile
        std::vector<double> cu(n_cus, 0.0);           // finish times; the next block goes to the earliest CU
        std::make_heap(cu.begin(), cu.end(), std::greater<double>());
        auto put = [&](double d) {
            std::pop_heap(cu.begin(), cu.end(), std::greater<double>());
            cu.back() += d;
            std::push_heap(cu.begin(), cu.end(), std::greater<double>());
        };
        for (int t = 0; t < vtiles; ++t) put(t * bm < f * 64 ? cost_with : cost_without);
        for (int b = 0; b < nb; ++b) put(cost_b);
        return *std::max_element(cu.begin(), cu.end());
    };
    double best = makespan(groups);
    for (int nb = 0; nb < groups; nb += std::max(1, groups / 64)) {
        const double m = makespan(nb);
        if (m < best - 1e-9) { best = m; *blocks = nb; *first = groups - nb; }
    }
}

// Fused launch sequence: per round P1 -> [S of the previous round] -> P2 -> P3, then the tail
// P1 (last round's PairNorm + logits) and the last S.
// Role D of P2 (kernels_round.hip): how many extra workgroups form update_gate's `variables` partial product for P3
// (0: not in this launch).  Only on CUs the clause tiles and the select workgroups leave idle; one 64-row group costs
// a workgroup 12-15 k cycles, a clause tile lasts ~52 k, select/SAT of a small graph ~28 k: the select workgroups
// take groups too unless they are the long parts of a large graph (sel_lead > 0).
static int role_d_blocks(const dsat_session* s, int actN, int actC, bool sel_on, int sel_lead, int n_sel, int* workers,
                         int* dsel, bool sel_work = true) {   // sel_work: the select workgroups take role-D groups too
    *workers = 0;
    *dsel = 0;
    if (!s->p3_part || s->prefer_tp || s->p2_rb != 4) return 0;
    const int p2_blocks = (actC + 63) / 64 + (sel_on ? (sel_lead > 0 ? sel_lead : n_sel) : 0);
    // CUs without a clause tile: all but the tiles' own in a single wave of workgroups, those of the LAST wave otherwise
    // (the role-D blocks are the last of the grid, so that is where they land; DSAT_P3_PART=1 confines role D to single waves)
    const int rem2 = p2_blocks % s->n_cus;
    const int nd = (actN + 63) / 64;
    const int free_cus = p2_blocks < s->n_cus ? s->n_cus - p2_blocks
                                              : (s->p3_part >= 2 && rem2 != 0 ? s->n_cus - rem2 : 0);
    const int sel_workers = sel_on && sel_lead == 0 && sel_work ? n_sel : 0;
    const int extra_blocks = std::min(std::max(nd - sel_workers, 1), std::max(free_cus, 0));
    if (free_cus < 8 || nd > 3 * extra_blocks + sel_workers) return 0;
    *workers = sel_workers + extra_blocks;
    *dsel = sel_workers > 0;
    return extra_blocks;
}

static int enqueue_step_fused(dsat_session* s, hipStream_t st, int rounds, int early_exit, float* trace_logits,
                              bool prof, int act_graphs = -1) {
    Launcher L{s, st, prof};
    hipError_t lerr = hipSuccess;   // first launch failure of the sequence
    auto note = [&](hipError_t e) { if (lerr == hipSuccess && e != hipSuccess) lerr = e; };
    const dsat_tap* tap = s->tap_on ? &s->tap : nullptr;
    const int* done_flag = early_exit ? &s->ctl->done : nullptr;
    const size_t nf = (size_t)s->g->N * F, cf = (size_t)s->g->C * F;
    auto tap_copy = [&](float* dst, const float* src, size_t n) {
        if (dst != nullptr && n > 0) note(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    };
    RoundArgs a1 = s->ra1, a2 = s->ra2, a3 = s->ra3;
    int actN = s->g->N, actC = s->g->C;
    if (act_graphs >= 0) {   // early retirement: the batch is the first act_graphs copies (identical formulas)
        const int np = s->g->N / s->g->G, mp = s->g->C / s->g->G;
        actN = act_graphs * np; actC = act_graphs * mp;
        for (RoundArgs* a : {&a1, &a2, &a3}) { a->N = actN; a->C = actC; a->G = act_graphs; a->n_sel = act_graphs; }
    }
    a1.check_done = a2.check_done = a3.check_done = early_exit;
    a1.trace_logits = trace_logits;
    int pb_first = 0, pb_blocks = -1;
    int cn3_prev = 0;   // role-C blocks the previous round's P3 carried: the clause state is already normalised ...
    int cn3_rows = 0;   // ... for rows below this (0: all of them)
    // the non-state columns of v1 (normal draw, noisy_num, noise_scale, denoised) of all rounds
    if (s->sel_exch != nullptr)   // the select parts' exchange words: an epoch (round + 1) is seen once per step
        note(hipMemsetAsync(s->sel_exch, 0, (size_t)s->n_sel * LOGIT_MAPS * sizeof(unsigned long long), st));
    L.run(KC_MISC, [&] { launch_aux(s->aux, actN, rounds, s->in_noisy, s->ctl, st); });   // :214-219, :239
    for (int r = 0; r <= rounds; ++r) {
        a1.round = r;
        a1.tail = r == rounds;
        a1.stamps = a1.tail ? nullptr : s->ra1.stamps;   // diagnostics: keep the last FULL launch's phase stamps
        const size_t half = std::max<size_t>(s->h_ctile_seg.empty() ? 1 : s->h_ctile_seg.back(), 1) * F;
        // idle CUs of P1 finish the clause PairNorm of round r-1 (only worth it while the variable
        // tiles leave CUs idle; otherwise P2 does it in its prologue)
        const int vtiles = (actN + ROUND_BM_VAR - 1) / ROUND_BM_VAR;
        // role B (output / query pair on 64-row tiles) in every launch but the tail, which is the pair's
        // output half alone
        a1.pair_blocks = 0;
        a1.pair_first = 0;
        a1.a2_tiles = s->p1_a2 ? (actN + 31) / 32 : 0;
        if (s->p1_a2) {                       // 32-row A tiles: role B covers every group, in the tail launch too --
            a1.pair_blocks = (actN + 63) / 64;
            // -- or, on grids of several waves of workgroups, only as many groups as balance the waves: the A tiles of the
            // first rows then run the pair themselves as split chains (role_a2, pair_here)
            if (s->a2_pair_here && s->ra1.bo_ws[0] != nullptr && s->ra1.sp_ws[0] != nullptr) {
                if (r == 0 || pb_blocks < 0)
                    plan_pair_blocks(s->n_cus, actN, &pb_first, &pb_blocks, 32, s->a2_cost[0], s->a2_cost[1], s->a2_cost[2]);
                a1.pair_blocks = pb_blocks;
                a1.pair_first = pb_first;
            }
        } else if (!a1.tail && s->pair_blocks != 0) {
            if (r == 0 || pb_blocks < 0) plan_pair_blocks(s->n_cus, actN, &pb_first, &pb_blocks);
            a1.pair_blocks = pb_blocks;
            a1.pair_first = pb_first;
        }
        a1.a1_split = s->p1_a1s && !s->p1_a2 && !a1.tail && a1.pair_first == 0 && a1.pair_blocks == (actN + 63) / 64;
        // CUs without a P1 workgroup: all but the workgroups' own in a single wave; with DSAT_CN_P1_TAIL=1 also those of
        // the last wave of a larger grid (the role-C blocks are the last of the grid)
        const int p1_blocks = (s->p1_a2 ? a1.a2_tiles : vtiles) + a1.pair_blocks;
        int free_cus = s->n_cus - p1_blocks;
        if (p1_blocks > s->n_cus) {
            // Several waves of workgroups: the CUs the LAST wave leaves idle can host role C too (its blocks are the last
            // of the grid) -- when that wave still holds A tiles, which last long enough to cover a block's row streaming
            // (config 5: 57 A tiles in the last wave, P3 75 -> 66 us for P1 + 0.5 us, 97.0 -> 99.0 steps/s); a last wave
            // of role-B workgroups alone is too short (config 3: P1 189 -> 228 us): P3's tail hosts it then (below)
            const int rem = p1_blocks % s->n_cus, idle1 = rem ? s->n_cus - rem : 0;
            const bool a_in_last = rem > a1.pair_blocks;
            const bool want = s->cn_p1_tail == 1 || (s->cn_p1_tail < 0 && a_in_last && idle1 >= 8 && (int64_t)actC <= (int64_t)idle1 * 1500);
            free_cus = want ? idle1 : 0;
        }
        const int cn = (r > 0 && r < rounds && actC > 0 && free_cus >= 8 && !s->no_cn && cn3_prev == 0) ? free_cus : 0;
        a1.cn_blocks = cn;
        float* vbuf[2] = {s->V, s->Vlog};   // carried state, double-buffered by round (Vlog is free on this path)
        a1.V_in = vbuf[(r & 1) ^ 1];
        a1.V_out = vbuf[r & 1];
        a1.V = a1.V_out;
        a1.psum_c_rd = s->psum_c + ((r + 1) & 1) * half;   // written by P2(r-1)
        L.run(KC_ROUND_P1, [&] { note(launch_round_p1(a1, st)); });   // also produces logits of round r-1
        if (tap != nullptr && r == tap->round) { tap_copy(tap->query, s->q, nf); tap_copy(tap->literals, s->lits, 2 * nf); }
        if (tap != nullptr && r == tap->round + 1) tap_copy(tap->variables, a1.V_out, nf);
        if (r == rounds) {
            // R15-R17 of the last round (with early exit the earlier rounds' run inside P2)
            L.run(KC_SELECT_SAT, [&] {
                launch_select_sat(s->g, s->sel_tab, act_graphs >= 0 ? act_graphs : s->n_sel, s->logits, s->ctl, s->pgl, s->best, s->out_logits,
                                  s->assign, s->graph_unsat, r - 1, early_exit, s->sel_exch, s->sel_ticks, st);
            });
            break;
        }
        a2.round = r;
        a2.s_enable = early_exit && r > 0 && !s->unsupervised;   // G extra blocks: select/SAT of round r-1
        // ... or G / 4: with hundreds of small graphs (config 3: 256) one 1024-thread workgroup per graph is a whole extra
        // wave of P2's grid for 100 variables each; four graphs share a workgroup then, 256 threads each
        // (not where the select workgroups double as role-D workers: that is the small-batch regime anyway)
        a2.sel_pack = 1;
        if (early_exit && r > 0 && s->unsupervised)   // supervised=False: the rule lives in the stand-alone kernel only
            L.run(KC_SELECT_SAT, [&] {
                launch_select_sat(s->g, s->sel_tab, act_graphs >= 0 ? act_graphs : s->n_sel, s->logits, s->ctl, s->pgl, s->best,
                                  s->out_logits, s->assign, s->graph_unsat, r - 1, early_exit, s->sel_exch, s->sel_ticks, st);
            });
        a2.cs_ready = cn > 0 || cn3_prev > 0;
        a2.cs_ready_rows = cn3_prev > 0 ? cn3_rows : 0;
        {
            a2.psum_c_wr = s->psum_c + (r & 1) * half;
            a2.psum_c_rd = s->psum_c + ((r + 1) & 1) * half;
        }
        a3.round = r;
        a3.V = a1.V_out;                    // the state P1 of this round wrote
        {   // role D: only on CUs the clause tiles (and the select workgroups) leave idle
            int workers = 0, dsel = 0;
            const bool pack = s->sel_pack_on && a2.s_enable && a2.sel_lead == 0 && a2.n_sel >= 64 &&
                              (actN + a2.n_sel - 1) / a2.n_sel <= 1024;
            if (pack) a2.sel_pack = 4;   // (packed select workgroups are not role-D workers)
            const int extra_blocks = role_d_blocks(s, actN, actC, a2.s_enable != 0, a2.sel_lead,
                                                   pack ? (a2.n_sel + 3) / 4 : a2.n_sel, &workers, &dsel, !pack);
            const bool part = extra_blocks > 0;
            a2.d_blocks = extra_blocks;
            a2.d_sel = dsel;
            a2.d_workers = workers;
            a2.V = a1.V_out;
            a2.d_ws = s->ra3.sp_ws[0];
            a2.d_ks0 = s->ra3.sp_ks[0];
            a2.d_nt0 = s->ra3.sp_nt[0];
            a2.upd_part = a3.upd_part = part ? s->hid[DSAT_MLP_UPDATE_GATE][0] : nullptr;
            a2.upd_ld = a3.upd_ld = s->hid_ld[DSAT_MLP_UPDATE_GATE];
        }
        const bool tapped = tap != nullptr && r == tap->round;
        a2.tap_cmsg = tapped ? tap->clause_messages : nullptr;
        a3.tap_grad = tapped ? tap->variables_grad : nullptr;
        a3.tap_vloss = tapped ? tap->variables_loss : nullptr;
        if (s->n_cpieces > 0)   // sums of the heavy clauses, spread over the chip (reads P1's sp / lits)
            L.run(KC_HEAVY, [&] {
                note(launch_heavy_gather(reinterpret_cast<const int2*>(s->cpieces), s->n_cpieces, s->g->c_lit, s->sp,
                                         s->lits, 2 * s->g->N, s->chpart, done_flag, st));
            });
        L.run(KC_ROUND_P2, [&] { note(launch_round_p2(a2, s->p2_rb, st)); });
        if (s->n_cchunks > 0)   // level-2 sums of the clause column sums P2 just wrote (large graphs)
            L.run(KC_SEGSUM2, [&] {
                note(launch_segsum2(reinterpret_cast<const int2*>(s->cchunks), s->n_cchunks, a2.psum_c_wr, s->psum2_c,
                                    done_flag, st));
            });
        if (tapped) {
            tap_copy(tap->clauses_loss4, s->closs4, cf);
            tap_copy(tap->variables_loss_all, s->vla, cf);
            tap_copy(tap->new_clause_value, s->ncv, cf);
        }
        if (tap != nullptr && r == tap->round + 1) tap_copy(tap->clause_state, s->Cs, cf);   // R10 of the tapped round
        if (s->n_pieces > 0)   // sums of the heavy literal rows, spread over the chip (reads P2's closs4 / vla)
            L.run(KC_HEAVY, [&] {
                note(launch_heavy_gather(reinterpret_cast<const int2*>(s->pieces), s->n_pieces, s->g->l_clause,
                                         s->closs4, s->vla, s->g->C, s->hpart, done_flag, st));
            });
        {   // Role C in P3: the clause PairNorm of this round, which P2 of the next round would otherwise do inside every
            // tile (12.7 k cycles per 64-row tile at config 3, where P1 has no idle CU for it) or P1's role C on its few
            // idle CUs.  The CUs P3's LAST wave of tiles leaves idle (all but the tiles' own when there is one wave) take
            // it, as blocks behind the tiles.  One wave (config 2): everything (L2-resident, ~60 cycles per row).  Several
            // waves: the rows stream from HBM then, 1.5 KB per row, and the blocks outlast the last wave's tiles (config 3:
            // P3 130 -> 147 us) -- but P2 loses more than that (247 -> 220 us): 56.7 -> 57.9 steps/s, config 5 95.0 -> 97.4.
            // cn_frac < 1 hands over only the first share of the rows (whole P2 tiles; the rest is normalised inside P2's
            // tiles as before): 0.5 and 0.75 gave less (57.7 / 57.8, 95.2 / 95.2), so the default is all rows.
            // (Role-C rows BETWEEN the tile rows, DSAT_CN_P3=2, were measured and lost: a block holds a whole CU while it
            // waits for memory -- config 3: P3 227 us against 147 us with the blocks behind the tiles.)
            const int p3tiles = (actN + s->p3_bm - 1) / s->p3_bm;
            // (only where P1 cannot: with idle CUs of its own -- a single wave of workgroups -- P1 hosts role C for free,
            // while behind P3's shorter tiles the same blocks cost 0.3-0.6 us per round: config 2 361.5 against 359.1 steps/s)
            const bool host_c = r + 1 < rounds && actC > 0 && !s->no_cn && s->cn_in_p3 > 0 && (free_cus < 8 || s->cn_in_p3 >= 3);
            const int rem = p3tiles % s->n_cus;
            const int idle3 = p3tiles < s->n_cus ? s->n_cus - p3tiles : (rem == 0 ? 0 : s->n_cus - rem);
            a3.cn_blocks = 0;
            a3.cn_period = 0;
            a3.cn_rows = 0;
            if (host_c && s->cn_in_p3 == 2 && p3tiles >= s->n_cus && s->stamps == nullptr) {   // experiment: interleaved rows
                const int nc = std::max(8, std::min(p3tiles, actC / 64));
                a3.cn_blocks = nc;
                a3.cn_period = std::max(1, ((p3tiles + 7) / 8) / ((nc + 7) / 8));
            } else if (host_c && idle3 >= 8) {
                const int rows = p3tiles < s->n_cus ? actC : std::min(actC, (int)((double)actC * s->cn_frac) / 64 * 64);
                if (rows > 0 && (int64_t)rows <= (int64_t)idle3 * 1200) {
                    a3.cn_blocks = idle3;
                    a3.cn_rows = rows < actC ? rows : 0;
                }
            }
            a3.psum_c_rd = a2.psum_c_wr;    // the column sums P2 of this round wrote (level-2 sums: segsum2 above)
            cn3_prev = a3.cn_blocks;
            cn3_rows = a3.cn_rows;
        }
        L.run(KC_ROUND_P3, [&] { note(launch_round_p3(a3, st)); });
        if (s->n_vchunks > 0)
            L.run(KC_SEGSUM2, [&] {
                note(launch_segsum2(reinterpret_cast<const int2*>(s->vchunks), s->n_vchunks, s->psum_v, s->psum2_v,
                                    done_flag, st));
            });
        if (tapped) tap_copy(tap->new_variables, s->newv, nf);
    }
    if (lerr != hipSuccess) return hip_fail(lerr, "fused round launch", __FILE__, __LINE__);
    return DSAT_OK;
}

// Run the step either eagerly or by replaying a captured graph.
static int run_step(dsat_session* s, hipStream_t st, int rounds, int early_exit, float* trace_logits, int mode) {
    const bool prof = s->profile != 0;
    if (!s->use_graph || prof || trace_logits || s->tap_on) {
        if (mode == 1) launch_sample_pre(s->x0, s->x1, s->in_noisy, s->g->N, s->ctl, rounds, st);
        if (s->use_fused) {
            int rc = enqueue_step_fused(s, st, rounds, early_exit, trace_logits, prof);
            if (rc != DSAT_OK) return rc;
        } else {
            enqueue_step(s, st, rounds, early_exit, trace_logits, prof);
        }
        if (mode == 1)
            launch_sample_post(s->x0, s->x1, s->out_logits, s->g, s->graph_unsat, s->fixed_step, s->x_fixed,
                               s->cum_acc, s->ctl, st);
        DSAT_HIP(hipGetLastError());
        return DSAT_OK;
    }
    auto key = std::make_tuple(rounds, early_exit, mode, s->use_fused + 2 * s->p1_a2 + 4 * s->prefer_tp + 8 * s->unsupervised);
    auto it = s->graphs.find(key);
    if (it == s->graphs.end()) {
        hipGraph_t graph = nullptr;
        DSAT_HIP(hipStreamBeginCapture(s->cap_stream, hipStreamCaptureModeThreadLocal));
        if (mode == 1) launch_sample_pre(s->x0, s->x1, s->in_noisy, s->g->N, s->ctl, rounds, s->cap_stream);
        int erc = DSAT_OK;
        if (s->use_fused) erc = enqueue_step_fused(s, s->cap_stream, rounds, early_exit, nullptr, false);
        else enqueue_step(s, s->cap_stream, rounds, early_exit, nullptr, false);
        if (mode == 1)
            launch_sample_post(s->x0, s->x1, s->out_logits, s->g, s->graph_unsat, s->fixed_step, s->x_fixed,
                               s->cum_acc, s->ctl, s->cap_stream);
        // the capture is always ended, also after a failed launch: an open capture would poison the stream
        const hipError_t ce = hipStreamEndCapture(s->cap_stream, &graph);
        if (erc != DSAT_OK) {
            if (graph != nullptr) (void)hipGraphDestroy(graph);
            return erc;
        }
        if (ce != hipSuccess) return hip_fail(ce, "hipStreamEndCapture", __FILE__, __LINE__);
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) return hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__);
        it = s->graphs.emplace(key, exec).first;
    }
    DSAT_HIP(hipGraphLaunch(it->second, st));
    return DSAT_OK;
}

static Ctl base_ctl(float noise_scale, uint32_t step, int n_steps, int rounds, uint64_t seed) {
    Ctl c;
    memset(&c, 0, sizeof(c));
    c.noise_scale = noise_scale; c.step = step; c.n_steps = n_steps; c.rounds = rounds; c.seed = seed;
    c.steps_taken = rounds - 1;
    return c;
}

static int diffusion_step_impl(dsat_session* s, float noise_scale, const float* noisy_num, const float* denoised,
                                    const float* normal, const int32_t* labels, uint64_t seed,
                                    uint32_t step_index, int32_t rounds, int32_t early_exit,
                                    float* logits_out, float* logits_rounds_out, float* prediction_out,
                                    int32_t* best_map_out, int32_t* steps_taken_out, void* stream) {
    DSAT_REQUIRE(s != nullptr, DSAT_E_INVALID, "dsat_diffusion_step: null session");
    DSAT_REQUIRE(noisy_num != nullptr && prediction_out != nullptr, DSAT_E_INVALID,
                 "dsat_diffusion_step: noisy_num and prediction_out are required");
    DSAT_REQUIRE(rounds >= 1 && rounds <= s->max_rounds, DSAT_E_INVALID,
                 "dsat_diffusion_step: rounds must be in [1, max_rounds of the session]");
    DSAT_REQUIRE(std::isfinite(noise_scale), DSAT_E_INVALID, "dsat_diffusion_step: noise_scale is not finite");
    { int rc = require_device(s->device, "dsat_diffusion_step"); if (rc != DSAT_OK) return rc; }
    if (s->tap_on)
        DSAT_REQUIRE(s->tap.round >= 0 && s->tap.round < rounds &&
                         (s->tap.clause_state == nullptr || !s->use_fused || s->tap.round < rounds - 1),
                     DSAT_E_INVALID,
                     "dsat_diffusion_step: tap round out of range (clause_state on the fused path needs round < rounds-1)");
    hipStream_t st = (hipStream_t)stream;
    const size_t N = s->g->N;
    DSAT_HIP(hipMemcpyAsync(s->in_noisy, noisy_num, N * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
    Ctl c = base_ctl(noise_scale, step_index, 1, rounds, seed);
    c.unsupervised = s->unsupervised;
    if (normal) {
        DSAT_HIP(hipMemcpyAsync(s->in_normal, normal, (size_t)rounds * N * 4 * sizeof(float),
                                hipMemcpyDeviceToDevice, st));
        c.inj_normal = s->in_normal;
    }
    if (denoised) {
        DSAT_HIP(hipMemcpyAsync(s->in_denoised, denoised, N * sizeof(float), hipMemcpyDeviceToDevice, st));
        c.denoised = s->in_denoised;
    }
    if (labels) {
        DSAT_HIP(hipMemcpyAsync(s->in_labels, labels, N * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        c.inj_labels = s->in_labels;
    }
    launch_ctl_set(s->ctl, c, st);
    int rc = run_step(s, st, rounds, early_exit ? 1 : 0, logits_rounds_out, 0);
    if (rc != DSAT_OK) return rc;
    DSAT_HIP(hipMemcpyAsync(prediction_out, s->out_logits, N * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (logits_out)
        DSAT_HIP(hipMemcpyAsync(logits_out, s->logits, N * LOGIT_MAPS * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (best_map_out)
        DSAT_HIP(hipMemcpyAsync(best_map_out, s->best, (size_t)s->g->G * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (steps_taken_out)
        DSAT_HIP(hipMemcpyAsync(steps_taken_out, &s->ctl->steps_taken, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (s->profile) return collect_profile(s, st);
    return DSAT_OK;
}

extern "C" int dsat_diffusion_step(dsat_session* s, float noise_scale, const float* noisy_num,
                                    const float* normal, const int32_t* labels, uint64_t seed,
                                    uint32_t step_index, int32_t rounds, int32_t early_exit,
                                    float* logits_out, float* logits_rounds_out, float* prediction_out,
                                    int32_t* best_map_out, int32_t* steps_taken_out, void* stream) {
    return diffusion_step_impl(s, noise_scale, noisy_num, nullptr, normal, labels, seed, step_index, rounds,
                               early_exit, logits_out, logits_rounds_out, prediction_out, best_map_out,
                               steps_taken_out, stream);
}

extern "C" int dsat_diffusion_step_self(dsat_session* s, float noise_scale, const float* noisy_num,
                                         const float* denoised_num, const float* normal, const int32_t* labels,
                                         uint64_t seed, uint32_t step_index, int32_t rounds, int32_t early_exit,
                                         float* logits_out, float* logits_rounds_out, float* prediction_out,
                                         int32_t* best_map_out, int32_t* steps_taken_out, void* stream) {
    DSAT_REQUIRE(denoised_num != nullptr, DSAT_E_INVALID, "dsat_diffusion_step_self: denoised_num is required");
    return diffusion_step_impl(s, noise_scale, noisy_num, denoised_num, normal, labels, seed, step_index, rounds,
                               early_exit, logits_out, logits_rounds_out, prediction_out, best_map_out,
                               steps_taken_out, stream);
}

namespace dsat {
__global__ void iota_kernel(int32_t* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
}  // namespace dsat

// diffusion() for a batch of identical copies with early retirement: after every denoising step the
// copies that have been solved (frozen by the reference, DiffusionSampler.py:154-170, and never looked at
// again, :181-184) leave the batch; the survivors move to the front and the next step's kernels run on
// that prefix only.  One host synchronisation per step (the new batch size sizes the next launches).
// Every copy's random draws are keyed by its original position, so its trajectory is the one it has
// without retirement -- except that the batch-global early exit (is_batch_sat) now means "all copies
// still in the batch are satisfied".
static int sample_retire_impl(dsat_session* s, int32_t n_steps, int32_t rounds, uint64_t seed, int32_t early_exit,
                              const float* uniform, const float* normal, const int32_t* labels,
                              uint8_t* assignment_out, uint8_t* solved_out, float* prediction_steps_out,
                              float* solved_fraction_out, hipStream_t st, bool self_conditioning) {
    const dsat_graph* g = s->g;
    const int G = g->G, np = g->N / G;
    const size_t N = g->N;
    Ctl c = base_ctl(1.f, 0, n_steps, rounds, seed);
    c.unsupervised = s->unsupervised;
    c.inj_normal = normal; c.normal_step_stride = (int64_t)rounds * N * 4;
    c.inj_labels = labels; c.labels_step_stride = (int64_t)N;
    c.inj_uniform = uniform; c.uniform_step_stride = (int64_t)N;
    c.pred_steps_out = prediction_steps_out;
    c.lid = s->lid; c.n_per = np;
    if (self_conditioning) { c.denoised = s->in_denoised; c.denoised_wr = s->in_denoised; c.denoised_from_step = 1; }
    launch_ctl_set(s->ctl, c, st);
    launch_sample_init(s->x0, s->x1, g->N, s->fixed_step, s->cum_acc, G, s->ctl, st);
    hipLaunchKernelGGL(dsat::iota_kernel, dim3((G + 255) / 256), dim3(256), 0, st, s->lid, G);
    if (prediction_steps_out != nullptr)
        DSAT_HIP(hipMemsetAsync(prediction_steps_out, 0, (size_t)n_steps * N * sizeof(float), st));
    float* x0[2] = {s->x0, s->x0b};
    float* x1[2] = {s->x1, s->x1b};
    int cur = 0, n_act = G;
    for (int t = 0; t < n_steps && n_act > 0; ++t) {
        launch_sample_pre_retire(x0[cur], x1[cur], s->in_noisy, n_act * np, s->ctl, rounds, st);
        int rc = enqueue_step_fused(s, st, rounds, early_exit ? 1 : 0, nullptr, false, n_act);
        if (rc != DSAT_OK) return rc;
        launch_sample_post_retire(x0[cur], x1[cur], s->out_logits, s->graph_unsat, s->fixed_step, s->x_fixed,
                                  n_act * np, (int64_t)N, s->ctl, st);
        const bool last = t + 1 == n_steps;
        launch_retire_plan(s->fixed_step, s->cum_acc, s->graph_unsat, s->lid, s->move_src, n_act, s->ctl, s->n_act_dev,
                           last ? 0 : 1, st);
        if (!last) {
            launch_retire_move(x0[cur], x1[cur], x0[cur ^ 1], x1[cur ^ 1], s->move_src, s->n_act_dev, n_act, np, st);
            cur ^= 1;
            DSAT_HIP(hipMemcpyAsync(s->n_act_host, s->n_act_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            DSAT_HIP(hipStreamSynchronize(st));
            n_act = s->n_act_host[0];
        }
    }
    // n_act: the slots of the last executed step (their lid entries are intact), or 0 when every copy was
    // solved before the last step; solved copies take their frozen bits, the others round(p_last)
    const int n_left = n_act;
    launch_sample_final_retire(s->out_logits, s->lid, n_left, np, s->fixed_step, s->x_fixed, assignment_out, (int)N,
                               s->cum_acc, solved_out, G, st);
    DSAT_HIP(hipGetLastError());
    if (solved_fraction_out) {
        std::vector<uint8_t> h(G);
        DSAT_HIP(hipMemcpyAsync(h.data(), s->cum_acc, G, hipMemcpyDeviceToHost, st));
        DSAT_HIP(hipStreamSynchronize(st));
        double acc = 0;
        for (uint8_t b : h) acc += b;
        *solved_fraction_out = (float)(acc / (double)G);
    }
    return DSAT_OK;
}

static int sample_impl(dsat_session* s, int32_t n_steps, int32_t rounds, uint64_t seed, int32_t early_exit,
                       const float* uniform, const float* normal, const int32_t* labels,
                       uint8_t* assignment_out, uint8_t* solved_out, float* prediction_steps_out,
                       float* solved_fraction_out, void* stream, bool self_conditioning) {
    DSAT_REQUIRE(s != nullptr, DSAT_E_INVALID, "dsat_sample: null session");
    DSAT_REQUIRE(n_steps >= 1 && n_steps <= (1 << 20), DSAT_E_INVALID, "dsat_sample: n_steps out of range");
    DSAT_REQUIRE(rounds >= 1 && rounds <= s->max_rounds, DSAT_E_INVALID,
                 "dsat_sample: rounds must be in [1, max_rounds of the session]");
    DSAT_REQUIRE(assignment_out != nullptr, DSAT_E_INVALID, "dsat_sample: assignment_out is required");
    DSAT_REQUIRE(!s->tap_on, DSAT_E_STATE, "dsat_sample: a debug tap is set (taps belong to dsat_diffusion_step)");
    { int rc = require_device(s->device, "dsat_sample"); if (rc != DSAT_OK) return rc; }
    if (s->retire)
        return sample_retire_impl(s, n_steps, rounds, seed, early_exit, uniform, normal, labels, assignment_out,
                                  solved_out, prediction_steps_out, solved_fraction_out, (hipStream_t)stream,
                                  self_conditioning);
    hipStream_t st = (hipStream_t)stream;
    const dsat_graph* g = s->g;
    const size_t N = g->N;
    Ctl c = base_ctl(1.f, 0, n_steps, rounds, seed);
    c.unsupervised = s->unsupervised;
    c.inj_normal = normal; c.normal_step_stride = (int64_t)rounds * N * 4;
    c.inj_labels = labels; c.labels_step_stride = (int64_t)N;
    c.inj_uniform = uniform; c.uniform_step_stride = (int64_t)N;
    c.pred_steps_out = prediction_steps_out;
    if (self_conditioning) {   // predictions of step t-1 feed step t; step 0 runs without (:93, :110-117)
        c.denoised = s->in_denoised; c.denoised_wr = s->in_denoised; c.denoised_from_step = 1;
    }
    launch_ctl_set(s->ctl, c, st);
    launch_sample_init(s->x0, s->x1, g->N, s->fixed_step, s->cum_acc, g->G, s->ctl, st);
    for (int t = 0; t < n_steps; ++t) {
        int rc = run_step(s, st, rounds, early_exit ? 1 : 0, nullptr, 1);
        if (rc != DSAT_OK) return rc;
    }
    launch_sample_final(s->out_logits, g, s->fixed_step, s->x_fixed, assignment_out, s->cum_acc, solved_out, st);
    DSAT_HIP(hipGetLastError());
    if (s->profile) {
        int rc = collect_profile(s, st);
        if (rc != DSAT_OK) return rc;
    }
    if (solved_fraction_out) {
        std::vector<uint8_t> h(g->G);
        DSAT_HIP(hipMemcpyAsync(h.data(), s->cum_acc, g->G, hipMemcpyDeviceToHost, st));
        DSAT_HIP(hipStreamSynchronize(st));
        double acc = 0;
        for (uint8_t b : h) acc += b;
        *solved_fraction_out = (float)(acc / (double)g->G);
    }
    return DSAT_OK;
}

extern "C" int dsat_sample(dsat_session* s, int32_t n_steps, int32_t rounds, uint64_t seed, int32_t early_exit,
                            const float* uniform, const float* normal, const int32_t* labels,
                            uint8_t* assignment_out, uint8_t* solved_out, float* prediction_steps_out,
                            float* solved_fraction_out, void* stream) {
    return sample_impl(s, n_steps, rounds, seed, early_exit, uniform, normal, labels, assignment_out, solved_out,
                       prediction_steps_out, solved_fraction_out, stream, false);
}

extern "C" int dsat_sample_self(dsat_session* s, int32_t n_steps, int32_t rounds, uint64_t seed,
                                 int32_t early_exit, const float* uniform, const float* normal,
                                 const int32_t* labels, uint8_t* assignment_out, uint8_t* solved_out,
                                 float* prediction_steps_out, float* solved_fraction_out, void* stream) {
    return sample_impl(s, n_steps, rounds, seed, early_exit, uniform, normal, labels, assignment_out, solved_out,
                       prediction_steps_out, solved_fraction_out, stream, true);
}

// ============================================================================ single operators
namespace dsat {
__global__ void pack_blocks_kernel(const float* __restrict__ x, int in_dim, float* __restrict__ dst, int kpad,
                                   int rows, int nb, int k0, int kp0, int k1, int kp1, int k2, int kp2, int k3,
                                   int kp3, int k4, int kp4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * kpad) return;
    const int row = (int)(i / kpad);
    int col = (int)(i % kpad);
    const int ks[5] = {k0, k1, k2, k3, k4}, kps[5] = {kp0, kp1, kp2, kp3, kp4};
    int src = 0;
    float v = 0.f;
    for (int b = 0; b < nb; ++b) {
        if (col < kps[b]) {
            if (col < ks[b]) v = x[(size_t)row * in_dim + src + col];
            break;
        }
        col -= kps[b];
        src += ks[b];
    }
    dst[i] = v;
}
}  // namespace dsat

extern "C" int dsat_op_mlp(dsat_session* s, int32_t mlp, const float* x, int32_t rows, float* y, void* stream) {
    DSAT_REQUIRE(s != nullptr && s->w->layout.feature_maps == F && s->w->layout.query_maps == F, DSAT_E_UNSUPPORTED,
                 "the single-operator entry points take the released widths (feature_maps = query_maps = 128) only");
    DSAT_REQUIRE(s && x && y, DSAT_E_INVALID, "dsat_op_mlp: null argument");
    DSAT_REQUIRE(mlp >= 0 && mlp < DSAT_N_MLP, DSAT_E_INVALID, "dsat_op_mlp: mlp out of range");
    DSAT_ON_DEVICE(s, "dsat_op_mlp");
    const int cap = (mlp == DSAT_MLP_CLAUSE) ? std::max(s->g->C, 1) : s->g->N;
    DSAT_REQUIRE(rows >= 0 && rows <= cap, DSAT_E_INVALID,
                 "dsat_op_mlp: rows exceed the session's capacity for this MLP (N, or C for the clause MLP)");
    if (rows == 0) return DSAT_OK;
    hipStream_t st = (hipStream_t)stream;
    const auto& layers = s->w->mlp[mlp];
    auto blocks = first_layer_blocks(mlp);
    int k[5] = {0, 0, 0, 0, 0}, kp[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < blocks.size(); ++i) { k[i] = blocks[i].k; kp[i] = blocks[i].kp; }
    const int kpad = layers[0].kpad;
    const int64_t n = (int64_t)rows * kpad;
    hipLaunchKernelGGL(dsat::pack_blocks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x,
                       layers[0].in_dim, s->scratch, kpad, rows, (int)blocks.size(), k[0], kp[0], k[1], kp[1],
                       k[2], kp[2], k[3], kp[3], k[4], kp[4]);
    Launcher L{s, st, false};
    float* h0 = s->hid[mlp][0];
    const int od = layers.back().out_dim;
    GemmArgs a = make_gemm(layers[0], rows, 1, single_out(h0, s->hid_ld[mlp], s->hid_ld[mlp]));
    add_block(a, s->scratch, kpad, kpad);
    launch_gemm(a, nullptr, st);
    run_mlp_tail(s, L, KC_MISC, mlp, 1, h0, s->hid_ld[mlp], rows, single_out(y, od, od), nullptr);
    DSAT_HIP(hipGetLastError());
    return DSAT_OK;
}

static int op_mlp_fused(dsat_session* s, int mlp, const float* x, int rows, float* y, int rb, int nwaves,
                        hipStream_t st, bool pack) {
    const auto& layers = s->w->mlp[mlp];
    auto blocks = first_layer_blocks(mlp);
    int k[5] = {0, 0, 0, 0, 0}, kp[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < blocks.size(); ++i) { k[i] = blocks[i].k; kp[i] = blocks[i].kp; }
    const int kpad = layers[0].kpad;
    if (pack) {
        const int64_t n = (int64_t)rows * kpad;
        hipLaunchKernelGGL(dsat::pack_blocks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x,
                           layers[0].in_dim, s->scratch, kpad, rows, (int)blocks.size(), k[0], kp[0], k[1], kp[1],
                           k[2], kp[2], k[3], kp[3], k[4], kp[4]);
    }
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    int w0 = kpad, w1 = 16;   // widest tile held by each ping-pong buffer
    a.nlayers = (int)layers.size();
    for (int l = 0; l < a.nlayers; ++l) {
        a.L[l].wp = reinterpret_cast<const float4*>(layers[l].wp);
        a.L[l].bias = layers[l].bias;
        a.L[l].kblocks = layers[l].kblocks();
        a.L[l].ntiles = layers[l].ntiles();
        a.L[l].tpw = pick_tpw(a.L[l].ntiles, nwaves, max_tpw_for_rb(rb));
        a.L[l].tpw_half = a.L[l].tpw;
        if (l + 1 < a.nlayers) {
            if (l & 1) w0 = std::max(w0, layers[l].npad); else w1 = std::max(w1, layers[l].npad);
        }
    }
    a.x = s->scratch; a.kpad0 = kpad; a.ld_buf0 = w0 + 8; a.ld_buf1 = w1 + 8; a.rows = rows;
    a.y = y; a.ldy = layers.back().out_dim; a.ncols_y = layers.back().out_dim;
    DSAT_HIP(launch_chain(a, rb, nwaves, st));
    return DSAT_OK;
}

// impl 2: the split-product chain (kernels_split_op.hip): input packed like op_mlp_fused, weights = the bf16 planes
static int op_mlp_split(dsat_session* s, int mlp, const float* x, int rows, float* y, int rb, hipStream_t st) {
    const auto& layers = s->w->mlp[mlp];
    DSAT_REQUIRE(layers.size() >= 2 && layers.size() <= 3, DSAT_E_UNSUPPORTED,
                 "dsat_op_mlp_ex(impl=2): the split-product chain takes MLPs of 2 or 3 layers");
    for (const PackedLayer& pl : layers)
        DSAT_REQUIRE(pl.ntiles() <= ROUND_NWAVES, DSAT_E_UNSUPPORTED,
                     "dsat_op_mlp_ex(impl=2): a layer wider than 256 columns (one 16-column tile per wave)");
    DSAT_REQUIRE(rb == 1 || rb == 2 || rb == 4, DSAT_E_INVALID, "dsat_op_mlp_ex(impl=2): rb must be 1, 2 or 4");
    auto blocks = first_layer_blocks(mlp);
    int k[5] = {0, 0, 0, 0, 0}, kp[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < blocks.size(); ++i) { k[i] = blocks[i].k; kp[i] = blocks[i].kp; }
    const int kpad = layers[0].kpad;
    const int64_t n = (int64_t)rows * kpad;
    hipLaunchKernelGGL(dsat::pack_blocks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x,
                       layers[0].in_dim, s->scratch, kpad, rows, (int)blocks.size(), k[0], kp[0], k[1], kp[1],
                       k[2], kp[2], k[3], kp[3], k[4], kp[4]);
    SplitOpArgs a;
    memset(&a, 0, sizeof(a));
    a.nlayers = (int)layers.size();
    a.ldh = 16;
    for (int l = 0; l < a.nlayers; ++l) {
        a.ws[l] = reinterpret_cast<const uint4*>(layers[l].ws);
        a.bias[l] = layers[l].bias;
        a.ks[l] = layers[l].ksteps();
        a.nt[l] = layers[l].ntiles();
        if (l > 0) a.ldh = std::max(a.ldh, layers[l].ksteps() * 32 + 16);
    }
    for (int l = a.nlayers; l < 3; ++l) { a.ws[l] = a.ws[0]; a.bias[l] = a.bias[0]; a.ks[l] = a.ks[0]; a.nt[l] = 0; }
    a.x = s->scratch; a.kpad0 = kpad; a.rows = rows;
    a.ld_in = layers[0].ksteps() * 32 + 16;
    a.y = y; a.ldy = layers.back().out_dim; a.ncols_y = layers.back().out_dim;
    hipError_t e = launch_split_op(a, rb, st);
    DSAT_REQUIRE(e != hipErrorInvalidValue, DSAT_E_UNSUPPORTED,
                 "dsat_op_mlp_ex(impl=2): the tile's plane images do not fit the CU's LDS at this rb");
    DSAT_HIP(e);
    return DSAT_OK;
}

extern "C" int dsat_op_mlp_ex(dsat_session* s, int32_t mlp, const float* x, int32_t rows, float* y,
                              int32_t impl, int32_t rb, int32_t nwaves, void* stream) {
    if (impl == 0) return dsat_op_mlp(s, mlp, x, rows, y, stream);
    DSAT_REQUIRE(s && x && y, DSAT_E_INVALID, "dsat_op_mlp_ex: null argument");
    DSAT_REQUIRE(s->w->layout.feature_maps == F && s->w->layout.query_maps == F, DSAT_E_UNSUPPORTED,
                 "the single-operator entry points take the released widths (feature_maps = query_maps = 128) only");
    DSAT_REQUIRE(mlp >= 0 && mlp < DSAT_N_MLP, DSAT_E_INVALID, "dsat_op_mlp_ex: mlp out of range");
    DSAT_ON_DEVICE(s, "dsat_op_mlp_ex");
    const int cap = (mlp == DSAT_MLP_CLAUSE) ? std::max(s->g->C, 1) : s->g->N;
    DSAT_REQUIRE(rows >= 0 && rows <= cap, DSAT_E_INVALID, "dsat_op_mlp_ex: rows exceed the session's capacity");
    if (rows == 0) return DSAT_OK;
    if (impl == 2) return op_mlp_split(s, mlp, x, rows, y, rb, (hipStream_t)stream);
    return op_mlp_fused(s, mlp, x, rows, y, rb, nwaves, (hipStream_t)stream, true);
}

// Average device time (us) of `iters` back-to-back runs of MLP `mlp` on `rows` rows (inputs:
// whatever the session scratch holds), measured with HIP events on `stream`.
extern "C" int dsat_bench_mlp(dsat_session* s, int32_t mlp, int32_t rows, int32_t impl, int32_t rb,
                              int32_t nwaves, int32_t iters, double* avg_us, void* stream) {
    DSAT_REQUIRE(s && avg_us && iters > 0, DSAT_E_INVALID, "dsat_bench_mlp: bad argument");
    DSAT_REQUIRE(mlp >= 0 && mlp < DSAT_N_MLP, DSAT_E_INVALID, "dsat_bench_mlp: mlp out of range");
    DSAT_ON_DEVICE(s, "dsat_bench_mlp");
    const int cap = (mlp == DSAT_MLP_CLAUSE) ? std::max(s->g->C, 1) : s->g->N;
    DSAT_REQUIRE(rows > 0 && rows <= cap, DSAT_E_INVALID, "dsat_bench_mlp: rows exceed the session's capacity");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    DSAT_HIP(hipEventCreate(&e0));
    DSAT_HIP(hipEventCreate(&e1));
    const auto& layers = s->w->mlp[mlp];
    const int od = layers.back().out_dim;
    float* y = s->hid[mlp][1];   // hidden buffer is at least rows x out_dim? use newv/ncv-sized scratch instead
    y = (mlp == DSAT_MLP_CLAUSE) ? s->hid[DSAT_MLP_CLAUSE][1] : s->hid[DSAT_MLP_LIT][1];
    (void)od;
    auto once = [&]() -> int {
        if (impl == 0) {
            Launcher L{s, st, false};
            float* h0 = s->hid[mlp][0];
            GemmArgs a = make_gemm(layers[0], rows, 1, single_out(h0, s->hid_ld[mlp], s->hid_ld[mlp]));
            add_block(a, s->scratch, layers[0].kpad, layers[0].kpad);
            launch_gemm(a, nullptr, st);
            run_mlp_tail(s, L, KC_MISC, mlp, 1, h0, s->hid_ld[mlp], rows, single_out(y, od, od), nullptr);
            return DSAT_OK;
        }
        return op_mlp_fused(s, mlp, nullptr, rows, y, rb, nwaves, st, false);
    };
    for (int i = 0; i < 3; ++i) { int rc = once(); if (rc != DSAT_OK) return rc; }
    DSAT_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) { int rc = once(); if (rc != DSAT_OK) return rc; }
    DSAT_HIP(hipEventRecord(e1, st));
    DSAT_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    DSAT_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = (double)ms * 1e3 / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return DSAT_OK;
}

extern "C" int dsat_op_clause_gather(dsat_session* s, const float* q, const float* lits, float* closs_out,
                                      float* cmsg_out, void* stream) {
    DSAT_REQUIRE(s && q && lits && closs_out && cmsg_out, DSAT_E_INVALID, "dsat_op_clause_gather: null argument");
    DSAT_ON_DEVICE(s, "dsat_op_clause_gather");
    launch_clause_gather(s->g, q, lits, closs_out, 1.f, cmsg_out, nullptr, (hipStream_t)stream);
    DSAT_HIP(hipGetLastError());
    return DSAT_OK;
}

extern "C" int dsat_op_literal_gather(dsat_session* s, const float* q, const float* closs, const float* vla,
                                       float* grad_out, float* vloss_out, void* stream) {
    DSAT_REQUIRE(s && q && closs && vla && grad_out && vloss_out, DSAT_E_INVALID,
                 "dsat_op_literal_gather: null argument");
    DSAT_ON_DEVICE(s, "dsat_op_literal_gather");
    launch_literal_gather(s->g, q, closs, 1.f, vla, grad_out, vloss_out, nullptr, (hipStream_t)stream);
    DSAT_HIP(hipGetLastError());
    return DSAT_OK;
}

extern "C" int dsat_op_pair_norm(dsat_session* s, int32_t side, const float* x, float* y, void* stream) {
    DSAT_REQUIRE(s != nullptr && s->w->layout.feature_maps == F && s->w->layout.query_maps == F, DSAT_E_UNSUPPORTED,
                 "the single-operator entry points take the released widths (feature_maps = query_maps = 128) only");
    DSAT_REQUIRE(s && x && y, DSAT_E_INVALID, "dsat_op_pair_norm: null argument");
    DSAT_REQUIRE(side == 0 || side == 1, DSAT_E_INVALID, "dsat_op_pair_norm: side must be 0 (variables) or 1 (clauses)");
    DSAT_ON_DEVICE(s, "dsat_op_pair_norm");
    const dsat_graph* g = s->g;
    hipStream_t st = (hipStream_t)stream;
    if (side == 0) {
        launch_colsum(g->var_seg, g->G, s->nch_v, x, s->pv, nullptr, st);
        launch_pairnorm_apply(1.f / (float)F, g->var_gid, g->var_seg, g->N, s->nch_v, x, s->pv, nullptr, 1.f, 0.f, 0, y, nullptr,
                              nullptr, st);
    } else {
        launch_colsum(g->cl_seg, g->G, s->nch_c, x, s->pc, nullptr, st);
        launch_pairnorm_apply(1.f / (float)F, g->cl_gid, g->cl_seg, g->C, s->nch_c, x, s->pc, nullptr, 1.f, 0.f, 0, y, nullptr,
                              nullptr, st);
    }
    DSAT_HIP(hipGetLastError());
    return DSAT_OK;
}

namespace dsat {
__global__ void round_assign_kernel(const float* __restrict__ logits, uint8_t* __restrict__ assign,
                                    int32_t* graph_unsat, int N, int G, Ctl* ctl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) assign[i] = (uint8_t)rintf(1.f / (1.f + expf(-logits[i])));
    if (i < G) graph_unsat[i] = 0;
    if (i == 0) { ctl->unsat_total = 0; ctl->ticket = 0; ctl->done = 0; }
}
}  // namespace dsat

extern "C" int dsat_op_is_batch_sat(dsat_session* s, const float* logits, int32_t* is_sat_out, void* stream) {
    DSAT_REQUIRE(s && logits && is_sat_out, DSAT_E_INVALID, "dsat_op_is_batch_sat: null argument");
    DSAT_ON_DEVICE(s, "dsat_op_is_batch_sat");
    const dsat_graph* g = s->g;
    hipStream_t st = (hipStream_t)stream;
    const int n = std::max(g->N, g->G);
    hipLaunchKernelGGL(dsat::round_assign_kernel, dim3((n + 255) / 256), dim3(256), 0, st, logits, s->assign,
                       s->graph_unsat, g->N, g->G, s->ctl);
    launch_sat_check(g, s->assign, s->graph_unsat, s->ctl, 0, 1, st);
    int done = 0;
    DSAT_HIP(hipMemcpyAsync(&done, &s->ctl->done, sizeof(int), hipMemcpyDeviceToHost, st));
    DSAT_HIP(hipStreamSynchronize(st));
    *is_sat_out = done;
    return DSAT_OK;
}

namespace dsat {
__global__ void graph_sat_flags_kernel(const int32_t* __restrict__ graph_unsat, uint8_t* __restrict__ out, int G) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) out[g] = graph_unsat[g] == 0 ? 1 : 0;
}
}  // namespace dsat

extern "C" int dsat_op_is_graph_sat(dsat_session* s, const float* logits, uint8_t* sat_out, void* stream) {
    DSAT_REQUIRE(s && logits && sat_out, DSAT_E_INVALID, "dsat_op_is_graph_sat: null argument");
    DSAT_ON_DEVICE(s, "dsat_op_is_graph_sat");
    const dsat_graph* g = s->g;
    hipStream_t st = (hipStream_t)stream;
    const int n = std::max(g->N, g->G);
    hipLaunchKernelGGL(dsat::round_assign_kernel, dim3((n + 255) / 256), dim3(256), 0, st, logits, s->assign,
                       s->graph_unsat, g->N, g->G, s->ctl);
    launch_sat_check(g, s->assign, s->graph_unsat, s->ctl, 0, 0, st);
    hipLaunchKernelGGL(dsat::graph_sat_flags_kernel, dim3((g->G + 255) / 256), dim3(256), 0, st, s->graph_unsat,
                       sat_out, g->G);
    DSAT_HIP(hipGetLastError());
    return DSAT_OK;
}

namespace dsat {
// one workgroup per graph: byte j of the graph's packed assignment = variables 8j .. 8j+7, bit k <-> variable 8j+k
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t* __restrict__ assign,
                                                        const int32_t* __restrict__ vseg,
                                                        const int32_t* __restrict__ pack_off,
                                                        uint8_t* __restrict__ out) {
    const int g = blockIdx.x;
    const int v0 = vseg[g], v1 = vseg[g + 1];
    const int nbytes = pack_off[g + 1] - pack_off[g];
    for (int j = threadIdx.x; j < nbytes; j += blockDim.x) {
        unsigned b = 0;
        for (int k = 0; k < 8; ++k) {
            const int v = v0 + 8 * j + k;
            if (v < v1 && assign[v] != 0) b |= 1u << k;
        }
        out[pack_off[g] + j] = (uint8_t)b;
    }
}
}  // namespace dsat

extern "C" int dsat_session_set_throughput(dsat_session* s, int32_t enable) {
    DSAT_REQUIRE(s != nullptr, DSAT_E_INVALID, "dsat_session_set_throughput: null session");
    s->prefer_tp = enable != 0;   // (role D of P2 spends CU-cycles to shorten one batch's P3: off when batches overlap)
    if (s->p1_a2_env == -1) {
        const bool large = (s->g->N + ROUND_BM_VAR - 1) / ROUND_BM_VAR + s->pair_blocks > s->n_cus;
        s->p1_a2 = s->p1_a2_ok && (enable != 0 || large);
    }
    return DSAT_OK;
}

static bool role_d_on(const dsat_session* s) {   // for the whole batch, select workgroups riding along
    int w = 0, d = 0;
    return role_d_blocks(s, s->g->N, s->g->C, true, s->ra2.sel_lead, s->ra2.n_sel, &w, &d) > 0;
}

extern "C" int32_t dsat_session_split_mask(const dsat_session* s) {
    if (s == nullptr || !s->use_fused) return 0;
    return (s->p2_split ? 1 : 0) | (s->p3_split ? 2 : 0) | (s->p1_a2 ? 4 : 0) | (s->ra1.sp_ws[0] != nullptr && s->pair_blocks ? 8 : 0) |
           (s->p3_split && s->p3_bm == 32 ? 16 : 0) | (s->p1_a1s && !s->p1_a2 ? 32 : 0) | (role_d_on(s) ? 64 : 0) |
           (s->ra1.bo_ws[0] != nullptr && s->pair_blocks ? 128 : 0);
}

extern "C" int64_t dsat_session_packed_bytes(const dsat_session* s, int32_t* offsets_out) {
    if (!s) return -1;
    if (offsets_out)
        for (size_t i = 0; i < s->h_pack_off.size(); ++i) offsets_out[i] = s->h_pack_off[i];
    return s->h_pack_off.empty() ? 0 : (int64_t)s->h_pack_off.back();
}

extern "C" int dsat_op_pack_assignment(dsat_session* s, const uint8_t* assignment, uint8_t* packed_out, void* stream) {
    DSAT_REQUIRE(s && assignment && packed_out, DSAT_E_INVALID, "dsat_op_pack_assignment: null argument");
    DSAT_ON_DEVICE(s, "dsat_op_pack_assignment");
    hipLaunchKernelGGL(dsat::pack_bits_kernel, dim3(s->g->G), dim3(256), 0, (hipStream_t)stream, assignment,
                       s->g->var_seg, s->pack_off, packed_out);
    DSAT_HIP(hipGetLastError());
    return DSAT_OK;
}

extern "C" int dsat_op_rng(dsat_session* s, uint64_t seed, uint32_t step_index, int32_t rounds,
                            float* normal_out, int32_t* labels_out, float* uniform_out, void* stream) {
    DSAT_REQUIRE(s != nullptr, DSAT_E_INVALID, "dsat_op_rng: null session");
    DSAT_REQUIRE(rounds >= 0, DSAT_E_INVALID, "dsat_op_rng: negative rounds");
    DSAT_ON_DEVICE(s, "dsat_op_rng");
    launch_rng(normal_out, labels_out, uniform_out, s->g->N, rounds, seed, step_index, (hipStream_t)stream);
    DSAT_HIP(hipGetLastError());
    return DSAT_OK;
}
