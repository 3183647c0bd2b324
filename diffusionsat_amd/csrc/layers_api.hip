// Free-standing layer objects of the C-ABI (include/dsat.h, "layers"): model/mlp.py's MLP and
// layers/normalization.py's PairNorm with the reference's own argument lists, for any widths, without a
// session.  The hot path never comes through here (the fused round kernels do that work); these are the
// layer classes the reference's model file is written in, kept callable one at a time.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "dsat_internal.h"

namespace dsat {

static inline int up_to(int x, int m) { return (x + m - 1) / m * m; }

// x[rows][in_dim] -> dst[rows][kpad], zero beyond in_dim (the GEMM reads 16-column chunks)
__global__ void pad_rows_kernel(const float* __restrict__ x, int in_dim, float* __restrict__ dst, int kpad,
                                int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t row = i / kpad;
    const int col = (int)(i - row * kpad);
    dst[i] = col < in_dim ? x[row * in_dim + col] : 0.f;
}

// ---- PairNorm for any width -------------------------------------------------------------------------
// mean[g] = sum_i weight_i * x[i] over the rows of graph g (weight_i = graph.values, 1/|g| when absent), in
// two fixed-order levels; then per row (x - mean) * rsqrt(mean((x - mean)^2 over the features) + eps).
__global__ __launch_bounds__(256) void pn_partial_kernel(const int32_t* __restrict__ seg, int nch, int width,
                                                         const float* __restrict__ x,
                                                         const float* __restrict__ row_weight,
                                                         float* __restrict__ partial) {
    const int g = blockIdx.x, ch = blockIdx.y;
    const int r0 = seg[g], r1 = seg[g + 1];
    const int per = (r1 - r0 + nch - 1) / nch;
    const int a = r0 + ch * per, b = min(a + per, r1);
    const float unit = r1 > r0 ? 1.f / (float)(r1 - r0) : 0.f;
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float s = 0.f;
        for (int r = a; r < b; ++r) s += (row_weight ? row_weight[r] : unit) * x[(size_t)r * width + c];
        partial[((size_t)g * nch + ch) * width + c] = s;
    }
}

__global__ __launch_bounds__(256) void pn_mean_kernel(int nch, int width, const float* __restrict__ partial,
                                                      float* __restrict__ mean) {
    const int g = blockIdx.x;
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float s = 0.f;
        for (int ch = 0; ch < nch; ++ch) s += partial[((size_t)g * nch + ch) * width + c];
        mean[(size_t)g * width + c] = s;
    }
}

// one wavefront per row
__global__ __launch_bounds__(256) void pn_apply_kernel(const int32_t* __restrict__ seg, int n_graphs, int rows,
                                                       int width, const float* __restrict__ x,
                                                       const float* __restrict__ mean, float eps,
                                                       float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int g = 0;
    if (mean != nullptr) {   // graph of the row: last g with seg[g] <= row
        int lo = 0, hi = n_graphs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (seg[mid] <= row) lo = mid; else hi = mid - 1;
        }
        g = lo;
    }
    const float* xr = x + (size_t)row * width;
    const float* mr = mean != nullptr ? mean + (size_t)g * width : nullptr;
    float ss = 0.f;
    for (int c = lane; c < width; c += 64) {
        const float v = xr[c] - (mr ? mr[c] : 0.f);
        ss += v * v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    const float rs = 1.f / sqrtf(ss / (float)width + eps);
    float* yr = y + (size_t)row * width;
    for (int c = lane; c < width; c += 64) yr[c] = (xr[c] - (mr ? mr[c] : 0.f)) * rs;
}

}  // namespace dsat

using namespace dsat;

struct dsat_mlp {
    int device = -1;
    std::vector<PackedLayer> layers;   // wt / bias only
    float* slab = nullptr;
    int max_hpad = 0;
};

extern "C" int dsat_mlp_create(int32_t n_layers, const int32_t* dims, const float* blob, int64_t n_floats,
                               dsat_mlp** out) {
    DSAT_REQUIRE(out != nullptr, DSAT_E_INVALID, "dsat_mlp_create: null out");
    *out = nullptr;
    DSAT_REQUIRE(n_layers >= 1 && n_layers <= 16 && dims != nullptr && blob != nullptr, DSAT_E_INVALID,
                 "dsat_mlp_create: need 1..16 layers, dims and a weight blob");
    int64_t need = 0;
    for (int l = 0; l <= n_layers; ++l)
        DSAT_REQUIRE(dims[l] >= 1 && dims[l] <= 8192, DSAT_E_UNSUPPORTED, "dsat_mlp_create: widths must be in 1..8192");
    for (int l = 0; l < n_layers; ++l) need += (int64_t)dims[l] * dims[l + 1] + dims[l + 1];
    DSAT_REQUIRE(n_floats == need, DSAT_E_INVALID,
                 "dsat_mlp_create: blob has " + std::to_string(n_floats) + " floats, the layers need " + std::to_string(need));
    std::vector<float> host((size_t)need);
    DSAT_HIP(hipMemcpy(host.data(), blob, (size_t)need * sizeof(float), hipMemcpyDefault));
    for (int64_t i = 0; i < need; ++i)
        DSAT_REQUIRE(std::isfinite(host[i]), DSAT_E_INVALID, "dsat_mlp_create: blob holds a non-finite value");
    std::unique_ptr<dsat_mlp> m(new dsat_mlp());
    DSAT_HIP(hipGetDevice(&m->device));
    m->layers.resize(n_layers);
    size_t total = 0;
    for (int l = 0; l < n_layers; ++l) {
        PackedLayer& pl = m->layers[l];
        pl.in_dim = dims[l]; pl.out_dim = pl.out_ext = dims[l + 1];
        pl.kpad = l == 0 ? up_to(dims[0], 16) : m->layers[l - 1].npad;   // a hidden tile is read at its padded width
        pl.npad = up_to(dims[l + 1], 32);
        total += (size_t)pl.npad * pl.kpad + pl.npad;
        if (l + 1 < n_layers) m->max_hpad = std::max(m->max_hpad, pl.npad);
    }
    std::vector<float> img(total, 0.f);
    size_t off = 0;
    const float* p = host.data();
    std::vector<size_t> wt_at(n_layers), b_at(n_layers);
    for (int l = 0; l < n_layers; ++l) {
        const PackedLayer& pl = m->layers[l];
        const float* kernel = p;
        const float* bias = p + (size_t)pl.in_dim * pl.out_dim;
        p = bias + pl.out_dim;
        wt_at[l] = off;
        for (int k = 0; k < pl.in_dim; ++k)
            for (int n = 0; n < pl.out_dim; ++n) img[off + (size_t)n * pl.kpad + k] = kernel[(size_t)k * pl.out_dim + n];
        off += (size_t)pl.npad * pl.kpad;
        b_at[l] = off;
        for (int n = 0; n < pl.out_dim; ++n) img[off + n] = bias[n];
        off += pl.npad;
    }
    DSAT_HIP(hipMalloc((void**)&m->slab, total * sizeof(float)));
    hipError_t e = hipMemcpy(m->slab, img.data(), total * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(m->slab);
        return hip_fail(e, "dsat_mlp_create upload", __FILE__, __LINE__);
    }
    for (int l = 0; l < n_layers; ++l) {
        m->layers[l].wt = m->slab + wt_at[l];
        m->layers[l].bias = m->slab + b_at[l];
    }
    *out = m.release();
    return DSAT_OK;
}

extern "C" void dsat_mlp_destroy(dsat_mlp* m) {
    if (!m) return;
    if (m->slab) (void)hipFree(m->slab);
    delete m;
}

extern "C" int64_t dsat_mlp_workspace_floats(const dsat_mlp* m, int32_t rows) {
    if (m == nullptr || rows < 0) return -1;
    return (int64_t)rows * (m->layers[0].kpad + 2 * (int64_t)m->max_hpad) + 64;
}

extern "C" int dsat_mlp_forward(const dsat_mlp* m, const float* x, int32_t rows, float* y, float* workspace,
                                void* stream) {
    DSAT_REQUIRE(m && x && y && workspace, DSAT_E_INVALID, "dsat_mlp_forward: null argument");
    DSAT_REQUIRE(rows >= 0, DSAT_E_INVALID, "dsat_mlp_forward: negative rows");
    int cur = -1;
    DSAT_HIP(hipGetDevice(&cur));
    DSAT_REQUIRE(cur == m->device, DSAT_E_STATE, "dsat_mlp_forward: the object lives on another HIP device than the current one");
    if (rows == 0) return DSAT_OK;
    hipStream_t st = (hipStream_t)stream;
    const int kpad0 = m->layers[0].kpad;
    float* in0 = workspace;
    float* hid[2] = {workspace + (size_t)rows * kpad0, workspace + (size_t)rows * (kpad0 + m->max_hpad)};
    const int64_t n = (int64_t)rows * kpad0;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, m->layers[0].in_dim,
                       in0, kpad0, n);
    const float* src = in0;
    int ld = kpad0;
    const int nl = (int)m->layers.size();
    for (int l = 0; l < nl; ++l) {
        const PackedLayer& pl = m->layers[l];
        const bool last = l == nl - 1;
        GemmArgs a;
        memset(&a, 0, sizeof(a));
        a.Wt = pl.wt; a.bias = pl.bias; a.kpad = pl.kpad; a.npad = pl.npad;
        a.M = rows; a.act = last ? 0 : 1;                  // model/mlp.py:24: hidden Dense(activation), last one linear
        a.split = 1 << 30;
        if (last) { a.out0 = a.out1 = y; a.ldo0 = a.ldo1 = pl.out_dim; a.nstore0 = pl.out_dim; }
        else      { a.out0 = a.out1 = hid[l & 1]; a.ldo0 = a.ldo1 = pl.npad; a.nstore0 = pl.npad; }   // padding columns: 0
        a.A[0] = src; a.lda[0] = ld; a.kchunks[0] = pl.kpad / 16; a.nblocks = 1;
        launch_gemm(a, nullptr, st);
        src = hid[l & 1];
        ld = pl.npad;
    }
    DSAT_HIP(hipGetLastError());
    return DSAT_OK;
}

static int pn_chunks(int rows, int n_graphs) {
    const int per_graph = rows / std::max(n_graphs, 1);
    return std::min(64, std::max(1, per_graph / 128));
}

extern "C" int64_t dsat_pair_norm_workspace_floats(int32_t rows, int32_t width, int32_t n_graphs) {
    if (rows < 0 || width < 1 || n_graphs < 0) return -1;
    const int g = std::max(n_graphs, 1);
    return (int64_t)(g + 1 + 63) / 64 * 64 + (int64_t)g * (pn_chunks(rows, g) + 1) * width + 64;
}

extern "C" int dsat_pair_norm(const float* x, int32_t rows, int32_t width, const int32_t* seg, int32_t n_graphs,
                              const float* row_weight, float epsilon, int32_t subtract_mean, float* y,
                              float* workspace, void* stream) {
    DSAT_REQUIRE(x && y, DSAT_E_INVALID, "dsat_pair_norm: null argument");
    DSAT_REQUIRE(rows >= 0 && width >= 1, DSAT_E_INVALID, "dsat_pair_norm: need rows >= 0 and width >= 1");
    DSAT_REQUIRE(n_graphs >= 0 && (n_graphs == 0 || seg != nullptr), DSAT_E_INVALID, "dsat_pair_norm: n_graphs > 0 needs seg");
    if (rows == 0) return DSAT_OK;
    hipStream_t st = (hipStream_t)stream;
    const int G = std::max(n_graphs, 1);
    std::vector<int32_t> one{0, rows};
    const int32_t* hseg = n_graphs > 0 ? seg : one.data();   // graph=None: one graph (normalization.py:55-57)
    DSAT_REQUIRE(hseg[0] == 0 && hseg[G] == rows, DSAT_E_INVALID, "dsat_pair_norm: seg must run from 0 to rows");
    for (int g = 0; g < G; ++g)
        DSAT_REQUIRE(hseg[g + 1] >= hseg[g], DSAT_E_INVALID, "dsat_pair_norm: seg must be non-decreasing");
    const float* mean = nullptr;
    int32_t* dseg = nullptr;
    if (subtract_mean) {
        DSAT_REQUIRE(workspace != nullptr, DSAT_E_INVALID, "dsat_pair_norm: subtract_mean needs the workspace");
        dseg = reinterpret_cast<int32_t*>(workspace);
        const int nch = pn_chunks(rows, G);
        float* partial = workspace + (size_t)(G + 1 + 63) / 64 * 64;
        float* mean_w = partial + (size_t)G * nch * width;
        // the table is host memory of the caller's: copied before the call returns (the one host wait of this operator)
        DSAT_HIP(hipMemcpyAsync(dseg, hseg, (size_t)(G + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
        DSAT_HIP(hipStreamSynchronize(st));
        hipLaunchKernelGGL(pn_partial_kernel, dim3(G, nch), dim3(256), 0, st, dseg, nch, (int)width, x, row_weight, partial);
        hipLaunchKernelGGL(pn_mean_kernel, dim3(G), dim3(256), 0, st, nch, (int)width, partial, mean_w);
        mean = mean_w;
    }
    hipLaunchKernelGGL(pn_apply_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, dseg, G, (int)rows, (int)width, x, mean,
                       epsilon, y);
    DSAT_HIP(hipGetLastError());
    return DSAT_OK;
}
