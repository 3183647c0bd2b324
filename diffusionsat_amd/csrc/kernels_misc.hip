// PairNorm, logit-map selection, SAT checks, RNG/aux inputs and the sampler's
// elementwise steps.  All reductions have a fixed order (no float atomics), so
// results are bitwise reproducible run to run.
#include "dsat_internal.h"
#include "philox.h"

namespace dsat {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------- PairNorm
// layers/normalization.py:43-71 (subtract_mean=True): per-graph column mean, then
// per-row RMS over the 128 features.  Graph rows are contiguous: seg[g]..seg[g+1].
// Pass 1: partial column sums, grid (G, nch); pass 2 (apply) finishes the mean.
int colsum_chunks(int G) {
    int nch = (768 + G - 1) / (G > 0 ? G : 1);
    if (nch < 1) nch = 1;
    if (nch > 16) nch = 16;
    return nch;
}

__global__ __launch_bounds__(256) void colsum_kernel(const int32_t* __restrict__ seg, int nch,
                                                      const float* __restrict__ x,
                                                      float* __restrict__ partial,
                                                      const int* __restrict__ done) {
    if (done != nullptr && *done != 0) return;
    __shared__ float4 red[8][32];
    const int g = blockIdx.x, ch = blockIdx.y;
    const int r0 = seg[g], r1 = seg[g + 1];
    const int len = r1 - r0;
    const int per = (len + nch - 1) / nch;
    const int a = r0 + ch * per;
    const int b = min(a + per, r1);
    const int sub = threadIdx.x & 31, hw = threadIdx.x >> 5;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = a + hw; r < b; r += 8) {
        const float4 v = ld4(x + (size_t)r * F + sub * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    red[hw][sub] = s;
    __syncthreads();
    if (hw == 0) {
        float4 t = red[0][sub];
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            const float4 u = red[i][sub];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        st4(partial + ((size_t)g * nch + ch) * F + sub * 4, t);
    }
}

// v = scale * PairNorm(x) + keep * old          (query_sat.py:265-266, :279-280)
// y_logit = v ; y_carry = carry ? v*0.2 + v*0.8 : v   (:347-348, forward value of the
// stop_gradient mix, applied to the state that enters the next round)
__global__ __launch_bounds__(256) void pairnorm_apply_kernel(
    float inv_f, const int32_t* __restrict__ gid, const int32_t* __restrict__ seg, int rows, int nch,
    const float* __restrict__ x, const float* __restrict__ partial, const float* __restrict__ old,
    float scale, float keep, int carry, float* __restrict__ y_carry, float* __restrict__ y_logit,
    const int* __restrict__ done) {
    if (done != nullptr && *done != 0) return;
    const int sub = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    const int g = gid[row];
    const float inv_cnt = 1.f / (float)(seg[g + 1] - seg[g]);
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ch = 0; ch < nch; ++ch) {
        const float4 p = ld4(partial + ((size_t)g * nch + ch) * F + sub * 4);
        m.x += p.x; m.y += p.y; m.z += p.z; m.w += p.w;
    }
    float4 v = ld4(x + (size_t)row * F + sub * 4);
    v.x -= m.x * inv_cnt; v.y -= m.y * inv_cnt; v.z -= m.z * inv_cnt; v.w -= m.w * inv_cnt;
    float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);  // within the 32-lane half
    const float rs = 1.f / sqrtf(ss * inv_f + 1e-6f);
    float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (old != nullptr) o4 = ld4(old + (size_t)row * F + sub * 4);
    float4 r;
    r.x = v.x * rs * scale + keep * o4.x; r.y = v.y * rs * scale + keep * o4.y;
    r.z = v.z * rs * scale + keep * o4.z; r.w = v.w * rs * scale + keep * o4.w;
    if (y_logit != nullptr) st4(y_logit + (size_t)row * F + sub * 4, r);
    if (carry) {
        r.x = __fadd_rn(__fmul_rn(r.x, 0.2f), __fmul_rn(r.x, 0.8f));
        r.y = __fadd_rn(__fmul_rn(r.y, 0.2f), __fmul_rn(r.y, 0.8f));
        r.z = __fadd_rn(__fmul_rn(r.z, 0.2f), __fmul_rn(r.z, 0.8f));
        r.w = __fadd_rn(__fmul_rn(r.w, 0.2f), __fmul_rn(r.w, 0.8f));
    }
    st4(y_carry + (size_t)row * F + sub * 4, r);
}

void launch_colsum(const int32_t* seg, int G, int nch, const float* x, float* partial,
                   const int* done, hipStream_t st) {
    if (G == 0) return;
    hipLaunchKernelGGL(colsum_kernel, dim3(G, nch), dim3(256), 0, st, seg, nch, x, partial, done);
}

void launch_pairnorm_apply(float inv_f, const int32_t* gid, const int32_t* seg, int rows, int nch, const float* x,
                           const float* partial, const float* old, float scale, float keep,
                           int carry, float* y_carry, float* y_logit, const int* done,
                           hipStream_t st) {
    if (rows == 0) return;
    hipLaunchKernelGGL(pairnorm_apply_kernel, dim3((rows + 7) / 8), dim3(256), 0, st, inv_f, gid, seg, rows,
                       nch, x, partial, old, scale, keep, carry, y_carry, y_logit, done);
}

// ---------------------------------------------------------------- fill
__global__ void fill_kernel(float* p, float v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
void launch_fill(float* p, float v, int64_t n, hipStream_t st) {
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, v, n);
}

// [rows][128] rows with `v` in the first `cols` columns and zero in the padding
__global__ void fill_cols_kernel(float* p, float v, int64_t n, int cols) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (int)(i & (F - 1)) < cols ? v : 0.f;
}
void launch_fill_cols(float* p, float v, int64_t rows, int cols, hipStream_t st) {
    const int64_t n = rows * F;
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, v, n, cols);
}

// ---------------------------------------------------------------- control block
__global__ void ctl_set_kernel(Ctl* dst, Ctl val) { *dst = val; }
void launch_ctl_set(Ctl* dst, const Ctl& val, hipStream_t st) {
    hipLaunchKernelGGL(ctl_set_kernel, dim3(1), dim3(1), 0, st, dst, val);
}

// ---------------------------------------------------------------- per-step inputs
// aux[r][n][0..15] = [normal(4) | noisy_num(2) | noise_scale | denoised(2)=0 | 0 x7]
// = the 9 non-state columns of v1 (query_sat.py:214-219, :239) for every round.
__global__ __launch_bounds__(256) void aux_kernel(float* __restrict__ aux, int N, int rounds,
                                                   const float* __restrict__ noisy_num,
                                                   const Ctl* __restrict__ ctl) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * rounds) return;
    const int r = (int)(i / N), n = (int)(i % N);
    const int ln = logical_var(ctl, n);      // draws belong to the variable's original position (retirement)
    float z[4];
    if (ctl->inj_normal != nullptr) {
        const int64_t nl = ctl->lid != nullptr ? ctl->normal_step_stride / ((int64_t)4 * rounds) : N;   // full N
        const float4 v = ld4(ctl->inj_normal + ctl->normal_step_stride * ctl->step + ((size_t)r * nl + ln) * 4);
        z[0] = v.x; z[1] = v.y; z[2] = v.z; z[3] = v.w;
    } else {
        u32x4 c = {(uint32_t)ln, (uint32_t)r, ctl->step, (uint32_t)STREAM_NORMAL};
        const u32x4 x = philox4x32_10(c, (uint32_t)ctl->seed, (uint32_t)(ctl->seed >> 32));
        box_muller(x.x, x.y, z[0], z[1]);
        box_muller(x.z, x.w, z[2], z[3]);
    }
    float* o = aux + i * AUX_W;
    st4(o, make_float4(z[0], z[1], z[2], z[3]));
    float d0 = 0.f, d1 = 0.f;
    if (ctl->denoised != nullptr && ctl->step >= ctl->denoised_from_step) { d0 = ctl->denoised[ln]; d1 = 1.f - d0; }   // query_sat.py:218
    st4(o + 4, make_float4(noisy_num[2 * n], noisy_num[2 * n + 1], ctl->noise_scale, d0));
    st4(o + 8, make_float4(d1, 0.f, 0.f, 0.f));
    st4(o + 12, make_float4(0.f, 0.f, 0.f, 0.f));
}
void launch_aux(float* aux, int N, int rounds, const float* noisy_num, const Ctl* ctl, hipStream_t st) {
    const int64_t n = (int64_t)N * rounds;
    if (n == 0) return;
    hipLaunchKernelGGL(aux_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, aux, N, rounds,
                       noisy_num, ctl);
}

// labels of query_sat.py:145 (random {0,1} when labels=None), injected or Philox
__global__ void labels_kernel(int32_t* __restrict__ labels, int N, const Ctl* __restrict__ ctl) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (ctl->inj_labels != nullptr) {
        labels[n] = ctl->inj_labels[ctl->labels_step_stride * ctl->step + n];
    } else {
        u32x4 c = {(uint32_t)n, 0u, ctl->step, (uint32_t)STREAM_LABELS};
        labels[n] = (int32_t)(philox4x32_10(c, (uint32_t)ctl->seed, (uint32_t)(ctl->seed >> 32)).x & 1u);
    }
}
void launch_labels(int32_t* labels, int N, const Ctl* ctl, hipStream_t st) {
    if (N == 0) return;
    hipLaunchKernelGGL(labels_kernel, dim3((N + 255) / 256), dim3(256), 0, st, labels, N, ctl);
}

// standalone draws for dsat_op_rng
__global__ void rng_normal_kernel(float* out, int N, int rounds, uint64_t seed, uint32_t step) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * rounds) return;
    u32x4 c = {(uint32_t)(i % N), (uint32_t)(i / N), step, (uint32_t)STREAM_NORMAL};
    const u32x4 x = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float z0, z1, z2, z3;
    box_muller(x.x, x.y, z0, z1);
    box_muller(x.z, x.w, z2, z3);
    st4(out + i * 4, make_float4(z0, z1, z2, z3));
}
__global__ void rng_labels_kernel(int32_t* out, int N, uint64_t seed, uint32_t step) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    u32x4 c = {(uint32_t)n, 0u, step, (uint32_t)STREAM_LABELS};
    out[n] = (int32_t)(philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32)).x & 1u);
}
__device__ __forceinline__ float philox_uniform(uint32_t n, uint32_t step, uint64_t seed) {
    u32x4 c = {n, 0u, step, (uint32_t)STREAM_UNIFORM};
    return (float)(philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32)).x >> 8) * 5.9604644775390625e-8f;
}
__global__ void rng_uniform_kernel(float* out, int N, uint64_t seed, uint32_t step) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    out[n] = philox_uniform((uint32_t)n, step, seed);
}
void launch_rng(float* normal, int32_t* labels, float* uniform, int N, int rounds, uint64_t seed,
                uint32_t step, hipStream_t st) {
    if (N == 0) return;
    if (normal != nullptr && rounds > 0) {
        const int64_t n = (int64_t)N * rounds;
        hipLaunchKernelGGL(rng_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, normal,
                           N, rounds, seed, step);
    }
    if (labels != nullptr)
        hipLaunchKernelGGL(rng_labels_kernel, dim3((N + 255) / 256), dim3(256), 0, st, labels, N, seed, step);
    if (uniform != nullptr)
        hipLaunchKernelGGL(rng_uniform_kernel, dim3((N + 255) / 256), dim3(256), 0, st, uniform, N, seed, step);
}

// ---------------------------------------------------------------- logit-map selection
// query_sat.py:289-292 (train_loss = normalised Bernoulli KL, :40-53), :317-320 argmin
// per graph, :328-329 gather.  One workgroup per graph; fixed-order tree reduction.
__device__ __forceinline__ float bern_kl(float a, float b) {
    return a * (logf(a) - logf(b)) + (1.f - a) * (log1pf(-a) - log1pf(-b));
}

__global__ __launch_bounds__(256) void select_kernel(
    const int32_t* __restrict__ seg, const float* __restrict__ logits,
    const int32_t* __restrict__ labels, Ctl* __restrict__ ctl, float* __restrict__ pgl,
    int32_t* __restrict__ best, float* __restrict__ out_logits, uint8_t* __restrict__ assign,
    int32_t* __restrict__ graph_unsat, const int32_t* __restrict__ cseg, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ c_lit, int N) {
    if (ctl->done != 0) return;
    __shared__ float red[256][LOGIT_MAPS + 1];
    __shared__ int s_best;
    const int g = blockIdx.x;
    const int v0 = seg[g], v1 = seg[g + 1];
    const float t = powf(ctl->noise_scale, 0.5f);
    const float tl = fminf(t + 0.01f, 1.f);
    const float norm = bern_kl(0.f * (1.f - tl) + tl * 0.5f, 0.f * (1.f - 1.f) + 1.f * 0.5f) + 1e-4f;
    float acc[LOGIT_MAPS];
#pragma unroll
    for (int k = 0; k < LOGIT_MAPS; ++k) acc[k] = 0.f;
    const bool unsup = ctl->unsupervised != 0;   // QuerySAT(supervised=False), model/query_sat.py:293-309
    if (unsup)
        for (int c = cseg[g] + threadIdx.x; c < cseg[g + 1]; c += 256) unsup_clause_loss8(logits, rowptr, c_lit, c, N, acc);
    for (int v = v0 + threadIdx.x; v < v1 && !unsup; v += 256) {
        const float a = (float)labels[v] * (1.f - tl) + tl * 0.5f;
        const float4 l0 = ld4(logits + (size_t)v * LOGIT_MAPS);
        const float4 l1 = ld4(logits + (size_t)v * LOGIT_MAPS + 4);
        const float lg[LOGIT_MAPS] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
        for (int k = 0; k < LOGIT_MAPS; ++k) {
            const float b = sigmoid_f(lg[k]) * (1.f - t) + t * 0.5f;
            acc[k] += bern_kl(a, b) / norm;
        }
    }
#pragma unroll
    for (int k = 0; k < LOGIT_MAPS; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < LOGIT_MAPS; ++k) red[threadIdx.x][k] += red[threadIdx.x + s][k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float inv = 1.f / (float)(v1 - v0);
        int bi = 0;
        float bv = unsup ? unsup_graph_loss(red[0][0]) : red[0][0] * inv;
        pgl[g * LOGIT_MAPS] = bv;
        for (int k = 1; k < LOGIT_MAPS; ++k) {
            const float x = unsup ? unsup_graph_loss(red[0][k]) : red[0][k] * inv;
            pgl[g * LOGIT_MAPS + k] = x;
            if (x < bv) { bv = x; bi = k; }   // first minimum wins, as TF's CPU ArgMin
        }
        best[g] = bi;
        s_best = bi;
        graph_unsat[g] = 0;
        if (g == 0) { ctl->unsat_total = 0; ctl->ticket = 0; }
    }
    __syncthreads();
    const int bi = s_best;
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
        const float x = logits[(size_t)v * LOGIT_MAPS + bi];
        out_logits[v] = x;
        assign[v] = (uint8_t)rintf(sigmoid_f(x));   // tf.round(tf.sigmoid(.)), half-to-even
    }
}
void launch_select(const dsat_graph* g, const float* logits, const int32_t* labels, Ctl* ctl, float* pgl,
                   int32_t* best, float* out_logits, uint8_t* assign, int32_t* graph_unsat,
                   hipStream_t st) {
    if (g->G == 0) return;
    hipLaunchKernelGGL(select_kernel, dim3(g->G), dim3(256), 0, st, g->var_seg, logits, labels, ctl, pgl,
                       best, out_logits, assign, graph_unsat, g->cl_seg, g->c_rowptr, g->c_lit, g->N);
}

// ---------------------------------------------------------------- SAT check
// is_batch_sat (utils/sat.py:118-124) and SATAccuracyTF.accuracy's per-graph count
// (metrics/sat_metrics.py:60-85) from the rounded assignment.  Integer atomics only.
// The last workgroup to finish publishes the verdict: done=1 when every clause of the
// batch is satisfied (early exit, query_sat.py:330-338), and steps_taken=round.
__global__ __launch_bounds__(256) void sat_check_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ c_lit,
    const int32_t* __restrict__ cl_gid, const uint8_t* __restrict__ assign,
    int32_t* __restrict__ graph_unsat, Ctl* __restrict__ ctl, int N, int C, int round, int set_done) {
    if (ctl->done != 0) return;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    bool unsat = false;
    if (c < C) {
        bool sat = false;
        for (int e = rowptr[c]; e < rowptr[c + 1]; ++e) {
            const int l = c_lit[e];
            const bool val = l < N ? (assign[l] != 0) : (assign[l - N] == 0);
            sat = sat || val;
        }
        unsat = !sat;
        if (unsat) atomicAdd(&graph_unsat[cl_gid[c]], 1);
    }
    const unsigned long long m = __ballot(unsat);
    if ((threadIdx.x & 63) == 0 && m != 0ull) atomicAdd(&ctl->unsat_total, (int)__popcll(m));
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = atomicAdd(&ctl->ticket, 1);
        if (t == (int)gridDim.x - 1) {
            const int total = atomicAdd(&ctl->unsat_total, 0);
            ctl->steps_taken = round;
            if (set_done && total == 0) ctl->done = 1;
        }
    }
}
void launch_sat_check(const dsat_graph* g, const uint8_t* assign, int32_t* graph_unsat, Ctl* ctl,
                      int round, int set_done, hipStream_t st) {
    const int blocks = g->C > 0 ? (g->C + 255) / 256 : 1;
    hipLaunchKernelGGL(sat_check_kernel, dim3(blocks), dim3(256), 0, st, g->c_rowptr, g->c_lit, g->cl_gid,
                       assign, graph_unsat, ctl, g->N, g->C, round, set_done);
}

__global__ void copy_if_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n,
                               const int* __restrict__ done) {
    if (done != nullptr && *done != 0) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
void launch_copy_if(const float* src, float* dst, int64_t n, const int* done, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(copy_if_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n, done);
}

// ---------------------------------------------------------------- sampler steps
// diffusion() of satuniformity/DiffusionSampler.py:78-191, one kernel per stage.
__global__ void sample_init_kernel(float* x0, float* x1, int N, int32_t* fixed_step, uint8_t* cum_acc,
                                   int G, Ctl* ctl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { x0[i] = 0.5f; x1[i] = 0.5f; }                      // :86
    if (i < G) { fixed_step[i] = -1; cum_acc[i] = 0; }              // :87-89
    if (i == 0) ctl->step = 0;
}

// :106-109  noise_scale = 1 - t/N ; x_noisy = randomized_rounding_tf(x) ; x = x_noisy
__global__ void sample_pre_kernel(float* __restrict__ x0, float* __restrict__ x1,
                                  float* __restrict__ noisy, int N, Ctl* __restrict__ ctl, int rounds) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = ctl->step;
    if (n == 0) {
        ctl->noise_scale = (float)(1.0 - (double)t / (double)ctl->n_steps);
        ctl->done = 0;
        ctl->steps_taken = rounds - 1;
    }
    if (n >= N) return;
    const float u = ctl->inj_uniform != nullptr ? ctl->inj_uniform[ctl->uniform_step_stride * t + n]
                                                : philox_uniform((uint32_t)n, t, ctl->seed);
    const float r = floorf(x0[n] + u);                              // query_sat.py:56-58
    noisy[2 * n] = r;
    noisy[2 * n + 1] = 1.f - r;
    x0[n] = r;
    x1[n] = 1.f - r;
}

// :115-129 + :154-170 per variable: p = sigmoid(prediction); theoretic reverse step;
// remember round(p) for graphs solved now for the first time.
__global__ void sample_post_kernel(float* __restrict__ x0, float* __restrict__ x1,
                                   const float* __restrict__ pred, const int32_t* __restrict__ var_gid,
                                   const int32_t* __restrict__ graph_unsat,
                                   const int32_t* __restrict__ fixed_step, float* __restrict__ x_fixed,
                                   int N, const Ctl* __restrict__ ctl) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t t = ctl->step;
    const double nsd = 1.0 - (double)t / (double)ctl->n_steps;
    const double inc = 1.0 / (double)ctl->n_steps;
    const float logit = pred[n];
    const float p = sigmoid_f(logit);
    if (ctl->pred_steps_out != nullptr) ctl->pred_steps_out[(size_t)t * N + n] = logit;
    if (ctl->denoised_wr != nullptr) ctl->denoised_wr[n] = p;
    // reverse_distribution_step_theoretic(x, [1-p, p], noise_scale, 1/N)   (:29-37)
    const float t1 = powf((float)nsd, 0.5f);
    const double rem = nsd - inc;
    const float t2 = powf((float)(rem > 0.0 ? rem : 0.0), 0.5f);
    const float xn0 = (1.f - p) * (1.f - t1) + t1 * 0.5f;
    const float xn1 = p * (1.f - t1) + t1 * 0.5f;
    const float alpha = (1.f - t1) / (1.f - t2);
    const float ti = 1.f - alpha;
    const float u0 = (x0[n] * (1.f - ti) + ti * 0.5f) * xn0;
    const float u1 = (x1[n] * (1.f - ti) + ti * 0.5f) * xn1;
    const float den = (u0 + u1) + 1e-8f;
    x0[n] = u0 / den;
    x1[n] = u1 / den;
    const int g = var_gid[n];
    if (fixed_step[g] < 0 && graph_unsat[g] == 0) x_fixed[n] = rintf(p);   // :154, :162-168
}

__global__ void sample_post_graph_kernel(int32_t* __restrict__ fixed_step, uint8_t* __restrict__ cum_acc,
                                         const int32_t* __restrict__ graph_unsat, int G, Ctl* ctl) {
    const uint32_t t = ctl->step;   // single workgroup: every thread reads t before it changes
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const bool sat = graph_unsat[g] == 0;
        if (sat) cum_acc[g] = 1;                                    // :130
        if (sat && fixed_step[g] < 0) fixed_step[g] = (int32_t)t;   // :169
    }
    __syncthreads();
    if (threadIdx.x == 0) ctl->step = t + 1;
}

// :181-184 predictions = round(p_last), overwritten by the frozen solutions
__global__ void sample_final_kernel(const float* __restrict__ pred, const int32_t* __restrict__ var_gid,
                                    const int32_t* __restrict__ fixed_step,
                                    const float* __restrict__ x_fixed, uint8_t* __restrict__ assignment,
                                    int N, const uint8_t* __restrict__ cum_acc,
                                    uint8_t* __restrict__ solved, int G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        const float v = fixed_step[var_gid[i]] >= 0 ? x_fixed[i] : rintf(sigmoid_f(pred[i]));
        assignment[i] = (uint8_t)v;
    }
    if (i < G && solved != nullptr) solved[i] = cum_acc[i];
}

// ---------------------------------------------------------------- sampler with early retirement
// The same steps for a batch of identical copies whose solved copies leave the batch between
// denoising steps (satuniformity/DiffusionSampler.py:154-170 freezes them; their later predictions are
// never used, :181-184).  The active copies are slots 0 .. n_act-1; lid[slot] = original position.
// x0 / x1 / noisy are per slot; fixed_step, cum_acc, x_fixed, denoised are per ORIGINAL position.
__global__ void sample_pre_retire_kernel(float* __restrict__ x0, float* __restrict__ x1,
                                         float* __restrict__ noisy, int N, Ctl* __restrict__ ctl, int rounds) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = ctl->step;
    if (n == 0) {
        ctl->noise_scale = (float)(1.0 - (double)t / (double)ctl->n_steps);
        ctl->done = 0;
        ctl->steps_taken = rounds - 1;
    }
    if (n >= N) return;
    const int ln = logical_var(ctl, n);
    const float u = ctl->inj_uniform != nullptr ? ctl->inj_uniform[ctl->uniform_step_stride * t + ln]
                                                : philox_uniform((uint32_t)ln, t, ctl->seed);
    const float r = floorf(x0[n] + u);
    noisy[2 * n] = r;
    noisy[2 * n + 1] = 1.f - r;
    x0[n] = r;
    x1[n] = 1.f - r;
}

__global__ void sample_post_retire_kernel(float* __restrict__ x0, float* __restrict__ x1,
                                          const float* __restrict__ pred, const int32_t* __restrict__ graph_unsat,
                                          const int32_t* __restrict__ fixed_step, float* __restrict__ x_fixed,
                                          int N, int64_t n_full, const Ctl* __restrict__ ctl) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t t = ctl->step;
    const double nsd = 1.0 - (double)t / (double)ctl->n_steps;
    const double inc = 1.0 / (double)ctl->n_steps;
    const float logit = pred[n];
    const float p = sigmoid_f(logit);
    const int ln = logical_var(ctl, n);
    if (ctl->pred_steps_out != nullptr) ctl->pred_steps_out[(size_t)t * n_full + ln] = logit;
    if (ctl->denoised_wr != nullptr) ctl->denoised_wr[ln] = p;
    const float t1 = powf((float)nsd, 0.5f);
    const double rem = nsd - inc;
    const float t2 = powf((float)(rem > 0.0 ? rem : 0.0), 0.5f);
    const float xn0 = (1.f - p) * (1.f - t1) + t1 * 0.5f;
    const float xn1 = p * (1.f - t1) + t1 * 0.5f;
    const float alpha = (1.f - t1) / (1.f - t2);
    const float ti = 1.f - alpha;
    const float u0 = (x0[n] * (1.f - ti) + ti * 0.5f) * xn0;
    const float u1 = (x1[n] * (1.f - ti) + ti * 0.5f) * xn1;
    const float den = (u0 + u1) + 1e-8f;
    x0[n] = u0 / den;
    x1[n] = u1 / den;
    const int g = n / ctl->n_per;
    if (fixed_step[ctl->lid[g]] < 0 && graph_unsat[g] == 0) x_fixed[ln] = rintf(p);
}

// One workgroup: record the copies solved in this step, then plan the next step's batch: unsolved
// slots keep their order and move to the front.  move_src[new slot] = old slot; lid is rewritten.
__global__ __launch_bounds__(1024) void retire_plan_kernel(int32_t* __restrict__ fixed_step,
                                                           uint8_t* __restrict__ cum_acc,
                                                           const int32_t* __restrict__ graph_unsat,
                                                           int32_t* __restrict__ lid, int32_t* __restrict__ move_src,
                                                           int n_act, Ctl* ctl, int32_t* __restrict__ n_act_out,
                                                           int reorder) {
    __shared__ int s_keep[1024];
    __shared__ int s_base;
    const uint32_t t = ctl->step;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_act; c0 += 1024) {
        const int sl = c0 + threadIdx.x;
        int keep = 0, lg = -1;
        if (sl < n_act) {
            lg = lid[sl];
            const bool sat = graph_unsat[sl] == 0;
            if (sat) cum_acc[lg] = 1;
            if (sat && fixed_step[lg] < 0) fixed_step[lg] = (int32_t)t;
            keep = fixed_step[lg] < 0 ? 1 : 0;
        }
        s_keep[threadIdx.x] = keep;
        __syncthreads();
        // exclusive scan of the chunk by thread 0 .. simple and at most 1024 entries per chunk
        if (threadIdx.x == 0) {
            int run = s_base;
            for (int i = 0; i < 1024; ++i) { const int k = s_keep[i]; s_keep[i] = run; run += k; }
            s_base = run;
        }
        __syncthreads();
        if (keep && reorder) move_src[s_keep[threadIdx.x]] = sl;
        __syncthreads();
    }
    if (!reorder) {   // last step: only record what was solved; slots and lid stay as the step ran them
        if (threadIdx.x == 0) { ctl->step = t + 1; *n_act_out = n_act; }
        return;
    }
    // second pass: new lid from the old one through move_src (separate array: slots only move down)
    __syncthreads();
    const int n_new = s_base;
    // lid[new] = lid[move_src[new]] with move_src[new] >= new: ascending order is safe per chunk of 1024
    for (int c0 = 0; c0 < n_new; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        int v = 0;
        if (i < n_new) v = lid[move_src[i]];
        __syncthreads();
        if (i < n_new) lid[i] = v;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ctl->step = t + 1;
        *n_act_out = n_new;
    }
}

// x of the surviving copies to their new slots (other buffer: slots only move down, but in parallel)
__global__ void retire_move_kernel(const float* __restrict__ x0_in, const float* __restrict__ x1_in,
                                   float* __restrict__ x0_out, float* __restrict__ x1_out,
                                   const int32_t* __restrict__ move_src, const int32_t* __restrict__ n_new_p, int n_per) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_new = *n_new_p;
    if (i >= n_new * n_per) return;
    const int g = i / n_per, k = i - g * n_per;
    const int src = move_src[g] * n_per + k;
    x0_out[i] = x0_in[src];
    x1_out[i] = x1_in[src];
}

// predictions = round(p_last) for copies still in the batch, the frozen bits for solved ones (:181-184)
__global__ void sample_final_retire_kernel(const float* __restrict__ pred, const int32_t* __restrict__ lid, int n_act,
                                           int n_per, const int32_t* __restrict__ fixed_step,
                                           const float* __restrict__ x_fixed, uint8_t* __restrict__ assignment,
                                           int n_full, const uint8_t* __restrict__ cum_acc,
                                           uint8_t* __restrict__ solved, int G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_full && fixed_step[i / n_per] >= 0) assignment[i] = (uint8_t)x_fixed[i];
    if (i < n_act * n_per) {
        const int g = i / n_per, lg = lid[g];
        if (fixed_step[lg] < 0) assignment[lg * n_per + (i - g * n_per)] = (uint8_t)rintf(sigmoid_f(pred[i]));
    }
    if (i < G && solved != nullptr) solved[i] = cum_acc[i];
}

void launch_sample_pre_retire(float* x0, float* x1, float* noisy, int N, Ctl* ctl, int rounds, hipStream_t st) {
    hipLaunchKernelGGL(sample_pre_retire_kernel, dim3((N + 255) / 256 > 0 ? (N + 255) / 256 : 1), dim3(256), 0, st,
                       x0, x1, noisy, N, ctl, rounds);
}
void launch_sample_post_retire(float* x0, float* x1, const float* pred, const int32_t* graph_unsat,
                               const int32_t* fixed_step, float* x_fixed, int N, int64_t n_full, const Ctl* ctl,
                               hipStream_t st) {
    hipLaunchKernelGGL(sample_post_retire_kernel, dim3((N + 255) / 256 > 0 ? (N + 255) / 256 : 1), dim3(256), 0, st,
                       x0, x1, pred, graph_unsat, fixed_step, x_fixed, N, n_full, ctl);
}
void launch_retire_plan(int32_t* fixed_step, uint8_t* cum_acc, const int32_t* graph_unsat, int32_t* lid,
                        int32_t* move_src, int n_act, Ctl* ctl, int32_t* n_act_out, int reorder, hipStream_t st) {
    hipLaunchKernelGGL(retire_plan_kernel, dim3(1), dim3(1024), 0, st, fixed_step, cum_acc, graph_unsat, lid, move_src,
                       n_act, ctl, n_act_out, reorder);
}
void launch_retire_move(const float* x0_in, const float* x1_in, float* x0_out, float* x1_out, const int32_t* move_src,
                        const int32_t* n_new_p, int n_act_old, int n_per, hipStream_t st) {
    const int64_t n = (int64_t)n_act_old * n_per;
    if (n <= 0) return;
    hipLaunchKernelGGL(retire_move_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x0_in, x1_in, x0_out,
                       x1_out, move_src, n_new_p, n_per);
}
void launch_sample_final_retire(const float* pred, const int32_t* lid, int n_act, int n_per,
                                const int32_t* fixed_step, const float* x_fixed, uint8_t* assignment, int n_full,
                                const uint8_t* cum_acc, uint8_t* solved, int G, hipStream_t st) {
    const int n = n_full > G ? n_full : G;
    hipLaunchKernelGGL(sample_final_retire_kernel, dim3((n + 255) / 256 > 0 ? (n + 255) / 256 : 1), dim3(256), 0, st,
                       pred, lid, n_act, n_per, fixed_step, x_fixed, assignment, n_full, cum_acc, solved, G);
}

void launch_sample_init(float* x0, float* x1, int N, int32_t* fixed_step, uint8_t* cum_acc, int G,
                        Ctl* ctl, hipStream_t st) {
    const int n = N > G ? N : G;
    hipLaunchKernelGGL(sample_init_kernel, dim3((n + 255) / 256 > 0 ? (n + 255) / 256 : 1), dim3(256), 0, st,
                       x0, x1, N, fixed_step, cum_acc, G, ctl);
}
void launch_sample_pre(float* x0, float* x1, float* noisy, int N, Ctl* ctl, int rounds, hipStream_t st) {
    hipLaunchKernelGGL(sample_pre_kernel, dim3((N + 255) / 256 > 0 ? (N + 255) / 256 : 1), dim3(256), 0, st,
                       x0, x1, noisy, N, ctl, rounds);
}
void launch_sample_post(float* x0, float* x1, const float* pred, const dsat_graph* g,
                        const int32_t* graph_unsat, int32_t* fixed_step, float* x_fixed,
                        uint8_t* cum_acc, Ctl* ctl, hipStream_t st) {
    hipLaunchKernelGGL(sample_post_kernel, dim3((g->N + 255) / 256 > 0 ? (g->N + 255) / 256 : 1), dim3(256),
                       0, st, x0, x1, pred, g->var_gid, graph_unsat, fixed_step, x_fixed, g->N, ctl);
    hipLaunchKernelGGL(sample_post_graph_kernel, dim3(1), dim3(256), 0, st, fixed_step, cum_acc,
                       graph_unsat, g->G, ctl);
}
void launch_sample_final(const float* pred, const dsat_graph* g, const int32_t* fixed_step,
                         const float* x_fixed, uint8_t* assignment, const uint8_t* cum_acc,
                         uint8_t* solved, hipStream_t st) {
    const int n = g->N > g->G ? g->N : g->G;
    hipLaunchKernelGGL(sample_final_kernel, dim3((n + 255) / 256 > 0 ? (n + 255) / 256 : 1), dim3(256), 0,
                       st, pred, g->var_gid, fixed_step, x_fixed, assignment, g->N, cum_acc, solved, g->G);
}

}  // namespace dsat
