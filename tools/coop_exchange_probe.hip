// Bound for "cut the 16-row tiles' weight stream by cooperating CUs" (round-2 verdict, item 2): the price of the
// exchange such a scheme needs, measured on the shape it would have inside round_p1_kernel.
//
// Scheme under test: G = 4 workgroups (one per CU, 1024 threads, equal blockIdx % 8 = one XCD under the observed
// round-robin placement) own 64 rows of lit_mlp (137 -> 512 -> 512 -> 256).  Layer 0 is split by columns, layers 1
// and 2 by K, so after each of those two layers every member holds fp32 partial sums of ALL output columns of the 64
// rows and the group needs a reduce-scatter: member m sends the three 64 x (N/4) blocks it does not own (layer 1:
// 3 x 32 KB, layer 2: 3 x 16 KB) and reads the three blocks the others computed for its own columns.  Each member
// streams 0.71 MB of weight planes instead of 2.85 MB; the MFMA work per CU is unchanged.
//
// The probe runs exactly that traffic pattern around a weight-streaming phase of the right length: 49 groups (196
// workgroups) + 60 workgroups that only stream (roles B / C of the real kernel), every workgroup one per CU:
//   stream 10 k-steps x 16 waves x 3 KiB  ->  publish 96 KB, wait for 3 flags, read 96 KB, add
//   stream  6 k-steps                      ->  publish 48 KB, wait, read 48 KB, add
// in the two publish forms cdna_hip_programming.md (Guideline 16) allows:
//   MODE 0: payload stored write-through (sc1), every storing wave drains, barrier, one lane stores the flag (sc1);
//           the consumer polls the flag with sc1 loads and reads the payload with sc1 loads (no fence anywhere);
//   MODE 1: plain payload stores + agent-scope release fence + flag; consumer: poll, agent-scope acquire fence,
//           plain loads.
// Output: cycles (shader clock) of each phase per workgroup, median and max over the groups, and a check of the sums.
//   coop_exchange_probe [groups = 49] [background workgroups = 60]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idiffusionsat_amd/csrc tools/coop_exchange_probe.hip -o tools/libs/coop_exchange_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NTH = 1024, G = 4, ROWS = 64;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

struct Args {
    const float4* weights;      // 8 MB of anything: the streamed planes
    float* slab;                // [groups][2 layers][src 4][dst 4][64 x 128] fp32 (layer 2 uses 64 x 64 of it)
    unsigned* flags;            // [groups][2][4]
    float* out;                 // [groups][4][64 x 128] reduced layer-1 block of each member (checked on the host)
    long long* stamps;          // [blocks][8]
    int* xcc;                   // [blocks]
    int groups, mode;
};

__device__ __forceinline__ void stream_phase(const float4* w, int ksteps, int wave, int lane, f4& sink) {
    // 3 x 16-byte loads per lane and k-step (the three planes of one 16-column tile), ring of depth 3, as the kernels do
    const float4* base = w + ((size_t)(blockIdx.x * 16 + wave) * 64 * 3 * 64) % (1u << 19) + lane;
    f4 r[3][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[p][j] = *reinterpret_cast<const f4*>(base + (p * 3 + j) * 64);
    for (int s = 0; s < ksteps; s += 3) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int j = 0; j < 3; ++j) sink += r[p][j];
            const int sn = s + p + 3;
#pragma unroll
            for (int j = 0; j < 3; ++j) r[p][j] = *reinterpret_cast<const f4*>(base + ((sn % 60) * 3 + j) * 64);
            // six MFMAs of 16 cycles per row block and k-step, four row blocks: the matrix pipe's share of the step
            __builtin_amdgcn_s_sleep(6);
        }
    }
}

template <int MODE>
__device__ __forceinline__ void exchange(const Args& a, int g, int m, int layer, int cols, unsigned epoch, const f4& mine,
                                         float* own_lds, int tid, long long* st, int si, float* result) {
    // publish: the three 64 x cols blocks for the other members; 16 B per lane and store
    const int n4 = ROWS * cols / 4;                 // float4 per block
    float* base = a.slab + ((size_t)(g * 2 + layer) * G * G) * ROWS * 128;
    for (int d = 0; d < G; ++d) {
        if (d == m) continue;
        float* dst = base + (size_t)(m * G + d) * ROWS * 128;
        for (int i = tid; i < n4; i += NTH) {
            f4 v = mine + (float)(i & 7);
            if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst + 4 * i), "v"(v) : "memory");
            else *reinterpret_cast<f4*>(dst + 4 * i) = v;
        }
    }
    for (int i = tid; i < n4; i += NTH) *reinterpret_cast<f4*>(own_lds + 4 * i) = mine + (float)(i & 7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __hip_atomic_store((gu32*)(a.flags + (g * 2 + layer) * G + m), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) st[si] = clock64();
    // wait for the three peers (one lane per peer polls, bounded)
    if (tid < G && tid != m) {
        gu32* f = (gu32*)(a.flags + (g * 2 + layer) * G + tid);
        for (int spin = 0; spin < (1 << 22); ++spin) {
            if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
    if (tid == 0) st[si + 1] = clock64();
    // reduce in member order (fixed: deterministic), own block from LDS
    for (int i = tid; i < n4; i += NTH) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        f4 v[G];
#pragma unroll
        for (int s = 0; s < G; ++s) {
            const float* src = base + (size_t)(s * G + m) * ROWS * 128 + 4 * i;
            if (s == m) v[s] = *reinterpret_cast<const f4*>(own_lds + 4 * i);
            else if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[s]) : "v"(src) : "memory");
            else v[s] = *reinterpret_cast<const f4*>(src);
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
#pragma unroll
        for (int s = 0; s < G; ++s) acc += v[s];
        if (result != nullptr) *reinterpret_cast<f4*>(result + 4 * i) = acc;
        else *reinterpret_cast<f4*>(own_lds + 4 * i) = acc;
    }
    __syncthreads();
    if (tid == 0) st[si + 2] = clock64();
}

template <int MODE>
__global__ __launch_bounds__(NTH) void probe(Args a, unsigned epoch) {
    __shared__ __attribute__((aligned(16))) float lds[ROWS * 128 + 16384];   // own block + what the real tile would hold
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x;
    long long* st = a.stamps + (size_t)b * 8;
    if (tid == 0) a.xcc[b] = (int)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);   // HW_REG_XCC_ID, 4 bits
    f4 sink = {0.f, 0.f, 0.f, 0.f};
    const int nb_group = 8 * 4 * ((a.groups + 7) / 8);
    if (b >= nb_group) {                      // background: streams for as long as a whole tile
        if (tid == 0) st[0] = clock64();
        stream_phase(a.weights, 48, wave, lane, sink);
        if (sink[0] == 12345.f) a.out[0] = sink[1];
        if (tid == 0) st[7] = clock64();
        return;
    }
    const int xcd = b % 8, j = b / 8, slot = j / 4, m = j % 4, g = slot * 8 + xcd;   // members of a group: equal b % 8
    if (g >= a.groups) return;
    if (tid == 0) st[0] = clock64();
    stream_phase(a.weights, 10, wave, lane, sink);                                 // layers 0 + 1 of a member
    const f4 mine = {1.f + m, 2.f, 3.f, sink[0] == 12345.f ? 1.f : 4.f};
    if (tid == 0) st[1] = clock64();
    exchange<MODE>(a, g, m, 0, 128, epoch, mine, lds, tid, st, 2, a.out + ((size_t)g * G + m) * ROWS * 128);
    stream_phase(a.weights, 6, wave, lane, sink);                                  // layer 2
    if (tid == 0) st[5] = clock64();
    long long dummy[3];
    exchange<MODE>(a, g, m, 1, 64, epoch, mine, lds, tid, dummy, 0, nullptr);
    if (sink[0] == 12345.f) a.out[1] = sink[1];
    if (tid == 0) st[7] = clock64();
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
static int run(int groups, int background) {
    const int nb_group = 8 * 4 * ((groups + 7) / 8), blocks = nb_group + background;
    if (blocks > 256) { printf("more than 256 workgroups: the group members would not all be resident\n"); return 1; }
    Args a{};
    float *w, *slab, *out; unsigned* flags; long long* stamps; int* xcc;
    CK(hipMalloc(&w, 9 << 20)); CK(hipMemset(w, 0, 9 << 20));
    const size_t slab_f = (size_t)groups * 2 * G * G * ROWS * 128;
    CK(hipMalloc(&slab, slab_f * 4)); CK(hipMalloc(&flags, groups * 2 * G * 4 + 64));
    CK(hipMalloc(&out, (size_t)groups * G * ROWS * 128 * 4)); CK(hipMalloc(&stamps, (size_t)blocks * 8 * 8)); CK(hipMalloc(&xcc, blocks * 4));
    a.weights = (const float4*)w; a.slab = slab; a.flags = flags; a.out = out; a.stamps = stamps; a.xcc = xcc; a.groups = groups; a.mode = MODE;
    float ms = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(slab, 0xFF, slab_f * 4));                      // NaNs: a stale read cannot pass for a fresh one
        CK(hipMemset(flags, 0, groups * 2 * G * 4 + 64));
        CK(hipMemset(stamps, 0, (size_t)blocks * 8 * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(NTH), 0, 0, a, (unsigned)(rep + 1));
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<long long> st((size_t)blocks * 8); std::vector<int> xc(blocks); std::vector<float> o((size_t)groups * G * ROWS * 128);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(xc.data(), xcc, blocks * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
    // every member's reduced block must be sum over s of (1 + s, 2, 3, 4) + 4 * (i & 7)
    size_t bad = 0;
    for (int g = 0; g < groups; ++g) for (int m = 0; m < G; ++m) for (int i = 0; i < ROWS * 128 / 4; ++i) {
        const float* p = &o[((size_t)g * G + m) * ROWS * 128 + 4 * i];
        const float k = 4.f * (i & 7);
        if (p[0] != 10.f + k || p[1] != 8.f + k || p[2] != 12.f + k || p[3] != 16.f + k) ++bad;
    }
    int split = 0;
    std::vector<long long> s1, pub, wait, red, tile;
    for (int b = 0; b < nb_group; ++b) {
        const int xcd = b % 8, j = b / 8, slot = j / 4, m = j % 4, g = slot * 8 + xcd;
        if (g >= groups) continue;
        const long long* s = &st[(size_t)b * 8];
        s1.push_back(s[1] - s[0]); pub.push_back(s[2] - s[1]); wait.push_back(s[3] - s[2]); red.push_back(s[4] - s[3]); tile.push_back(s[7] - s[0]);
        if (m == 0) for (int q = 1; q < G; ++q) split += xc[b] != xc[(slot * 4 + q) * 8 + xcd];
    }
    auto med = [](std::vector<long long>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<long long>& v) { return *std::max_element(v.begin(), v.end()); };
    printf("mode %d (%s): %d groups + %d background workgroups, kernel %.1f us; wrong sums %zu; members on another XCD than member 0: %d\n",
           MODE, MODE == 0 ? "sc1 stores, drained, sc1 flag; sc1 loads" : "plain stores + release fence; acquire fence + plain loads",
           groups, background, ms * 1e3, bad, split);
    printf("  cycles per workgroup, median (max): stream 10 k-steps %lld (%lld) | exchange 1 (96 KB out, 96 KB in): publish %lld (%lld), wait for peers %lld (%lld), read + add %lld (%lld) | whole pseudo-tile (16 k-steps, 2 exchanges) %lld (%lld)\n",
           med(s1), mx(s1), med(pub), mx(pub), med(wait), mx(wait), med(red), mx(red), med(tile), mx(tile));
    return 0;
}

int main(int argc, char** argv) {
    const int groups = argc > 1 ? atoi(argv[1]) : 49, bg = argc > 2 ? atoi(argv[2]) : 32;
    if (run<0>(groups, bg)) return 1;
    if (run<1>(groups, bg)) return 1;
    return 0;
}
