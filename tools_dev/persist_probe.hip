// What would ONE persistent launch per draft step take BESIDE the target's kernels?  (tools_dev/persist_beside.py drives this next to real target runs.)
// The draft step of the bench (TinyLlama-1.1B Q4_K_M, one token) is 203 dependent launches moving 0.67 GB: 1.3 ms alone, 2.5-2.8 ms beside the target.
// This library replays that step's SHAPE - the same number of dependent phases, each reading its matrix's bytes, tiny phases run by one workgroup - two ways on a
// high-priority stream: (A) one launch per phase, (B) one persistent kernel of W workgroups x 256 threads with a grid barrier between phases.  No arithmetic of the
// product is in here; it answers a scheduling question only.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools_dev/persist_probe.hip -o tools_dev/libpersist_probe.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

struct bar_t { unsigned xcd[8 * 32]; unsigned all[32]; };

static __device__ __forceinline__ void grid_barrier(bar_t * b, unsigned epoch, unsigned n_wg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned x = blockIdx.x & 7, n_x = (n_wg - x + 7) / 8;
        const unsigned old = __hip_atomic_fetch_add(&b->xcd[x * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (old == n_x * epoch - 1) __hip_atomic_fetch_add(&b->all[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned groups = n_wg < 8 ? n_wg : 8;
        while (__hip_atomic_load(&b->all[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < groups * epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

struct phase_t { unsigned long long off16, n16; };          // the phase's bytes inside the weight buffer, in 16-byte units; n16 == 0: a one-workgroup phase

constexpr int U = 4;
static __device__ __forceinline__ void phase_body(const uint4 * __restrict__ w, phase_t ph, const unsigned * vec_in, unsigned * vec_out, unsigned wg, unsigned n_wg) {
    unsigned acc = 0;
    if (ph.n16 == 0) {                                        // norm / quantise / attention-sized phase: one workgroup reads the 8 KB vector and writes it back
        if (wg == 0) for (int i = threadIdx.x; i < 2048; i += 256) __hip_atomic_store(vec_out + i, __hip_atomic_load(vec_in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const unsigned a = __hip_atomic_load(vec_in + (threadIdx.x & 2047), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const size_t per = (ph.n16 + n_wg - 1) / n_wg, lo = (size_t) wg * per, hi = lo + per < ph.n16 ? lo + per : ph.n16;
    const uint4 * p = w + ph.off16;
    for (size_t t = lo + threadIdx.x; t < hi; t += 256 * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = t + u * 256 < hi ? p[t + u * 256] : uint4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    acc ^= a;
    if (threadIdx.x < 8) __hip_atomic_store(vec_out + ((wg * 8 + threadIdx.x) & 2047), acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256) k_phase(const uint4 * w, phase_t ph, const unsigned * vec_in, unsigned * vec_out) { phase_body(w, ph, vec_in, vec_out, blockIdx.x, gridDim.x); }

__global__ void __launch_bounds__(256) k_persistent(const uint4 * w, const phase_t * phases, int n_phase, unsigned * vec0, unsigned * vec1, bar_t * b) {
    for (int i = 0; i < n_phase; i++) {
        phase_body(w, phases[i], (i & 1) ? vec1 : vec0, (i & 1) ? vec0 : vec1, blockIdx.x, gridDim.x);
        if (i + 1 < n_phase) grid_barrier(b, (unsigned) i + 1, gridDim.x);
    }
}

static uint4 * g_w; static phase_t * g_ph_dev; static std::vector<phase_t> g_ph; static unsigned * g_v0, * g_v1; static bar_t * g_bar; static hipStream_t g_s; static hipEvent_t g_e0, g_e1;
static size_t g_bytes;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

// TinyLlama-1.1B Q4_K_M: per layer norm | q+k+v | attention | quantise | wo | norm | up+gate | act+quantise | down, then norm | output | top-k | top-k
extern "C" int pp_init(void) {
    const double MB = 1 << 20;
    const double layer[9] = { 0, 3.0 * MB, 0, 0, 2.36 * MB, 0, 12.98 * MB, 0, 7.9 * MB };
    size_t off = 0;
    auto add = [&](double bytes) { phase_t p; p.off16 = off; p.n16 = (unsigned long long) (bytes / 16); off += p.n16; g_ph.push_back(p); };
    add(0);
    for (int l = 0; l < 22; l++) for (double b : layer) add(b);
    add(0); add(53.7 * MB); add(0); add(0);
    g_bytes = off * 16;
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&g_s, hipStreamNonBlocking, hi));
    CK(hipMalloc(&g_w, g_bytes)); CK(hipMemset(g_w, 1, g_bytes));
    CK(hipMalloc(&g_ph_dev, g_ph.size() * sizeof(phase_t))); CK(hipMemcpy(g_ph_dev, g_ph.data(), g_ph.size() * sizeof(phase_t), hipMemcpyHostToDevice));
    CK(hipMalloc(&g_v0, 8192)); CK(hipMalloc(&g_v1, 8192)); CK(hipMemset(g_v0, 0, 8192)); CK(hipMemset(g_v1, 0, 8192));
    CK(hipMalloc(&g_bar, sizeof(bar_t)));
    CK(hipEventCreate(&g_e0)); CK(hipEventCreate(&g_e1));
    return (int) g_ph.size();
}
extern "C" double pp_bytes(void) { return (double) g_bytes; }
// one step; returns its GPU milliseconds (events on the probe's stream), < 0 on error.  mode 0: a launch per phase (grids of `wg` workgroups), 1: persistent kernel of `wg` workgroups
// `wrap` (bit mask): what the engine's draft call has around its graph - 1: four small H2D copies from pinned memory ahead of it, 2: a 128 KB + three small D2H copies to pinned
// memory behind it, 4: a one-thread kernel ahead of everything, 8: the copies of 1 / 2 done by kernels on mapped pinned memory instead of hipMemcpyAsync
static int g_wrap; static uint8_t * g_pin; static uint8_t * g_pin_dev; static uint8_t * g_devbuf;
__global__ void k_one(unsigned * p) { p[0] += 1; }
__global__ void k_copy(const uint4 * src, uint4 * dst, size_t n16) { for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) dst[i] = src[i]; }
extern "C" int pp_wrap(int wrap) {
    g_wrap = wrap;
    if (!g_pin) { CK(hipHostMalloc((void **) &g_pin, 1 << 20, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void **) &g_pin_dev, g_pin, 0)); CK(hipMalloc(&g_devbuf, 1 << 20)); }
    return 0;
}
extern "C" float pp_step(int mode, int wg) {
    float ms = 0;
    if (g_wrap) {
        CK(hipEventRecord(g_e0, g_s));
        if (g_wrap & 4) hipLaunchKernelGGL(k_one, dim3(1), dim3(1), 0, g_s, g_v0);
        if (g_wrap & 1) {
            if (g_wrap & 8) hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, g_s, (const uint4 *) g_pin_dev, (uint4 *) g_devbuf, (size_t) 16);
            else for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(g_devbuf + 4096 * i, g_pin + 4096 * i, i == 3 ? 128 : 16, hipMemcpyHostToDevice, g_s));
        }
        hipGraphExec_t ge = nullptr;
        {
            static std::vector<std::pair<int, hipGraphExec_t>> graphs;
            for (auto & g : graphs) if (g.first == wg) ge = g.second;
            if (!ge) {
                hipGraph_t gr = nullptr;
                CK(hipStreamBeginCapture(g_s, hipStreamCaptureModeThreadLocal));
                for (size_t i = 0; i < g_ph.size(); i++) hipLaunchKernelGGL(k_phase, dim3(g_ph[i].n16 ? wg : 1), dim3(256), 0, g_s, g_w, g_ph[i], (i & 1) ? g_v1 : g_v0, (i & 1) ? g_v0 : g_v1);
                CK(hipStreamEndCapture(g_s, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0)); CK(hipGraphDestroy(gr));
                graphs.push_back({wg, ge});
            }
        }
        CK(hipGraphLaunch(ge, g_s));
        if (g_wrap & 2) {
            if (g_wrap & 8) hipLaunchKernelGGL(k_copy, dim3(8), dim3(256), 0, g_s, (const uint4 *) g_devbuf, (uint4 *) (g_pin_dev + 65536), (size_t) 8192 + 16);
            else {
                CK(hipMemcpyAsync(g_pin + 65536, g_devbuf + 65536, 128000, hipMemcpyDeviceToHost, g_s));
                for (int i = 0; i < 3; i++) CK(hipMemcpyAsync(g_pin + 4096 * i + 512, g_devbuf + 4096 * i, 32, hipMemcpyDeviceToHost, g_s));
            }
        }
        CK(hipEventRecord(g_e1, g_s)); CK(hipEventSynchronize(g_e1)); CK(hipEventElapsedTime(&ms, g_e0, g_e1));
        return ms;
    }
    if (mode == 1) { CK(hipMemsetAsync(g_bar, 0, sizeof(bar_t), g_s)); CK(hipEventRecord(g_e0, g_s)); }
    if (mode == 0) {                                          // replayed as a captured graph, like the engine's draft step
        static std::vector<std::pair<int, hipGraphExec_t>> graphs;
        hipGraphExec_t ge = nullptr;
        for (auto & g : graphs) if (g.first == wg) ge = g.second;
        if (!ge) {
            hipGraph_t gr = nullptr;
            CK(hipStreamBeginCapture(g_s, hipStreamCaptureModeThreadLocal));
            for (size_t i = 0; i < g_ph.size(); i++) hipLaunchKernelGGL(k_phase, dim3(g_ph[i].n16 ? wg : 1), dim3(256), 0, g_s, g_w, g_ph[i], (i & 1) ? g_v1 : g_v0, (i & 1) ? g_v0 : g_v1);
            CK(hipStreamEndCapture(g_s, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0)); CK(hipGraphDestroy(gr));
            graphs.push_back({wg, ge});
            CK(hipGraphLaunch(ge, g_s)); CK(hipStreamSynchronize(g_s));
        }
        CK(hipEventRecord(g_e0, g_s));
        CK(hipGraphLaunch(ge, g_s));
    } else hipLaunchKernelGGL(k_persistent, dim3(wg), dim3(256), 0, g_s, g_w, g_ph_dev, (int) g_ph.size(), g_v0, g_v1, g_bar);
    CK(hipEventRecord(g_e1, g_s)); CK(hipEventSynchronize(g_e1)); CK(hipEventElapsedTime(&ms, g_e0, g_e1));
    return ms;
}
