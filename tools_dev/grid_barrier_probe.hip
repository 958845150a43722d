// What does a decode layer gain from ONE persistent launch that walks its phases behind grid barriers, instead of one launch per phase?
//   part 1  the barrier itself: W workgroups x 256 threads, per-XCD counters + one device counter, us per barrier
//   part 2  P dependent streaming phases of B megabytes each (phase p reads buffer p AND a 32 KB vector every workgroup of phase p-1 wrote a slice of):
//           A  one launch per phase, back to back on a stream
//           B  persistent kernel, grid barrier between phases
//           C  as B, and every wave requests the first tile of ITS share of the next phase before it arrives at the barrier (weights do not depend on
//              the activations, so HBM keeps streaming while the barrier settles)
// build: hipcc --offload-arch=gfx950 -O3 tools_dev/grid_barrier_probe.hip -o tools_dev/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct bar_t { unsigned xcd[8 * 32]; unsigned all[32]; };          // counters 128 bytes apart

static __device__ __forceinline__ void grid_barrier(bar_t * b, unsigned epoch, unsigned n_wg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned x = blockIdx.x & 7, n_x = (n_wg - x + 7) / 8;
        const unsigned old = __hip_atomic_fetch_add(&b->xcd[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == n_x * epoch - 1) __hip_atomic_fetch_add(&b->all[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned groups = n_wg < 8 ? n_wg : 8;
        while (__hip_atomic_load(&b->all[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < groups * epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_barriers(bar_t * b, int n, unsigned epoch0) {
    for (int i = 0; i < n; i++) grid_barrier(b, epoch0 + i + 1, gridDim.x);
}

// one phase: the workgroup's slab of the buffer (contiguous per workgroup, tiles of 256 threads x 16 B x U), xor-folded; result mixed with the vector
constexpr int U = 8;
static __device__ __forceinline__ unsigned phase_body(const uint4 * __restrict__ buf, size_t n16, const unsigned * vec_in, unsigned * vec_out, uint4 * pre, bool have_pre) {
    const size_t per_wg = (n16 / gridDim.x) & ~(size_t) (256 * U - 1);
    const uint4 * p = buf + (size_t) blockIdx.x * per_wg + threadIdx.x;
    unsigned acc = 0;
    // the activations: 8192 words written by the previous phase, read device-coherently
    unsigned a = 0;
    {   // eight 16-byte loads per thread, all in flight at once, device-coherent (sc1: past this XCD's L2)
        uint4 w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(w[i]) : "v"((const uint4 *) vec_in + threadIdx.x + 256 * i) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) a ^= w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
    }
    size_t t = 0;
    if (have_pre) {
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= pre[u].x ^ pre[u].w;
        t = 256 * U;
    }
    for (; t < per_wg; t += 256 * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = p[t + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].w;
    }
    acc ^= a;
    // every workgroup writes its slice of the next vector (write-through)
    for (int i = threadIdx.x; i < 8192; i += 256) if ((i >> 5) % gridDim.x == blockIdx.x) __hip_atomic_store(vec_out + i, acc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return acc;
}

__global__ void __launch_bounds__(256) k_phase(const uint4 * buf, size_t n16, const unsigned * vec_in, unsigned * vec_out) {
    uint4 pre[U];
    phase_body(buf, n16, vec_in, vec_out, pre, false);
}

template <bool PRE>
__global__ void __launch_bounds__(256) k_persistent(const uint4 * const * bufs, int n_phase, size_t n16, unsigned * vec0, unsigned * vec1, bar_t * b, unsigned epoch0) {
    uint4 pre[U];
    bool have = false;
    const size_t per_wg = (n16 / gridDim.x) & ~(size_t) (256 * U - 1);
    for (int ph = 0; ph < n_phase; ph++) {
        phase_body(bufs[ph], n16, (ph & 1) ? vec1 : vec0, (ph & 1) ? vec0 : vec1, pre, have);
        if (ph + 1 < n_phase) {
            if (PRE) {
                const uint4 * p = bufs[ph + 1] + (size_t) blockIdx.x * per_wg + threadIdx.x;
#pragma unroll
                for (int u = 0; u < U; u++) pre[u] = p[u * 256];
                have = true;
            }
            grid_barrier(b, epoch0 + ph + 1, gridDim.x);
        }
    }
}

int main(int argc, char ** argv) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    bar_t * b; CK(hipMalloc(&b, sizeof(bar_t))); CK(hipMemset(b, 0, sizeof(bar_t)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    unsigned epoch = 0;
    int occ_b = 0, occ_p = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, k_barriers, 256, 0)); CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_p, k_persistent<true>, 256, 0));
    printf("co-resident workgroups per CU: barrier kernel %d, persistent kernel %d\n", occ_b, occ_p);
    for (int wg : { 256, 512, 1024, 2048 }) {
        if (wg > occ_b * 256) continue;
        CK(hipMemsetAsync(b, 0, sizeof(bar_t), s)); CK(hipStreamSynchronize(s)); epoch = 0;
        const int n = 200;
        hipLaunchKernelGGL(k_barriers, dim3(wg), dim3(256), 0, s, b, 10, epoch); epoch += 10; CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_barriers, dim3(wg), dim3(256), 0, s, b, n, epoch); epoch += n;
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid barrier, %4d workgroups x 256: %.2f us each\n", wg, 1e3 * ms / n);
    }
    unsigned * vec0, * vec1; CK(hipMalloc(&vec0, 8192 * 4)); CK(hipMalloc(&vec1, 8192 * 4)); CK(hipMemset(vec0, 0, 8192 * 4)); CK(hipMemset(vec1, 0, 8192 * 4));
    const int P = 8, NBUF = 16;
    for (size_t mb : { 4, 37, 132, 264 }) {
        const size_t bytes = mb << 20, n16 = bytes / 16;
        std::vector<uint4 *> bufs(NBUF);
        for (auto & p : bufs) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); }
        const uint4 ** dl; CK(hipMalloc(&dl, 2 * P * sizeof(void *)));
        std::vector<const uint4 *> hl(2 * P);
        for (int i = 0; i < 2 * P; i++) hl[i] = bufs[i % NBUF];
        CK(hipMemcpy(dl, hl.data(), 2 * P * sizeof(void *), hipMemcpyHostToDevice));
        for (int wg : { 256, 512, 1024, 2048 }) {
            if (wg > occ_p * 256) continue;                    // a grid barrier needs every workgroup resident
            const int reps = 6;
            double tA = 0, tB = 0, tC = 0;
            for (int r = 0; r < reps + 1; r++) {
                const int off = (r & 1) * P;
                CK(hipEventRecord(e0, s));
                for (int ph = 0; ph < P; ph++) hipLaunchKernelGGL(k_phase, dim3(wg), dim3(256), 0, s, hl[off + ph], n16, (ph & 1) ? vec1 : vec0, (ph & 1) ? vec0 : vec1);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r) tA += ms;
                CK(hipMemsetAsync(b, 0, sizeof(bar_t), s));
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(k_persistent<false>, dim3(wg), dim3(256), 0, s, dl + off, P, n16, vec0, vec1, b, 0u);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r) tB += ms;
                CK(hipMemsetAsync(b, 0, sizeof(bar_t), s));
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(k_persistent<true>, dim3(wg), dim3(256), 0, s, dl + off, P, n16, vec0, vec1, b, 0u);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r) tC += ms;
            }
            const double ua = 1e3 * tA / reps / P, ub = 1e3 * tB / reps / P, uc = 1e3 * tC / reps / P;
            printf("%4zu MB x %d phases, %4d workgroups: per phase  launches %.2f us (%.2f TB/s) | persistent %.2f us (%.2f TB/s) | persistent + prefetch across the barrier %.2f us (%.2f TB/s)\n",
                   mb, P, wg, ua, bytes / ua / 1e6, ub, bytes / ub / 1e6, uc, bytes / uc / 1e6);
        }
        for (auto & p : bufs) CK(hipFree(p));
        CK(hipFree(dl));
    }
    return 0;
}
