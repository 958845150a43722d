// Can every wave of a kernel see, cheaply, a word that a kernel of ANOTHER stream rewrote with a device-scope atomic?  (the abort word of a run, pi_device.h)
//   part 1  staleness: a one-thread writer kernel on stream A exchanges the word; after it has completed, a 2048-workgroup reader kernel on stream B reads it
//           three ways (scalar load, vector sc1 load = relaxed agent-scope atomic load, plain vector load); every workgroup reports what it saw
//   part 2  cost: a streaming kernel at the mat-vec's granularity (128-thread workgroups, 5 x 16 B per thread, no loop) with the check in front of its loads,
//           behind them, or absent
// build: hipcc --offload-arch=gfx950 -O3 tools_dev/abort_probe.hip -o tools_dev/abort_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_write(int * w, int v) { __hip_atomic_exchange(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_cas(int * w, int expect, int v) { __hip_atomic_compare_exchange_strong(w, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(64) k_read(const int * w, int * out, int nblk) {
    int s;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s) : "s"(w) : "memory");
    const int a = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int p = *(volatile const int *) w;
    if (threadIdx.x == 0) { out[blockIdx.x] = s; out[nblk + blockIdx.x] = a; out[2 * nblk + blockIdx.x] = p; }
}

// MODE 0: no check; 1: vector sc1 load, tested BEFORE the loads are issued; 2: scalar load, tested before; 3: vector sc1 load issued first, tested after the loads are issued
template <int T, int L, int MODE>
__global__ void __launch_bounds__(T) k_stream(const uint4 * __restrict__ p, size_t n16, uint32_t * out, const int * abort_w) {
    int ab = 0;
    if constexpr (MODE == 1 || MODE == 3) ab = __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (MODE == 2) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ab) : "s"(abort_w) : "memory");
    if constexpr (MODE == 1 || MODE == 2) { if (ab & 1) return; }
    const size_t base = (size_t) blockIdx.x * T * L + threadIdx.x;
    uint4 v[L]; uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < L; u++) { const size_t i = base + (size_t) u * T; v[u] = p[i < n16 ? i : n16 - 1]; }
    if constexpr (MODE == 3) { if (ab & 1) return; }
#pragma unroll
    for (int u = 0; u < L; u++) acc ^= v[u].x ^ v[u].w;
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static double time_stream(const std::vector<uint4 *> & bufs, size_t n16, uint32_t * out, const int * w, hipStream_t s) {
    constexpr int T = 128, L = 5;
    const unsigned grid = (unsigned) ((n16 + T * L - 1) / (T * L));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_stream<T, L, MODE>), dim3(grid), dim3(T), 0, s, bufs[i % bufs.size()], n16, out, w);
    CK(hipStreamSynchronize(s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int n = 60;
    CK(hipEventRecord(a, s));
    for (int i = 0; i < n; i++) hipLaunchKernelGGL((k_stream<T, L, MODE>), dim3(grid), dim3(T), 0, s, bufs[i % bufs.size()], n16, out, w);
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1e3 * ms / n;
}

int main() {
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const int nblk = 2048;
    int * out; CK(hipMalloc(&out, 3 * nblk * 4));
    std::vector<int> h(3 * nblk);
    for (int kind = 0; kind < 2; kind++) {
        int * w = nullptr;
        if (kind == 0) CK(hipMalloc(&w, 256)); else CK(hipExtMallocWithFlags((void **) &w, 256, hipDeviceMallocFinegrained));
        CK(hipMemset(w, 0, 256));
        long stale[3] = { 0, 0, 0 }; int trials = 1500;
        for (int i = 1; i <= trials; i++) {
            if (i & 1) hipLaunchKernelGGL(k_write, dim3(1), dim3(1), 0, sa, w, i); else hipLaunchKernelGGL(k_cas, dim3(1), dim3(1), 0, sa, w, i - 1, i);
            CK(hipStreamSynchronize(sa));
            hipLaunchKernelGGL(k_read, dim3(nblk), dim3(64), 0, sb, w, out, nblk);
            CK(hipStreamSynchronize(sb));
            CK(hipMemcpy(h.data(), out, 3 * nblk * 4, hipMemcpyDeviceToHost));
            for (int m = 0; m < 3; m++) for (int b = 0; b < nblk; b++) if (h[m * nblk + b] != i) stale[m]++;
        }
        printf("%s: stale reads out of %ld: scalar %ld, vector sc1 %ld, plain %ld\n", kind ? "fine-grained" : "hipMalloc", (long) trials * nblk, stale[0], stale[1], stale[2]);
        // cost beside a stream: 132 MB buffers, rotated beyond the Infinity Cache
        const size_t bytes = (size_t) 132 << 20, n16 = bytes / 16;
        std::vector<uint4 *> bufs(4);
        for (auto & b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
        CK(hipMemset(w, 0, 256));
        for (int rep = 0; rep < 2; rep++)
            printf("  %s stream 132 MB: no check %.2f us | vector sc1 before loads %.2f | scalar before loads %.2f | vector sc1, tested behind the loads %.2f\n", kind ? "fine-grained" : "hipMalloc",
                   time_stream<0>(bufs, n16, (uint32_t *) out, w, sb), time_stream<1>(bufs, n16, (uint32_t *) out, w, sb), time_stream<2>(bufs, n16, (uint32_t *) out, w, sb), time_stream<3>(bufs, n16, (uint32_t *) out, w, sb));
        int one = 1; CK(hipMemcpy(w, &one, 4, hipMemcpyHostToDevice));
        printf("  aborted (word = 1): vector before %.2f us | scalar before %.2f | vector behind the loads %.2f\n", time_stream<1>(bufs, n16, (uint32_t *) out, w, sb), time_stream<2>(bufs, n16, (uint32_t *) out, w, sb), time_stream<3>(bufs, n16, (uint32_t *) out, w, sb));
        for (auto & b : bufs) CK(hipFree(b));
        CK(hipFree(w));
    }
    return 0;
}
