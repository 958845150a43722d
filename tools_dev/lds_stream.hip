// How fast can a workgroup-phase structure stream weights THROUGH LDS?  (proxy for an LDS-staged decode mat-vec: weights arrive by LDS-DMA, no VGPRs hold loads in flight)
//   A: workgroup = 256 threads, CHUNK bytes: every wave issues its share of global_load_lds_dwordx4, waits, barrier, then all waves read the chunk back from LDS (xor)
//   B: plain register streaming (8 x 16 B per thread in flight), for reference
// build: hipcc --offload-arch=gfx950 -O3 tools_dev/lds_stream.hip -o tools_dev/lds_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CHUNK>
__global__ void __launch_bounds__(256) k_lds(const uint8_t * __restrict__ w, size_t nbytes, uint32_t * out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t base = (size_t) blockIdx.x * CHUNK;
    if (base >= nbytes) return;
    const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) lds;
    constexpr int PIECES = CHUNK / 1024;                                   // 1 KB per DMA instruction
#pragma unroll
    for (int i = 0; i < PIECES / 4; i++) {
        const int piece = i * 4 + wave;
        const uint8_t * src = w + base + (size_t) piece * 1024 + lane * 16;
        const uint32_t dst = lds0 + piece * 1024;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < CHUNK / (256 * 16); i++) { const uint4 v = *(const uint4 *) (lds + (size_t) (i * 256 + threadIdx.x) * 16); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_reg(const uint4 * __restrict__ p, size_t n16, uint32_t * out) {
    const size_t stride = (size_t) gridDim.x * 256; size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; uint32_t acc = 0;
    for (; i + 7 * stride < n16; i += 8 * stride) { uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u].x ^ v[u].w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// C: register streaming at the GRANULARITY of the decode mat-vec: workgroups of T threads, each thread L x 16 B, no loop (one workgroup = T * L * 16 bytes, then it exits)
template <int T, int L>
__global__ void __launch_bounds__(T) k_small(const uint4 * __restrict__ p, size_t n16, uint32_t * out) {
    const size_t base = (size_t) blockIdx.x * T * L + threadIdx.x;
    uint4 v[L]; uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < L; u++) { const size_t i = base + (size_t) u * T; v[u] = p[i < n16 ? i : n16 - 1]; }
#pragma unroll
    for (int u = 0; u < L; u++) acc ^= v[u].x ^ v[u].w;
    if (acc == 0x12345678u) out[0] = acc;
}
// D: the decode mat-vec's own load pattern without its arithmetic: workgroup = 2 waves (K = 8192: two tiles of 16 blocks), R rows per workgroup; lane (q, c) of wave t loads, for each
//    row, the block header (16 B, the four lanes of a quad the same), and two 16-byte pieces at 16 + 16c and 80 + 16c of block 16t + q (144-byte blocks, rows 32 blocks = 4608 B apart)
//    PLANES = 1: the same bytes re-laid per tile as [16 headers][64 x 16 B][64 x 16 B] so that every wave instruction reads one contiguous piece
template <int R, int PLANES>
__global__ void __launch_bounds__(128) k_pattern(const uint8_t * __restrict__ w, size_t nrows, uint32_t * out) {
    const int lane = threadIdx.x & 63, t = threadIdx.x >> 6, q = lane >> 2, c = lane & 3;
    uint4 v[R][3]; uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < R; i++) {
        size_t row = (size_t) blockIdx.x * R + i; if (row >= nrows) row = nrows - 1;
        const uint8_t * tile = w + row * 4608 + (size_t) t * 2304;
        if (PLANES) { v[i][0] = *(const uint4 *) (tile + q * 16); v[i][1] = *(const uint4 *) (tile + 256 + lane * 16); v[i][2] = *(const uint4 *) (tile + 1280 + lane * 16); }
        else { const uint8_t * b = tile + q * 144; v[i][0] = *(const uint4 *) b; v[i][1] = *(const uint4 *) (b + 16 + 16 * c); v[i][2] = *(const uint4 *) (b + 80 + 16 * c); }
    }
#pragma unroll
    for (int i = 0; i < R; i++) acc ^= v[i][0].x ^ v[i][1].y ^ v[i][2].w;
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    uint32_t * out; CK(hipMalloc(&out, 1 << 22));
    for (size_t nbytes : { (size_t) 37748736, (size_t) 132120576, (size_t) 264241152 }) {
        const int ncopy = (int) (700e6 / nbytes) + 1;
        std::vector<uint8_t *> bufs(ncopy);
        for (auto & b : bufs) { CK(hipMalloc(&b, nbytes + 65536)); CK(hipMemset(b, 1, nbytes)); }
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        auto timeit = [&](auto launch, const char * name) {
            for (int i = 0; i < ncopy; i++) launch(bufs[i]);
            CK(hipDeviceSynchronize());
            const int n = 4 * ncopy; CK(hipEventRecord(a, 0));
            for (int i = 0; i < n; i++) launch(bufs[i % ncopy]);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("%6.1f MB  %-44s %7.2f us  %6.0f GB/s\n", nbytes / 1e6, name, ms / n * 1e3, nbytes / (ms / n * 1e-3) / 1e9);
        };
        timeit([&](uint8_t * p) { hipLaunchKernelGGL(k_reg, dim3(2048), dim3(256), 0, 0, (const uint4 *) p, nbytes / 16, out); }, "registers, 8 x 16 B per thread, 2048 WGs");
        CK(hipFuncSetAttribute((const void *) k_lds<16384>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384));
        CK(hipFuncSetAttribute((const void *) k_lds<32768>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
        CK(hipFuncSetAttribute((const void *) k_lds<49152>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
        timeit([&](uint8_t * p) { hipLaunchKernelGGL((k_lds<16384>), dim3((unsigned) ((nbytes + 16383) / 16384)), dim3(256), 16384, 0, p, nbytes, out); }, "LDS-DMA, 16 KB per workgroup (9 WGs/CU)");
        timeit([&](uint8_t * p) { hipLaunchKernelGGL((k_lds<32768>), dim3((unsigned) ((nbytes + 32767) / 32768)), dim3(256), 32768, 0, p, nbytes, out); }, "LDS-DMA, 32 KB per workgroup (4 WGs/CU)");
        timeit([&](uint8_t * p) { hipLaunchKernelGGL((k_lds<49152>), dim3((unsigned) ((nbytes + 49151) / 49152)), dim3(256), 49152, 0, p, nbytes, out); }, "LDS-DMA, 48 KB per workgroup (3 WGs/CU)");
        auto small = [&](auto tc, auto lc, const char * name) {
            constexpr int T = decltype(tc)::value, L = decltype(lc)::value;
            const size_t n16 = nbytes / 16; const unsigned grid = (unsigned) ((n16 + (size_t) T * L - 1) / ((size_t) T * L));
            timeit([&](uint8_t * p) { hipLaunchKernelGGL((k_small<T, L>), dim3(grid), dim3(T), 0, 0, (const uint4 *) p, n16, out); }, name);
        };
        small(std::integral_constant<int, 128>{}, std::integral_constant<int, 5>{}, "registers, 128 threads x 5 loads (10 KB per WG)");
        small(std::integral_constant<int, 128>{}, std::integral_constant<int, 10>{}, "registers, 128 threads x 10 loads (20 KB per WG)");
        small(std::integral_constant<int, 256>{}, std::integral_constant<int, 5>{}, "registers, 256 threads x 5 loads (20 KB per WG)");
        small(std::integral_constant<int, 256>{}, std::integral_constant<int, 10>{}, "registers, 256 threads x 10 loads (40 KB per WG)");
        small(std::integral_constant<int, 512>{}, std::integral_constant<int, 10>{}, "registers, 512 threads x 10 loads (80 KB per WG)");
        small(std::integral_constant<int, 1024>{}, std::integral_constant<int, 10>{}, "registers, 1024 threads x 10 loads (160 KB per WG)");
        {
            const size_t nrows = nbytes / 4608;
            timeit([&](uint8_t * p) { hipLaunchKernelGGL((k_pattern<2, 0>), dim3((unsigned) ((nrows + 1) / 2)), dim3(128), 0, 0, p, nrows, out); }, "decode pattern, 2 rows per WG (144-byte blocks)");
            timeit([&](uint8_t * p) { hipLaunchKernelGGL((k_pattern<2, 1>), dim3((unsigned) ((nrows + 1) / 2)), dim3(128), 0, 0, p, nrows, out); }, "decode pattern, 2 rows per WG, per-tile planes");
            timeit([&](uint8_t * p) { hipLaunchKernelGGL((k_pattern<4, 0>), dim3((unsigned) ((nrows + 3) / 4)), dim3(128), 0, 0, p, nrows, out); }, "decode pattern, 4 rows per WG (144-byte blocks)");
            timeit([&](uint8_t * p) { hipLaunchKernelGGL((k_pattern<4, 1>), dim3((unsigned) ((nrows + 3) / 4)), dim3(128), 0, 0, p, nrows, out); }, "decode pattern, 4 rows per WG, per-tile planes");
        }
        for (auto bb : bufs) CK(hipFree(bb));
    }
    return 0;
}
