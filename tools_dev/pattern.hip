// Access-pattern micro-benchmark for the multi-token MFMA kernel: how fast does HBM stream a Q4_K matrix when a wave reads
// 16 rows x one 144-byte block per step (what a 16x16 MFMA tile needs), versus the same bytes laid out so that those 16 blocks
// are contiguous?  No arithmetic: loads are only XOR-ed together.
//   pattern 0: row-major blocks (rows nb*144 bytes apart); lane (r, g) reads hdr + its 16-byte share; S waves per row group
//   pattern 1: same, but each lane reads its full 64-byte half (S = 1)
//   pattern 2: "tile-major": [row group][block][2304 contiguous bytes], lane reads 16 B x (36/S pieces of 1 KB... ) contiguous
// build: hipcc --offload-arch=gfx950 -O3 tools_dev/pattern.hip -o tools_dev/pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PATTERN, int S, int D>
__global__ void __launch_bounds__(64 * S) k_pat(const uint8_t * __restrict__ w, int nb, float * out) {
    const int lane = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int r = lane & 15, gq = lane >> 4, p = gq >> 1;
    const size_t rg = blockIdx.x;
    uint32_t acc = 0;
    if (PATTERN == 2) {
        // tile-major: block b of the row group = 2304 contiguous bytes; wave s reads its 1/S of it, 16 B per lane per load
        constexpr int PER = 2304 / S;                         // bytes per wave per block (S = 1: 2304 = 2.25 KB -> 3 loads, last partial)
        constexpr int NL = (PER + 1023) / 1024;
        const uint8_t * base = w + rg * (size_t) nb * 2304 + (size_t) s * PER;
        uint4 ring[D][NL];
        auto ld = [&](int b, uint4 (&R)[NL]) {
            const int bc = b < nb ? b : nb - 1;
#pragma unroll
            for (int i = 0; i < NL; i++) { const int off = i * 1024 + lane * 16; R[i] = *(const uint4 *) (base + (size_t) bc * 2304 + (off < PER ? off : 0)); }
        };
#pragma unroll
        for (int u = 0; u < D; u++) ld(u, ring[u]);
        for (int b0 = 0; b0 < nb; b0 += D) {
#pragma unroll
            for (int u = 0; u < D; u++) {
#pragma unroll
                for (int i = 0; i < NL; i++) acc += ring[u][i].x ^ ring[u][i].y ^ ring[u][i].z ^ ring[u][i].w;
                ld(b0 + u + D, ring[u]);
            }
        }
    } else {
        constexpr int NQ = PATTERN == 1 ? 4 : 4 / S;
        const uint8_t * base = w + (rg * 16 + r) * (size_t) nb * 144;
        const int share = 16 + 64 * p + (PATTERN == 1 ? 0 : (64 / S) * s);
        struct W { uint4 hdr; uint4 q[NQ]; };
        W ring[D];
        auto ld = [&](int b, W & R) {
            const int bc = b < nb ? b : nb - 1;
            R.hdr = *(const uint4 *) (base + (size_t) bc * 144);
#pragma unroll
            for (int i = 0; i < NQ; i++) R.q[i] = *(const uint4 *) (base + (size_t) bc * 144 + share + 16 * i);
        };
#pragma unroll
        for (int u = 0; u < D; u++) ld(u, ring[u]);
        for (int b0 = 0; b0 < nb; b0 += D) {
#pragma unroll
            for (int u = 0; u < D; u++) {
                acc += ring[u].hdr.x ^ ring[u].hdr.w;
#pragma unroll
                for (int i = 0; i < NQ; i++) acc += ring[u].q[i].x ^ ring[u].q[i].y ^ ring[u].q[i].z ^ ring[u].q[i].w;
                ld(b0 + u + D, ring[u]);
            }
        }
    }
    if (acc == 0x12345678) out[blockIdx.x] = 1.0f;
}

int main() {
    void * out; CK(hipMalloc(&out, 1 << 24));
    const int shapes[][2] = { { 28672, 8192 }, { 8192, 8192 }, { 8192, 28672 } };
    for (auto & sh : shapes) {
        const int M = sh[0], K = sh[1], nb = K / 256;
        const size_t nbytes = (size_t) M * nb * 144;
        const int ncopy = (int) (600e6 / nbytes) + 1;
        std::vector<void *> bufs(ncopy);
        for (auto & b : bufs) { CK(hipMalloc(&b, nbytes + 4096)); CK(hipMemset(b, 1, nbytes)); }
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        auto run = [&](auto pat, auto sc, auto dc, const char * name) {
            constexpr int P = decltype(pat)::value, S = decltype(sc)::value, D = decltype(dc)::value;
            const unsigned grid = M / 16;
            for (int i = 0; i < ncopy; i++) hipLaunchKernelGGL((k_pat<P, S, D>), dim3(grid), dim3(64 * S), 0, 0, (const uint8_t *) bufs[i], nb, (float *) out);
            CK(hipDeviceSynchronize());
            float tot = 0; const int n = 3 * ncopy;
            for (int i = 0; i < n; i++) {
                CK(hipEventRecord(a, 0));
                hipLaunchKernelGGL((k_pat<P, S, D>), dim3(grid), dim3(64 * S), 0, 0, (const uint8_t *) bufs[i % ncopy], nb, (float *) out);
                CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); tot += ms;
            }
            printf("M=%6d K=%6d %-34s S=%d D=%d: %7.2f us (%6.0f GB/s)\n", M, K, name, S, D, tot / n * 1e3, nbytes / (tot / n * 1e-3) / 1e9);
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>; using I6 = std::integral_constant<int, 6>; using I8 = std::integral_constant<int, 8>;
        run(I0{}, I4{}, I6{}, "row-major, 16 B shares");
        run(I0{}, I2{}, I6{}, "row-major, 32 B shares");
        run(I1{}, I1{}, I4{}, "row-major, 64 B halves");
        run(I1{}, I1{}, I8{}, "row-major, 64 B halves");
        run(I2{}, I1{}, I4{}, "tile-major contiguous");
        run(I2{}, I1{}, I8{}, "tile-major contiguous");
        run(I2{}, I2{}, I6{}, "tile-major contiguous");
        run(I2{}, I4{}, I6{}, "tile-major contiguous");
        for (auto & bb : bufs) CK(hipFree(bb));
    }
    return 0;
}
