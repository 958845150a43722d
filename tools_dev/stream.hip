// HBM streaming micro-benchmark: what does a plain read of a weight matrix cost at mat-vec sizes (cold, rotating buffers)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// each wave reads `per_wave` contiguous bytes as uint4 per lane per step, DEPTH loads in flight per lane
template <int DEPTH>
__global__ void __launch_bounds__(256) k_stream(const uint4 * __restrict__ src, size_t per_wave_u4, float * out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint4 * p = src + wave * per_wave_u4;
    uint32_t acc = 0;
    for (size_t i = lane; i < per_wave_u4; i += 64 * DEPTH) {
        uint4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) { size_t j = i + 64 * d; v[d] = p[j < per_wave_u4 ? j : lane]; }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
    }
    if (acc == 0x12345678) out[wave] = 1.0f;
}

int main(int argc, char ** argv) {
    const size_t sizes[] = { 9437184, 25362432, 107520000 };
    void * out; CK(hipMalloc(&out, 1 << 24));
    for (size_t nbytes : sizes) {
        const double pool = argc > 1 ? atof(argv[1]) * 1e6 : 600e6; const int ncopy = (int) (pool / nbytes) + 1;
        std::vector<void *> bufs(ncopy);
        for (auto & b : bufs) { CK(hipMalloc(&b, nbytes + 4096)); CK(hipMemset(b, 1, nbytes)); }
        for (size_t per_wave : { (size_t) 2304, (size_t) 9216 }) {
            const size_t waves = nbytes / per_wave;
            const unsigned grid = (unsigned) ((waves + 3) / 4);
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            auto run = [&](auto depth, const char * name) {
                constexpr int D = decltype(depth)::value;
                for (int i = 0; i < ncopy; i++) hipLaunchKernelGGL(k_stream<D>, dim3(grid), dim3(256), 0, 0, (const uint4 *) bufs[i], per_wave / 16, (float *) out);
                CK(hipDeviceSynchronize());
                float best = 1e9, tot = 0; const int n = 3 * ncopy;
                for (int i = 0; i < n; i++) {
                    CK(hipEventRecord(a, 0));
                    hipLaunchKernelGGL(k_stream<D>, dim3(grid), dim3(256), 0, 0, (const uint4 *) bufs[i % ncopy], per_wave / 16, (float *) out);
                    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b)); tot += ms; if (ms < best) best = ms;
                }
                printf("bytes=%9zu per_wave=%6zu waves=%7zu depth=%d: avg %7.2f us (%6.0f GB/s)  min %7.2f us\n", nbytes, per_wave, waves, D, tot / n * 1e3, nbytes / (tot / n * 1e-3) / 1e9, best * 1e3);
            };
            run(std::integral_constant<int, 1>{}, "d1");
            run(std::integral_constant<int, 4>{}, "d4");
        }
        for (auto & b : bufs) CK(hipFree(b));
    }
    return 0;
}
