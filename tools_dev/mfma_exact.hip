// Is v_mfma_f32_16x16x32_f16 exact on integer data of the sizes the K-quant dot products need?
//   A: |a| <= amax (scale x weight), B: |b| <= 127 (Q8_K activations), C: |c| <= cmax, all integers; every partial sum < 2^24.
// Also pins the lane maps (A: row l&15, k = 8(l>>4)+j; B: col l&15, same k; D: col l&15, row 4(l>>4)+reg).
// build: hipcc --offload-arch=gfx950 -O2 tools_dev/mfma_exact.hip -o tools_dev/mfma_exact
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void k(const _Float16 * A, const _Float16 * B, const float * C, float * D, int n) {
    const int l = threadIdx.x;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        const _Float16 * a = A + (size_t) t * 16 * 32, * b = B + (size_t) t * 32 * 16;
        half8 fa, fb;
        for (int j = 0; j < 8; j++) { fa[j] = a[(l & 15) * 32 + 8 * (l >> 4) + j]; fb[j] = b[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
        f32x4 c;
        for (int i = 0; i < 4; i++) c[i] = C[(size_t) t * 256 + (4 * (l >> 4) + i) * 16 + (l & 15)];
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c, 0, 0, 0);
        for (int i = 0; i < 4; i++) D[(size_t) t * 256 + (4 * (l >> 4) + i) * 16 + (l & 15)] = c[i];
    }
}

int main() {
    const int n = 4096;
    int bad_total = 0;
    const int amaxs[] = { 945, 1953, 2048, 4096 };
    for (int cfg = 0; cfg < 8; cfg++) {
        const int amax = amaxs[cfg & 3];
        const bool mult16 = amax == 4096;                  // 16 * sh * w' (Q6_K high part): multiples of 16
        const bool extreme = cfg >= 4;                     // all operands at their extremes, same sign
        std::vector<_Float16> A((size_t) n * 512), B((size_t) n * 512);
        std::vector<float> C((size_t) n * 256), D((size_t) n * 256);
        srand(1234 + cfg);
        for (auto & v : A) { int x = extreme ? amax : (rand() % (2 * amax + 1)) - amax; if (mult16) x = x / 16 * 16; v = (_Float16) (float) x; }
        for (auto & v : B) { int x = extreme ? 127 : (rand() % 255) - 127; v = (_Float16) (float) x; }
        for (auto & v : C) { int x = (rand() % (1 << 22)) - (extreme ? 0 : (1 << 21)); v = (float) x; }
        _Float16 * dA, * dB; float * dC, * dD;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, n);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0; double maxabs = 0;
        for (int t = 0; t < n; t++)
            for (int r = 0; r < 16; r++)
                for (int c = 0; c < 16; c++) {
                    int64_t s = (int64_t) C[(size_t) t * 256 + r * 16 + c];
                    for (int kk = 0; kk < 32; kk++) s += (int64_t) (float) A[(size_t) t * 512 + r * 32 + kk] * (int64_t) (float) B[(size_t) t * 512 + kk * 16 + c];
                    if ((double) llabs(s) > maxabs) maxabs = (double) llabs(s);
                    if ((float) s != D[(size_t) t * 256 + r * 16 + c]) { if (bad < 3) printf("  mismatch t=%d r=%d c=%d want %lld got %.1f\n", t, r, c, (long long) s, D[(size_t) t * 256 + r * 16 + c]); bad++; }
                }
        printf("cfg %d amax %d extreme %d: %d mismatches of %d, max |sum| %.0f (2^24 = 16777216)\n", cfg, amax, (int) extreme, bad, n * 256, maxabs);
        bad_total += bad;
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    }
    printf(bad_total ? "MFMA f16 NOT exact\n" : "MFMA f16 exact on all configurations\n");
    return bad_total != 0;
}
