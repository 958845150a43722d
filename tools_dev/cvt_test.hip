#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(const float * x, uint16_t * a, uint16_t * b, int n) {
    int i = threadIdx.x;
    if (2 * i + 1 < n) {
        // packed path: two adjacent conversions (the compiler may pick v_cvt_pk_f16_f32)
        __half2 h = __floats2half2_rn(x[2 * i], x[2 * i + 1]);
        a[2 * i] = __half_as_ushort(__low2half(h)); a[2 * i + 1] = __half_as_ushort(__high2half(h));
        uint32_t r0, r1;
        asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r0) : "v"(x[2 * i]));
        asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r1) : "v"(x[2 * i + 1]));
        b[2 * i] = (uint16_t) r0; b[2 * i + 1] = (uint16_t) r1;
    }
}
int main() {
    const int n = 64; float hx[n];
    // ties between adjacent f16 values: mantissa m + 0.5 ulp, for odd and even m, plus near-ties
    for (int i = 0; i < n; i++) { uint32_t bits = 0x39f72000u + (uint32_t) (i / 4) * 0x2000u + ((i % 4 == 0) ? 0x1000u : (i % 4 == 1) ? 0x0fffu : (i % 4 == 2) ? 0x1001u : 0u); memcpy(&hx[i], &bits, 4); }
    float * dx; uint16_t * da, * db; hipMalloc(&dx, sizeof hx); hipMalloc(&da, 2 * n); hipMalloc(&db, 2 * n);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, da, db, n);
    uint16_t ha[n], hb[n]; hipMemcpy(ha, da, 2 * n, hipMemcpyDeviceToHost); hipMemcpy(hb, db, 2 * n, hipMemcpyDeviceToHost);
    int diff = 0;
    for (int i = 0; i < n; i++) { uint32_t bits; memcpy(&bits, &hx[i], 4); if (ha[i] != hb[i]) { diff++; printf("x=%08x pk=%04x single=%04x\n", bits, ha[i], hb[i]); } }
    printf("differences: %d of %d\n", diff, n);
    return 0;
}
