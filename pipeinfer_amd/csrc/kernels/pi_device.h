// Device-side helpers shared by the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define PI_WAVE 64

static __device__ __forceinline__ float pi_h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
// RNE, as _cvtss_sh(x, 0).  The empty asm keeps the conversion a separate instruction: the backend otherwise folds
// f16(a * b) into v_fma_mixlo_f16 (ONE rounding) even with -ffp-contract=off, while the reference rounds the fp32 product
// first — the two differ when the fp32 product lands exactly on an f16 tie.
static __device__ __forceinline__ uint16_t pi_f2h(float f) { asm volatile("" : "+v"(f)); return __half_as_ushort(__float2half_rn(f)); }

// butterfly sum over W consecutive lanes (W power of two <= 64); every lane ends with the total
template <int W>
static __device__ __forceinline__ float pi_lanes_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, PI_WAVE);
    return v;
}
static __device__ __forceinline__ float pi_lanes_sum_rt(float v, int w) {
    for (int o = w >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, PI_WAVE);
    return v;
}
static __device__ __forceinline__ double pi_wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, PI_WAVE);
    return v;
}
static __device__ __forceinline__ float pi_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, PI_WAVE));
    return v;
}

// signed 4 x int8 dot product accumulate: v_dot4c_i32_i8
static __device__ __forceinline__ int pi_dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// fp16 lookup tables of the reference (ggml.c:2190-2201), computed on the host with the host libm and uploaded once
struct pi_tables {
    const uint16_t * exp_f16;    // f16(expf(x))      indexed by f16 bits of x
    const uint16_t * silu_f16;   // f16(x/(1+expf(-x)))
    const uint16_t * gelu_f16;   // f16(tanh-gelu(x))
};
