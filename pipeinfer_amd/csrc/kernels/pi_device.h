// Device-side helpers shared by the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <type_traits>

#define PI_WAVE 64

static __device__ __forceinline__ float pi_h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
// RNE, as _cvtss_sh(x, 0).  The empty asm keeps the conversion a separate instruction: the backend otherwise folds
// f16(a * b) into v_fma_mixlo_f16 (ONE rounding) even with -ffp-contract=off, while the reference rounds the fp32 product
// first — the two differ when the fp32 product lands exactly on an f16 tie.
static __device__ __forceinline__ uint16_t pi_f2h(float f) { asm volatile("" : "+v"(f)); return __half_as_ushort(__float2half_rn(f)); }

// Partner exchange for butterfly reductions without the LDS crossbar (`__shfl_xor` is a ds_bpermute: ~100 cycles a step, and the
// small kernels are chains of such steps).  O = 32 / 16: gfx950's v_permlane32_swap / v_permlane16_swap + a select; O = 8 / 4: DPP
// row rotations — lane i gets lane (i + O) % 16 of its row, which is not i ^ O, but for a reduction that every lane finishes with the
// same commutative op the operand *sets* and the association are those of the xor butterfly (the partner's partial is symmetric under
// the steps already taken), so sums are bit-identical and max / arg-max are the same element; O = 2 / 1: DPP quad permutes (= xor).
// Blocks are one-dimensional, so lane = threadIdx.x % 64.
template <int O>
static __device__ __forceinline__ int pi_partner(int v) {
    static_assert(O == 32 || O == 16 || O == 8 || O == 4 || O == 2 || O == 1, "power of two below the wave size");
    if constexpr (O == 32) { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (threadIdx.x & 32) ? r[0] : r[1]; }
    else if constexpr (O == 16) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (threadIdx.x & 16) ? r[0] : r[1]; }
    else if constexpr (O == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);      // row_ror:8
    else if constexpr (O == 4) return __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true);      // row_ror:4
    else if constexpr (O == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    else return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);                             // quad_perm [1,0,3,2]
}
template <int O> static __device__ __forceinline__ float pi_partner(float v) { return __builtin_bit_cast(float, pi_partner<O>(__builtin_bit_cast(int, v))); }
template <int O> static __device__ __forceinline__ double pi_partner(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned) pi_partner<O>((int) (unsigned) b), hi = (unsigned) pi_partner<O>((int) (unsigned) ((unsigned long long) b >> 32));
    return __builtin_bit_cast(double, (long long) (((unsigned long long) hi << 32) | lo));
}
template <int O> static __device__ __forceinline__ unsigned long long pi_partner(unsigned long long v) {
    const unsigned lo = (unsigned) pi_partner<O>((int) (unsigned) v), hi = (unsigned) pi_partner<O>((int) (unsigned) (v >> 32));
    return ((unsigned long long) hi << 32) | lo;
}
// the value of lane i ^ 4 exactly (groups of 8 lanes: the (l, l+4) fold of an AVX hsum): two row shifts and a select
static __device__ __forceinline__ int pi_xor4(int v) {
    const int up = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0xF, true), dn = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shl:4, row_shr:4
    return (threadIdx.x & 4) ? dn : up;
}
static __device__ __forceinline__ float pi_xor4(float v) { return __builtin_bit_cast(float, pi_xor4(__builtin_bit_cast(int, v))); }
// applies f(partner distance as integral_constant) for O, O/2, ..., 1
template <int O, class F> static __device__ __forceinline__ void pi_butterfly(F && f) {
    f(std::integral_constant<int, O>{});
    if constexpr (O > 1) pi_butterfly<O / 2>(f);
}

// butterfly sum over W consecutive lanes (W power of two <= 64); every lane ends with the total
template <int W>
static __device__ __forceinline__ float pi_lanes_sum(float v) {
    static_assert(W != 8, "the distance-4 step is a row rotation: only valid after the distance-8 step (W >= 16) or absent (W <= 4)");
    if constexpr (W > 1) pi_butterfly<W / 2>([&](auto o) { v += pi_partner<decltype(o)::value>(v); });
    return v;
}
static __device__ __forceinline__ float pi_lanes_sum_rt(float v, int w) {
    for (int o = w >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, PI_WAVE);
    return v;
}
static __device__ __forceinline__ double pi_wave_sum_f64(double v) {
    pi_butterfly<32>([&](auto o) { v += pi_partner<decltype(o)::value>(v); });
    return v;
}
static __device__ __forceinline__ float pi_wave_max(float v) {
    pi_butterfly<32>([&](auto o) { v = fmaxf(v, pi_partner<decltype(o)::value>(v)); });
    return v;
}

// ---- device-visible abort word of a run (runtime.hip: pi_op_abort_begin / pi_op_abort_cancel) ----------------------------------------
// The reference's workers poll the cancel channel between every two graph nodes (ggml.c:15748-15811, llama.cpp:5665-5687); here every hot-path
// kernel of a context reads that context's abort word — (run id << 1 | aborted), rewritten by device-scope atomics only — and returns when bit 0
// is set.  The load bypasses L1 (sc1, agent scope: what a poll of another agent's atomic needs) and is issued first; kernels that stream
// weights look at the value only after their first loads are in flight (vmcnt retires in order, so the wait costs them nothing).
static __device__ __forceinline__ int pi_abort_peek(const int32_t * w) { return w ? __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; }
// The word is the first int32 of a 64-byte block (include/pipeinfer_ops.h): words 8..14 count, per kernel family (kernel_timing.h: pi_kt_family), the LAUNCHES that found
// the run aborted and returned — one thread of a launch adds one, on the abort path only.  bench.py's roofline keeps those launches apart from the ones that did the work.
static __device__ __forceinline__ void pi_abort_note(const int32_t * w, int fam) {
    if ((threadIdx.x | threadIdx.y | threadIdx.z | blockIdx.x | blockIdx.y | blockIdx.z) == 0)
        __hip_atomic_fetch_add(const_cast<int32_t *>(w) + 8 + fam, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ bool pi_aborted(const int32_t * w, int fam = 6 /* PI_KT_OTHER */) {
    if ((pi_abort_peek(w) & 1) == 0) return false;
    pi_abort_note(w, fam);
    return true;
}

// KV cell of token i of a pass.  dyn (device memory, refreshed per call so that one captured graph serves every call): {kv_head, n_kv, n_tokens, has_table, cell[0..n_tokens)}.
// A run's tokens sit in consecutive cells from kv_head (find_slot, llama.cpp:1593-1644); a pass that carries several runs at once (context.cpp: coalescing) brings a table.
static __device__ __forceinline__ int pi_cell_of(const int32_t * dyn, int kv_head, int i) { return dyn ? (dyn[3] ? dyn[4 + i] : dyn[0] + i) : kv_head + i; }

// signed 4 x int8 dot product accumulate: v_dot4c_i32_i8
static __device__ __forceinline__ int pi_dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// Q8_K quantisation of the 256 values a wave holds (lane l: values 4l .. 4l+3), quantize_row_q8_K (ggml-quants.c:2329-2364): max = the signed value of the FIRST largest
// |x|, iscale = -128 / max, q = min(127, rne(iscale * x)), d = 1 / iscale.  Returns d; q[] = the lane's four quants; s16 = the sum of the lane's 16-value group (valid in
// every lane of the quad).  The ONE implementation behind k_quantize_q8_K, k_norm_quant, k_act_quant (elementwise.hip) and the mat-vec prologues (mmvq.hip).
static __device__ __forceinline__ float pi_wave_q8K(const float xs[4], int lane, int q[4], int & s16) {
    float amax = 0.0f, mx = 0.0f; int idx = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { const float a = fabsf(xs[i]); if (a > amax) { amax = a; mx = xs[i]; idx = lane * 4 + i; } }
    pi_butterfly<32>([&](auto o) {
        constexpr int O = decltype(o)::value;
        const float oa = pi_partner<O>(amax), om = pi_partner<O>(mx); const int oi = pi_partner<O>(idx);
        if (oa > amax || (oa == amax && oi < idx)) { amax = oa; mx = om; idx = oi; }
    });
    q[0] = q[1] = q[2] = q[3] = 0;
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -128.f / mx;
#pragma unroll
        for (int i = 0; i < 4; i++) { const int r = __float2int_rn(iscale * xs[i]); q[i] = r < 127 ? r : 127; }
        d = 1 / iscale;
    }
    int s = q[0] + q[1] + q[2] + q[3];
    s += pi_partner<1>(s); s += pi_partner<2>(s);
    s16 = s;
    return d;
}
// stores the wave's block into Q8_K planes (pi_q8_layout): qs k-major — values 4l..4l+3 are dword (l & 7) of 32-group (l >> 3) —, bsums of the sixteen 16-groups, d
static __device__ __forceinline__ void pi_wave_q8K_store(const int q[4], int s16, float d, int lane, int64_t row_blk, int8_t * qs, float * dd, int16_t * bsums) {
    ((uint32_t *) (qs + row_blk * 256))[(lane & 7) * 8 + (lane >> 3)] =
        (uint32_t) (q[0] & 0xff) | (uint32_t) (q[1] & 0xff) << 8 | (uint32_t) (q[2] & 0xff) << 16 | (uint32_t) (q[3] & 0xff) << 24;
    if ((lane & 3) == 0) bsums[row_blk * 16 + (lane >> 2)] = (int16_t) s16;
    if (lane == 0) dd[row_blk] = d;
}

// fp16 lookup tables of the reference (ggml.c:2190-2201), computed on the host with the host libm and uploaded once
struct pi_tables {
    const uint16_t * exp_f16;    // f16(expf(x))      indexed by f16 bits of x
    const uint16_t * silu_f16;   // f16(x/(1+expf(-x)))
    const uint16_t * gelu_f16;   // f16(tanh-gelu(x))
};
