// Small per-token operators: embedding gather, norms, activation quantisers, RoPE + KV store, activations, top-k.
// Every rounding point follows the reference CPU path (see include/pipeinfer_ops.h for file:line); this file is
// compiled with -ffp-contract=off so a*b+c is never fused where the reference rounds twice.
#include "ops_common.h"
#include "kernel_timing.h"

#include <cmath>
#include <cstdlib>
#include <vector>

// =================================================================================================================
// get_rows: dequantise n_tok embedding rows (ggml.c:10161-10188; dequantize_row_* ggml-quants.c:1074,1168,1917,2117,2263)
// =================================================================================================================
static __device__ __forceinline__ void k4_scale_min(int j, const uint8_t * q, int & sc, int & m) {
    if (j < 4) { sc = q[j] & 63; m = q[j + 4] & 63; }
    else       { sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}

__global__ void k_get_rows(pi_weight w, const int32_t * __restrict__ tokens, float * __restrict__ out, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    const int t = blockIdx.y;
    const int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= w.ne0) return;
    const int64_t row = tokens[t];
    float y = 0.0f;
    switch (w.type) {
        case PI_F32: y = ((const float *) w.p0)[row * w.ne0 + e]; break;
        case PI_F16: y = pi_h2f(((const uint16_t *) w.p0)[row * w.ne0 + e]); break;
        case PI_Q4_0: {
            const int64_t blk = e >> 5, grp = row * w.ng + (blk >> 2); const int bi = (int) (blk & 3), l = (int) (e & 31);
            const uint8_t q = w.p0[grp * 64 + ((l & 15) >> 2) * 16 + bi * 4 + (l & 3)];        // dword c = byte/4 of the block's 16 quant bytes
            const int x = (l < 16 ? (q & 0xF) : (q >> 4)) - 8;
            y = x * pi_h2f(((const uint16_t *) w.p1)[grp * 4 + bi]);
        } break;
        case PI_Q4_1: case PI_Q5_0: case PI_Q5_1: {                                              // dequantize_row_q4_1 / q5_0 / q5_1, ggml-quants.c:1094-1166
            const int64_t blk = e >> 5, grp = row * w.ng + (blk >> 2); const int bi = (int) (blk & 3), l = (int) (e & 31);
            const uint8_t q = w.p0[grp * 64 + ((l & 15) >> 2) * 16 + bi * 4 + (l & 3)];
            const int hb = (int) ((((const uint32_t *) w.p2)[grp * 4 + bi] >> l) & 1) << 4;       // (0 for Q4_1)
            const int x = (l < 16 ? (q & 0xF) : (q >> 4)) | hb;
            const float d = pi_h2f(((const uint16_t *) w.p1)[grp * 8 + bi]), m = pi_h2f(((const uint16_t *) w.p1)[grp * 8 + 4 + bi]);
            y = w.type == PI_Q5_0 ? (x - 16) * d : x * d + m;
        } break;
        case PI_Q8_0: {
            const int64_t blk = e >> 5, grp = row * w.ng + (blk >> 2); const int bi = (int) (blk & 3), l = (int) (e & 31);
            y = (int) ((const int8_t *) w.p0)[grp * 128 + (l >> 2) * 16 + bi * 4 + (l & 3)] * pi_h2f(((const uint16_t *) w.p1)[grp * 4 + bi]);
        } break;
        case PI_Q4_K: case PI_Q5_K: {
            const int ts = w.type == PI_Q4_K ? 144 : 176;
            const uint8_t * blk = w.p0 + (row * w.nb + (e >> 8)) * ts;
            const int ee = (int) (e & 255), s = ee >> 5, l = ee & 31, j = s >> 1, h = s & 1;
            const float d = pi_h2f(blk[0] | (blk[1] << 8)), mn = pi_h2f(blk[2] | (blk[3] << 8));
            int sc, m; k4_scale_min(s, blk + 4, sc, m);
            const uint8_t * qs = blk + (w.type == PI_Q4_K ? 16 : 48);
            const uint8_t qb = qs[(j >> 1) * 64 + (l >> 2) * 8 + (j & 1) * 4 + (l & 3)];          // p-half order: [p = j/2][lane k = l/4][e = j%2]
            int q = h ? qb >> 4 : qb & 0xF;
            if (w.type == PI_Q5_K) q += ((blk[16 + l] >> (2 * j + h)) & 1) << 4;
            const float d1 = d * sc, m1 = mn * m;
            y = d1 * q - m1;
        } break;
        case PI_Q2_K: {                                                                          // dequantize_row_q2_K, ggml-quants.c:1495-1540
            const int64_t b = row * w.nb + (e >> 8);
            const int ee = (int) (e & 255), n = ee >> 7, j = (ee >> 5) & 3, l = ee & 31, c = l >> 3, h = (l >> 2) & 1;
            const int q = (w.p0[b * 64 + 16 * c + 8 * n + 4 * h + (l & 3)] >> (2 * j)) & 3;        // qs[32n + l]
            const uint8_t scb = w.p1[b * 16 + (ee >> 4)];
            const uint32_t dd = ((const uint32_t *) w.p2)[b];
            const float dl = pi_h2f((uint16_t) (dd & 0xffff)) * (float) (scb & 0xF), ml = pi_h2f((uint16_t) (dd >> 16)) * (float) (scb >> 4);
            y = dl * (float) q - ml;
        } break;
        case PI_Q3_K: {                                                                          // dequantize_row_q3_K, ggml-quants.c:1711-1759
            const int64_t b = row * w.nb + (e >> 8);
            const int ee = (int) (e & 255), n = ee >> 7, j = (ee >> 5) & 3, l = ee & 31, c = l >> 3, h = (l >> 2) & 1;
            const uint8_t lo = w.p0[b * 96 + 16 * c + 8 * n + 4 * h + (l & 3)];                  // qs[32n + l]
            const uint8_t hm = w.p0[b * 96 + 64 + 8 * c + 4 * h + (l & 3)];                      // hmask[l]
            const int q = (int) ((lo >> (2 * j)) & 3) - (((hm >> (4 * n + j)) & 1) ? 0 : 4);
            const float dl = pi_h2f(((const uint16_t *) w.p2)[b]) * (float) ((const int8_t *) w.p1)[b * 16 + (ee >> 4)];
            y = dl * (float) q;
        } break;
        case PI_Q6_K: {
            const int64_t b = row * w.nb + (e >> 8);
            const int ee = (int) (e & 255), n = ee >> 7, g = (ee >> 5) & 3, l = ee & 31;
            const uint8_t lo = w.p0[b * 192 + n * 64 + (l >> 2) * 8 + (g & 1) * 4 + (l & 3)];     // ql[64n + 32(g&1) + l]
            const uint8_t hb = w.p0[b * 192 + 128 + (l >> 2) * 8 + n * 4 + (l & 3)];               // qh[32n + l]
            const int q6 = ((g & 2) ? lo >> 4 : lo & 0xF) | (((hb >> (2 * g)) & 3) << 4);
            const int8_t q = (int8_t) (q6 - 32);
            const float d = pi_h2f(((const uint16_t *) w.p2)[b]);
            y = d * (float) ((const int8_t *) w.p1)[b * 16 + (ee >> 4)] * (float) q;
        } break;
    }
    out[(int64_t) t * w.ne0 + e] = y;
}

extern "C" int pi_op_get_rows(const pi_weight * w, const int32_t * tokens, int n_tok, float * out, void * stream) {
    if (n_tok <= 0) return 0;
    PI_KT_LAUNCH(PI_KT_OTHER, 0, n_tok, (double) n_tok * ((double) w->nbytes / (double) w->ne1 + 4.0 * w->ne0), k_get_rows, dim3((unsigned) ((w->ne0 + 255) / 256), n_tok), dim3(256), 0, (hipStream_t) stream, *w, tokens, out, pi_abort_word());
    PI_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================================
// norms (ggml.c:9028-9074 rms, 8959-9008 layer norm) followed by ggml_mul(weight) and ggml_add(bias) (llama.cpp:3579-3609)
// one 256-thread workgroup per token row; sums in double like the reference
// =================================================================================================================
static __device__ __forceinline__ double block_sum_f64(double v, double * sh) {
    v = pi_wave_sum_f64(v);
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wv] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; i++) s += sh[i];
    return s;
}

__global__ void __launch_bounds__(256) k_norm(const float * __restrict__ x, const float * __restrict__ w, const float * __restrict__ b,
                                              float * __restrict__ y, int n, float eps, int rms, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    __shared__ double sh[8];
    const float * xr = x + (int64_t) blockIdx.x * n;
    float * yr = y + (int64_t) blockIdx.x * n;
    float mean = 0.0f;
    double acc = 0.0;
    if (rms) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) { const float v = xr[i]; acc += (double) (v * v); }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double) xr[i];
        mean = (float) (block_sum_f64(acc, sh) / n);
        acc = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) { const float v = xr[i] - mean; acc += (double) (v * v); }
    }
    const float var = (float) (block_sum_f64(acc, sh) / n);
    const float scale = 1.0f / sqrtf(var + eps);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float v = (xr[i] - mean) * scale;      // rms: mean == 0 exactly, x - 0 == x
        if (w) v = v * w[i];
        if (b) v = v + b[i];
        yr[i] = v;
    }
}

extern "C" int pi_op_norm(const float * x, const float * w, const float * b, float * y, int n, int n_tok, float eps, int rms, void * stream) {
    if (n_tok <= 0) return 0;
    PI_KT_LAUNCH(PI_KT_OTHER, 0, n_tok, (double) n_tok * n * 8 + (double) n * (w ? 4 : 0) + (double) n * (b ? 4 : 0), k_norm, dim3(n_tok), dim3(256), 0, (hipStream_t) stream, x, w, b, y, n, eps, rms, pi_abort_word());
    PI_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================================
// activation quantisers.  One wave per 256 values, 4 consecutive values per lane.
//   Q8_K (ggml-quants.c:2329-2364): max = signed value of the FIRST largest |x|; iscale = -128/max; q = min(127, rne(iscale*x));
//                                   d = 1/iscale (f32); bsums over groups of 16.
//   Q8_0 (ggml-quants.c:702-760, AVX arm): d = amax/127 (stored f16); id = 127/amax; q = rne(x*id).
// Output planes: see pi_q8_layout.
// =================================================================================================================
// MFMA operand planes (pi_q8_frag_layout): the lane that holds values 32j + 4k + (0..3) of a block writes them as f16 (i0, i2, i1, i3) into
// the B-fragment slot of its token, for both sub-block pairings; lanes 16c write the pair sums of the block sums and the scale as floats
struct q8k_frags { uint8_t * fa, * fb, * aux; };         // all null: not wanted (single-token calls)
static __device__ __forceinline__ void wave_emit_frags(const q8k_frags & F, const int q[4], int s4, float d, int lane, int t, int64_t blk, int64_t nb) {
    if (!F.fa) return;
    const int k = lane & 7, j = lane >> 3, tl = t & 15;
    const int64_t cb = (int64_t) (t >> 4) * nb + blk;                                       // (chunk, block)
    const uint32_t h02 = (uint32_t) __half_as_ushort(__float2half_rn((float) q[0])) | (uint32_t) __half_as_ushort(__float2half_rn((float) q[2])) << 16;
    const uint32_t h13 = (uint32_t) __half_as_ushort(__float2half_rn((float) q[1])) | (uint32_t) __half_as_ushort(__float2half_rn((float) q[3])) << 16;
    // fragment of accumulator lane k: MFMA lane (token, c) holds sub-blocks 2c and 2c+1 (8 bytes each) — the lane that unpacks qs group c of a weight row
    *(uint2 *) (F.fa + (((cb * 8 + k) * 64 + tl + 16 * (j >> 1)) * 16 + (j & 1) * 8)) = make_uint2(h02, h13);
    // aux: [16 tokens] d as float, then the sixteen 16-value sums of the block as f16 (|sum| <= 16 * 127: exact), lane (token, c) = sums 4c..4c+3:
    // the operand of the mins product  m[2c]*(bs[4c]+bs[4c+1]) + m[2c+1]*(bs[4c+2]+bs[4c+3])  (ggml-quants.c:5339-5345)
    // s4: sum of the 16-group lane/4 (valid in every lane of the quad)
    if ((lane & 3) == 0) *(uint16_t *) (F.aux + cb * 1024 + 64 + (tl + 16 * (lane >> 4)) * 8 + ((lane >> 2) & 3) * 2) = __half_as_ushort(__float2half_rn((float) s4));
    if (lane == 0) *(float *) (F.aux + cb * 1024 + tl * 4) = d;
    // 16 bytes of zeros behind the block sums (bytes 576..591 of the 1 KB image): the operand of the matrix-core kernel's idle token rows in its packed <= 8-token form
    // (mmq_mfma.hip: k_mmq_g2) — every token of the chunk writes the same zeros
    if (lane == 1) *(uint4 *) (F.aux + cb * 1024 + 576) = make_uint4(0u, 0u, 0u, 0u);
}

__global__ void __launch_bounds__(64) k_quantize_q8_K(const float * __restrict__ x, int64_t k, int8_t * __restrict__ qs, float * __restrict__ dd, int16_t * __restrict__ bsums, const q8k_frags F, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w, 3)) return;
    const int t = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x;
    const int64_t nb = k >> 8;
    const float4 v = *(const float4 *) (x + (int64_t) t * k + (int64_t) blk * 256 + lane * 4);
    const float xs[4] = { v.x, v.y, v.z, v.w };
    int q[4], s;
    const float d = pi_wave_q8K(xs, lane, q, s);
    pi_wave_q8K_store(q, s, d, lane, (int64_t) t * nb + blk, qs, dd, bsums);
    wave_emit_frags(F, q, s, d, lane, t, blk, nb);
}

// block scale of Q8_0 (f16 d) or Q8_1 (ss != null: float d and s = d * sum of the 32 quants, quantize_row_q8_1 ggml-quants.c:940-985)
static __device__ __forceinline__ void q8_0_store_scale(const int q[4], float d, int kk, int64_t grp, int bi, uint16_t * dd, float * ss) {
    if (!ss) { if (kk == 0) dd[grp * 4 + bi] = pi_f2h(d); return; }
    int sum = q[0] + q[1] + q[2] + q[3];
    sum += pi_partner<1>(sum); sum += pi_partner<2>(sum); sum += pi_xor4(sum);            // the 8 lanes of the block (integers: any order)
    if (kk == 0) { ((float *) dd)[grp * 4 + bi] = d; ss[grp * 4 + bi] = d * (float) sum; }
}

// one wave = 8 blocks of 32 = two groups of 4 blocks; values past k belong to the zero padding of the last group
__global__ void __launch_bounds__(64) k_quantize_q8_0(const float * __restrict__ x, int64_t k, int64_t ng, uint32_t * __restrict__ qs, uint16_t * __restrict__ dd, float * __restrict__ ss, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w, 3)) return;
    const int t = blockIdx.y, lane = threadIdx.x;
    const int64_t e0 = (int64_t) blockIdx.x * 256 + lane * 4;
    if (e0 >= ng * 128) return;
    float xs[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (e0 < k) { const float4 v = *(const float4 *) (x + (int64_t) t * k + e0); xs[0] = v.x; xs[1] = v.y; xs[2] = v.z; xs[3] = v.w; }
    float amax = fmaxf(fmaxf(fabsf(xs[0]), fabsf(xs[1])), fmaxf(fabsf(xs[2]), fabsf(xs[3])));
    amax = fmaxf(amax, pi_partner<1>(amax)); amax = fmaxf(amax, pi_partner<2>(amax)); amax = fmaxf(amax, pi_xor4(amax));
    const float d = amax / 127.f;
    const float id = amax != 0.0f ? 127.f / amax : 0.0f;
    int q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { int r = __float2int_rn(xs[i] * id); q[i] = r > 127 ? 127 : (r < -128 ? -128 : r); }
    const int64_t blk = e0 >> 5, grp = (int64_t) t * ng + (blk >> 2); const int bi = (int) (blk & 3), kk = lane & 7;
    qs[(grp * 8 + kk) * 4 + bi] = (uint32_t) (q[0] & 0xff) | (uint32_t) (q[1] & 0xff) << 8 | (uint32_t) (q[2] & 0xff) << 16 | (uint32_t) (q[3] & 0xff) << 24;
    q8_0_store_scale(q, d, kk, grp, bi, dd, ss);
}

// ---- per-256 (Q8_K) / per-8x32 (Q8_0) quantisation of four consecutive values held by each lane of a wave ----------------
// Same arithmetic as k_quantize_q8_K / k_quantize_q8_0 above; `blk` is the index of the 256-value chunk inside the row.
static __device__ __forceinline__ void wave_quantize_q8_K(const float xs[4], int lane, int t, int64_t blk, int64_t nb, int8_t * qs, float * dd, int16_t * bsums, const q8k_frags & F) {
    int q[4], s;
    const float d = pi_wave_q8K(xs, lane, q, s);
    pi_wave_q8K_store(q, s, d, lane, (int64_t) t * nb + blk, qs, dd, bsums);
    wave_emit_frags(F, q, s, d, lane, t, blk, nb);
}
static __device__ __forceinline__ void wave_quantize_q8_0(const float xs[4], int lane, int64_t tok_ng, int64_t e0, uint32_t * qs, uint16_t * dd, float * ss) {
    float amax = fmaxf(fmaxf(fabsf(xs[0]), fabsf(xs[1])), fmaxf(fabsf(xs[2]), fabsf(xs[3])));
    amax = fmaxf(amax, pi_partner<1>(amax)); amax = fmaxf(amax, pi_partner<2>(amax)); amax = fmaxf(amax, pi_xor4(amax));
    const float d = amax / 127.f;
    const float id = amax != 0.0f ? 127.f / amax : 0.0f;
    int q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { int r = __float2int_rn(xs[i] * id); q[i] = r > 127 ? 127 : (r < -128 ? -128 : r); }
    const int64_t blk = e0 >> 5, grp = tok_ng + (blk >> 2); const int bi = (int) (blk & 3), kk = lane & 7;
    qs[(grp * 8 + kk) * 4 + bi] = (uint32_t) (q[0] & 0xff) | (uint32_t) (q[1] & 0xff) << 8 | (uint32_t) (q[2] & 0xff) << 16 | (uint32_t) (q[3] & 0xff) << 24;
    q8_0_store_scale(q, d, kk, grp, bi, dd, ss);
}

// norm (rms / layer) * weight (+ bias) straight into the quantised activation planes; optionally also the fp32 result.
// One workgroup per token; every wave normalises and quantises whole 256-value chunks (a Q8_K block, or eight Q8_0 blocks).
__global__ void __launch_bounds__(1024) k_norm_quant(const float * __restrict__ x, const float * __restrict__ w, const float * __restrict__ b, float * __restrict__ yf,
                                                      int n, float eps, int rms, int vdt, int64_t ng, uint8_t * __restrict__ qs, uint8_t * __restrict__ dd, int16_t * __restrict__ bsums, const q8k_frags F, const int32_t * __restrict__ abort_w) {
    __shared__ double sh[16];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const float * xr = x + (int64_t) t * n;
    auto block_sum = [&](double v) {
        v = pi_wave_sum_f64(v);
        __syncthreads();
        if (lane == 0) sh[wave] = v;
        __syncthreads();
        double s = 0.0;
        for (int i = 0; i < nw; i++) s += sh[i];
        return s;
    };
    // Every thread's elements (4 consecutive values of chunk wave + c * nw) are the same in the statistics pass and in the
    // quantisation pass: with <= NQ_REG chunks per wave they are loaded ONCE and stay in registers, and the norm weight / bias are
    // requested together with x, so the kernel has a single dependent round trip to memory instead of three.
    constexpr int NQ_REG = 4;
    const int64_t n_pad = vdt == PI_Q8_K ? n : ng * 128;
    const int per = (int) (((n_pad + 255) / 256 + nw - 1) / nw);                 // chunks per wave (a last half chunk counts: rows of 128 (2k + 1) values with 32-blocks)
    const bool in_regs = per <= NQ_REG;
    float4 xr4[NQ_REG], wr4[NQ_REG], br4[NQ_REG];
    if (in_regs) {
#pragma unroll
        for (int cc = 0; cc < NQ_REG; cc++) {
            const int64_t e0 = ((int64_t) wave + (int64_t) cc * nw) * 256 + lane * 4;
            const bool in = cc < per && e0 < n;
            xr4[cc] = in ? *(const float4 *) (xr + e0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            wr4[cc] = in && w ? *(const float4 *) (w + e0) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            br4[cc] = in && b ? *(const float4 *) (b + e0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    // the abort word is requested BEHIND the kernel's own loads (loads retire in order: in front of them its round trip — past L1, to where the device-scope atomics
    // live — would gate x) and looked at only in front of the stores: this kernel is ONE chain of dependent round trips (a 1 024-thread workgroup per token), a test on
    // entry adds one (measured +0.9 us on 6.8), and an aborted norm saves nothing worth saving
    const int aborted = pi_abort_peek(abort_w);
    float mean = 0.0f;
    double acc = 0.0;
    if (in_regs) {
        // same per-thread order as the loops below: elements tid*4 .. +3, then + blockDim*4, ...
        if (rms) {
#pragma unroll
            for (int cc = 0; cc < NQ_REG; cc++) if (cc < per && ((int64_t) wave + (int64_t) cc * nw) * 256 + lane * 4 < n) {
                const float4 v = xr4[cc]; acc += (double) (v.x * v.x); acc += (double) (v.y * v.y); acc += (double) (v.z * v.z); acc += (double) (v.w * v.w); }
        } else {
#pragma unroll
            for (int cc = 0; cc < NQ_REG; cc++) if (cc < per && ((int64_t) wave + (int64_t) cc * nw) * 256 + lane * 4 < n) {
                const float4 v = xr4[cc]; acc += (double) v.x; acc += (double) v.y; acc += (double) v.z; acc += (double) v.w; }
            mean = (float) (block_sum(acc) / n);
            acc = 0.0;
#pragma unroll
            for (int cc = 0; cc < NQ_REG; cc++) if (cc < per && ((int64_t) wave + (int64_t) cc * nw) * 256 + lane * 4 < n) {
                const float4 v = xr4[cc];
                const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
                acc += (double) (a0 * a0); acc += (double) (a1 * a1); acc += (double) (a2 * a2); acc += (double) (a3 * a3);
            }
        }
    } else if (rms) {
        for (int i = tid * 4; i < n; i += blockDim.x * 4) { const float4 v = *(const float4 *) (xr + i); acc += (double) (v.x * v.x); acc += (double) (v.y * v.y); acc += (double) (v.z * v.z); acc += (double) (v.w * v.w); }
    } else {
        for (int i = tid * 4; i < n; i += blockDim.x * 4) { const float4 v = *(const float4 *) (xr + i); acc += (double) v.x; acc += (double) v.y; acc += (double) v.z; acc += (double) v.w; }
        mean = (float) (block_sum(acc) / n);
        acc = 0.0;
        for (int i = tid * 4; i < n; i += blockDim.x * 4) {
            const float4 v = *(const float4 *) (xr + i);
            const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
            acc += (double) (a0 * a0); acc += (double) (a1 * a1); acc += (double) (a2 * a2); acc += (double) (a3 * a3);
        }
    }
    const float var = (float) (block_sum(acc) / n);
    const float scale = 1.0f / sqrtf(var + eps);
    if (aborted & 1) { pi_abort_note(abort_w, 2); return; }                                                   // (no barrier below this line)
    int cc = 0;
    for (int64_t c0 = (int64_t) wave * 256; c0 < n_pad; c0 += (int64_t) nw * 256, cc++) {    // whole 256-value chunks per wave
        const int64_t e0 = c0 + lane * 4;
        float ys[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        if (e0 < n) {
            float4 v, wv, bv;
            if (in_regs) {
                // register arrays are indexed by a loop counter: select with a compile-time index (dynamic indexing would spill)
                v = xr4[0]; wv = wr4[0]; bv = br4[0];
#pragma unroll
                for (int q = 1; q < NQ_REG; q++) if (cc == q) { v = xr4[q]; wv = wr4[q]; bv = br4[q]; }
            } else {
                v = *(const float4 *) (xr + e0);
                wv = w ? *(const float4 *) (w + e0) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                bv = b ? *(const float4 *) (b + e0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            const float xv[4] = { v.x, v.y, v.z, v.w }, ww[4] = { wv.x, wv.y, wv.z, wv.w }, bb[4] = { bv.x, bv.y, bv.z, bv.w };
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float y = (xv[i] - mean) * scale;
                if (w) y = y * ww[i];
                if (b) y = y + bb[i];
                ys[i] = y;
            }
            if (yf) *(float4 *) (yf + (int64_t) t * n + e0) = make_float4(ys[0], ys[1], ys[2], ys[3]);
        }
        if (vdt == PI_Q8_K) wave_quantize_q8_K(ys, lane, t, c0 >> 8, n >> 8, (int8_t *) qs, (float *) dd, bsums, F);
        else if (e0 < n_pad) wave_quantize_q8_0(ys, lane, (int64_t) t * ng, e0, (uint32_t *) qs, (uint16_t *) dd, vdt == PI_Q8_1 ? (float *) bsums : nullptr);
    }
}

// The same operator as SEVERAL workgroups per token row (Q8_K, rows of whole 1024-value strips): workgroup g of a row holds, like every other workgroup of the row, the whole row in
// registers (float4 number tid + 256 i, i < n / 1024: the row is 8-32 KB, read from L2 by all of them), computes the row's statistic itself — every workgroup the same sums in the
// same order, hence the same scale — and quantises only ITS four 256-value chunks (wave w: chunk w + 4 g, whose values are its i = g loads).  One four-wave barrier instead of two
// sixteen-wave ones, one chunk per wave instead of two, and the stores of a row spread over n / 1024 CUs: the kernel is a chain of latencies (launch, one round trip for x, the
// reduction, the quantiser's butterflies, the store drain), and this shortens the last three.  The double sums run in another order than k_norm_quant's (and than the
// reference's sequential loop, ggml.c:9028-9074): like there, that moves the fp32 scale by at most an ulp in rare cases (the one tolerance of the contract).
__global__ void __launch_bounds__(256) k_norm_quant_mw(const float * __restrict__ x, const float * __restrict__ w, const float * __restrict__ b, float * __restrict__ yf,
                                                       int n, float eps, int rms, uint8_t * __restrict__ qs, uint8_t * __restrict__ dd, int16_t * __restrict__ bsums, const q8k_frags F, const int32_t * __restrict__ abort_w) {
    __shared__ double sh[4];
    constexpr int NLMAX = 16;
    const int t = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = n >> 10;
    const float * xr = x + (int64_t) t * n;
    float4 xv[NLMAX];
#pragma unroll
    for (int i = 0; i < NLMAX; i++) xv[i] = i < nl ? *(const float4 *) (xr + ((int64_t) tid + 256 * i) * 4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int64_t e0 = ((int64_t) tid + 256 * g) * 4;                                       // this thread's own four values: chunk wave + 4 g, lane's quad
    const float4 wv = w ? *(const float4 *) (w + e0) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    const float4 bv = b ? *(const float4 *) (b + e0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int aborted = pi_abort_peek(abort_w);                                             // (behind the kernel's own loads, looked at in front of the stores: see k_norm_quant)
    auto block_sum = [&](double v) {
        v = pi_wave_sum_f64(v);
        __syncthreads();
        if (lane == 0) sh[wave] = v;
        __syncthreads();
        return ((sh[0] + sh[1]) + sh[2]) + sh[3];
    };
    float mean = 0.0f;
    double acc = 0.0;
    if (rms) {
#pragma unroll
        for (int i = 0; i < NLMAX; i++) if (i < nl) { const float4 v = xv[i]; acc += (double) (v.x * v.x); acc += (double) (v.y * v.y); acc += (double) (v.z * v.z); acc += (double) (v.w * v.w); }
    } else {
#pragma unroll
        for (int i = 0; i < NLMAX; i++) if (i < nl) { const float4 v = xv[i]; acc += (double) v.x; acc += (double) v.y; acc += (double) v.z; acc += (double) v.w; }
        mean = (float) (block_sum(acc) / n);
        acc = 0.0;
#pragma unroll
        for (int i = 0; i < NLMAX; i++) if (i < nl) {
            const float4 v = xv[i];
            const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
            acc += (double) (a0 * a0); acc += (double) (a1 * a1); acc += (double) (a2 * a2); acc += (double) (a3 * a3);
        }
    }
    const float var = (float) (block_sum(acc) / n);
    const float scale = 1.0f / sqrtf(var + eps);
    if (aborted & 1) { pi_abort_note(abort_w, 2); return; }
    float4 v = xv[0];
#pragma unroll
    for (int i = 1; i < NLMAX; i++) if (g == i) v = xv[i];                                  // (compile-time register indices: a dynamic one would spill)
    const float xs[4] = { v.x, v.y, v.z, v.w }, ww[4] = { wv.x, wv.y, wv.z, wv.w }, bb[4] = { bv.x, bv.y, bv.z, bv.w };
    float ys[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float y = (xs[i] - mean) * scale;
        if (w) y = y * ww[i];
        if (b) y = y + bb[i];
        ys[i] = y;
    }
    if (yf) *(float4 *) (yf + (int64_t) t * n + e0) = make_float4(ys[0], ys[1], ys[2], ys[3]);
    wave_quantize_q8_K(ys, lane, t, (int64_t) wave + 4 * g, n >> 8, (int8_t *) qs, (float *) dd, bsums, F);
}

// where the quantisers put the MFMA operand planes of a Q8_K buffer (null for single-token calls and for Q8_0)
static q8k_frags frags_of(int vdt, int64_t k, int n_tok, uint8_t * base) {
    q8k_frags F = { nullptr, nullptr, nullptr };
    if (vdt != PI_Q8_K || n_tok < pi_mmq_min_nt()) return F;
    size_t fa, fb, fx;
    pi_q8_frag_layout(k, n_tok, &fa, &fb, &fx);
    F.fa = base + fa; F.fb = base + fb; F.aux = base + fx;
    return F;
}

extern "C" int pi_op_norm_quantize(const float * x, const float * w, const float * b, float * y_f32, int n, int n_tok, float eps, int rms, int vdt, void * q8, void * stream) {
    if (n_tok <= 0) return 0;
    if ((vdt == PI_Q8_K && n % 256) || (vdt != PI_Q8_K && n % 32) || (vdt != PI_Q8_K && vdt != PI_Q8_0 && vdt != PI_Q8_1)) { pi_set_error("pi_op_norm_quantize: n=%d not valid for activation type %d", n, vdt); return -1; }
    size_t oq, od, ob;
    pi_q8_layout(vdt, n, n_tok, &oq, &od, &ob);
    uint8_t * base = (uint8_t *) q8;
    const int64_t ng = (n / 32 + 3) / 4;
    const int chunks = (int) ((vdt == PI_Q8_K ? n : ng * 128) + 255) / 256;
    const int threads = 64 * (chunks < 16 ? chunks : 16);
    // rows of whole 1024-value strips: n / 1024 four-wave workgroups per row (PI_NORMQ_MW=0: the one-workgroup kernel everywhere)
    static const bool mw_on = !(getenv("PI_NORMQ_MW") && atoi(getenv("PI_NORMQ_MW")) == 0);
    if (mw_on && vdt == PI_Q8_K && n % 1024 == 0 && n / 1024 <= 16) {
        const double qb = (double) n / 256 * 292;
        PI_KT_LAUNCH(PI_KT_NORMQ, 0, n_tok, (double) n_tok * (4.0 * n + qb + (y_f32 ? 4.0 * n : 0.0)) + 4.0 * n * ((w ? 1 : 0) + (b ? 1 : 0)), k_norm_quant_mw, dim3((unsigned) (n / 1024), n_tok), dim3(256), 0, (hipStream_t) stream,
                     x, w, b, y_f32, n, eps, rms, base + oq, base + od, (int16_t *) (base + ob), frags_of(vdt, n, n_tok, base), pi_abort_word());
        PI_CHECK_LAUNCH();
        return 0;
    }
    // algorithmic bytes: the row in (fp32), weight (+ bias), the quantised planes out (Q8_K: 292 B per 256 values; 32-blocks: 34 / 40 B per 32)
    const double qb = vdt == PI_Q8_K ? (double) n / 256 * 292 : (double) n / 32 * (vdt == PI_Q8_1 ? 40 : 34);
    PI_KT_LAUNCH(PI_KT_NORMQ, 0, n_tok, (double) n_tok * (4.0 * n + qb + (y_f32 ? 4.0 * n : 0.0)) + 4.0 * n * ((w ? 1 : 0) + (b ? 1 : 0)), k_norm_quant, dim3(n_tok), dim3(threads), 0, (hipStream_t) stream, x, w, b, y_f32, n, eps, rms, vdt, ng, base + oq, base + od, (int16_t *) (base + ob), frags_of(vdt, n, n_tok, base), pi_abort_word());
    PI_CHECK_LAUNCH();
    return 0;
}

// act(gate) [* up] straight into the quantised activation planes: which = 0 silu(gate)*up, 1 gelu(gate).  One wave per 256 values.
__global__ void __launch_bounds__(256) k_act_quant(const float * __restrict__ g, const float * __restrict__ u, int which, int64_t n, int vdt, int64_t ng,
                                                    const uint16_t * __restrict__ tab, uint8_t * __restrict__ qs, uint8_t * __restrict__ dd, int16_t * __restrict__ bsums, const q8k_frags F, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w, 4)) return;
    const int t = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c0 = ((int64_t) blockIdx.x * 4 + wave) * 256, e0 = c0 + lane * 4;
    const int64_t n_pad = vdt == PI_Q8_K ? n : ng * 128;
    if (c0 >= n_pad) return;
    float ys[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (e0 < n) {
        const float4 gv = *(const float4 *) (g + (int64_t) t * n + e0);
        const float gs[4] = { gv.x, gv.y, gv.z, gv.w };
        float us[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
        if (which == 0) { const float4 uv = *(const float4 *) (u + (int64_t) t * n + e0); us[0] = uv.x; us[1] = uv.y; us[2] = uv.z; us[3] = uv.w; }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (which == 2) { const float a = gs[i] > 0.0f ? gs[i] : 0.0f; ys[i] = a * a; }                 // ggml_relu then ggml_sqr (LLM_FFN_RELU_SQR, ggml.c:1415, 1407)
            else { const float a = pi_h2f(tab[pi_f2h(gs[i])]); ys[i] = which == 0 ? a * us[i] : a; }
        }
    }
    if (vdt == PI_Q8_K) wave_quantize_q8_K(ys, lane, t, c0 >> 8, n >> 8, (int8_t *) qs, (float *) dd, bsums, F);
    else if (e0 < n_pad) wave_quantize_q8_0(ys, lane, (int64_t) t * ng, e0, (uint32_t *) qs, (uint16_t *) dd, vdt == PI_Q8_1 ? (float *) bsums : nullptr);
}

extern "C" int pi_op_act_quantize(const float * gate, const float * up, int which, int64_t n, int n_tok, int vdt, void * q8, void * stream) {
    if (n_tok <= 0) return 0;
    if ((vdt == PI_Q8_K && n % 256) || (vdt != PI_Q8_K && n % 32) || (vdt != PI_Q8_K && vdt != PI_Q8_0 && vdt != PI_Q8_1)) { pi_set_error("pi_op_act_quantize: n=%lld not valid for activation type %d", (long long) n, vdt); return -1; }
    size_t oq, od, ob;
    pi_q8_layout(vdt, n, n_tok, &oq, &od, &ob);
    uint8_t * base = (uint8_t *) q8;
    const int64_t ng = (n / 32 + 3) / 4;
    const int64_t chunks = ((vdt == PI_Q8_K ? n : ng * 128) + 255) / 256;
    const pi_tables & tb = pi_get_tables();
    const double qb = vdt == PI_Q8_K ? (double) n / 256 * 292 : (double) n / 32 * (vdt == PI_Q8_1 ? 40 : 34);
    PI_KT_LAUNCH(PI_KT_ACTQ, 0, n_tok, (double) n_tok * (4.0 * n * (which == 0 ? 2 : 1) + qb), k_act_quant, dim3((unsigned) ((chunks + 3) / 4), n_tok), dim3(256), 0, (hipStream_t) stream, gate, up, which, n, vdt, ng,
                 which == 0 ? tb.silu_f16 : tb.gelu_f16, base + oq, base + od, (int16_t *) (base + ob), frags_of(vdt, n, n_tok, base), pi_abort_word());
    PI_CHECK_LAUNCH();
    return 0;
}

extern "C" int pi_op_quantize(int vdt, const float * x, int64_t k, int n_tok, void * q8, void * stream) {
    if (n_tok <= 0) return 0;
    size_t oq, od, ob;
    pi_q8_layout(vdt, k, n_tok, &oq, &od, &ob);
    uint8_t * base = (uint8_t *) q8;
    if (vdt == PI_Q8_K) {
        if (k % 256) { pi_set_error("pi_op_quantize: Q8_K needs k %% 256 == 0 (k=%lld)", (long long) k); return -1; }
        PI_KT_LAUNCH(PI_KT_QUANT, 0, n_tok, (double) n_tok * (4.0 * k + (double) k / 256 * 292), k_quantize_q8_K, dim3((unsigned) (k / 256), n_tok), dim3(64), 0, (hipStream_t) stream, x, k, (int8_t *) (base + oq), (float *) (base + od), (int16_t *) (base + ob), frags_of(vdt, k, n_tok, base), pi_abort_word());
    } else if (vdt == PI_Q8_0 || vdt == PI_Q8_1) {
        if (k % 32) { pi_set_error("pi_op_quantize: Q8_0 / Q8_1 need k %% 32 == 0 (k=%lld)", (long long) k); return -1; }
        const int64_t ng = (k / 32 + 3) / 4;
        PI_KT_LAUNCH(PI_KT_QUANT, 0, n_tok, (double) n_tok * (4.0 * k + (double) k / 32 * (vdt == PI_Q8_1 ? 40 : 34)), k_quantize_q8_0, dim3((unsigned) ((ng * 128 + 255) / 256), n_tok), dim3(64), 0, (hipStream_t) stream, x, k, ng, (uint32_t *) (base + oq), (uint16_t *) (base + od),
                     vdt == PI_Q8_1 ? (float *) (base + ob) : nullptr, pi_abort_word());
    } else { pi_set_error("pi_op_quantize: unsupported activation type %d", vdt); return -1; }
    PI_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================================
// RoPE (ggml.c:11006-11168) + KV store as f16 (llama.cpp:3546-3577).  cos/sin come from a host-built table so that
// theta (iterated `theta *= theta_scale` in fp32) and libm cosf/sinf are exactly the reference's.
// =================================================================================================================
struct pi_rope_table { float * cs; int n_pos; int head_dim; int n_rot; };   // cs: [n_pos][head_dim/2][2] = (cos, sin)

extern "C" void * pi_rope_table_create(int n_pos, int head_dim, float freq_base, float freq_scale, int neox) {
    return pi_rope_table_create_ex(n_pos, head_dim, head_dim, freq_base, freq_scale, neox);
}
extern "C" void * pi_rope_table_create_ex(int n_pos, int head_dim, int n_rot, float freq_base, float freq_scale, int neox) {
    if (n_rot <= 0 || n_rot % 2 || head_dim % n_rot || (n_rot != head_dim && neox <= 0)) { pi_set_error("pi_rope_table_create: n_rot %d does not divide head_dim %d (or is not NeoX mode)", n_rot, head_dim); return nullptr; }
    std::vector<float> h((size_t) n_pos * head_dim);
    const float theta_scale = powf(freq_base, -2.0f / n_rot);             // ggml.c:11062: n_dims is the rope's, not the head's
    for (int p = 0; p < n_pos; p++) {
        float theta_base = neox < 0 ? 0.0f : (float) p;               // neox < 0: no rotation (cosf(0) = 1, sinf(0) = 0: x0 * 1 - x1 * 0 == x0)
        if (neox > 0) theta_base *= freq_scale;                       // ggml.c:11136
        for (int i = 0; i < head_dim / 2; i++) {
            const float theta = freq_scale * theta_base;              // rope_yarn with ext_factor == 0 (ggml.c:10979-10989)
            h[((size_t) p * (head_dim / 2) + i) * 2 + 0] = cosf(theta) * 1.0f;
            h[((size_t) p * (head_dim / 2) + i) * 2 + 1] = sinf(theta) * 1.0f;
            theta_base *= theta_scale;
        }
    }
    pi_rope_table * t = new pi_rope_table{ nullptr, n_pos, head_dim, n_rot };
    if (hipMalloc((void **) &t->cs, h.size() * 4) != hipSuccess || hipMemcpy(t->cs, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        pi_set_error("pi_rope_table_create: device allocation/copy failed"); delete t; return nullptr;
    }
    return t;
}
extern "C" void pi_rope_table_free(void * t_) { pi_rope_table * t = (pi_rope_table *) t_; if (t) { (void) hipFree(t->cs); delete t; } }

// grid (n_tok, n_head + 2*n_head_kv) — one workgroup of hd/2 threads per head vector
__global__ void k_rope_kv(const float * __restrict__ q, const float * __restrict__ k, const float * __restrict__ v, int64_t qs, int64_t ks, int64_t vs,
                          const int32_t * __restrict__ pos, const float * __restrict__ cs, int neox, int n_rot, int hd, int n_head, int n_head_kv, int kv_head_,
                          uint16_t * __restrict__ q16, uint16_t * __restrict__ kc, uint16_t * __restrict__ vc, const int32_t * __restrict__ dyn, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    const int t = blockIdx.x, hh = blockIdx.y, i = threadIdx.x;           // i: pair index
    const int cell = pi_cell_of(dyn, kv_head_, t);
    const int n_embd = n_head * hd, n_gqa = n_head_kv * hd;
    if (hh >= n_head + n_head_kv) {                                         // V: plain f32 -> f16 copy into the cache row
        const int h = hh - n_head - n_head_kv;
        const float * src = v + (int64_t) t * vs + h * hd;
        uint16_t * dst = vc + (int64_t) cell * n_gqa + h * hd;
        dst[2 * i] = pi_f2h(src[2 * i]); dst[2 * i + 1] = pi_f2h(src[2 * i + 1]);
        return;
    }
    const bool is_q = hh < n_head;
    const int h = is_q ? hh : hh - n_head;
    const float * src = is_q ? q + (int64_t) t * qs + h * hd : k + (int64_t) t * ks + h * hd;
    uint16_t * dst = is_q ? q16 + (int64_t) t * n_embd + h * hd : kc + (int64_t) cell * n_gqa + h * hd;
    const float2 c = *(const float2 *) (cs + ((int64_t) pos[t] * (hd / 2) + i) * 2);
    // NeoX: block ib = i / (n_rot/2) of n_rot values, pair (ib n_rot + k, ib n_rot + k + n_rot/2), table entry i = ib (n_rot/2) + k (ggml.c:11138-11157)
    const int hr = n_rot >> 1, ib = i / hr, kk = i - ib * hr;
    const int i0 = neox ? ib * n_rot + kk : 2 * i, i1 = neox ? i0 + hr : 2 * i + 1;
    const float x0 = src[i0], x1 = src[i1];
    dst[i0] = pi_f2h(x0 * c.x - x1 * c.y);
    dst[i1] = pi_f2h(x0 * c.y + x1 * c.x);
}

// build_persimmon's attention inputs (llama.cpp:4372-4460): the fused QKV rows are [head][q | k | v][64]; Q and K get a LayerNorm over the 64 values of their head
// (attn_q_norm / attn_k_norm, llm_build_norm: norm, * w, + b), then NeoX RoPE over the FIRST 32 values of the head (n_dims = 32: pairs (i, i + 16), theta_scale from 32) while
// the other 32 pass, then the f16 rounding of the cache / of the KQ product.  One workgroup of 64 threads per (token, head); the statistics are the reference's sequential
// double sums (ggml.c:8959-9008), by one thread each for Q and K.  cs: table of pi_rope_table_create_ex(n_pos, 32, 32, ...) = [pos][16][2]
__global__ void __launch_bounds__(64) k_persimmon_qkv(const float * __restrict__ qkv, int n_head, const float * __restrict__ qw, const float * __restrict__ qb,
                                                     const float * __restrict__ kw, const float * __restrict__ kb, float eps, const int32_t * __restrict__ pos,
                                                     const float * __restrict__ cs, int kv_head_, const int32_t * __restrict__ dyn,
                                                     uint16_t * __restrict__ q16, uint16_t * __restrict__ kc, uint16_t * __restrict__ vc, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    __shared__ float sx[2][64];
    __shared__ float st[4];
    const int t = blockIdx.x, h = blockIdx.y, e = threadIdx.x;
    const int n_embd = n_head * 64;
    const float * base = qkv + (int64_t) t * 3 * n_embd + (int64_t) h * 192;
    const float xq = base[e], xk = base[64 + e], xv = base[128 + e];
    sx[0][e] = xq; sx[1][e] = xk;
    __syncthreads();
    if (e < 2) {
        const float * x = sx[e];
        double sum = 0.0;
        for (int i = 0; i < 64; i++) sum += (double) x[i];
        const float mean = (float) (sum / 64);
        double sum2 = 0.0;
        for (int i = 0; i < 64; i++) { const float v = x[i] - mean; sum2 += (double) (v * v); }
        const float var = (float) (sum2 / 64);
        st[2 * e] = mean; st[2 * e + 1] = 1.0f / sqrtf(var + eps);
    }
    __syncthreads();
    float yq = (xq - st[0]) * st[1]; yq = yq * qw[e]; yq = yq + qb[e];
    float yk = (xk - st[2]) * st[3]; yk = yk * kw[e]; yk = yk + kb[e];
    __syncthreads();
    sx[0][e] = yq; sx[1][e] = yk;
    __syncthreads();
    const int cell = pi_cell_of(dyn, kv_head_, t);
    uint16_t * dq = q16 + (int64_t) t * n_embd + h * 64, * dk = kc + (int64_t) cell * n_embd + h * 64, * dv = vc + (int64_t) cell * n_embd + h * 64;
    dv[e] = pi_f2h(xv);
    if (e < 16) {
        const float2 c = *(const float2 *) (cs + ((int64_t) pos[t] * 16 + e) * 2);
        { const float x0 = sx[0][e], x1 = sx[0][e + 16]; dq[e] = pi_f2h(x0 * c.x - x1 * c.y); dq[e + 16] = pi_f2h(x0 * c.y + x1 * c.x); }
        { const float x0 = sx[1][e], x1 = sx[1][e + 16]; dk[e] = pi_f2h(x0 * c.x - x1 * c.y); dk[e + 16] = pi_f2h(x0 * c.y + x1 * c.x); }
    } else if (e >= 32) { dq[e] = pi_f2h(yq); dk[e] = pi_f2h(yk); }
}
extern "C" int pi_op_persimmon_qkv_store(const float * qkv, int n_head, int hd, const float * qw, const float * qb, const float * kw, const float * kb, float eps, const int32_t * pos,
                                         const void * table_, int n_tok, int kv_head, uint16_t * q16, uint16_t * kc, uint16_t * vc, const int32_t * dyn_dev, void * stream) {
    if (n_tok <= 0) return 0;
    const pi_rope_table * t = (const pi_rope_table *) table_;
    if (hd != 64 || t->head_dim != 32 || t->n_rot != 32) { pi_set_error("pi_op_persimmon_qkv_store: head width %d (the reference fixes 64, llama.cpp:3105-3108) with a rope table of %d/%d (needs 32/32)", hd, t->head_dim, t->n_rot); return -1; }
    PI_KT_LAUNCH(PI_KT_OTHER, 0, n_tok, (double) n_tok * n_head * hd * 3 * 6, k_persimmon_qkv, dim3(n_tok, n_head), dim3(64), 0, (hipStream_t) stream, qkv, n_head, qw, qb, kw, kb, eps, pos, t->cs, kv_head, dyn_dev, q16, kc, vc, pi_abort_word());
    PI_CHECK_LAUNCH();
    return 0;
}

extern "C" int pi_op_rope_kv_store(const float * q, const float * k, const float * v, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                   const int32_t * pos, const void * table_, int neox,
                                   int hd, int n_head, int n_head_kv, int n_tok, int kv_head, uint16_t * q16, uint16_t * kc, uint16_t * vc,
                                   const int32_t * dyn_dev, void * stream) {
    if (n_tok <= 0) return 0;
    const pi_rope_table * t = (const pi_rope_table *) table_;
    if (t->head_dim != hd) { pi_set_error("pi_op_rope_kv_store: table head_dim %d != %d", t->head_dim, hd); return -1; }
    PI_KT_LAUNCH(PI_KT_OTHER, 0, n_tok, (double) n_tok * (n_head + 2 * n_head_kv) * hd * 6, k_rope_kv, dim3(n_tok, n_head + 2 * n_head_kv), dim3(hd / 2), 0, (hipStream_t) stream,
                       q, k, v, q_stride, k_stride, v_stride, pos, t->cs, neox, t->n_rot, hd, n_head, n_head_kv, kv_head, q16, kc, vc, dyn_dev, pi_abort_word());
    PI_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================================
// activations through the fp16 tables (ggml.c:1433-1441, 1490-1498), ggml_mul, ggml_add
// =================================================================================================================
__global__ void k_silu_mul(const float * __restrict__ g, const float * __restrict__ u, float * __restrict__ o, int64_t n, const uint16_t * __restrict__ tab, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = pi_h2f(tab[pi_f2h(g[i])]) * u[i];
}
__global__ void k_gelu(const float * __restrict__ x, float * __restrict__ o, int64_t n, const uint16_t * __restrict__ tab, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = pi_h2f(tab[pi_f2h(x[i])]);
}
__global__ void k_add(const float * __restrict__ a, const float * __restrict__ b, float * __restrict__ o, int64_t n, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}

__global__ void k_add_bias(float * __restrict__ x, const float * __restrict__ b, int64_t n, int64_t total, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) x[i] = x[i] + b[i % n];
}
__global__ void k_clamp(float * __restrict__ x, float lo, float hi, int64_t n, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = fmaxf(fminf(x[i], hi), lo);                             // ggml.c:10928: MAX(MIN(x, max), min)
}

extern "C" int pi_op_add_bias(float * x, const float * b, int64_t n, int n_tok, void * stream) {
    const int64_t total = n * n_tok;
    if (total <= 0) return 0;
    PI_KT_LAUNCH(PI_KT_OTHER, 0, n_tok, 8.0 * total + 4.0 * n, k_add_bias, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, (hipStream_t) stream, x, b, n, total, pi_abort_word());
    PI_CHECK_LAUNCH(); return 0;
}
extern "C" int pi_op_clamp(float * x, float lo, float hi, int64_t n, void * stream) {
    if (n <= 0) return 0;
    PI_KT_LAUNCH(PI_KT_OTHER, 0, 1, 8.0 * n, k_clamp, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream, x, lo, hi, n, pi_abort_word());
    PI_CHECK_LAUNCH(); return 0;
}
extern "C" int pi_op_silu_mul(const float * g, const float * u, float * o, int64_t n, void * stream) {
    if (n <= 0) return 0;
    PI_KT_LAUNCH(PI_KT_OTHER, 0, 1, 12.0 * n, k_silu_mul, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream, g, u, o, n, pi_get_tables().silu_f16, pi_abort_word());
    PI_CHECK_LAUNCH(); return 0;
}
extern "C" int pi_op_gelu(const float * x, float * o, int64_t n, void * stream) {
    if (n <= 0) return 0;
    PI_KT_LAUNCH(PI_KT_OTHER, 0, 1, 8.0 * n, k_gelu, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream, x, o, n, pi_get_tables().gelu_f16, pi_abort_word());
    PI_CHECK_LAUNCH(); return 0;
}
extern "C" int pi_op_add(const float * a, const float * b, float * o, int64_t n, void * stream) {
    if (n <= 0) return 0;
    PI_KT_LAUNCH(PI_KT_OTHER, 0, 1, 12.0 * n, k_add, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream, a, b, o, n, pi_abort_word());
    PI_CHECK_LAUNCH(); return 0;
}

// =================================================================================================================
// top-k per token over the vocabulary: ordering (logit desc, id asc) == std::max_element's first-max rule for k = 1
// (llama.cpp:7496-7510) and a stable version of the std::sort in llama_sample_softmax (llama.cpp:7019-7046);
// plus max and sum_i expf(l_i - max) so p_j = expf(l_j - max) / sum on the host.
// =================================================================================================================
// stage 1: grid (n_tok, NCH) — each workgroup finds the k best of its slice of the vocabulary and its partial max / sum-exp
// stage 2: one workgroup per token merges NCH*k candidates (same ordering) and the partial sums.
#define TOPK_NCH 32
// (both stages read their values once into registers; the k selection rounds then run on registers and LDS only)
#define TOPK_VPT 8                                                     // values per thread held in registers by stage 1 (slices up to 2048)
__global__ void __launch_bounds__(256) k_topk1(const float * __restrict__ logits, int n_vocab, int k, int32_t * __restrict__ cid, float * __restrict__ cval, float * __restrict__ cmax, float * __restrict__ csum, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    __shared__ float sv[2][4]; __shared__ int si[2][4];
    const int t = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (n_vocab + TOPK_NCH - 1) / TOPK_NCH, lo = ch * per, hi = min(n_vocab, lo + per);
    const float * l = logits + (int64_t) t * n_vocab;
    const bool in_regs = per <= 256 * TOPK_VPT;
    float rv[TOPK_VPT];
#pragma unroll
    for (int j = 0; j < TOPK_VPT; j++) { const int i = lo + tid + 256 * j; rv[j] = (in_regs && i < hi) ? l[i] : -INFINITY; }
    float prev_v = INFINITY; int prev_i = -1;
    float first = -INFINITY;
    for (int r = 0; r < k; r++) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < TOPK_VPT; j++) {
                const int i = lo + tid + 256 * j; const float v = rv[j];
                const bool after = i < hi && (v < prev_v || (v == prev_v && i > prev_i));
                if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
            }
        } else {
            for (int i = lo + tid; i < hi; i += 256) {
                const float v = l[i];
                const bool after = v < prev_v || (v == prev_v && i > prev_i);
                if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
            }
        }
        pi_butterfly<32>([&](auto o) {
            const float ov = pi_partner<decltype(o)::value>(bv); const int oi = pi_partner<decltype(o)::value>(bi);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        });
        if (lane == 0) { sv[r & 1][wv] = bv; si[r & 1][wv] = bi; }     // double-buffered: one barrier per round
        __syncthreads();
        bv = sv[r & 1][0]; bi = si[r & 1][0];
#pragma unroll
        for (int w2 = 1; w2 < 4; w2++) { const float ov = sv[r & 1][w2]; const int oi = si[r & 1][w2]; if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; } }
        if (tid == 0) { cid[((int64_t) t * TOPK_NCH + ch) * k + r] = bi; cval[((int64_t) t * TOPK_NCH + ch) * k + r] = bv; }
        prev_v = bv; prev_i = bi;
        if (r == 0) first = prev_v;
    }
    float s = 0.0f;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < TOPK_VPT; j++) { const int i = lo + tid + 256 * j; if (i < hi) s += expf(rv[j] - first); }
    } else {
        for (int i = lo + tid; i < hi; i += 256) s += expf(l[i] - first);
    }
    s = pi_lanes_sum<64>(s);
    __syncthreads();
    if (lane == 0) sv[0][wv] = s;
    __syncthreads();
    if (tid == 0) { cmax[t * TOPK_NCH + ch] = first; csum[t * TOPK_NCH + ch] = (sv[0][0] + sv[0][1]) + (sv[0][2] + sv[0][3]); }
}
__global__ void __launch_bounds__(64) k_topk2(int k, const int32_t * __restrict__ cid, const float * __restrict__ cval, const float * __restrict__ cmax, const float * __restrict__ csum,
                                               int32_t * __restrict__ ids, float * __restrict__ vals, float * __restrict__ sumexp, const int32_t * __restrict__ abort_w) {
    if (pi_aborted(abort_w)) return;
    const int t = blockIdx.x, lane = threadIdx.x;
    const int n = TOPK_NCH * k;                                        // <= 256: four candidates per lane
    float cv[4]; int ci[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const int i = lane + 64 * j; cv[j] = i < n ? cval[(int64_t) t * n + i] : -INFINITY; ci[j] = i < n ? cid[(int64_t) t * n + i] : 0x7fffffff; }
    const float pm = lane < TOPK_NCH ? cmax[t * TOPK_NCH + lane] : -INFINITY, ps = lane < TOPK_NCH ? csum[t * TOPK_NCH + lane] : 0.0f;
    float prev_v = INFINITY; int prev_i = -1;
    for (int r = 0; r < k; r++) {
        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float v = cv[j]; const int id = ci[j];
            const bool after = lane + 64 * j < n && (v < prev_v || (v == prev_v && id > prev_i));
            if (after && (v > bv || (v == bv && id < bi))) { bv = v; bi = id; }
        }
        pi_butterfly<32>([&](auto o) {
            const float ov = pi_partner<decltype(o)::value>(bv); const int oi = pi_partner<decltype(o)::value>(bi);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        });
        if (lane == 0) { ids[t * k + r] = bi; vals[t * k + r] = bv; }
        prev_v = bv; prev_i = bi;
    }
    static_assert(TOPK_NCH <= 64, "one partial per lane");
    float mx = pi_wave_max(pm);
    float s = (lane < TOPK_NCH && pm != -INFINITY) ? ps * expf(pm - mx) : 0.0f;
    s = pi_lanes_sum<64>(s);
    if (lane == 0) sumexp[t] = s;
}

// bytes of caller-owned scratch pi_op_topk needs for n_tok rows (one buffer per stream: contexts decode concurrently)
extern "C" size_t pi_op_topk_scratch_bytes(int n_tok) { return (size_t) n_tok * TOPK_NCH * (2 * 8 + 2) * 4; }
extern "C" int pi_op_topk(const float * logits, int n_vocab, int n_tok, int k, int32_t * ids, float * vals, float * sumexp, void * scratch, void * stream) {
    if (n_tok <= 0) return 0;
    if (k > 8) { pi_set_error("pi_op_topk: k=%d > 8", k); return -1; }
    if (!scratch) { pi_set_error("pi_op_topk: scratch is NULL (needs pi_op_topk_scratch_bytes(n_tok) bytes)"); return -1; }
    int32_t * cid = (int32_t *) scratch; float * cval = (float *) (cid + (size_t) n_tok * TOPK_NCH * k);
    float * cmax = cval + (size_t) n_tok * TOPK_NCH * k, * csum = cmax + (size_t) n_tok * TOPK_NCH;
    PI_KT_LAUNCH(PI_KT_OTHER, 0, n_tok, 4.0 * n_tok * n_vocab, k_topk1, dim3(n_tok, TOPK_NCH), dim3(256), 0, (hipStream_t) stream, logits, n_vocab, k, cid, cval, cmax, csum, pi_abort_word());
    PI_KT_LAUNCH(PI_KT_OTHER, 0, n_tok, 0.0, k_topk2, dim3(n_tok), dim3(64), 0, (hipStream_t) stream, k, cid, cval, cmax, csum, ids, vals, sumexp, pi_abort_word());
    PI_CHECK_LAUNCH(); return 0;
}
