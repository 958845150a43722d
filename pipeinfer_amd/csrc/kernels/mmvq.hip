// Quantised mat-vec / tiny-batch mat-mat for the decode and verify path (n_tok <= 8 per launch).
//
//   out[t][row] = residual[t][row] + sum_blocks dot(W[row][blk], act[t][blk])
//
// Reference: ggml_compute_forward_mul_mat (ggml.c:9388-9597) calling ggml_vec_dot_{q4_K,q5_K,q6_K}_q8_K and
// ggml_vec_dot_{q4_0,q8_0}_q8_0 (ggml-quants.c:5216, 5864, 6594, 2426, 3521).  The integer part of every block dot
// product is exact (v_dot4_i32_i8); only the order in which the per-block fp32 terms are added differs from the CPU.
//
// Work decomposition (wave64):
//   * K-quants: 4 lanes per 256-value super-block, so one wave spans 16 super-blocks (4096 values) of a row with
//     three 16-byte loads per lane (header / two quant slices), all 16-byte aligned.
//   * Q4_0/Q8_0: one lane per 32-value block, a wave spans 64 blocks (2048 values).
//   * Rows longer than one wave span are split over NKW "K-waves" of the workgroup whose partial sums meet in LDS.
//     Rows shorter than a span pack several rows into one wave (LPR lanes per row).
//   * Each lane keeps ITS slice of the quantised activations of all NT tokens in registers for the whole launch and
//     streams weight rows past them: weights are read exactly once from HBM, activations once per wave from L2.
#include "ops_common.h"

struct mmvq_args {
    pi_weight w;
    const int8_t *  aq;       // [n_tok][K]
    const void *    ad;       // Q8_K: f32 [n_tok][K/256]   Q8_0: f16 [n_tok][K/32]
    const int16_t * abs_;     // Q8_K: [n_tok][K/16]
    float *         out;      // [n_tok][M]
    const float *   res;      // nullable
    int M, K;
    int nkw;                  // K-waves per row
    int nrg;                  // row groups (waves along rows) per workgroup
    int rpw;                  // rows per wave (multiple of rows-per-iteration)
    int lpr;                  // lanes per row (<= 64, power of two)
};

template <int TYPE> struct qtraits;
template <> struct qtraits<PI_Q4_K> { static constexpr int LPB = 4, NY = 16, KQ = 1; };   // lanes per block, act dwords per lane, K-quant?
template <> struct qtraits<PI_Q5_K> { static constexpr int LPB = 4, NY = 16, KQ = 1; };
template <> struct qtraits<PI_Q6_K> { static constexpr int LPB = 4, NY = 16, KQ = 1; };
template <> struct qtraits<PI_Q4_0> { static constexpr int LPB = 1, NY = 8,  KQ = 0; };
template <> struct qtraits<PI_Q8_0> { static constexpr int LPB = 1, NY = 8,  KQ = 0; };

// per-lane activation slice of one token
template <int TYPE> struct act_slice {
    int   y[qtraits<TYPE>::NY];
    float d;
    int   s[4];               // type specific sums (see load)
};

template <int TYPE>
static __device__ __forceinline__ void load_act(act_slice<TYPE> & a, const mmvq_args & g, int t, int64_t gb, int part) {
    if constexpr (TYPE == PI_Q4_K || TYPE == PI_Q5_K) {
        // part j: values 64j .. 64j+63 (sub-blocks 2j, 2j+1)
        const int4 * p = (const int4 *) (g.aq + (int64_t) t * g.K + gb * 256 + part * 64);
#pragma unroll
        for (int i = 0; i < 4; i++) { const int4 v = p[i]; a.y[4 * i] = v.x; a.y[4 * i + 1] = v.y; a.y[4 * i + 2] = v.z; a.y[4 * i + 3] = v.w; }
        a.d = ((const float *) g.ad)[(int64_t) t * (g.K >> 8) + gb];
        const short4 b = *(const short4 *) (g.abs_ + ((int64_t) t * (g.K >> 8) + gb) * 16 + part * 4);
        a.s[0] = (int) (short) (b.x + b.y);           // the reference adds the pair in int16 (_mm_hadd_epi16)
        a.s[1] = (int) (short) (b.z + b.w);
    } else if constexpr (TYPE == PI_Q6_K) {
        // part = 2n + hl: four runs of 16 values at 128n + 32g + 16hl, g = 0..3; scale index of run g = 8n + 2g + hl
        const int n = part >> 1, hl = part & 1;
        const int8_t * base = g.aq + (int64_t) t * g.K + gb * 256 + 128 * n + 16 * hl;
#pragma unroll
        for (int r = 0; r < 4; r++) { const int4 v = *(const int4 *) (base + 32 * r); a.y[4 * r] = v.x; a.y[4 * r + 1] = v.y; a.y[4 * r + 2] = v.z; a.y[4 * r + 3] = v.w; }
        a.d = ((const float *) g.ad)[(int64_t) t * (g.K >> 8) + gb];
        const int16_t * bs = g.abs_ + ((int64_t) t * (g.K >> 8) + gb) * 16 + 8 * n + hl;
#pragma unroll
        for (int r = 0; r < 4; r++) a.s[r] = bs[2 * r];
    } else {
        const int4 * p = (const int4 *) (g.aq + (int64_t) t * g.K + gb * 32);
        const int4 v0 = p[0], v1 = p[1];
        a.y[0] = v0.x; a.y[1] = v0.y; a.y[2] = v0.z; a.y[3] = v0.w; a.y[4] = v1.x; a.y[5] = v1.y; a.y[6] = v1.z; a.y[7] = v1.w;
        a.d = pi_h2f(((const uint16_t *) g.ad)[(int64_t) t * (g.K >> 5) + gb]);
        int sum = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) sum = pi_dot4(a.y[i], 0x01010101, sum);
        a.s[0] = sum;
    }
}

// per-lane weight slice of one row
template <int TYPE> struct w_slice { int4 a, b, c, d, e; uint16_t dh; };

template <int TYPE>
static __device__ __forceinline__ void load_w(w_slice<TYPE> & s, const pi_weight & w, int64_t row, int64_t gb, int part) {
    if constexpr (TYPE == PI_Q4_K) {
        const uint8_t * blk = w.p0 + (row * w.nb + gb) * 144;
        s.a = *(const int4 *) blk;                                  // d, dmin, scales[12]
        s.b = *(const int4 *) (blk + 16 + 32 * part);
        s.c = *(const int4 *) (blk + 32 + 32 * part);
    } else if constexpr (TYPE == PI_Q5_K) {
        const uint8_t * blk = w.p0 + (row * w.nb + gb) * 176;
        s.a = *(const int4 *) blk;
        s.d = *(const int4 *) (blk + 16);                           // qh[0..15]
        s.e = *(const int4 *) (blk + 32);                           // qh[16..31]
        s.b = *(const int4 *) (blk + 48 + 32 * part);
        s.c = *(const int4 *) (blk + 64 + 32 * part);
    } else if constexpr (TYPE == PI_Q6_K) {
        const int64_t b = row * w.nb + gb;
        const int n = part >> 1, hl = part & 1;
        s.a = *(const int4 *) (w.p0 + b * 128 + 64 * n + 16 * hl);        // ql, groups 0 (low nibble) and 2 (high)
        s.b = *(const int4 *) (w.p0 + b * 128 + 64 * n + 32 + 16 * hl);   // ql, groups 1 and 3
        s.c = *(const int4 *) (w.p1 + b * 64 + 32 * n + 16 * hl);         // qh
        s.d = *(const int4 *) (w.p2 + b * 16);                            // 16 int8 scales
        s.dh = ((const uint16_t *) w.p3)[b];
    } else if constexpr (TYPE == PI_Q4_0) {
        const int64_t b = row * w.nb + gb;
        s.a = *(const int4 *) (w.p0 + b * 16);
        s.dh = ((const uint16_t *) w.p1)[b];
    } else {
        const int64_t b = row * w.nb + gb;
        s.a = *(const int4 *) (w.p0 + b * 32);
        s.b = *(const int4 *) (w.p0 + b * 32 + 16);
        s.dh = ((const uint16_t *) w.p1)[b];
    }
}

static __device__ __forceinline__ int byte_of(const int4 & v, int i) {      // i in 0..15, runtime
    const int wsel = i >> 2;
    const uint32_t word = wsel == 0 ? (uint32_t) v.x : wsel == 1 ? (uint32_t) v.y : wsel == 2 ? (uint32_t) v.z : (uint32_t) v.w;
    return (word >> (8 * (i & 3))) & 0xff;
}

// 6-bit scale/min of sub-block `is` (get_scale_min_k4, ggml-quants.c:1446-1453); hdr bytes 4..15 are scales[0..11]
static __device__ __forceinline__ void k4_unpack(const int4 & hdr, int is, int & sc, int & m) {
    if (is < 4) { sc = byte_of(hdr, 4 + is) & 63; m = byte_of(hdr, 8 + is) & 63; }
    else {
        const int hi = byte_of(hdr, 8 + is);                        // scales[is + 4]
        sc = (hi & 0xF) | ((byte_of(hdr, is) >> 6) << 4);           // scales[is - 4]  (hdr byte 4 + is - 4)
        m  = (hi >> 4)  | ((byte_of(hdr, 4 + is) >> 6) << 4);       // scales[is]
    }
}

template <int TYPE>
static __device__ __forceinline__ float dot_slice(const w_slice<TYPE> & s, const act_slice<TYPE> & a, int part) {
    if constexpr (TYPE == PI_Q4_K || TYPE == PI_Q5_K) {
        int sc0, m0, sc1, m1;
        k4_unpack(s.a, 2 * part, sc0, m0);
        k4_unpack(s.a, 2 * part + 1, sc1, m1);
        const int q[8] = { s.b.x, s.b.y, s.b.z, s.b.w, s.c.x, s.c.y, s.c.z, s.c.w };
        int lo = 0, hi = 0;
        if constexpr (TYPE == PI_Q4_K) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                lo = pi_dot4(q[i] & 0x0F0F0F0F, a.y[i], lo);
                hi = pi_dot4((q[i] >> 4) & 0x0F0F0F0F, a.y[8 + i], hi);
            }
        } else {
            const int h[8] = { s.d.x, s.d.y, s.d.z, s.d.w, s.e.x, s.e.y, s.e.z, s.e.w };
            const int sh = 2 * part;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int hb = (int) ((uint32_t) h[i] >> sh);
                lo = pi_dot4((q[i] & 0x0F0F0F0F) | ((hb & 0x01010101) << 4), a.y[i], lo);
                hi = pi_dot4(((q[i] >> 4) & 0x0F0F0F0F) | (((hb >> 1) & 0x01010101) << 4), a.y[8 + i], hi);
            }
        }
        const int isum = sc0 * lo + sc1 * hi;
        const int msum = m0 * a.s[0] + m1 * a.s[1];
        const float d    = pi_h2f((uint16_t) (s.a.x & 0xffff)) * a.d;
        const float dmin = pi_h2f((uint16_t) ((uint32_t) s.a.x >> 16)) * a.d;
        return d * (float) isum - dmin * (float) msum;
    } else if constexpr (TYPE == PI_Q6_K) {
        const int n = part >> 1, hl = part & 1;
        const int l1[4] = { s.a.x, s.a.y, s.a.z, s.a.w }, l2[4] = { s.b.x, s.b.y, s.b.z, s.b.w }, qh[4] = { s.c.x, s.c.y, s.c.z, s.c.w };
        int dg[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t h = (uint32_t) qh[i];
            dg[0] = pi_dot4((l1[i] & 0x0F0F0F0F)        | (int) ((h & 0x03030303u) << 4),        a.y[i],      dg[0]);
            dg[1] = pi_dot4((l2[i] & 0x0F0F0F0F)        | (int) (((h >> 2) & 0x03030303u) << 4), a.y[4 + i],  dg[1]);
            dg[2] = pi_dot4(((l1[i] >> 4) & 0x0F0F0F0F) | (int) (((h >> 4) & 0x03030303u) << 4), a.y[8 + i],  dg[2]);
            dg[3] = pi_dot4(((l2[i] >> 4) & 0x0F0F0F0F) | (int) (((h >> 6) & 0x03030303u) << 4), a.y[12 + i], dg[3]);
        }
        int isum = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int sc = (int) (int8_t) byte_of(s.d, 8 * n + 2 * r + hl);
            isum += sc * (dg[r] - 32 * a.s[r]);
        }
        return (pi_h2f(s.dh) * a.d) * (float) isum;
    } else if constexpr (TYPE == PI_Q4_0) {
        const int q[4] = { s.a.x, s.a.y, s.a.z, s.a.w };
        int isum = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            isum = pi_dot4(q[i] & 0x0F0F0F0F, a.y[i], isum);
            isum = pi_dot4((q[i] >> 4) & 0x0F0F0F0F, a.y[4 + i], isum);
        }
        isum -= 8 * a.s[0];
        return (pi_h2f(s.dh) * a.d) * (float) isum;
    } else {
        const int q[8] = { s.a.x, s.a.y, s.a.z, s.a.w, s.b.x, s.b.y, s.b.z, s.b.w };
        int isum = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) isum = pi_dot4(q[i], a.y[i], isum);
        return (pi_h2f(s.dh) * a.d) * (float) isum;
    }
}

template <int TYPE, int NT>
__global__ void __launch_bounds__(qtraits<TYPE>::KQ ? 512 : 1024) k_mmvq(const mmvq_args g) {
    extern __shared__ float lds[];                       // [nrg][nkw][rpw][NT] partial sums when nkw > 1
    constexpr int LPB = qtraits<TYPE>::LPB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kw = wave % g.nkw, rg = wave / g.nkw;
    const int rpi = 64 / g.lpr;                          // rows per wave iteration
    const int sub = lane / g.lpr, l = lane % g.lpr;
    const int blk = l / LPB, part = l % LPB;
    const int64_t gb = (int64_t) kw * (64 / LPB) + blk;  // block index within the row
    const bool live = gb < g.w.nb;

    act_slice<TYPE> act[NT];
    if (live) {
#pragma unroll
        for (int t = 0; t < NT; t++) load_act<TYPE>(act[t], g, t, gb, part);
    }

    const int64_t row0 = ((int64_t) blockIdx.x * g.nrg + rg) * g.rpw;
    for (int it = 0; it < g.rpw; it += rpi) {
        const int64_t row = row0 + it + sub;
        float acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = 0.0f;
        if (live && row < g.M) {
            w_slice<TYPE> ws;
            load_w<TYPE>(ws, g.w, row, gb, part);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = dot_slice<TYPE>(ws, act[t], part);
        }
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = pi_lanes_sum_rt(acc[t], g.lpr);
        if (l == 0 && row < g.M) {
            if (g.nkw == 1) {
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const int64_t o = (int64_t) t * g.M + row;
                    g.out[o] = g.res ? g.res[o] + acc[t] : acc[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; t++) lds[(((rg * g.nkw + kw) * g.rpw) + it + sub) * NT + t] = acc[t];
            }
        }
    }
    if (g.nkw > 1) {
        __syncthreads();
        const int n_out = g.nrg * g.rpw * NT;
        for (int i = threadIdx.x; i < n_out; i += blockDim.x) {
            const int t = i % NT, r = (i / NT) % g.rpw, rgi = i / (NT * g.rpw);
            const int64_t row = ((int64_t) blockIdx.x * g.nrg + rgi) * g.rpw + r;
            if (row >= g.M) continue;
            float s = 0.0f;
            for (int k = 0; k < g.nkw; k++) s += lds[(((rgi * g.nkw + k) * g.rpw) + r) * NT + t];
            const int64_t o = (int64_t) t * g.M + row;
            g.out[o] = g.res ? g.res[o] + s : s;
        }
    }
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

template <int TYPE, int NT>
static int launch_mmvq(mmvq_args & g, hipStream_t stream) {
    constexpr int LPB = qtraits<TYPE>::LPB;
    const int span = 64 / LPB;                                   // blocks per wave
    const int nb = (int) g.w.nb;
    if (nb >= span) { g.nkw = (nb + span - 1) / span; g.lpr = 64; }
    else            { g.nkw = 1; g.lpr = LPB * next_pow2(nb); }
    constexpr int MAX_KW = qtraits<TYPE>::KQ ? 8 : 16;          // K-quants keep 19 act dwords/token in VGPRs: 512 threads max
    if (g.nkw > MAX_KW) { pi_set_error("pi_op_mul_mat: K=%d too long for one workgroup (nkw=%d)", g.K, g.nkw); return -1; }
    const int rpi = 64 / g.lpr;
    g.nrg = g.nkw >= 4 ? 1 : (g.nkw >= 2 ? 2 : 4);               // 4..8 waves per workgroup (more for very long rows)
    // rows per wave: keep ~2-4k waves in the launch (256 CUs x 8-16 resident waves), at least one iteration
    int rpw = rpi;
    while ((int64_t) g.M * g.nkw / ((int64_t) (rpw / rpi) * 2) >= 2048 && rpw < 16 * rpi) rpw *= 2;
    g.rpw = rpw;
    const int rows_per_wg = g.nrg * g.rpw;
    const unsigned grid = (unsigned) ((g.M + rows_per_wg - 1) / rows_per_wg);
    const size_t lds = g.nkw > 1 ? (size_t) g.nrg * g.nkw * g.rpw * NT * sizeof(float) : 0;
    hipLaunchKernelGGL((k_mmvq<TYPE, NT>), dim3(grid), dim3(64 * g.nkw * g.nrg), lds, stream, g);
    PI_CHECK_LAUNCH();
    return 0;
}

template <int TYPE>
static int dispatch_nt(mmvq_args & g, int nt, hipStream_t stream) {
    switch (nt) {
        case 1: return launch_mmvq<TYPE, 1>(g, stream);
        case 2: return launch_mmvq<TYPE, 2>(g, stream);
        case 3: return launch_mmvq<TYPE, 3>(g, stream);
        case 4: return launch_mmvq<TYPE, 4>(g, stream);
        case 5: return launch_mmvq<TYPE, 5>(g, stream);
        case 6: return launch_mmvq<TYPE, 6>(g, stream);
        case 7: return launch_mmvq<TYPE, 7>(g, stream);
        case 8: return launch_mmvq<TYPE, 8>(g, stream);
    }
    return -1;
}

extern "C" int pi_op_mul_mat(const pi_weight * w, const void * q8, int n_tok, float * out, const float * res, void * stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    if (n_tok <= 0) return 0;
    const int vdt = pi_vec_dot_type(w->type);
    if (vdt != PI_Q8_K && vdt != PI_Q8_0) { pi_set_error("pi_op_mul_mat: weight type %d has no quantised mat-vec", w->type); return -1; }
    size_t oq, od, ob;
    pi_q8_layout(vdt, w->ne0, n_tok, &oq, &od, &ob);
    const uint8_t * base = (const uint8_t *) q8;
    const int64_t K = w->ne0, M = w->ne1;
    for (int t0 = 0; t0 < n_tok; t0 += 8) {                   // verify batches are <= 8 wide; prompts loop over chunks
        const int nt = n_tok - t0 < 8 ? n_tok - t0 : 8;
        mmvq_args g;
        g.w = *w;
        g.aq = (const int8_t *) (base + oq) + (int64_t) t0 * K;
        g.ad = vdt == PI_Q8_K ? (const void *) ((const float *) (base + od) + (int64_t) t0 * (K / 256)) : (const void *) ((const uint16_t *) (base + od) + (int64_t) t0 * (K / 32));
        g.abs_ = vdt == PI_Q8_K ? (const int16_t *) (base + ob) + (int64_t) t0 * (K / 16) : nullptr;
        g.out = out + (int64_t) t0 * M;
        g.res = res ? res + (int64_t) t0 * M : nullptr;
        g.M = (int) M; g.K = (int) K;
        int rc = -1;
        switch (w->type) {
            case PI_Q4_K: rc = dispatch_nt<PI_Q4_K>(g, nt, stream); break;
            case PI_Q5_K: rc = dispatch_nt<PI_Q5_K>(g, nt, stream); break;
            case PI_Q6_K: rc = dispatch_nt<PI_Q6_K>(g, nt, stream); break;
            case PI_Q4_0: rc = dispatch_nt<PI_Q4_0>(g, nt, stream); break;
            case PI_Q8_0: rc = dispatch_nt<PI_Q8_0>(g, nt, stream); break;
        }
        if (rc) return rc;
    }
    return 0;
}
