// Quantised mat-vec / tiny-batch mat-mat for the decode and verify path (n_tok <= 8 per launch), BIT-EXACT with the
// reference's x86 AVX2 arm.
//
//   out[t][row] = residual[t][row] + vec_dot(W[row], act[t])
//
// Reference: ggml_compute_forward_mul_mat (ggml.c:9388-9597) calling ggml_vec_dot_q4_K_q8_K (ggml-quants.c:5318-5381),
// q5_K (:5967-6050), q6_K (:6741-6810), q4_0_q8_0 (:2500-2523), q8_0_q8_0 (:3584-3605).  That code keeps EIGHT fp32
// accumulators per row — lane k owns bytes 4k..4k+3 of every 32-value group — updates each with one fma per block
// in block order, and folds them with hsum_float_8 (:67-73).  Activation quantisation (Q8_K/Q8_0) and the fp16 paths
// downstream amplify a 1-ulp difference in these sums into 1e-2 logit differences, so the summation order is part of
// the contract: this kernel keeps exactly those eight chains.
//
// Mapping (wave64): a wave is 8 rows x 8 lanes; lane (r, k) IS accumulator lane k of row r.  It walks the row's blocks in
// order; the k-major weight layout (ops_common.h) gives it its share of a block with one 16-byte load, and the block's
// integer sub-sums are exact (v_dot4_i32_i8).  The quantised activations of the tile of blocks in flight are staged in LDS
// once per workgroup (all rows and lanes re-read them; 8 of every 64 lanes hit distinct banks, the rest broadcast).
// Weights are streamed from HBM exactly once, straight into registers.
#include "ops_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define MMVQ_TILE 16            // groups of four 32-blocks staged in LDS per step (Q4_0/Q8_0 kernel)

#include "mmvq_args.h"

static __device__ __forceinline__ int bytes_sub(int v, int c) {            // per-byte v - c for bytes whose result fits int8 (no borrow across bytes)
    return (int) ((((uint32_t) v | 0x80808080u) - (uint32_t) c) ^ 0x80808080u);
}
static __device__ __forceinline__ int ub(uint32_t w, int i) { return (int) ((w >> (8 * i)) & 0xff); }

// folds the 8 lanes of a row exactly like hsum_float_8: (a4+a0, a5+a1, a6+a2, a7+a3) -> (r0+r2, r1+r3) -> s0+s1
static __device__ __forceinline__ float hsum8_lanes(float a) {
    a = a + pi_xor4(a);
    a = a + pi_partner<2>(a);
    a = a + pi_partner<1>(a);
    return a;
}

// ================================================================================================================
// K-quants x Q8_K
//
// The integer part of a block's dot product is exact in any order; only the fp32 accumulation (eight chains per row and
// token, one fma per block in block order) is ordered.  The kernel separates the two:
//
//   produce  work item = (row r, block b) handled by a quad of lanes; lane c of the quad owns accumulator lanes k = 2c and
//            2c+1 (two 16-byte k-records of the block = 64 weights), unpacks the 6-bit scales once for both, and leaves
//            (float) sumi[k] and the block scale product d_x * d_y in a per-wave LDS exchange buffer.  A wave takes 16
//            items at a time — with 4096-wide rows that is one whole row per step — and R rows per tile of 16 blocks;
//            weights stream from HBM exactly once with the next item's loads issued before the current item is touched.
//   chain    lane (p, k), p = (row, token) pair, extends chain k of pair p by the tile's blocks in order:
//            acc = fma(d, sumi, acc) — 16 dependent fmas per tile for up to 8 pairs at once.
//
// Waves never synchronise with each other (the exchange buffer is wave-private), so a launch is M/R independent waves.
// VALU work per 16 weight bytes is what bounds this kernel once HBM is streaming (v_dot4 + scale multiply + nibble
// unpack ~ 60 instructions); see DESIGN.md for the count against the HBM rate.
// ================================================================================================================
#define KQ_TB 16                // blocks per tile
#define MMVQ_ST_NT 4            // widest batch that gets the loop-free (single-tile / tile-per-wave) instances
template <int TYPE> struct kq_raw { uint4 hdr, qa, qb, h; uint32_t d; };     // one lane's share of a block (k = 2c, 2c+1)
template <int TYPE> constexpr bool kq_has_mins = TYPE == PI_Q4_K || TYPE == PI_Q5_K || TYPE == PI_Q2_K;   // Q6_K and Q3_K are scale-only (symmetric) formats

template <int TYPE>
static __device__ __forceinline__ void kq_load(const mmvq_seg & w, uint32_t blk, int c, kq_raw<TYPE> & o) {
    if constexpr (TYPE == PI_Q4_K || TYPE == PI_Q5_K) {
        constexpr int BS = TYPE == PI_Q4_K ? 144 : 176, QOFF = TYPE == PI_Q4_K ? 16 : 48;
        const uint8_t * p = w.p0 + (size_t) blk * BS;
        o.hdr = *(const uint4 *) p;
        // p-half order (runtime.hip: k_repack_k45): vector p holds dwords m = 2p, 2p+1 of accumulator lanes k = 2c and 2c+1
        const uint4 l0 = *(const uint4 *) (p + QOFF + 16 * c), l1 = *(const uint4 *) (p + QOFF + 64 + 16 * c);
        o.qa = make_uint4(l0.x, l0.y, l1.x, l1.y);
        o.qb = make_uint4(l0.z, l0.w, l1.z, l1.w);
        if constexpr (TYPE == PI_Q5_K) { const uint2 hh = *(const uint2 *) (p + 16 + 8 * c); o.h.x = hh.x; o.h.y = hh.y; }
    } else if constexpr (TYPE == PI_Q2_K) {                                  // runtime.hip: k_repack_q2_K
        o.qa  = *(const uint4 *) (w.p0 + (size_t) blk * 64 + 16 * c);        // qs: half 0 (elements 8c.., 8c+4..), half 1 (same)
        o.hdr = *(const uint4 *) (w.p1 + (size_t) blk * 16);                 // sixteen (scale | min << 4)
        o.d   = ((const uint32_t *) w.p2)[blk];                              // d | dmin << 16
    } else if constexpr (TYPE == PI_Q3_K) {                                  // runtime.hip: k_repack_q3_K
        const uint8_t * p = w.p0 + (size_t) blk * 96;
        o.qa  = *(const uint4 *) (p + 16 * c);                               // qs: half 0 (elements 8c.., 8c+4..), half 1 (same)
        const uint2 hh = *(const uint2 *) (p + 64 + 8 * c); o.h.x = hh.x; o.h.y = hh.y;     // hmask of elements 8c.., 8c+4..
        o.hdr = *(const uint4 *) (w.p1 + (size_t) blk * 16);                 // sixteen int8 scales, already - 32
        o.d   = ((const uint16_t *) w.p2)[blk];
    } else {
        const uint8_t * p = w.p0 + (size_t) blk * 192;
        const uint4 l0 = *(const uint4 *) (p + 16 * c), l1 = *(const uint4 *) (p + 64 + 16 * c);      // halves n = 0, 1: (ql[64n + 4k], ql[64n + 32 + 4k]) of k = 2c, 2c+1
        o.qa  = make_uint4(l0.x, l0.y, l1.x, l1.y);
        o.qb  = make_uint4(l0.z, l0.w, l1.z, l1.w);
        o.h   = *(const uint4 *) (p + 128 + 16 * c);
        o.hdr = *(const uint4 *) (w.p1 + (size_t) blk * 16);
        o.d   = ((const uint16_t *) w.p2)[blk];
    }
}

// Q2_K: group g = 4j + s is (qs dword of half j >> 2s) & 3 per byte, unsigned 0..3 (ggml-quants.c:3745-3748)
static __device__ __forceinline__ void kq_unpack_q2(uint32_t q_half0, uint32_t q_half1, int * wq) {
#pragma unroll
    for (int g = 0; g < 8; g++) wq[g] = (int) (((g < 4 ? q_half0 : q_half1) >> (2 * (g & 3))) & 0x03030303u);
}

// Q3_K: group g = 4j + s of the block is (qs dword of half j >> 2s) & 3 per byte, | high bit g of the hmask dword << 2, minus 4
// (ggml-quants.c:4486-4520: the AVX2 arm subtracts 4 where the high bit is NOT set — the same thing)
static __device__ __forceinline__ void kq_unpack_q3(uint32_t q_half0, uint32_t q_half1, uint32_t hm, int * wq) {
#pragma unroll
    for (int g = 0; g < 8; g++) {
        const uint32_t q = g < 4 ? q_half0 : q_half1;
        wq[g] = bytes_sub((int) (((q >> (2 * (g & 3))) & 0x03030303u) | (((hm >> g) & 0x01010101u) << 2)), 0x04040404);
    }
}

// 32 weights of accumulator lane k as 8 dwords in activation order (dword i pairs with sub-block / group i)
template <int TYPE>
static __device__ __forceinline__ void kq_unpack_w(const uint4 qv, uint32_t h0, uint32_t h1, int * wq) {
    const int q[4] = { (int) qv.x, (int) qv.y, (int) qv.z, (int) qv.w };
    if constexpr (TYPE == PI_Q4_K) {
#pragma unroll
        for (int j = 0; j < 4; j++) { wq[2 * j] = q[j] & 0x0F0F0F0F; wq[2 * j + 1] = (q[j] >> 4) & 0x0F0F0F0F; }
    } else if constexpr (TYPE == PI_Q5_K) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t h = h0 >> (2 * j);
            wq[2 * j]     = (q[j] & 0x0F0F0F0F)        | (int) ((h & 0x01010101u) << 4);
            wq[2 * j + 1] = ((q[j] >> 4) & 0x0F0F0F0F) | (int) ((h & 0x02020202u) << 3);
        }
    } else {
        const uint32_t hh[2] = { h0, h1 };
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const uint32_t h = hh[n];
            const int la = q[2 * n], lb = q[2 * n + 1];
            wq[4 * n + 0] = bytes_sub((la & 0x0F0F0F0F)        | (int) ((h & 0x03030303u) << 4),        0x20202020);
            wq[4 * n + 1] = bytes_sub((lb & 0x0F0F0F0F)        | (int) (((h >> 2) & 0x03030303u) << 4), 0x20202020);
            wq[4 * n + 2] = bytes_sub(((la >> 4) & 0x0F0F0F0F) | (int) (((h >> 4) & 0x03030303u) << 4), 0x20202020);
            wq[4 * n + 3] = bytes_sub(((lb >> 4) & 0x0F0F0F0F) | (int) (((h >> 6) & 0x03030303u) << 4), 0x20202020);
        }
    }
}

template <int TYPE, int NP> struct kq_lds {                                  // floats per wave
    static constexpr int SSTR = KQ_TB * 8 + 8, PSTR = KQ_TB + 1, MSTR = KQ_TB * 4 + 4;      // strides padded to spread banks
    static constexpr int S = 0, P = S + NP * SSTR, SM = P + NP * PSTR,
                         PM = SM + (TYPE == PI_Q4_K ? NP * MSTR : TYPE == PI_Q5_K ? NP * PSTR : TYPE == PI_Q2_K ? NP * SSTR : 0),
                         FLOATS = (PM + (kq_has_mins<TYPE> ? NP * PSTR : 0) + 3) & ~3;
};

static __device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int TYPE> struct kq_act { int y[2][8]; float dy; int bsA, bsB; };      // one token's share of a block for lanes 2c, 2c+1

template <int TYPE>
static __device__ __forceinline__ void kq_load_act(const mmvq_args & g, size_t ab, int c, kq_act<TYPE> & a) {
    const uint4 * yp = (const uint4 *) (g.aq + ab * 256 + 64 * c);
    const uint4 y0 = yp[0], y1 = yp[1], y2 = yp[2], y3 = yp[3];
    a.y[0][0] = (int) y0.x; a.y[0][1] = (int) y0.y; a.y[0][2] = (int) y0.z; a.y[0][3] = (int) y0.w;
    a.y[0][4] = (int) y1.x; a.y[0][5] = (int) y1.y; a.y[0][6] = (int) y1.z; a.y[0][7] = (int) y1.w;
    a.y[1][0] = (int) y2.x; a.y[1][1] = (int) y2.y; a.y[1][2] = (int) y2.z; a.y[1][3] = (int) y2.w;
    a.y[1][4] = (int) y3.x; a.y[1][5] = (int) y3.y; a.y[1][6] = (int) y3.z; a.y[1][7] = (int) y3.w;
    a.dy = ((const float *) g.ad)[ab];
    a.bsA = a.bsB = 0;
    if constexpr (kq_has_mins<TYPE>) {                                       // int16 pair sums like _mm_hadd_epi16
        const short4 bs = *(const short4 *) (g.abs_ + ab * 32 + 8 * c);
        if constexpr (TYPE == PI_Q2_K) { a.bsA = (int) (((uint32_t) (uint16_t) bs.x) | ((uint32_t) (uint16_t) bs.y << 16)); a.bsB = (int) (((uint32_t) (uint16_t) bs.z) | ((uint32_t) (uint16_t) bs.w << 16)); }   // the four raw sums
        else { a.bsA = (int) (short) (bs.x + bs.y); a.bsB = (int) (short) (bs.z + bs.w); }
    }
}

// ST = 1: instance for rows of a single tile (K <= 4096) at decode width — without the tile loop and its second buffers it fits
// 128 VGPRs = 4 waves per SIMD, and the bytes in flight per CU are what a single-round launch lives on
// ST = 2: decode instance for longer rows — "tile per wave": the workgroup is one row group, wave t takes tile t exactly like the
// single-tile instance (so every tile of the row is in flight at once and nothing loops), the products meet in LDS and wave 0 runs
// the ordered chains through the tiles in block order
// LSTR: floats of LDS per wave (== kq_lds<TYPE, NP>::FLOATS, or the larger of two types' when waves of two weight types share a workgroup: k_mmvq_kq2)
template <int TYPE, int NT, int R, int ST, int LSTR>
static __device__ __forceinline__ void mmvq_kq_body(const mmvq_args & g, uint8_t * lds) {
    static_assert(ST == 0 || (ST != 0 && NT <= MMVQ_ST_NT), "loop-free instances: decode and small verify batches");
    constexpr bool TPW = ST == 2;
    constexpr int NP = R * NT, NIT = (NP + 7) / 8;
    constexpr int TG = 4;                                                    // tokens whose activations are held in registers at once
    using L = kq_lds<TYPE, NP>;
    static_assert(LSTR >= L::FLOATS, "LDS stride per wave");
    // the wave index is made a scalar: with it in a VGPR, g.seg[si] turns into per-lane global loads of the kernel arguments — two
    // dependent memory round trips before the first weight load of every wave
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float * xs = (float *) lds + (size_t) wave * LSTR;
    const int q = lane >> 2, c = lane & 3;                                   // producer role: quad q, accumulator lanes 2c, 2c+1
    const int s = lane >> 3, k = lane & 7;                                   // chain role: pair slot s, accumulator lane k
    const int wave_id = TPW ? (int) blockIdx.x : (int) (blockIdx.x * (blockDim.x >> 6) + wave);   // index of the row group
    const int si = (wave_id >= g.seg[1].wave0) + (wave_id >= g.seg[2].wave0);
    const mmvq_seg & sg = g.seg[si];
    const int64_t row0 = (int64_t) (wave_id - sg.wave0) * R;
    if (row0 >= sg.M) return;                                                // waves never synchronise with each other (TPW: the whole group leaves)
    const int nb = g.nb;
    const int rows = sg.M - row0 < R ? (int) (sg.M - row0) : R;

    float acc[NIT], accm[NIT];
#pragma unroll
    for (int ci = 0; ci < NIT; ci++) { acc[ci] = 0.0f; accm[ci] = 0.0f; }

    // quad q -> (row offset, block) of a tile; the block is the same for every step of the tile, so the activations are too
    auto tile_map = [&](int b0, int & nbt, int & rpi, int & r_off, int & b) {
        nbt = nb - b0 < KQ_TB ? nb - b0 : KQ_TB;
        rpi = nbt == KQ_TB ? 1 : KQ_TB / nbt;                                // rows per step
        r_off = nbt == KQ_TB ? 0 : q / nbt;
        b = nbt == KQ_TB ? q : q - r_off * nbt;
    };
    // ---- ordered part: lane (p, k) extends chain k of pair p by a tile's blocks, in block order ----
    auto chain = [&](const float * xt, int nbt, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int bb = 0; bb < KQ_TB; bb++) {
            if (FULL || bb < nbt) {
#pragma unroll
                for (int ci = 0; ci < NIT; ci++) {
                    const int p = ci * 8 + s;
                    if (NP % 8 == 0 || p < NP) {
                        if constexpr (TYPE == PI_Q2_K) acc[ci] = fmaf(xt[L::PM + p * L::PSTR + bb], xt[L::SM + p * L::SSTR + bb * 8 + k], acc[ci]);   // the mins go into the same chain, first
                        acc[ci] = fmaf(xt[L::P + p * L::PSTR + bb], xt[L::S + p * L::SSTR + bb * 8 + k], acc[ci]);
                        if constexpr (TYPE == PI_Q4_K)      accm[ci] = fmaf(xt[L::PM + p * L::PSTR + bb], xt[L::SM + p * L::MSTR + bb * 4 + (k & 3)], accm[ci]);
                        else if constexpr (TYPE == PI_Q5_K) accm[ci] = accm[ci] + xt[L::PM + p * L::PSTR + bb] * xt[L::SM + p * L::PSTR + bb];   // summs += dmin * hsum (mul, then add)
                    }
                }
            }
        }
    };
    // every weight load of a tile (R items) is issued before the previous tile is consumed: loads are unconditional, indices clamped
    auto tile_load = [&](int b0, kq_raw<TYPE> * raw) {
        int nbt, rpi, r_off, b;
        tile_map(b0 < nb ? b0 : 0, nbt, rpi, r_off, b);
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int r = r_off + i * rpi;
            kq_load<TYPE>(sg, (uint32_t) ((row0 + (r < rows ? r : rows - 1)) * nb + (b0 < nb ? b0 : 0) + b), c, raw[i]);
        }
    };
    // two tile buffers used alternately (no copy between them: a conditional `raw = nxt` made the compiler move two dwords of the
    // just-requested tile right behind its loads, i.e. wait for the next tile before touching the current one)
    // decode widths (NT <= 2) also keep two sets of activations and request a tile's activations BEFORE its weights: vmcnt retires
    // in order, so the (L2-resident) activations are usable while the weights are still on their way, instead of being requested
    // only after the weights have arrived — one L2 round trip less on every tile
    constexpr bool ACT2 = NT <= 2;
    constexpr bool WIDE = NT >= 7;
    constexpr int NG_MAX = NT < TG ? NT : TG;
    auto act_load = [&](int b0, kq_act<TYPE> (&dst)[NG_MAX]) __attribute__((always_inline)) {
        int nbt, rpi, r_off, b;
        tile_map(b0, nbt, rpi, r_off, b);
#pragma unroll
        for (int tt = 0; tt < NG_MAX; tt++) kq_load_act<TYPE>(g, (size_t) tt * nb + b0 + b, c, dst[tt]);
    };
    kq_raw<TYPE> bufA[R], bufB[R];
    kq_act<TYPE> actA[NG_MAX], actB[NG_MAX];
    auto tile_step = [&](int b0, kq_raw<TYPE> (&raw)[R], kq_raw<TYPE> (&nxt)[R], kq_act<TYPE> (&actc)[NG_MAX], kq_act<TYPE> (&actn)[NG_MAX], auto pf_tag) __attribute__((always_inline)) {
        constexpr bool PF = decltype(pf_tag)::value;                          // rows of more than one tile prefetch the next one
        int nbt, rpi, r_off, b;
        tile_map(b0, nbt, rpi, r_off, b);
        const bool quad_on = r_off < rpi;
        // the common case — a full tile of 16 blocks, all R rows present — runs without the per-item and per-block guards
        auto tile_body = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            if constexpr (WIDE) { if (nb > KQ_TB) tile_load(b0 + KQ_TB, nxt); }   // wide batches: next tile first, one code instance (below)
    #pragma unroll
            for (int t0 = 0; t0 < NT; t0 += TG) {                                // token groups: activations stay in registers across rows
                kq_act<TYPE> actl[NG_MAX];                                       // (unused, and removed, when ACT2)
                if constexpr (!ACT2) {
    #pragma unroll
                    for (int tt = 0; tt < NG_MAX; tt++) if (t0 + tt < NT) kq_load_act<TYPE>(g, (size_t) (t0 + tt) * nb + b0 + b, c, actl[tt]);
                } else if constexpr (PF) {
                    act_load(b0 + KQ_TB < nb ? b0 + KQ_TB : b0, actn);
                }
                kq_act<TYPE> (&act)[NG_MAX] = ACT2 ? actc : actl;
                // the next tile is requested AFTER the (last group's) activations: vmcnt retires in order, so a wait for loads issued
                // behind the prefetch would also wait for the prefetch
                // — and unconditionally: a branch here makes the compiler's wait counts assume the worst of both paths (drain all);
                // the last tile re-requests itself (an L2 hit nobody waits for)
                if (PF && !WIDE && t0 + TG >= NT) tile_load(b0 + KQ_TB < nb ? b0 + KQ_TB : b0, nxt);
    #pragma unroll
                for (int i = 0; i < R; i++) {
                    const int r = r_off + i * rpi;
                    if (FULL || i * rpi < rows) {                                // uniform: some quad still has a row in this step
                        const bool valid = FULL || (quad_on && r < rows);
                        // ---- block-level unpack, shared by this lane's two accumulator lanes ----
                        int wq[2][8], sc[8], m0 = 0, m1 = 0;
                        float dx, dminx = 0.0f;
                        if constexpr (TYPE == PI_Q4_K || TYPE == PI_Q5_K) {
                            const uint4 hdr = raw[i].hdr;
                            dx = pi_h2f((uint16_t) (hdr.x & 0xffff)); dminx = pi_h2f((uint16_t) (hdr.x >> 16));
                            // 6-bit scales/mins (get_scale_min_k4, ggml-quants.c:1446-1453) unpacked four at a time like the reference's utmp[]
                            const uint32_t scA = hdr.y & 0x3f3f3f3fu, mA = hdr.z & 0x3f3f3f3fu;
                            const uint32_t scB = (hdr.w & 0x0f0f0f0fu) | (((hdr.y >> 6) & 0x03030303u) << 4);
                            const uint32_t mB  = ((hdr.w >> 4) & 0x0f0f0f0fu) | (((hdr.z >> 6) & 0x03030303u) << 4);
    #pragma unroll
                            for (int j = 0; j < 4; j++) { sc[j] = ub(scA, j); sc[4 + j] = ub(scB, j); }
                            const uint32_t mw = (c & 2) ? mB : mA;               // this lane is mins lane c: m[2c], m[2c+1]
                            const int sh = (c & 1) * 16;
                            m0 = (int) ((mw >> sh) & 0xff); m1 = (int) ((mw >> (sh + 8)) & 0xff);
                        } else if constexpr (TYPE == PI_Q2_K) {
                            dx = pi_h2f((uint16_t) (raw[i].d & 0xffff)); dminx = pi_h2f((uint16_t) (raw[i].d >> 16));
                            const uint32_t s4[4] = { raw[i].hdr.x, raw[i].hdr.y, raw[i].hdr.z, raw[i].hdr.w };
                            const int hi = c >> 1;
    #pragma unroll
                            for (int j = 0; j < 8; j++) {                        // scale of group j for these lanes: low nibble of byte 2j + hi
                                const int idx = 8 * (j >> 2) + 2 * (j & 3);
                                sc[j] = (int) ((s4[idx >> 2] >> (8 * ((idx & 3) + hi))) & 0xf);
                            }
                            const uint32_t mw = ((c == 0 ? s4[0] : c == 1 ? s4[1] : c == 2 ? s4[2] : s4[3]) >> 4) & 0x0f0f0f0fu;      // mins 4c..4c+3
                            m0 = (int) mw;                                       // (used byte-wise below)
                        } else {
                            dx = pi_h2f((uint16_t) raw[i].d);
                            const uint32_t s4[4] = { raw[i].hdr.x, raw[i].hdr.y, raw[i].hdr.z, raw[i].hdr.w };
                            const int hi = c >> 1;                               // k = 2c, 2c+1 share k >> 2
    #pragma unroll
                            for (int j = 0; j < 8; j++) {                        // j = 4n + g: scale index 8n + 2g + hi
                                const int idx = 8 * (j >> 2) + 2 * (j & 3);
                                sc[j] = (int) (int8_t) ((s4[idx >> 2] >> (8 * ((idx & 3) + hi))) & 0xff);
                            }
                        }
                        if constexpr (TYPE == PI_Q3_K) {
                            kq_unpack_q3(raw[i].qa.x, raw[i].qa.z, raw[i].h.x, wq[0]);
                            kq_unpack_q3(raw[i].qa.y, raw[i].qa.w, raw[i].h.y, wq[1]);
                        } else if constexpr (TYPE == PI_Q2_K) {
                            kq_unpack_q2(raw[i].qa.x, raw[i].qa.z, wq[0]);
                            kq_unpack_q2(raw[i].qa.y, raw[i].qa.w, wq[1]);
                        } else {
                            kq_unpack_w<TYPE>(raw[i].qa, raw[i].h.x, TYPE == PI_Q6_K ? raw[i].h.y : 0u, wq[0]);
                            kq_unpack_w<TYPE>(raw[i].qb, TYPE == PI_Q6_K ? raw[i].h.z : raw[i].h.y, TYPE == PI_Q6_K ? raw[i].h.w : 0u, wq[1]);
                        }
    #pragma unroll
                        for (int tt = 0; tt < NG_MAX; tt++) {
                            if (t0 + tt < NT) {
                                const kq_act<TYPE> & a = act[tt];
                                int sumi0 = 0, sumi1 = 0;
    #pragma unroll
                                for (int j = 0; j < 8; j++) { sumi0 += __mul24(sc[j], pi_dot4(wq[0][j], a.y[0][j], 0)); sumi1 += __mul24(sc[j], pi_dot4(wq[1][j], a.y[1][j], 0)); }
                                const int p = r * NT + t0 + tt;
                                if (valid) {
                                    *(float2 *) (xs + L::S + p * L::SSTR + b * 8 + 2 * c) = make_float2((float) sumi0, (float) sumi1);
                                    if (c == 0) xs[L::P + p * L::PSTR + b] = a.dy * dx;
                                }
                                if constexpr (TYPE == PI_Q2_K) {
                                    // madd_epi16(mins, bsums): lane k = m[2k]*bs[2k] + m[2k+1]*bs[2k+1]; this quad lane holds k = 2c, 2c+1
                                    const int p0 = __mul24(ub((uint32_t) m0, 0), (int) (short) (a.bsA & 0xffff)) + __mul24(ub((uint32_t) m0, 1), (int) (short) ((uint32_t) a.bsA >> 16));
                                    const int p1 = __mul24(ub((uint32_t) m0, 2), (int) (short) (a.bsB & 0xffff)) + __mul24(ub((uint32_t) m0, 3), (int) (short) ((uint32_t) a.bsB >> 16));
                                    if (valid) {
                                        *(float2 *) (xs + L::SM + p * L::SSTR + b * 8 + 2 * c) = make_float2((float) p0, (float) p1);
                                        if (c == 0) xs[L::PM + p * L::PSTR + b] = -a.dy * dminx;
                                    }
                                } else if constexpr (kq_has_mins<TYPE>) {
                                    // mins: prod[c] = m[2c]*(bsums[4c]+bsums[4c+1]) + m[2c+1]*(bsums[4c+2]+bsums[4c+3])
                                    int prod = __mul24(m0, a.bsA) + __mul24(m1, a.bsB);
                                    if constexpr (TYPE == PI_Q4_K) {
                                        if (valid) xs[L::SM + p * L::MSTR + b * 4 + c] = (float) prod;
                                    } else {
                                        prod += pi_partner<1>(prod); prod += pi_partner<2>(prod);
                                        if (valid && c == 0) xs[L::SM + p * L::PSTR + b] = (float) prod;
                                    }
                                    if (valid && c == 0) xs[L::PM + p * L::PSTR + b] = -a.dy * dminx;
                                }
                            }
                        }
                    }
                }
            }
            if constexpr (!TPW) {
                wave_lds_sync();
                chain(xs, nbt, full_tag);
                wave_lds_sync();
            }
        };
        if (nbt == KQ_TB && rows == R) tile_body(std::true_type{}); else tile_body(std::false_type{});
    };
    // (the first tile is requested inside each branch: requested before it, the two paths want it in different registers and the
    // multi-tile path starts by waiting for all of it in order to copy it)
    if constexpr (TPW) {
        const int b0 = wave * KQ_TB;                                           // the launch has one wave per tile
        if constexpr (ACT2) act_load(b0, actA);
        tile_load(b0, bufA);
        tile_step(b0, bufA, bufB, actA, actB, std::false_type{});
        __syncthreads();
        if (wave != 0) return;
        for (int t = 0, b = 0; b < nb; t++, b += KQ_TB) {
            const float * xt = (const float *) lds + (size_t) t * LSTR;
            if (nb - b >= KQ_TB) chain(xt, KQ_TB, std::true_type{}); else chain(xt, nb - b, std::false_type{});
        }
    } else if constexpr (WIDE) {
        // NT >= 7 is VALU-bound and at the register limit: the three step instances of the scheme below cost it a wave of occupancy
        // (256 VGPRs), so it keeps one instance with a copy between the two buffers
        tile_load(0, bufA);
        for (int b0 = 0; b0 < nb; b0 += KQ_TB) {
            tile_step(b0, bufA, bufB, actA, actB, std::false_type{});
            if (nb > KQ_TB) {
#pragma unroll
                for (int i = 0; i < R; i++) bufA[i] = bufB[i];
            }
        }
    } else if (ST || nb <= KQ_TB) {
        if constexpr (ACT2) act_load(0, actA);
        tile_load(0, bufA);
        tile_step(0, bufA, bufB, actA, actB, std::false_type{});
    } else if constexpr (!ST) {
        if constexpr (ACT2) act_load(0, actA);
        tile_load(0, bufA);
        for (int b0 = 0; b0 < nb; b0 += 2 * KQ_TB) {
            tile_step(b0, bufA, bufB, actA, actB, std::true_type{});
            if (b0 + KQ_TB < nb) tile_step(b0 + KQ_TB, bufB, bufA, actB, actA, std::true_type{});
        }
    }
    // ---- fold the eight chains of each pair ----
#pragma unroll
    for (int ci = 0; ci < NIT; ci++) {
        const int p = ci * 8 + s;
        float v = hsum8_lanes(acc[ci]);
        if constexpr (TYPE == PI_Q4_K) {
            float mv = (k < 4) ? accm[ci] : 0.0f;
            mv = mv + pi_partner<2>(mv);                                        // (m0+m2, m1+m3)
            mv = mv + pi_partner<1>(mv);                                        // (m0+m2) + (m1+m3)
            v = v + mv;
        } else if constexpr (TYPE == PI_Q5_K) {
            v = v + accm[ci];
        }
        const int r = p / NT, t = p % NT;
        if (p < NP && k == 0 && r < rows) {
            if (sg.store) xs[p] = v;                                           // RoPE needs the row pair: collect the wave's results first
            else {
                const int64_t o = (int64_t) t * sg.M + row0 + r;
                sg.out[o] = sg.res ? sg.res[o] + v : v;
            }
        }
    }
    if (sg.store) {
        // llm_build_kv_store / ggml_rope mode 0 (llama.cpp:3546-3577, ggml.c:11062-11163) as the epilogue of the projection: rows 2i, 2i+1
        // of a head are neighbours in this wave (R even, row0 multiple of R); the rounding points are those of k_rope_kv
        wave_lds_sync();
        if (sg.store == 3) {
            for (int j = lane; j < rows * NT; j += 64) {
                const int r = j / NT, t = j % NT;
                sg.dst16[(int64_t) pi_cell_of(g.dyn, g.kv_head, g.tok0 + t) * sg.stride16 + row0 + r] = pi_f2h(xs[j]);
            }
        } else {
            for (int j = lane; j < (rows / 2) * NT; j += 64) {
                const int pr = j / NT, t = j % NT;
                const float x0 = xs[(2 * pr) * NT + t], x1 = xs[(2 * pr + 1) * NT + t];
                const int64_t row = row0 + 2 * pr;
                const int i = (int) (row % g.hd) >> 1;                       // pair index inside the head
                const float2 cs = *(const float2 *) (g.rope_cs + ((int64_t) g.pos[g.tok0 + t] * (g.hd / 2) + i) * 2);
                uint16_t * d = sg.dst16 + (int64_t) (sg.store == 2 ? pi_cell_of(g.dyn, g.kv_head, g.tok0 + t) : g.tok0 + t) * sg.stride16 + row;
                d[0] = pi_f2h(x0 * cs.x - x1 * cs.y);
                d[1] = pi_f2h(x0 * cs.y + x1 * cs.x);
            }
        }
    }
}

template <int TYPE, int NT, int R, int ST>
__global__ void __launch_bounds__(ST == 2 ? (NT <= 2 ? 1024 : 128) : 256) __attribute__((amdgpu_waves_per_eu(ST && NT <= 2 ? 4 : 1))) k_mmvq_kq(const mmvq_args g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if (pi_aborted(g.abort_w, 0)) return;
    mmvq_kq_body<TYPE, NT, R, ST, kq_lds<TYPE, R * NT>::FLOATS>(g, lds);
}

// Two weight types in ONE launch (decode widths): the attention projections of the Q4_K_M / Q5_K_M mixtures keep attn_q / attn_k in Q4_K and put attn_v into Q5_K or
// Q6_K (llama.cpp:7923-7933) — 1024 rows of the 70B model, a launch of its own that runs at 0.04-0.10 of the HBM peak because it is all launch ramp.  Segments whose bit
// is set in g.type_b_mask run the TB body, the others the TA body; the choice is uniform per wave (per workgroup in the tile-per-wave instance), both bodies compute
// exactly what their single-type kernels compute.
template <int TA, int TB, int NT, int R, int ST>
__global__ void __launch_bounds__(ST == 2 ? (NT <= 2 ? 1024 : 128) : 256) __attribute__((amdgpu_waves_per_eu(ST && NT <= 2 ? 4 : 1))) k_mmvq_kq2(const mmvq_args g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if (pi_aborted(g.abort_w, 0)) return;
    constexpr int FA = kq_lds<TA, R * NT>::FLOATS, FB = kq_lds<TB, R * NT>::FLOATS, LSTR = FA > FB ? FA : FB;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave_id = ST == 2 ? (int) blockIdx.x : (int) (blockIdx.x * (blockDim.x >> 6) + wave);
    const int si = (wave_id >= g.seg[1].wave0) + (wave_id >= g.seg[2].wave0);
    if ((g.type_b_mask >> si) & 1) mmvq_kq_body<TB, NT, R, ST, LSTR>(g, lds);
    else                           mmvq_kq_body<TA, NT, R, ST, LSTR>(g, lds);
}

// ================================================================================================================
// Q4_0 / Q8_0 x Q8_0 at decode widths: the workgroup is 8 rows, wave w takes the groups [w*G, (w+1)*G) of those rows — every
// group of the rows is in flight at once (8x the waves of k_mmvq_q0 below) — and keeps its (d_x*d_y, integer dot) pairs in
// registers; the ordered fma chains (ggml-quants.c:2500-2523, 3584-3605: one fma per 32-block, blocks in order) then pass from
// wave to wave through LDS, 4*G dependent fma per hand-over.
// ================================================================================================================
template <int TYPE, int NT, int G>
__global__ void __launch_bounds__(1024) k_mmvq_q0r(const mmvq_args g) {
    const mmvq_seg & sg = g.seg[0];
    __shared__ float hand[NT][64];
    if (pi_aborted(g.abort_w, 0)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int r = lane >> 3, k = lane & 7;
    const int64_t row = (int64_t) blockIdx.x * 8 + r;
    const bool live = row < sg.M;
    const int64_t rowc = live ? row : sg.M - 1;
    const int ng = g.nb, g_lo = wave * G;
    // activations (L2-resident) are requested before the weights: vmcnt retires in order
    uint4 yv[NT][G]; uint2 ydv[NT][G];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int gi = 0; gi < G; gi++) {
            const int gg = g_lo + gi < ng ? g_lo + gi : ng - 1;
            yv[t][gi]  = *(const uint4 *) (g.aq + ((size_t) t * ng + gg) * 128 + 16 * k);
            ydv[t][gi] = *(const uint2 *) (g.ad + ((size_t) t * ng + gg) * 8);
        }
    uint4 wx[G]; uint2 wd[G];
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        const int64_t grp = rowc * ng + (g_lo + gi < ng ? g_lo + gi : ng - 1);
        if constexpr (TYPE == PI_Q4_0) wx[gi] = *(const uint4 *) (sg.p0 + grp * 64 + 16 * (k & 3));
        else                           wx[gi] = *(const uint4 *) (sg.p0 + grp * 128 + 16 * k);
        wd[gi] = *(const uint2 *) (sg.p1 + grp * 8);
    }
    float sc[NT][G][4], pr[NT][G][4];
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        const uint32_t x4[4] = { wx[gi].x, wx[gi].y, wx[gi].z, wx[gi].w };
        int xs[4]; float dx[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            if constexpr (TYPE == PI_Q4_0) xs[b] = bytes_sub((int) (((k & 4) ? (x4[b] >> 4) : x4[b]) & 0x0F0F0F0Fu), 0x08080808);
            else                           xs[b] = (int) x4[b];
            dx[b] = pi_h2f((uint16_t) (((b < 2 ? wd[gi].x : wd[gi].y) >> (16 * (b & 1))) & 0xffff));
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int y4[4] = { (int) yv[t][gi].x, (int) yv[t][gi].y, (int) yv[t][gi].z, (int) yv[t][gi].w };
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const float dy = pi_h2f((uint16_t) (((b < 2 ? ydv[t][gi].x : ydv[t][gi].y) >> (16 * (b & 1))) & 0xffff));
                sc[t][gi][b] = dx[b] * dy;
                pr[t][gi][b] = (float) pi_dot4(xs[b], y4[b], 0);
            }
        }
    }
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = 0.0f;
    for (int w = 0; w < nw; w++) {                                           // ordered part: wave w continues where wave w - 1 stopped
        if (w == wave) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (w > 0) acc[t] = hand[t][lane];
#pragma unroll
                for (int gi = 0; gi < G; gi++)
                    if (g_lo + gi < ng) {
#pragma unroll
                        for (int b = 0; b < 4; b++) acc[t] = fmaf(sc[t][gi][b], pr[t][gi][b], acc[t]);
                    }
                hand[t][lane] = acc[t];
            }
        }
        __syncthreads();
    }
    if (wave != nw - 1) return;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float v = hsum8_lanes(acc[t]);
        if (live && k == 0) {
            const int64_t o = (int64_t) t * sg.M + row;
            sg.out[o] = sg.res ? sg.res[o] + v : v;
        }
    }
}

// ================================================================================================================
// Q4_0 / Q8_0 x Q8_0: lane (r, k) walks groups of four 32-blocks
// ================================================================================================================
template <int TYPE, int NT>
__global__ void __launch_bounds__(256) k_mmvq_q0(const mmvq_args g) {
    const mmvq_seg & sg = g.seg[0];
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if (pi_aborted(g.abort_w, 0)) return;
    // LDS tile: qs [NT][TILE groups][128] | d [NT][TILE][4] f16
    uint8_t * l_qs = lds;
    uint16_t * l_d = (uint16_t *) (lds + (size_t) NT * MMVQ_TILE * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 3, k = lane & 7;
    const int64_t row = (int64_t) blockIdx.x * (blockDim.x >> 3) + wave * 8 + r;
    const bool live = row < sg.M;
    const int64_t rowc = live ? row : sg.M - 1;
    const int ng = g.nb;
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = 0.0f;

    for (int g0 = 0; g0 < ng; g0 += MMVQ_TILE) {
        const int ngt = min(MMVQ_TILE, ng - g0);
        uint4 w_x[MMVQ_TILE]; uint2 w_dv[MMVQ_TILE];                      // all weight loads of the tile up front (see k_mmvq_kq)
#pragma unroll
        for (int gi = 0; gi < MMVQ_TILE; gi++) {
            {
                const int64_t grp = rowc * ng + g0 + (gi < ngt ? gi : ngt - 1);
                if constexpr (TYPE == PI_Q4_0) w_x[gi] = *(const uint4 *) (sg.p0 + grp * 64 + 16 * (k & 3));
                else                           w_x[gi] = *(const uint4 *) (sg.p0 + grp * 128 + 16 * k);
                w_dv[gi] = *(const uint2 *) (sg.p1 + grp * 8);
            }
        }
        __syncthreads();
        for (int u = tid; u < NT * ngt * 8; u += blockDim.x) {
            const int t = u / (ngt * 8), rem = u % (ngt * 8);
            *(uint4 *) (l_qs + ((size_t) t * MMVQ_TILE * 128) + rem * 16) = *(const uint4 *) (g.aq + ((size_t) t * ng + g0) * 128 + (size_t) rem * 16);
        }
        for (int u = tid; u < NT * ngt; u += blockDim.x) {
            const int t = u / ngt, gi = u % ngt;
            *(uint2 *) (l_d + ((size_t) t * MMVQ_TILE + gi) * 4) = *(const uint2 *) (g.ad + ((size_t) t * ng + g0 + gi) * 8);
        }
        __syncthreads();
        if (!live) continue;
#pragma unroll
        for (int gi = 0; gi < MMVQ_TILE; gi++) {
          if (gi < ngt) {
            const uint4 xv = w_x[gi];
            const uint2 dv = w_dv[gi];
            const uint32_t x4[4] = { xv.x, xv.y, xv.z, xv.w };
            int xs[4]; float dx[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if constexpr (TYPE == PI_Q4_0) xs[b] = bytes_sub((int) (((k & 4) ? (x4[b] >> 4) : x4[b]) & 0x0F0F0F0Fu), 0x08080808);
                else                           xs[b] = (int) x4[b];
                dx[b] = pi_h2f((uint16_t) (((b < 2 ? dv.x : dv.y) >> (16 * (b & 1))) & 0xffff));
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const uint4 yv = *(const uint4 *) (l_qs + ((size_t) t * MMVQ_TILE + gi) * 128 + 16 * k);
                const uint2 dyv = *(const uint2 *) (l_d + ((size_t) t * MMVQ_TILE + gi) * 4);
                const int y4[4] = { (int) yv.x, (int) yv.y, (int) yv.z, (int) yv.w };
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const float dy = pi_h2f((uint16_t) (((b < 2 ? dyv.x : dyv.y) >> (16 * (b & 1))) & 0xffff));
                    acc[t] = fmaf(dx[b] * dy, (float) pi_dot4(xs[b], y4[b], 0), acc[t]);
                }
            }
          }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float v = hsum8_lanes(acc[t]);
        if (live && k == 0) {
            const int64_t o = (int64_t) t * sg.M + row;
            sg.out[o] = sg.res ? sg.res[o] + v : v;
        }
    }
}

// ================================================================================================================
// Legacy 32-blocks Q4_1 / Q5_0 / Q5_1 (x Q8_1 / Q8_0 / Q8_1): the Q4_0 walk above with the formats' extras —
//   ggml_vec_dot_q4_1_q8_1 (ggml-quants.c:2808-2842), q5_0_q8_0 (:3056-3078), q5_1_q8_1 (:3372-3398), AVX2 arms:
//   acc[k] = fma(d_x * d_y, (float) sum_{i<4} x[4k+i] * y[4k+i], acc[k]) per block in block order (x unsigned for the _1 formats, 5-bit - 16 for
//   Q5_0), and for the _1 formats a NINTH, scalar chain  summs += m_x * s_y  (a multiply, then an add) over the same blocks;
//   result = hsum_float_8(acc) + summs.  Weights: nibbles as Q4_0, (d, m) and qh in planes of their own (runtime.hip: k_repack_legacy).
// Coverage kernel: one instance for every width (no decode-width specialisation like k_mmvq_q0r).
// ================================================================================================================
template <int TYPE, int NT>
__global__ void __launch_bounds__(256) k_mmvq_leg(const mmvq_args g) {
    constexpr bool Y81 = TYPE == PI_Q4_1 || TYPE == PI_Q5_1, HB = TYPE != PI_Q4_1;
    constexpr int DSTR = Y81 ? 16 : 8;                                       // bytes of activation scales per group: 4 floats or 4 f16
    const mmvq_seg & sg = g.seg[0];
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if (pi_aborted(g.abort_w, 0)) return;
    uint8_t * l_qs = lds;                                                    // qs [NT][TILE groups][128] | d [NT][TILE][DSTR] | s [NT][TILE][16]
    uint8_t * l_d = lds + (size_t) NT * MMVQ_TILE * 128;
    float * l_s = (float *) (l_d + (size_t) NT * MMVQ_TILE * DSTR);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 3, k = lane & 7;
    const int64_t row = (int64_t) blockIdx.x * (blockDim.x >> 3) + wave * 8 + r;
    const bool live = row < sg.M;
    const int64_t rowc = live ? row : sg.M - 1;
    const int ng = g.nb, nbr = g.K / 32;                                     // groups per row, real (unpadded) blocks per row
    float acc[NT], summs[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { acc[t] = 0.0f; summs[t] = 0.0f; }

    for (int g0 = 0; g0 < ng; g0 += MMVQ_TILE) {
        const int ngt = min(MMVQ_TILE, ng - g0);
        uint4 w_x[MMVQ_TILE], w_dm[MMVQ_TILE], w_h[MMVQ_TILE];
#pragma unroll
        for (int gi = 0; gi < MMVQ_TILE; gi++) {
            const int64_t grp = rowc * ng + g0 + (gi < ngt ? gi : ngt - 1);
            w_x[gi] = *(const uint4 *) (sg.p0 + grp * 64 + 16 * (k & 3));
            w_dm[gi] = *(const uint4 *) (sg.p1 + grp * 16);
            if constexpr (HB) w_h[gi] = *(const uint4 *) (sg.p2 + grp * 16);
        }
        __syncthreads();
        for (int u = tid; u < NT * ngt * 8; u += blockDim.x) {
            const int t = u / (ngt * 8), rem = u % (ngt * 8);
            *(uint4 *) (l_qs + ((size_t) t * MMVQ_TILE * 128) + rem * 16) = *(const uint4 *) (g.aq + ((size_t) t * ng + g0) * 128 + (size_t) rem * 16);
        }
        for (int u = tid; u < NT * ngt; u += blockDim.x) {
            const int t = u / ngt, gi = u % ngt;
            if constexpr (Y81) {
                *(uint4 *) (l_d + ((size_t) t * MMVQ_TILE + gi) * 16) = *(const uint4 *) (g.ad + ((size_t) t * ng + g0 + gi) * 16);
                *(uint4 *) (l_s + ((size_t) t * MMVQ_TILE + gi) * 4) = *(const uint4 *) (g.abs_ + ((size_t) t * ng + g0 + gi) * 16);
            } else *(uint2 *) (l_d + ((size_t) t * MMVQ_TILE + gi) * 8) = *(const uint2 *) (g.ad + ((size_t) t * ng + g0 + gi) * 8);
        }
        __syncthreads();
        if (!live) continue;
#pragma unroll
        for (int gi = 0; gi < MMVQ_TILE; gi++) {
          if (gi < ngt) {
            const uint32_t x4[4] = { w_x[gi].x, w_x[gi].y, w_x[gi].z, w_x[gi].w };
            const uint32_t h4[4] = { w_h[gi].x, w_h[gi].y, w_h[gi].z, w_h[gi].w };
            int xs[4]; float dx[4], mx[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                uint32_t v = ((k & 4) ? (x4[b] >> 4) : x4[b]) & 0x0F0F0F0Fu;
                if constexpr (HB) v |= ((((h4[b] >> (4 * k)) & 0xFu) * 0x00204081u) & 0x01010101u) << 4;     // bits 4k..4k+3 of qh -> bit 4 of bytes 0..3
                xs[b] = TYPE == PI_Q5_0 ? bytes_sub((int) v, 0x10101010) : (int) v;
                dx[b] = pi_h2f((uint16_t) (((b < 2 ? w_dm[gi].x : w_dm[gi].y) >> (16 * (b & 1))) & 0xffff));
                mx[b] = pi_h2f((uint16_t) (((b < 2 ? w_dm[gi].z : w_dm[gi].w) >> (16 * (b & 1))) & 0xffff));
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const uint4 yv = *(const uint4 *) (l_qs + ((size_t) t * MMVQ_TILE + gi) * 128 + 16 * k);
                const int y4[4] = { (int) yv.x, (int) yv.y, (int) yv.z, (int) yv.w };
                float dy[4], sy[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                if constexpr (Y81) {
                    const float4 dv = *(const float4 *) (l_d + ((size_t) t * MMVQ_TILE + gi) * 16), sv = *(const float4 *) (l_s + ((size_t) t * MMVQ_TILE + gi) * 4);
                    dy[0] = dv.x; dy[1] = dv.y; dy[2] = dv.z; dy[3] = dv.w; sy[0] = sv.x; sy[1] = sv.y; sy[2] = sv.z; sy[3] = sv.w;
                } else {
                    const uint2 dyv = *(const uint2 *) (l_d + ((size_t) t * MMVQ_TILE + gi) * 8);
#pragma unroll
                    for (int b = 0; b < 4; b++) dy[b] = pi_h2f((uint16_t) (((b < 2 ? dyv.x : dyv.y) >> (16 * (b & 1))) & 0xffff));
                }
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    if constexpr (Y81) { if ((g0 + gi) * 4 + b < nbr) { const float prod = mx[b] * sy[b]; summs[t] = summs[t] + prod; } }   // the padding blocks of the last group are not part of the row
                    acc[t] = fmaf(dx[b] * dy[b], (float) pi_dot4(xs[b], y4[b], 0), acc[t]);
                }
            }
          }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        float v = hsum8_lanes(acc[t]);
        if constexpr (Y81) v = v + summs[t];
        if (live && k == 0) {
            const int64_t o = (int64_t) t * sg.M + row;
            sg.out[o] = sg.res ? sg.res[o] + v : v;
        }
    }
}

// ---- launch accounting (kernel_timing.h; implementation in runtime.hip) ----
#include "kernel_timing.h"
// algorithmic bytes of a mat-vec / mat-mat launch: the weight matrices once + quantised activations + fp32 output (+ residual)
static double kt_bytes(int type, int nt, const mmvq_launch & L) {
    const double abytes = (double) nt * ((type == PI_Q4_1 || type == PI_Q5_1) ? (double) L.g.K / 32 * 40 : (type == PI_Q4_0 || type == PI_Q8_0 || type == PI_Q5_0) ? (double) L.g.K / 32 * 34 : (double) L.g.K / 256 * 292);
    const double obytes = (double) nt * L.m_total * 4 * (L.any_res ? 2 : 1);
    return L.wbytes + abytes + obytes;
}
static bool kt_begin(int type, int nt, const mmvq_launch & L, hipEvent_t & ea, hipEvent_t & eb, int fam = PI_KT_MATVEC) { return pi_kt_begin(fam, type, nt, kt_bytes(type, nt, L), ea, eb); }
static void kt_end(int type, int nt, const mmvq_launch & L, hipEvent_t ea, hipEvent_t eb, int fam = PI_KT_MATVEC) { pi_kt_end(fam, type, nt, kt_bytes(type, nt, L), ea, eb); }

template <int TYPE, int NT>
static int launch_mmvq(mmvq_launch & L, hipStream_t stream) {
    mmvq_args & g = L.g;
    // sampled launches go through hipExtLaunchKernelGGL: its start/stop events carry the kernel's own begin/end timestamps (what
    // rocprofv3 reports as the duration); events recorded around a launch would add the dispatch gap (~3 us) to every sample
    hipEvent_t ea = nullptr, eb = nullptr;
    const bool sample = kt_begin(TYPE, L.g.nt, L, ea, eb);
    if constexpr (TYPE == PI_Q4_1 || TYPE == PI_Q5_0 || TYPE == PI_Q5_1) {
        const int64_t M = g.seg[0].M;
        int wpb = 1;
        while (wpb < 4 && M / (8 * wpb * 2) >= 512) wpb *= 2;
        const unsigned grid = (unsigned) ((M + 8 * wpb - 1) / (8 * wpb));
        const size_t lds = (size_t) NT * MMVQ_TILE * (128 + 16 + 16);
        if (sample) hipExtLaunchKernelGGL((k_mmvq_leg<TYPE, NT>), dim3(grid), dim3(64 * wpb), lds, stream, ea, eb, 0, g);
        else        hipLaunchKernelGGL((k_mmvq_leg<TYPE, NT>), dim3(grid), dim3(64 * wpb), lds, stream, g);
    } else if constexpr (TYPE == PI_Q4_0 || TYPE == PI_Q8_0) {
        const int64_t M = g.seg[0].M;
        bool launched = false;
        if constexpr (NT <= 2) {                                   // decode widths: one wave per range of groups (k_mmvq_q0r)
            static const bool no_r = getenv("PI_MMVQ_NO_Q0R") != nullptr;      // tuning / tests
            const unsigned grid = (unsigned) ((M + 7) / 8);
            auto go_r = [&](auto gc) {                                 // G groups per wave, at most 16 waves per workgroup
                constexpr int G = decltype(gc)::value;
                const int nwv = (g.nb + G - 1) / G;
                if (sample) hipExtLaunchKernelGGL((k_mmvq_q0r<TYPE, NT, G>), dim3(grid), dim3(64 * nwv), 0, stream, ea, eb, 0, g);
                else        hipLaunchKernelGGL((k_mmvq_q0r<TYPE, NT, G>), dim3(grid), dim3(64 * nwv), 0, stream, g);
                launched = true;
            };
            if (!no_r && g.nb >= 8) {                                  // (registers: 8 / 12 groups per wave only fit at NT = 1)
                if (g.nb <= 64) go_r(std::integral_constant<int, 4>{});
                else if constexpr (NT == 1) {
                    if (g.nb <= 128) go_r(std::integral_constant<int, 8>{});
                    else if (g.nb <= 192) go_r(std::integral_constant<int, 12>{});
                }
            }
        }
        if (!launched) {
            // waves per workgroup: keep >= ~512 workgroups for the 256 CUs when the matrix is tall enough
            int wpb = 1;
            while (wpb < 4 && M / (8 * wpb * 2) >= 512) wpb *= 2;
            const unsigned grid = (unsigned) ((M + 8 * wpb - 1) / (8 * wpb));
            const size_t lds = (size_t) NT * MMVQ_TILE * (128 + 8);
            if (sample) hipExtLaunchKernelGGL((k_mmvq_q0<TYPE, NT>), dim3(grid), dim3(64 * wpb), lds, stream, ea, eb, 0, g);
            else        hipLaunchKernelGGL((k_mmvq_q0<TYPE, NT>), dim3(grid), dim3(64 * wpb), lds, stream, g);
        }
    } else {
        // rows per wave: pairs (row, token) share the chain phase (8 per pass) and the activation registers, so more rows
        // amortise them; fewer rows give more waves.  Keep >= ~2048 waves when the matrices allow.  PI_MMVQ_R overrides.
        static const int r_env = getenv("PI_MMVQ_R") ? atoi(getenv("PI_MMVQ_R")) : 0;
        // measured on the whole decode: at 7B sizes 2 rows per wave beat 4 (3.57 vs 3.75 ms per token) although 4 win on isolated launches; at 70B sizes
        // (K >= 8192, >= 8192 rows) 4 rows win both ways: 28672 x 8192 29.5 -> 28.1 us, 8192 x 28672 34.3 -> 32.1 us, 86.5 -> 88.0 tok/s (same-box A/B)
        constexpr int RMAX = 4;
        int R = g.K >= 8192 ? 4 : 2;
        while (R > 1 && L.m_total / R < 2048) R >>= 1;
        if (r_env) R = r_env < RMAX ? r_env : RMAX;
        if (R == 3) R = 2;
        if (L.rope && R < 2) R = 2;                               // the RoPE epilogue needs both rows of a pair in one wave
        int waves = 0;
        for (int i = 0; i < MMVQ_MAX_SEG; i++) {
            if (i < L.nseg) { g.seg[i].wave0 = waves; waves += (g.seg[i].M + R - 1) / R; }
            else { g.seg[i] = g.seg[0]; g.seg[i].wave0 = 0x7fffffff; }
        }
        auto go = [&](auto rc) {
            constexpr int RR = decltype(rc)::value;
            if constexpr (RR <= RMAX) {
                constexpr int wpb = RR * NT > 8 ? 2 : 4;
                const unsigned grid = (unsigned) ((waves + wpb - 1) / wpb);
                const size_t lds = (size_t) wpb * kq_lds<TYPE, RR * NT>::FLOATS * 4;
                if constexpr (NT <= MMVQ_ST_NT) {
                    const int ntiles = (g.nb + KQ_TB - 1) / KQ_TB;
                    static const bool no_tpw = getenv("PI_MMVQ_NO_TPW") != nullptr;    // tuning / tests
                    if constexpr (NT <= 4) {                                           // measured per width: up to 4 tiles at NT <= 2, two tiles at NT 3-4 (70B K = 8192: -15 %), the loop otherwise
                        // (Q5_K at TWO tokens is the exception — round 6, us per launch, tile per wave against the loop: 9216 x 8192 27.6 / 23.0, 8192 x 8192 24.0 / 18.3,
                        //  32768 x 8192 71.1 / 53.9; at one token the tile-per-wave form wins as for the other types: 14.1 / 19.8.  profiles/r06_pass_width_dispatch.log)
                        constexpr bool tpw_loses = TYPE == PI_Q5_K && NT == 2;
                        if (ntiles > 1 && ntiles <= (NT <= 2 ? 4 : 2) && !no_tpw && !tpw_loses) {                 // measured: 2-3 tiles win (K = 8192: 11.3 vs 14.5 us), 7 tiles lose (K = 28672: 36.9 vs 34.0; NT = 2: 57 vs 43) — wave 0's serial chain grows with the row
                            const size_t lds_t = (size_t) ntiles * kq_lds<TYPE, RR * NT>::FLOATS * 4;
                            if (sample) hipExtLaunchKernelGGL((k_mmvq_kq<TYPE, NT, RR, 2>), dim3(waves), dim3(64 * ntiles), lds_t, stream, ea, eb, 0, g);
                            else        hipLaunchKernelGGL((k_mmvq_kq<TYPE, NT, RR, 2>), dim3(waves), dim3(64 * ntiles), lds_t, stream, g);
                            return;
                        }
                    }
                    if (g.nb <= KQ_TB) {
                        if (sample) hipExtLaunchKernelGGL((k_mmvq_kq<TYPE, NT, RR, 1>), dim3(grid), dim3(64 * wpb), lds, stream, ea, eb, 0, g);
                        else        hipLaunchKernelGGL((k_mmvq_kq<TYPE, NT, RR, 1>), dim3(grid), dim3(64 * wpb), lds, stream, g);
                        return;
                    }
                }
                if (sample) hipExtLaunchKernelGGL((k_mmvq_kq<TYPE, NT, RR, 0>), dim3(grid), dim3(64 * wpb), lds, stream, ea, eb, 0, g);
                else        hipLaunchKernelGGL((k_mmvq_kq<TYPE, NT, RR, 0>), dim3(grid), dim3(64 * wpb), lds, stream, g);
            }
        };
        if (R == 1) go(std::integral_constant<int, 1>{}); else if (R == 2) go(std::integral_constant<int, 2>{}); else go(std::integral_constant<int, 4>{});
    }
    if (sample) kt_end(TYPE, L.g.nt, L, ea, eb);
    PI_CHECK_LAUNCH();
    return 0;
}

// two weight types in one launch (k_mmvq_kq2): TA = Q4_K, TB = Q5_K or Q6_K, decode widths, rows of one to four tiles.  Same instance choice as launch_mmvq.
static bool mmvq2_applies(int type_a, int type_b, int nt, int nb) {
    static const bool off = getenv("PI_MMVQ_NO_MIXED") != nullptr;          // tuning / tests
    return !off && type_a == PI_Q4_K && (type_b == PI_Q5_K || type_b == PI_Q6_K) && nt >= 1 && nt <= 2 && nb <= 4 * KQ_TB;
}
template <int TB, int NT>
static int launch_mmvq2(mmvq_launch & L, hipStream_t stream) {
    mmvq_args & g = L.g;
    hipEvent_t ea = nullptr, eb = nullptr;
    const bool sample = kt_begin(PI_Q4_K, L.g.nt, L, ea, eb);                 // (statistics: counted with the Q4_K launches, all bytes of the launch)
    int R = g.K >= 8192 ? 4 : 2;
    while (R > 1 && L.m_total / R < 2048) R >>= 1;
    if (R < 2) R = 2;                                                       // (the RoPE epilogue needs both rows of a pair in one wave)
    int waves = 0;
    for (int i = 0; i < MMVQ_MAX_SEG; i++) {
        if (i < L.nseg) { g.seg[i].wave0 = waves; waves += (g.seg[i].M + R - 1) / R; }
        else { g.seg[i] = g.seg[0]; g.seg[i].wave0 = 0x7fffffff; }
    }
    const int ntiles = (g.nb + KQ_TB - 1) / KQ_TB;
    auto go = [&](auto rc) {
        constexpr int RR = decltype(rc)::value;
        constexpr int FA = kq_lds<PI_Q4_K, RR * NT>::FLOATS, FB = kq_lds<TB, RR * NT>::FLOATS, LSTR = FA > FB ? FA : FB;
        if (ntiles > 1) {
            const size_t lds_t = (size_t) ntiles * LSTR * 4;
            if (sample) hipExtLaunchKernelGGL((k_mmvq_kq2<PI_Q4_K, TB, NT, RR, 2>), dim3(waves), dim3(64 * ntiles), lds_t, stream, ea, eb, 0, g);
            else        hipLaunchKernelGGL((k_mmvq_kq2<PI_Q4_K, TB, NT, RR, 2>), dim3(waves), dim3(64 * ntiles), lds_t, stream, g);
        } else {
            constexpr int wpb = 4;
            const unsigned grid = (unsigned) ((waves + wpb - 1) / wpb);
            if (sample) hipExtLaunchKernelGGL((k_mmvq_kq2<PI_Q4_K, TB, NT, RR, 1>), dim3(grid), dim3(64 * wpb), (size_t) wpb * LSTR * 4, stream, ea, eb, 0, g);
            else        hipLaunchKernelGGL((k_mmvq_kq2<PI_Q4_K, TB, NT, RR, 1>), dim3(grid), dim3(64 * wpb), (size_t) wpb * LSTR * 4, stream, g);
        }
    };
    if (R == 2) go(std::integral_constant<int, 2>{}); else go(std::integral_constant<int, 4>{});
    if (sample) kt_end(PI_Q4_K, L.g.nt, L, ea, eb);
    PI_CHECK_LAUNCH();
    return 0;
}

template <int TYPE>
static int dispatch_nt(mmvq_launch & L, int nt, hipStream_t stream) {
    switch (nt) {
        case 1: return launch_mmvq<TYPE, 1>(L, stream);
        case 2: return launch_mmvq<TYPE, 2>(L, stream);
        case 3: return launch_mmvq<TYPE, 3>(L, stream);
        case 4: return launch_mmvq<TYPE, 4>(L, stream);
        case 5: return launch_mmvq<TYPE, 5>(L, stream);
        case 6: return launch_mmvq<TYPE, 6>(L, stream);
        case 7: return launch_mmvq<TYPE, 7>(L, stream);
        case 8: return launch_mmvq<TYPE, 8>(L, stream);
    }
    return -1;
}

// out[i][t][row] = residual[i][t][row] + vec_dot(W_i[row], act[t]) for n_w (<= 3) weight matrices of the same type and width
// sharing one quantised activation.  Q4_0/Q8_0 weights are launched one by one.
// RoPE / KV-store epilogue of a grouped launch: per weight where its rows go (mmvq_seg::store), plus what the epilogue reads
struct mmvq_rope { int store[MMVQ_MAX_SEG]; uint16_t * dst16[MMVQ_MAX_SEG]; int stride16[MMVQ_MAX_SEG]; const int32_t * pos; const float * cs; const int32_t * dyn; int kv_head, hd; };

static int mul_mat_multi_impl(int n_w, const pi_weight * const * ws, const void * q8, int n_tok, float * const * outs, const float * const * ress, const mmvq_rope * rope, void * stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    if (n_tok <= 0 || n_w <= 0) return 0;
    if (n_w > MMVQ_MAX_SEG) { pi_set_error("pi_op_mul_mat_multi: n_w=%d > %d", n_w, MMVQ_MAX_SEG); return -1; }
    const pi_weight * w0 = ws[0];
    const int vdt = pi_vec_dot_type(w0->type);
    if (vdt != PI_Q8_K && vdt != PI_Q8_0 && vdt != PI_Q8_1) { pi_set_error("pi_op_mul_mat: weight type %d has no quantised mat-vec", w0->type); return -1; }
    // one type per launch — or Q4_K together with Q5_K / Q6_K at decode widths (k_mmvq_kq2)
    int type_b = -1, type_b_mask = 0;
    for (int i = 0; i < n_w; i++) {
        if (ws[i]->ne0 != w0->ne0) { pi_set_error("pi_op_mul_mat_multi: weights differ in width"); return -1; }
        if (ws[i]->type == PI_Q4_K) continue;
        if (type_b >= 0 && ws[i]->type != type_b) { type_b = -2; break; }
        type_b = ws[i]->type; type_b_mask |= 1 << i;
    }
    const bool all_b = type_b >= 0 && type_b_mask == (1 << n_w) - 1;            // a single type that is not Q4_K
    const bool mixed = type_b >= 0 && !all_b;
    // ... or, at 3..8 tokens, on the matrix cores (k_mmq_kq2t): the Q4_K matrices first, the second type's behind them
    bool mixed_mfma = false; int n_a = 0;
    if (mixed && vdt == PI_Q8_K) {
        while (n_a < n_w && ws[n_a]->type == PI_Q4_K) n_a++;
        bool suffix = n_a > 0 && type_b_mask == (((1 << n_w) - 1) & ~((1 << n_a) - 1)), copies = true;
        int64_t rows_a = 0, rows_b = 0;
        for (int i = 0; i < n_w; i++) { copies = copies && ws[i]->mq != nullptr; if (i < n_a) rows_a += ws[i]->ne1; else rows_b += ws[i]->ne1; }
        mixed_mfma = suffix && copies && n_tok >= pi_mmq_min_nt() && pi_mmq2_applies(type_b, n_tok, rows_a, rows_b);
    }
    if (type_b == -2 || (mixed && !mixed_mfma && !mmvq2_applies(PI_Q4_K, type_b, n_tok, (int) (w0->ne0 / 256)))) {
        bool same = true;
        for (int i = 1; i < n_w; i++) same = same && ws[i]->type == w0->type;
        if (!same) {
            if (rope) { pi_set_error("pi_op_mul_mat_multi: weights differ in type (Q4_K with Q5_K or Q6_K share a launch at 1-2 tokens, and at 3-8 on the matrix cores)"); return -1; }
            // plain outputs: the matrices that cannot share a launch go out one by one
            for (int i = 0; i < n_w; i++) { const int rc = mul_mat_multi_impl(1, ws + i, q8, n_tok, outs + i, ress ? ress + i : nullptr, nullptr, stream_); if (rc) return rc; }
            return 0;
        }
    }
    if (vdt != PI_Q8_K && n_w > 1) {
        for (int i = 0; i < n_w; i++) { const int rc = mul_mat_multi_impl(1, ws + i, q8, n_tok, outs + i, ress ? ress + i : nullptr, nullptr, stream_); if (rc) return rc; }
        return 0;
    }
    size_t oq, od, ob;
    pi_q8_layout(vdt, w0->ne0, n_tok, &oq, &od, &ob);
    const uint8_t * base = (const uint8_t *) q8;
    const int64_t K = w0->ne0;
    const int64_t nbq = vdt == PI_Q8_K ? K / 256 : w0->ng;
    // batches of PI_MMQ_MIN_NT tokens and more go to the matrix cores, 16 tokens per pass over the weights (mmq_mfma.hip); everything else
    // (single tokens, types without an MFMA kernel) takes the VALU mat-vec, 8 tokens per pass
    int64_t rows_all = 0;
    for (int i = 0; i < n_w; i++) rows_all += ws[i]->ne1;
    bool mfma = !mixed && vdt == PI_Q8_K && pi_mmq_wants(w0->type, n_tok, rows_all, K);
    for (int i = 0; i < n_w; i++) mfma = mfma && ws[i]->mq != nullptr;
    size_t ofa = 0, ofb = 0, oax = 0;
    if (mfma || mixed_mfma) pi_q8_frag_layout(K, n_tok, &ofa, &ofb, &oax);
    const int step = mfma ? pi_mmq_step(w0->type, n_tok, rows_all) : 8;
    for (int t0 = 0, adv = 0; t0 < n_tok; t0 += adv) {
        // (a two-type matrix-core launch walks up to 8 tokens — a verify pass — or, in a prompt pass, 64: four chunks, the pass's last launch with whatever is left)
        const int step_now = mixed_mfma ? (n_tok > 8 ? 64 : 8) : step;
        const int nt = n_tok - t0 < step_now ? n_tok - t0 : step_now;
        adv = nt;
        mmvq_launch L;
        L.nseg = n_w; L.m_total = 0; L.m_min = 1 << 30; L.wbytes = 0.0; L.any_res = false;
        L.g.K = (int) K; L.g.nb = (int) nbq;
        L.g.pos = rope ? rope->pos : nullptr; L.g.rope_cs = rope ? rope->cs : nullptr; L.g.dyn = rope ? rope->dyn : nullptr;
        L.g.kv_head = rope ? rope->kv_head : 0; L.g.hd = rope ? rope->hd : 2; L.g.tok0 = t0;
        L.g.frag = nullptr; L.g.aux = nullptr; L.g.nt = nt; L.g.abort_w = pi_abort_word(); L.g.type_b_mask = mixed ? type_b_mask : 0;
        L.rope = rope != nullptr;
        for (int i = 0; i < n_w; i++) {
            const pi_weight * w = ws[i];
            mmvq_seg & sg = L.g.seg[i];
            sg.p0 = w->p0; sg.p1 = w->p1; sg.p2 = w->p2; sg.mq = w->mq; sg.M = (int) w->ne1; sg.wave0 = 0;
            sg.store = rope ? rope->store[i] : 0; sg.dst16 = rope ? rope->dst16[i] : nullptr; sg.stride16 = rope ? rope->stride16[i] : 0;
            sg.out = outs && outs[i] ? outs[i] + (int64_t) t0 * w->ne1 : nullptr;
            sg.res = ress && ress[i] ? ress[i] + (int64_t) t0 * w->ne1 : nullptr;
            L.m_total += w->ne1; L.wbytes += (double) w->nbytes; L.any_res |= sg.res != nullptr;
        }
        for (int i = n_w; i < MMVQ_MAX_SEG; i++) { L.g.seg[i] = L.g.seg[0]; L.g.seg[i].wave0 = 0x7fffffff; }
        if (vdt == PI_Q8_K) {
            L.g.aq = base + oq + (size_t) t0 * nbq * 256; L.g.ad = base + od + (size_t) t0 * nbq * 4; L.g.abs_ = base + ob + (size_t) t0 * nbq * 32;
        } else {
            L.g.aq = base + oq + (size_t) t0 * nbq * 128;
            if (vdt == PI_Q8_1) { L.g.ad = base + od + (size_t) t0 * nbq * 16; L.g.abs_ = base + ob + (size_t) t0 * nbq * 16; }
            else { L.g.ad = base + od + (size_t) t0 * nbq * 8; L.g.abs_ = nullptr; }
        }
        int rc = -1;
        if (mixed_mfma) {
            L.g.frag = base + ofa + (size_t) (t0 / 16) * nbq * 8192;
            L.g.aux = base + oax + (size_t) (t0 / 16) * nbq * 1024;
            L.g.type_b_mask = 0;
            hipEvent_t ea = nullptr, eb = nullptr;
            const bool sample = kt_begin(PI_Q4_K, nt, L, ea, eb, PI_KT_MMQ);
            rc = pi_mmq_launch2(type_b, n_a, n_tok > 8, L, stream, ea, eb);
            if (sample) kt_end(PI_Q4_K, nt, L, ea, eb, PI_KT_MMQ);
            if (rc) return rc;
            continue;
        }
        if (mixed) {
            if (type_b == PI_Q5_K) rc = nt == 1 ? launch_mmvq2<PI_Q5_K, 1>(L, stream) : launch_mmvq2<PI_Q5_K, 2>(L, stream);
            else                   rc = nt == 1 ? launch_mmvq2<PI_Q6_K, 1>(L, stream) : launch_mmvq2<PI_Q6_K, 2>(L, stream);
            if (rc) return rc;
            continue;
        }
        if (mfma) {
            L.g.frag = base + ofa + (size_t) (t0 / 16) * nbq * 8192;
            L.g.aux = base + oax + (size_t) (t0 / 16) * nbq * 1024;
            L.g.nt = nt;
            hipEvent_t ea = nullptr, eb = nullptr;
            const bool sample = kt_begin(w0->type, nt, L, ea, eb, PI_KT_MMQ);
            rc = pi_mmq_launch(w0->type, L, stream, ea, eb);
            if (sample) kt_end(w0->type, nt, L, ea, eb, PI_KT_MMQ);
            if (rc) return rc;
            continue;
        }
        switch (w0->type) {
            case PI_Q2_K: rc = dispatch_nt<PI_Q2_K>(L, nt, stream); break;
            case PI_Q3_K: rc = dispatch_nt<PI_Q3_K>(L, nt, stream); break;
            case PI_Q4_K: rc = dispatch_nt<PI_Q4_K>(L, nt, stream); break;
            case PI_Q5_K: rc = dispatch_nt<PI_Q5_K>(L, nt, stream); break;
            case PI_Q6_K: rc = dispatch_nt<PI_Q6_K>(L, nt, stream); break;
            case PI_Q4_0: rc = dispatch_nt<PI_Q4_0>(L, nt, stream); break;
            case PI_Q4_1: rc = dispatch_nt<PI_Q4_1>(L, nt, stream); break;
            case PI_Q5_0: rc = dispatch_nt<PI_Q5_0>(L, nt, stream); break;
            case PI_Q5_1: rc = dispatch_nt<PI_Q5_1>(L, nt, stream); break;
            case PI_Q8_0: rc = dispatch_nt<PI_Q8_0>(L, nt, stream); break;
        }
        if (rc) return rc;
    }
    return 0;
}

extern "C" int pi_op_mul_mat_multi(int n_w, const pi_weight * const * ws, const void * q8, int n_tok, float * const * outs, const float * const * ress, void * stream) {
    return mul_mat_multi_impl(n_w, ws, q8, n_tok, outs, ress, nullptr, stream);
}

extern "C" int pi_op_mul_mat(const pi_weight * w, const void * q8, int n_tok, float * out, const float * res, void * stream) {
    return mul_mat_multi_impl(1, &w, q8, n_tok, &out, &res, nullptr, stream);
}

// 1 when pi_op_mul_mat_qkv_rope applies: Llama-style RoPE (pairs 2i, 2i+1), K-quant projections, even row counts
extern "C" int pi_op_mul_mat_qkv_rope_ok(const pi_weight * wq, const pi_weight * wk, const pi_weight * wv, int neox, int hd) {
    static const bool off = getenv("PI_NO_ROPE_EPILOGUE") != nullptr;      // tests: read once, not per call
    if (neox || hd % 2 || off) return 0;
    const pi_weight * w[3] = { wq, wk, wv };
    for (int i = 0; i < 3; i++) if (pi_vec_dot_type(w[i]->type) != PI_Q8_K || w[i]->ne1 % 2 || w[i]->ne0 != wq->ne0) return 0;
    return 1;
}

struct pi_rope_table_view { float * cs; int n_pos; int head_dim; int n_rot; };   // == pi_rope_table (elementwise.hip)

// q16 = f16(rope(Wq . act)), K cache cells <- f16(rope(Wk . act)), V cache cells <- f16(Wv . act): the attention projections,
// ggml_rope and llm_build_kv_store (llama.cpp:3546-3577, 4064-4100) with the rotation and the cache writes done by the mat-vec
// waves themselves (no fp32 q/k/v round trip, no separate launch).  Weights of the same type share a launch.
extern "C" int pi_op_mul_mat_qkv_rope(const pi_weight * wq, const pi_weight * wk, const pi_weight * wv, const void * q8, int n_tok, const int32_t * pos, const void * table_,
                                      int hd, int n_head, int n_head_kv, int kv_head, const int32_t * dyn_dev, uint16_t * q16, uint16_t * kc, uint16_t * vc, void * stream) {
    if (!pi_op_mul_mat_qkv_rope_ok(wq, wk, wv, 0, hd)) { pi_set_error("pi_op_mul_mat_qkv_rope: not applicable to these weights"); return -1; }
    const pi_rope_table_view * tb = (const pi_rope_table_view *) table_;
    if (tb->head_dim != hd) { pi_set_error("pi_op_mul_mat_qkv_rope: table head_dim %d != %d", tb->head_dim, hd); return -1; }
    if (wq->ne1 != (int64_t) n_head * hd || wk->ne1 != (int64_t) n_head_kv * hd || wv->ne1 != wk->ne1) { pi_set_error("pi_op_mul_mat_qkv_rope: projection shapes"); return -1; }
    const pi_weight * ws[3] = { wq, wk, wv };
    const int store[3] = { 1, 2, 3 };
    uint16_t * dst[3] = { q16, kc, vc };
    const int stride[3] = { n_head * hd, n_head_kv * hd, n_head_kv * hd };
    bool done[3] = { false, false, false };
    {   // attn_q / attn_k in Q4_K with attn_v in Q5_K or Q6_K (the Q4_K_M mixture, llama.cpp:7923-7933): ONE launch at decode widths
        int tb_type = -1, n_a = 0; bool ok2 = true;
        for (int j = 0; j < 3; j++) { if (ws[j]->type == PI_Q4_K) n_a++; else if (tb_type < 0 || tb_type == ws[j]->type) tb_type = ws[j]->type; else ok2 = false; }
        // (and from three tokens on the matrix cores, when the Q4_K matrices come first: k_mmq_kq2t)
        int64_t rows_a = 0, rows_b = 0; bool copies = true, a_first = true;
        for (int j = 0; j < 3; j++) { copies = copies && ws[j]->mq != nullptr; if (ws[j]->type == PI_Q4_K) { rows_a += ws[j]->ne1; a_first = a_first && j < n_a; } else rows_b += ws[j]->ne1; }
        const bool on_mfma = a_first && copies && n_tok >= pi_mmq_min_nt() && pi_mmq2_applies(tb_type, n_tok, rows_a, rows_b);
        if (ok2 && n_a > 0 && n_a < 3 && (mmvq2_applies(PI_Q4_K, tb_type, n_tok, (int) (wq->ne0 / 256)) || on_mfma)) {
            mmvq_rope r;
            r.pos = pos; r.cs = tb->cs; r.dyn = dyn_dev; r.kv_head = kv_head; r.hd = hd;
            for (int j = 0; j < 3; j++) { r.store[j] = store[j]; r.dst16[j] = dst[j]; r.stride16[j] = stride[j]; }
            return mul_mat_multi_impl(3, ws, q8, n_tok, nullptr, nullptr, &r, stream);
        }
    }
    for (int i = 0; i < 3; i++) {
        if (done[i]) continue;
        const pi_weight * gw[3]; mmvq_rope r; int ng = 0;
        r.pos = pos; r.cs = tb->cs; r.dyn = dyn_dev; r.kv_head = kv_head; r.hd = hd;
        for (int j = i; j < 3; j++)
            if (!done[j] && ws[j]->type == ws[i]->type) { gw[ng] = ws[j]; r.store[ng] = store[j]; r.dst16[ng] = dst[j]; r.stride16[ng] = stride[j]; ng++; done[j] = true; }
        for (int j = ng; j < MMVQ_MAX_SEG; j++) { r.store[j] = 0; r.dst16[j] = nullptr; r.stride16[j] = 0; }
        const int rc = mul_mat_multi_impl(ng, gw, q8, n_tok, nullptr, nullptr, &r, stream);
        if (rc) return rc;
    }
    return 0;
}
