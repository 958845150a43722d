// Multi-token quantised mat-mat for verify batches and prompts (2 <= n_tok <= 16 per launch) on the matrix cores, BIT-EXACT with the
// reference's x86 AVX2 arm — the same contract as mmvq.hip:
//
//   out[t][row] = residual[t][row] + vec_dot(W[row], act[t])
//
// Reference: ggml_compute_forward_mul_mat (ggml.c:9388-9597) -> ggml_vec_dot_q4_K_q8_K (ggml-quants.c:5318-5381), q5_K (:5967-6050),
// q6_K (:6741-6810).  Per 256-block the AVX2 code forms eight int32 lane sums  sumi[k] = sum_j sc_j * sum_{i<4} w[32j+4k+i] * y[32j+4k+i]
// (lane k = bytes 4k..4k+3 of every 32-value sub-block j), converts them to float and extends eight fp32 chains with ONE fma per block,
// in block order:  acc[k] = fma(d_x * d_y, (float) sumi[k], acc[k]);  mins go through their own four (Q4_K) / one (Q5_K) chains.
//
// What is exact in any order is the integer part — and that part is a matrix product.  v_mfma_f32_16x16x32_f16 computes it without
// rounding: one operand is (scale x weight) as f16 (<= 63*15 = 945 for Q4_K, <= 63*31 = 1953 for Q5_K; for Q6_K the int8 scale is split
// sc = 16*sh + sl so that both 16*sh*w' and sl*w' are f16 integers), the other the int8 activation as f16; products and partial sums are
// integers below 2^24, so every fp32 accumulation inside the MFMA is exact (tools_dev/mfma_exact.hip checks the instruction on hardware
// at these magnitudes).  One MFMA per (block, lane k): K = 32 = the (j, i) index; A = 16 tokens, B = 16 weight rows.  The fp32 chains
// stay on the VALU, one fma per block in block order: the reference's rounding points are kept, so the result has the same bits as the
// VALU kernel and as the reference.
//
// Mapping.  The MFMA is issued "transposed" (tokens are the rows of A, weight rows the columns of B) so that the C tile puts weight row
// r on lane (r, c) with tokens 4c..4c+3 in its four registers, and lane (r, c) builds the K-slice of sub-blocks 2c, 2c+1 — the low and high
// nibbles of ONE dword per accumulator lane k (qs group c, bytes 4k..4k+3).  The tile-major weight copy (runtime.hip: k_repack_mq_q4_K,
// PI_MQ_TILE bytes per 16 rows x one block) hands each lane exactly what it needs: (d, dmin), a 4-byte record with its two scales and its two
// mins already unpacked from the 6-bit packing, and its 8 dwords of quants.  VALU per (block, k): one shift, four v_and_or_b32, four
// v_pk_fma_f16; per block: 8 for the scales / mins (v_perm_b32 builds the f16 pairs 1024 + s), 8 for d_x*d_y and -dmin*d_y, and the chains.
// The mins go through the matrix cores too: acc_m[c] needs m[2c]*(bs[4c]+bs[4c+1]) + m[2c+1]*(bs[4c+2]+bs[4c+3]) — an MFMA whose B operand
// is (m[2c], m[2c], m[2c+1], m[2c+1]) on the lanes of K-slice c and zero elsewhere, A = the tokens' sixteen 16-value sums as f16 (<= 2032).
// A workgroup is R row groups of 16 rows x S waves; wave s of a row group owns accumulator lanes k in [s*8/S, (s+1)*8/S) and mins lanes
// c in [s*4/S, (s+1)*4/S) for ALL blocks of the row, so waves never exchange anything until the final fold — no per-block barrier.  S = 1
// (tall matrices: enough row groups to fill the chip) needs the fewest instructions per byte; S = 2 / 4 buy waves for short matrices at
// the price of repeating the per-block scalar work.  Reading 16 rows x one block per step from the row-major planes of the decode kernel
// streams at 1.3-1.6 TB/s with no arithmetic at all, the same bytes tile-major at ~5 TB/s (tools_dev/pattern.hip).  What IS shared is the
// activation side: the f16 fragments, the tokens' d and block sums (written by the quantisers in exactly this image, elementwise.hip:
// wave_emit_frags) are copied global -> LDS by DMA one tile of TB blocks ahead, once per workgroup.
// Weights: each wave keeps a ring of 2*TB blocks in registers (its share: 8 + 32/S bytes per lane and block), refilled right after use.
#include "mmvq_args.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

using pi_half8 = __attribute__((ext_vector_type(8))) _Float16;
using pi_half2 = __attribute__((ext_vector_type(2))) _Float16;
using pi_f32x4 = __attribute__((ext_vector_type(4))) float;
typedef uint32_t pi_u32x4 __attribute__((ext_vector_type(4)));

int pi_mmq_min_nt() {
    // measured on 70B shapes (tools_dev/perf_mmvq.py, us per launch, 8192x8192 / 28672x8192 / 8192x28672; one token: 11.2 / 30.4 / 34.0): the VALU
    // mat-vec takes 14.9 / 42.2 / 46.5 at 2 tokens and 19.3 / 49.2 / 53.8 at 3, the matrix-core kernel 16.9 / 40.4 / 48.9 and 17.2 / 37.7 / 44.9,
    // and the same at every width up to 8 (16: 19.5 / 39.6 / 57.6): it takes over from 3 tokens
    static const int v = [] { const char * e = getenv("PI_MMQ_MIN_NT"); const int n = e ? atoi(e) : 3; return n < 2 ? 2 : n; }();
    return v;
}
extern "C" int pi_op_mmq_min_tokens(void) { return pi_mmq_min_nt(); }
// tokens one launch of the matrix-core kernel walks: 64 (four chunks of 16 over one weight stream, register-staged activations); PI_MMQ_CHUNKS=2 / 1: 32 / 16
int pi_mmq_max_nt() {
    static const int v = [] { const char * e = getenv("PI_MMQ_CHUNKS"); return e && atoi(e) == 1 ? 16 : e && atoi(e) == 2 ? 32 : 64; }();
    return v;
}
// tokens per launch of a prompt pass over matrices of `rows` rows in all: 64 (four chunks over one weight stream, register-staged activations) — except for the tall launches
// (> 1152 row groups: the fused up + gate), whose one-wave row groups walk two chunks at 2.7 TB/s and would have to be split over four waves for four (measured, 70B up + gate
// per 64 tokens: 2 x 99 us in two-chunk launches, 218-260 us in one four-chunk launch; every shorter launch gains: ffn_down Q6_K 2 x 108 -> 143, Q4_K 2 x 83 -> 122, wo 2 x 27 -> 41)
int pi_mmq_step(int type, int n_tok, int64_t rows) {
    (void) type;
    static const int force = getenv("PI_MMQ_STEP") ? atoi(getenv("PI_MMQ_STEP")) : 0;
    if (force) return force;
    if (n_tok <= 16) return 16;
    const int64_t groups = (rows + 15) / 16 * 256 / pi_cu_count();
    return groups > 1152 ? (pi_mmq_max_nt() < 32 ? pi_mmq_max_nt() : 32) : pi_mmq_max_nt();
}
bool pi_mmq_supported(int type) {
    static const bool q3 = !(getenv("PI_MMQ_Q3") && atoi(getenv("PI_MMQ_Q3")) == 0);      // PI_MMQ_Q3=0: Q3_K batches stay on the VALU kernel (A/B)
    return (type == PI_Q3_K && q3) || type == PI_Q4_K || type == PI_Q5_K || type == PI_Q6_K;
}
// does a batch of n_tok tokens over matrices of `rows` rows in all go to the matrix cores?  Q4_K / Q5_K from pi_mmq_min_nt() tokens; Q6_K builds two operands per
// accumulator lane and only won from 5 tokens in round 2 (8192 x 28672: VALU 56.9 / 64.9 / 105 / 206 us at 3 / 4 / 8 / 16 tokens, matrix cores 76 at any width up to 16),
// and on short matrices (attn_v of a GQA model: 64 row groups) only past the VALU kernel's 8 tokens per pass
bool pi_mmq_wants(int type, int n_tok, int64_t rows, int64_t k) {
    if (!pi_mmq_supported(type) || n_tok < pi_mmq_min_nt()) return false;
    // (round 4: from 4 tokens on matrices of >= 4096 rows — in the bench the VALU kernel took 72.3 us for the 193 MB ffn_down at 4 tokens, the matrix cores take 65.7)
    // (round 6, re-measured per shape — profiles/r06_pass_width_dispatch.log, us per launch VALU / matrix cores: long rows win from 3 tokens (8192 x 28672: 65.2 / 59.7 at 3,
    //  8192 x 32768: 73.0 / 65.9), K <= 11008 only from 5 (9216 x 8192: 25.3 / 29.4 at 4, 41.2 / 27.7 at 5; 8192 x 8192: 23.1 / 24.3, 30.7 / 22.4; 4096 x 11008: 19.3 / 21.2,
    //  23.1 / 20.5).  PI_MMQ_Q6_MIN=n: one threshold for every shape, as until round 5 with n = 4)
    static const int q6_min = getenv("PI_MMQ_Q6_MIN") ? atoi(getenv("PI_MMQ_Q6_MIN")) : 0;
    // tall launches (one-wave row groups: > 1152 row groups, the output heads): from 3 as well — 32000 x 8192 58.4 / 55.4 at 3, 59.5 / 53.6 at 4; 32000 x 4096 31.0 / 32.7, 33.1 / 31.2
    if (type == PI_Q6_K) return n_tok >= (q6_min ? q6_min : (k >= 16384 || rows > 18432) ? 3 : 5) && (rows >= 4096 || n_tok > 8);
    // short launches: a row group walks its K / 256 blocks one after the other (13.4 us at K = 8192 however few rows there are), while the VALU kernel's
    // tile-per-wave instances put every tile of a short matrix in flight at once.  Measured (us, 3 / 4 / 8 tokens, matrix cores vs VALU): 1024 x 8192 Q5_K (attn_v of
    // a GQA model) 14.5 / 14.5 / 14.5 vs 6.4 / 7.0 / 9.9; 2048 x 8192 13.7 / 13.5 / 13.6 vs 8.5 / 10.4 / 14.9; 4096 x 8192 14.4 / 14.1 / 13.9 vs 12.0 / 13.6 / 21.3;
    // 8192 x 8192 17.1 / 16.3 / 16.3 vs 19.4 / 22.1 / 44.8; 4096 x 4096 9.4 / 9.0 / 8.9 vs 7.9 / 8.4 / 12.1; 2048 x 2048 6.3 / 6.0 / 5.9 vs 5.8 / 5.9 / 7.6
    if (n_tok <= 8 && rows * n_tok < 16384) return false;
    return true;
}

static __device__ __forceinline__ pi_half2 h2_from_bits(uint32_t u) { return __builtin_bit_cast(pi_half2, u); }
static __device__ __forceinline__ uint32_t h2_bits(pi_half2 v) { return __builtin_bit_cast(uint32_t, v); }
static __device__ __forceinline__ uint32_t pk_fma(uint32_t a, uint32_t b, uint32_t c) { return h2_bits(__builtin_elementwise_fma(h2_from_bits(a), h2_from_bits(b), h2_from_bits(c))); }
static __device__ __forceinline__ uint32_t pk_mul(uint32_t a, uint32_t b) { return h2_bits(h2_from_bits(a) * h2_from_bits(b)); }
static __device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return h2_bits(h2_from_bits(a) + h2_from_bits(b)); }
// the mins product: its operands are four f16 per lane — (bs[4c] .. bs[4c+3]) of a token and (m, m, m', m') of a row — so it is the K = 16 instruction (the K = 32 form carried four
// zeros per lane in each operand: moves to build them, and twice the matrix-core time).  Lane (i, g) holds k = 4g .. 4g+3 there and k = 8g .. 8g+3 (+ zeros) here: the same pairs
// meet, and the sum of exact integers below 2^24 does not depend on their order.
using pi_f32x2 = __attribute__((ext_vector_type(2))) float;
static __device__ __forceinline__ pi_f32x2 mq_pair(float a, float b) { pi_f32x2 v = { a, b }; asm volatile("" : "+v"(v)); return v; }
using pi_half4 = __attribute__((ext_vector_type(4))) _Float16;
static __device__ __forceinline__ pi_f32x4 mq_mfma_k16(uint2 a, uint2 b, pi_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(pi_half4, a), __builtin_bit_cast(pi_half4, b), c, 0, 0, 0);
}

// LDS image of one block of activations: 8 fragments + 1 KB aux.  A fragment is 1 KB (16 tokens) or, for batches of <= 8 tokens (NT8), the 512 bytes
// of tokens 0..7: lane (token, g) at (8 g + token) * 16
// s_waitcnt vmcnt(n) only
#define MQ_STR_(x) #x
#define MQ_STR(x) MQ_STR_(x)
#define MQ_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")

template <int S> struct mq_q { uint4 v[2 / S]; };
template <> struct mq_q<4> { uint2 v[1]; };
template <int TYPE, int S> struct mq_w { uint32_t dd, rec, rec2; mq_q<S> q; mq_q<S> h; };       // one lane's share of one block: (d, dmin), its scales / mins record, 8 / S quant dwords (Q5_K: + as many dwords of fifth bits)

// NC: chunks of 16 tokens per pass over the weights (prompt batches: NC = 2 walks 32 tokens with ONE weight stream — the operand of a (block, lane k) is built once and
// fed to one MFMA per chunk; every chunk keeps its own chains, so each token's bits are those of a 16-token pass)
// PIPE (S >= 2, one chunk, Q4_K / Q5_K): the block step as a two-stage software pipeline — see "pipelined form" below
// The kernel body as a device function (inlined into k_mmq_kq and into the two-type k_mmq_kq2t): workgroup `wg` of the launch's part that starts at row group `group0`
// and ends at `n_groups` (both counted over the whole launch: the segments' wave0 are) walks row groups group0 + wg * R ..
// STG (prompt passes, NC = 4: 33..64 tokens over ONE weight stream): the activation image of a block — NC x 9 KB — is staged through registers (global_load -> ds_write,
// two LDS buffers, one barrier per block) instead of by LDS-DMA: three DMA buffers of that size would leave the first 64 KB of LDS, the range an M0-based DMA
// destination can address.  The reference decodes a prompt as ONE batch of up to n_batch = 512 tokens (llama.cpp:5493-5496, speculative.cpp:235-239).
template <int TYPE, int S, int R, int TB, bool NT8, int NC = 1, bool PIPE = false, bool STG = false>
static __device__ __forceinline__ void mmq_kq_body(const mmvq_args & g, const int n_groups, const int wg, const int group0) {
    static_assert(!PIPE || (NC == 1 && S >= 2 && (TYPE == PI_Q4_K || TYPE == PI_Q5_K)), "pipelined form: one chunk, split row groups, Q4_K / Q5_K");
    static_assert(NC == 1 || ((NC == 2 || NC == 4) && !NT8), "several chunks only for full 16-token chunks");
    static_assert(STG == (NC == 4) && (!STG || (TB == 1 && !PIPE)), "four chunks: register-staged, one block per tile");
    static_assert(TYPE == PI_Q3_K || TYPE == PI_Q4_K || TYPE == PI_Q5_K || TYPE == PI_Q6_K, "types are added one by one");
    constexpr bool Q5 = TYPE == PI_Q5_K, Q6 = TYPE == PI_Q6_K, Q3 = TYPE == PI_Q3_K;
    constexpr uint32_t TILE = Q6 ? PI_MQ_TILE_Q6 : Q5 ? PI_MQ_TILE_Q5 : Q3 ? PI_MQ_TILE_Q3 : PI_MQ_TILE, QOFF = Q6 ? 1600 : (Q5 || Q3) ? 832 : 320;
    constexpr uint32_t QPART = Q3 ? 512 : 1024;                                          // bytes between the two halves (k = 0..3 / 4..7) of the quants
    static_assert(S == 1 || S == 2 || S == 4, "waves per row group");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int KS = 8 / S, CS = 4 / S, NW = S * R, D = 2 * TB, LW = (S == 1 ? 4 : 3) + ((Q5 || Q3) ? (S == 1 ? 2 : 1) : 0) + (Q6 ? (S == 1 ? 2 : 1) : 0);  // LW: vector loads per block of weights
    constexpr int FRAG = NT8 ? 512 : 1024, CH_LDS = 8 * FRAG + 1024, MQ_BLK_LDS = NC * CH_LDS, NP1 = NT8 ? 5 : 9, NP = NC * NP1;   // NP: DMA pieces of 1 KB per block (all chunks)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rgl = wave / S, s = wave % S;                                               // row group inside the workgroup, share of its accumulator lanes
    const int r = lane & 15, gq = lane >> 4;                                              // weight row r, K-slice gq: sub-blocks 2gq, 2gq+1; C tile: tokens 4 gq ..
    const int rg_raw = group0 + wg * R + rgl;
    const bool active = rg_raw < n_groups;
    const int rg = active ? rg_raw : n_groups - 1;
    const int si = (rg >= g.seg[1].wave0) + (rg >= g.seg[2].wave0);
    const mmvq_seg & sg = g.seg[si];
    const int64_t row0 = (int64_t) (rg - sg.wave0) * 16;
    const int rows = sg.M - row0 < 16 ? (int) (sg.M - row0) : 16;
    const int nb = g.nb, nt = g.nt, ntile = (nb + TB - 1) / TB;
    // tuning toggles (1 no compute, 2 no staging, 4 no barrier, 8 no chains; wrong results): only in a -DPI_MMQ_DBG_BUILD build — as run-time branches
    // they sit between every MFMA and its chain and keep the scheduler from overlapping one accumulator lane's MFMA with the next one's operand build
#ifdef PI_MMQ_DBG_BUILD
    const int dbg = g.dbg;
#else
    constexpr int dbg = 0;
#endif

    // (q & 0x000F000F) | 0x6400 = 1024 + n is scaled by s; (q & 0x00F000F0) | 0x6400 = 1024 + 16 n by s / 16 (exact: a power of two) — no shift of
    // the high nibbles, and (1024 + x n) * s' - 1024 * s' == n * s in one v_pk_fma_f16
    uint32_t magic = 0x64006400u;                                                         // f16 pair (1024, 1024); kept in a VGPR so that (q & mask) | magic is one v_and_or_b32
    asm volatile("" : "+v"(magic));
    const uint32_t own[4] = { gq == 0 ? 0xffffffffu : 0u, gq == 1 ? 0xffffffffu : 0u, gq == 2 ? 0xffffffffu : 0u, gq == 3 ? 0xffffffffu : 0u };

    float acc[NC][KS][4], accm[NC][CS][4];
#pragma unroll
    for (int ch = 0; ch < NC; ch++) {
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[ch][kk][i] = 0.0f;
#pragma unroll
        for (int cc = 0; cc < CS; cc++)
#pragma unroll
            for (int i = 0; i < 4; i++) accm[ch][cc][i] = 0.0f;
    }

    // tile-major copy (runtime.hip: k_repack_mq_q4_K): tile (row group, block) = PI_MQ_TILE contiguous bytes
    const uint8_t * const wbase = sg.mq + (size_t) (rg - sg.wave0) * nb * TILE;
    const uint32_t dd_off = 4 * r, rec_off = 64 + (Q6 ? 8 : 4) * lane, q_off = QOFF + 16 * (Q3 ? 2 * r + (gq >> 1) : lane) + (S == 1 ? 0 : S == 2 ? QPART * s : QPART * (s >> 1) + 8 * (s & 1));      // (Q3_K: the row's half gq >> 1 — both K-slices of a half read the same dwords)
    const uint32_t h_off = (Q6 ? 576 + 64 * r + 32 * (gq >> 1) : 320 + 32 * r) + (S == 1 ? 0 : S == 2 ? 16 * s : 8 * s);       // Q6_K: qh of the row's half n = gq >> 1;       // Q5_K: the row's qh bytes 4k..4k+3 of this wave's accumulator lanes (the same for the four K-slices)
    auto load_w = [&](int b, mq_w<TYPE, S> & W) __attribute__((always_inline)) {
        const uint32_t bo = (uint32_t) (b < nb ? b : nb - 1) * TILE;     // unconditional, clamped (see mmvq.hip on branches between a load and its use)
        W.dd = *(const uint32_t *) (wbase + (bo + dd_off));
        if constexpr (Q6) { const uint2 rr = *(const uint2 *) (wbase + (bo + rec_off)); W.rec = rr.x; W.rec2 = rr.y; }
        else W.rec = *(const uint32_t *) (wbase + (bo + rec_off));
        if constexpr (S == 4) { W.q.v[0] = *(const uint2 *) (wbase + (bo + q_off)); if constexpr (Q5 || Q6 || Q3) W.h.v[0] = *(const uint2 *) (wbase + (bo + h_off)); }
        else {
#pragma unroll
            for (int i = 0; i < 2 / S; i++) { W.q.v[i] = *(const uint4 *) (wbase + (bo + q_off + QPART * i)); if constexpr (Q5 || Q6 || Q3) W.h.v[i] = *(const uint4 *) (wbase + (bo + h_off + 16 * i)); }
        }
    };
    auto hdword = [&](const mq_w<TYPE, S> & W, int kk) __attribute__((always_inline)) -> uint32_t {
        if constexpr (S == 4) return kk ? W.h.v[0].y : W.h.v[0].x;
        else { const uint4 v = W.h.v[kk >> 2]; return (kk & 3) == 0 ? v.x : (kk & 3) == 1 ? v.y : (kk & 3) == 2 ? v.z : v.w; }
    };
    auto qdword = [&](const mq_w<TYPE, S> & W, int kk) __attribute__((always_inline)) -> uint32_t {
        if constexpr (S == 4) return kk ? W.q.v[0].y : W.q.v[0].x;
        else { const uint4 v = W.q.v[kk >> 2]; return (kk & 3) == 0 ? v.x : (kk & 3) == 1 ? v.y : (kk & 3) == 2 ? v.z : v.w; }
    };
    // activation tile T -> LDS buffer: 9 pieces of 1 KB per block, dealt round-robin to the workgroup's waves, one DMA instruction each
    // (global_load_lds_dwordx4: lane l's 16 bytes land at M0 + 16 l).  Issued from inline asm: the compiler's wait-count pass drains
    // vmcnt before every LDS read once it sees an LDS-DMA, which would serialise the weight ring; the wave counts its DMAs itself (MQ_WAIT_VM)
    auto dma_tile = [&](int T, uint32_t buf) __attribute__((always_inline)) {
        if (T < ntile && !(dbg & 2)) {
#pragma unroll
            for (int i = 0; i < (NP * TB + NW - 1) / NW; i++) {
                const int piece = i * NW + wave;
                if (piece < NP * TB) {
                    const int u = piece / NP, ch = (piece % NP) / NP1, part = piece % NP1;
                    const int b = T * TB + u < nb ? T * TB + u : nb - 1;
                    const size_t cb = (size_t) ch * nb + b;                                   // chunk ch of this launch: its planes follow the first chunk's (pi_q8_frag_layout)
                    const uint8_t * src;
                    if (part == NP1 - 1) src = g.aux + cb * 1024 + lane * 16;
                    else if (NT8) src = g.frag + ((cb * 8 + 2 * part + (lane >> 5)) * 64 + (lane & 7) + 16 * ((lane & 31) >> 3)) * 16;   // two fragments per piece, tokens 0..7 only
                    else src = g.frag + (cb * 8 + part) * 1024 + lane * 16;
                    const uint32_t dst = buf + u * MQ_BLK_LDS + ch * CH_LDS + part * 1024;
                    uint32_t keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                }
            }
        }
    };
    // workgroup barrier without the fence of __syncthreads (which waits for every outstanding load): LDS reads of this wave have returned
    // (lgkmcnt), the DMA writes it issued have landed (the caller's MQ_WAIT_VM)
    auto wg_barrier = [&]() __attribute__((always_inline)) { if (!(dbg & 6)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- Q6_K (ggml-quants.c:6741-6810): sumi[k] = sum over the eight 32-groups of sc * sum_{i<4} (u - 32) * y, sc an int8: sc * (u - 32) does not fit f16's 11 bits,
    //      so sc = 16 * sh + sl is split and the lane sum comes from TWO accumulating MFMAs with operands 16 sh (u - 32) and sl (u - 32) (both exact: |.| <= 4096 in
    //      steps of 16, and <= 480).  K-slice (n, hh) = the groups g = hh (low nibbles) and hh + 2 (high nibbles) of half n; the activation fragment pairs 32-groups
    //      (2c, 2c+1), so the two 8-byte halves of the operand come from two fragment lanes.  No mins (symmetric format). ----
    // ---- Q3_K (ggml-quants.c:4443-4533): sumi[k] = sum over the eight 32-groups g of sc[2g + (k >> 2)] * sum_{i<4} w * y with w = (2 low bits | high bit << 2) - 4 in
    //      [-4, 3] and sc in [-32, 31]: sc * w fits f16 exactly, ONE MFMA per (block, accumulator lane), no mins.  K-slice gq = groups 2gq, 2gq+1 = two bit pairs of ONE dword of
    //      the row's half gq >> 1 (a qs byte holds four groups), high bits from bit (2gq + e) of the hmask bytes; the fragments pair groups (2c, 2c+1) like Q4_K's sub-blocks ----
    auto operand_q3 = [&](const mq_w<TYPE, S> & W, int kk) __attribute__((always_inline)) -> uint4 {
        const int half = S == 1 ? (kk >> 2) : S == 2 ? s : (s >> 1);                                       // accumulator lanes k < 4: the first 16 weights of a group, k >= 4: the second
        const uint32_t selA = 0x04000400u + (uint32_t) half * 0x00010001u, selB = selA + 0x00020002u;     // record bytes: (group 2gq: sub 0, sub 1), (group 2gq+1: sub 0, sub 1), each value + 32
        const uint32_t SA = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, selA), 0xE420E420u);           // (1024 + sc + 32) - 1056 = sc
        const uint32_t SB = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, selB), 0xE420E420u);
        const uint32_t q = qdword(W, kk) >> (4 * (gq & 1)), hm = hdword(W, kk) >> (2 * gq);
        const uint32_t t0 = (q & 0x03030303u) | ((hm & 0x01010101u) << 2), t1 = ((q >> 2) & 0x03030303u) | ((hm & 0x02020202u) << 1);      // n = low bits | high bit << 2, per byte
        // 1024 + n as f16 pairs (bytes i0, i2 / i1, i3), minus 1028 = n - 4, times the scale (<= 128 in magnitude: exact)
        uint4 bw;
        bw.x = pk_mul(pk_add((t0 & 0x00FF00FFu) | magic, 0xE404E404u), SA);
        bw.y = pk_mul(pk_add(((t0 >> 8) & 0x00FF00FFu) | magic, 0xE404E404u), SA);
        bw.z = pk_mul(pk_add((t1 & 0x00FF00FFu) | magic, 0xE404E404u), SB);
        bw.w = pk_mul(pk_add(((t1 >> 8) & 0x00FF00FFu) | magic, 0xE404E404u), SB);
        return bw;
    };
    auto compute_q3 = [&](const mq_w<TYPE, S> & W, const uint8_t * blk) __attribute__((always_inline)) {
        const float dx_h = pi_h2f((uint16_t) (W.dd & 0xffff)); const pi_f32x2 dP = mq_pair(dx_h, dx_h); const float dx = dP.x;
        const pi_f32x4 z = { 0.0f, 0.0f, 0.0f, 0.0f };
        uint4 bw[KS];
#pragma unroll
        for (int kk = 0; kk < KS; kk++) bw[kk] = operand_q3(W, kk);
#pragma unroll
        for (int ch = 0; ch < NC; ch++) {
            const float4 dy4 = *(const float4 *) (blk + ch * CH_LDS + 8 * FRAG + 16 * gq);
            const float d[4] = { dy4.x * dx, dy4.y * dx, dy4.z * dx, dy4.w * dx };                        // y[i].d * GGML_FP16_TO_FP32(x[i].d)
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
                const uint4 af = *(const uint4 *) (blk + ch * CH_LDS + (s * KS + kk) * FRAG + (NT8 ? (gq * 8 + (r & 7)) * 16 : lane * 16));
                const pi_f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pi_half8, af), __builtin_bit_cast(pi_half8, bw[kk]), z, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) acc[ch][kk][i] = fmaf(d[i], c[i], acc[ch][kk][i]);
            }
            if constexpr (NC > 1) __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto compute_q6 = [&](const mq_w<TYPE, S> & W, const uint8_t * blk) __attribute__((always_inline)) {
        const int n = gq >> 1, hh = gq & 1;
        const float dx_h = pi_h2f((uint16_t) (W.dd & 0xffff)); const pi_f32x2 dP = mq_pair(dx_h, dx_h); const float dx = dP.x;
        float d[NC][4];
#pragma unroll
        for (int ch = 0; ch < NC; ch++) {
            const float4 dy4 = *(const float4 *) (blk + ch * CH_LDS + 8 * FRAG + 16 * gq);
            d[ch][0] = dy4.x * dx; d[ch][1] = dy4.y * dx; d[ch][2] = dy4.z * dx; d[ch][3] = dy4.w * dx;   // y[i].d * GGML_FP16_TO_FP32(x[i].d)
        }
        const pi_f32x4 z = { 0.0f, 0.0f, 0.0f, 0.0f };
        const uint32_t aoff = (NT8 ? ((2 * n) * 8 + (r & 7)) : (r + 32 * n)) * 16 + 8 * hh;                // fragment lane (token, 2n), half hh; lane (token, 2n+1) is 16 lanes (NT8: 8) further
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            const int half = S == 1 ? (kk >> 2) : S == 2 ? s : (s >> 1);                                   // accumulator lanes k < 4 use the first scale of a 32-group, k >= 4 the second
            const uint32_t selA = 0x04000400u + (uint32_t) half * 0x00010001u, selB = selA + 0x00020002u;
            const uint32_t SLa = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, selA), 0xE400E400u);                     // (sl, sl) of the low-nibble group
            const uint32_t SHa = pk_fma(__builtin_amdgcn_perm(0x64646464u, W.rec2, selA), 0x4C004C00u, 0xF408F408u);       // (1024 + sh + 8) * 16 - 16512 = 16 sh
            const uint32_t SLb = pk_fma(__builtin_amdgcn_perm(0x64646464u, W.rec, selB), 0x2C002C00u, 0xD400D400u);        // (1024 + sl) / 16 - 64 = sl / 16 (the high-nibble operand is 16 w)
            const uint32_t SHb = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec2, selB), 0xE408E408u);                     // (1024 + sh + 8) - 1032 = sh
            const uint32_t q = qdword(W, kk), q8 = q >> 8;
            const uint32_t h = hdword(W, kk) >> (2 * hh), h8 = h >> 8;
            // 1024 + u (u = nibble | top2 << 4) resp. 1024 + 16 u, minus 1056 resp. 1536 = u - 32 resp. 16 (u - 32)
            const uint32_t w0 = pk_add((q & 0x000F000Fu) | magic | ((h & 0x00030003u) << 4), 0xE420E420u);
            const uint32_t w1 = pk_add((q8 & 0x000F000Fu) | magic | ((h8 & 0x00030003u) << 4), 0xE420E420u);
            const uint32_t w2 = pk_add((q & 0x00F000F0u) | magic | ((h & 0x00300030u) << 4), 0xE600E600u);
            const uint32_t w3 = pk_add((q8 & 0x00F000F0u) | magic | ((h8 & 0x00300030u) << 4), 0xE600E600u);
            const uint4 b1 = make_uint4(pk_mul(w0, SHa), pk_mul(w1, SHa), pk_mul(w2, SHb), pk_mul(w3, SHb));
            const uint4 b2 = make_uint4(pk_mul(w0, SLa), pk_mul(w1, SLa), pk_mul(w2, SLb), pk_mul(w3, SLb));
#pragma unroll
            for (int ch = 0; ch < NC; ch++) {
                const uint8_t * fr = blk + ch * CH_LDS + (s * KS + kk) * FRAG + aoff;
                const uint2 a_lo = *(const uint2 *) fr, a_hi = *(const uint2 *) (fr + (NT8 ? 8 : 16) * 16);
                const uint4 af = make_uint4(a_lo.x, a_lo.y, a_hi.x, a_hi.y);
                pi_f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pi_half8, af), __builtin_bit_cast(pi_half8, b1), z, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pi_half8, af), __builtin_bit_cast(pi_half8, b2), c, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) acc[ch][kk][i] = fmaf(d[ch][i], c[i], acc[ch][kk][i]);
            }
        }
    };
    // ---- four chunks (STG): the block's weight operands are built ONCE — they do not depend on the chunk — and the chunks then run one after the other, each with its own
    //      d / fragments / MFMAs / chain updates, behind a scheduling barrier: the generic form below keeps every chunk's d and fragments live at once (402 VGPRs at S = 2) ----
    auto compute_pp = [&](const mq_w<TYPE, S> & W, const uint8_t * blk) __attribute__((always_inline)) {
        const pi_f32x4 z = { 0.0f, 0.0f, 0.0f, 0.0f };
        if constexpr (Q6) {
            const int n = gq >> 1, hh = gq & 1;
            const float dx_h = pi_h2f((uint16_t) (W.dd & 0xffff)); const pi_f32x2 dP = mq_pair(dx_h, dx_h); const float dx = dP.x;
            const uint32_t aoff = (uint32_t) (r + 32 * n) * 16 + 8 * hh;
            uint4 b1[KS], b2[KS];
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
                const int half = S == 1 ? (kk >> 2) : S == 2 ? s : (s >> 1);
                const uint32_t selA = 0x04000400u + (uint32_t) half * 0x00010001u, selB = selA + 0x00020002u;
                const uint32_t SLa = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, selA), 0xE400E400u);
                const uint32_t SHa = pk_fma(__builtin_amdgcn_perm(0x64646464u, W.rec2, selA), 0x4C004C00u, 0xF408F408u);
                const uint32_t SLb = pk_fma(__builtin_amdgcn_perm(0x64646464u, W.rec, selB), 0x2C002C00u, 0xD400D400u);
                const uint32_t SHb = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec2, selB), 0xE408E408u);
                const uint32_t q = qdword(W, kk), q8 = q >> 8;
                const uint32_t h = hdword(W, kk) >> (2 * hh), h8 = h >> 8;
                const uint32_t w0 = pk_add((q & 0x000F000Fu) | magic | ((h & 0x00030003u) << 4), 0xE420E420u);
                const uint32_t w1 = pk_add((q8 & 0x000F000Fu) | magic | ((h8 & 0x00030003u) << 4), 0xE420E420u);
                const uint32_t w2 = pk_add((q & 0x00F000F0u) | magic | ((h & 0x00300030u) << 4), 0xE600E600u);
                const uint32_t w3 = pk_add((q8 & 0x00F000F0u) | magic | ((h8 & 0x00300030u) << 4), 0xE600E600u);
                b1[kk] = make_uint4(pk_mul(w0, SHa), pk_mul(w1, SHa), pk_mul(w2, SHb), pk_mul(w3, SHb));
                b2[kk] = make_uint4(pk_mul(w0, SLa), pk_mul(w1, SLa), pk_mul(w2, SLb), pk_mul(w3, SLb));
            }
#pragma unroll
            for (int ch = 0; ch < NC; ch++) {
                const float4 dy4 = *(const float4 *) (blk + ch * CH_LDS + 8 * FRAG + 16 * gq);
                const float d[4] = { dy4.x * dx, dy4.y * dx, dy4.z * dx, dy4.w * dx };
#pragma unroll
                for (int kk = 0; kk < KS; kk++) {
                    const uint8_t * fr = blk + ch * CH_LDS + (s * KS + kk) * FRAG + aoff;
                    const uint2 a_lo = *(const uint2 *) fr, a_hi = *(const uint2 *) (fr + 16 * 16);
                    const uint4 af = make_uint4(a_lo.x, a_lo.y, a_hi.x, a_hi.y);
                    pi_f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pi_half8, af), __builtin_bit_cast(pi_half8, b1[kk]), z, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pi_half8, af), __builtin_bit_cast(pi_half8, b2[kk]), c, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[ch][kk][i] = fmaf(d[i], c[i], acc[ch][kk][i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const pi_f32x2 dP = mq_pair(pi_h2f((uint16_t) (W.dd & 0xffff)), pi_h2f((uint16_t) (W.dd >> 16))); const float dx = dP.x, dminx = dP.y;
            const uint32_t S1 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04000400u), 0xE400E400u);
            const uint32_t S2 = pk_fma(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04010401u), 0x2C002C00u, 0xD400D400u);
            const uint32_t C1 = pk_mul(S1, 0xE400E400u), C2 = pk_mul(S2, 0xE400E400u);
            uint4 bw[KS];
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
                const uint32_t q = qdword(W, kk), qh = q >> 8;
                if constexpr (Q5) {
                    const uint32_t t = hdword(W, kk) >> (2 * gq), th = t >> 8;
                    bw[kk].x = pk_fma((q & 0x000F000Fu) | magic | ((t & 0x00010001u) << 4), S1, C1);
                    bw[kk].y = pk_fma((qh & 0x000F000Fu) | magic | ((th & 0x00010001u) << 4), S1, C1);
                    bw[kk].z = pk_fma((q & 0x00F000F0u) | magic | ((t & 0x00020002u) << 7), S2, C2);
                    bw[kk].w = pk_fma((qh & 0x00F000F0u) | magic | ((th & 0x00020002u) << 7), S2, C2);
                } else {
                    bw[kk].x = pk_fma((q & 0x000F000Fu) | magic, S1, C1);
                    bw[kk].y = pk_fma((qh & 0x000F000Fu) | magic, S1, C1);
                    bw[kk].z = pk_fma((q & 0x00F000F0u) | magic, S2, C2);
                    bw[kk].w = pk_fma((qh & 0x00F000F0u) | magic, S2, C2);
                }
            }
            const uint32_t M0 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04020402u), 0xE400E400u);
            const uint32_t M1 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04030403u), 0xE400E400u);
#pragma unroll
            for (int ch = 0; ch < NC; ch++) {
                const float4 dy4 = *(const float4 *) (blk + ch * CH_LDS + 8 * FRAG + 16 * gq);
                const float dy[4] = { dy4.x, dy4.y, dy4.z, dy4.w };
                float d[4], nd[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { d[i] = dy[i] * dx; nd[i] = -dy[i] * dminx; }
#pragma unroll
                for (int kk = 0; kk < KS; kk++) {
                    const uint4 af = *(const uint4 *) (blk + ch * CH_LDS + (s * KS + kk) * FRAG + lane * 16);
                    const pi_f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pi_half8, af), __builtin_bit_cast(pi_half8, bw[kk]), z, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[ch][kk][i] = fmaf(d[i], c[i], acc[ch][kk][i]);
                }
                const uint2 bs = *(const uint2 *) (blk + ch * CH_LDS + 8 * FRAG + 64 + lane * 8);
                const uint2 am = make_uint2(bs.x, bs.y);
                if constexpr (Q5) {
                    if (s == 0) {
                        const uint2 bm = make_uint2(M0, M1);
                        const pi_f32x4 cm = mq_mfma_k16(am, bm, z);
#pragma unroll
                        for (int i = 0; i < 4; i++) { const float prod = nd[i] * cm[i]; accm[ch][0][i] = accm[ch][0][i] + prod; }
                    }
                } else {
#pragma unroll
                    for (int cc = 0; cc < CS; cc++) {
                        const int c = s * CS + cc;
                        const uint2 bm = make_uint2(M0 & own[c], M1 & own[c]);
                        const pi_f32x4 cm = mq_mfma_k16(am, bm, z);
#pragma unroll
                        for (int i = 0; i < 4; i++) accm[ch][cc][i] = fmaf(nd[i], cm[i], accm[ch][cc][i]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto compute = [&](const mq_w<TYPE, S> & W, const uint8_t * blk, bool live) __attribute__((always_inline)) {
        if constexpr (Q3) { compute_q3(W, blk); return; }                   // (every width: the operands are built once per block, the chunks run one after the other)
        if constexpr (STG) { compute_pp(W, blk); return; }
        if constexpr (Q6) { compute_q6(W, blk); return; }
        // ---- block scalars: d, dmin of the row; this lane's two scales and two mins, bytes of one record ----
        const uint32_t ddl = live ? W.dd : 0u;
        const pi_f32x2 dP = mq_pair(pi_h2f((uint16_t) (ddl & 0xffff)), pi_h2f((uint16_t) (ddl >> 16))); const float dx = dP.x, dminx = dP.y;
        // integer -> f16 pair without a conversion: bytes (v, 0x64, v, 0x64) = (1024 + v, 1024 + v)
        const uint32_t S1 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04000400u), 0xE400E400u);                       // (s, s) of sub-block 2gq   (low nibbles)
        const uint32_t S2 = pk_fma(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04010401u), 0x2C002C00u, 0xD400D400u);          // (1024 + s) / 16 - 64 = s / 16 of sub-block 2gq+1 (high nibbles)
        const uint32_t C1 = pk_mul(S1, 0xE400E400u), C2 = pk_mul(S2, 0xE400E400u);                                             // -1024 * s'
        float dy[NC][4], d[NC][4];
#pragma unroll
        for (int ch = 0; ch < NC; ch++) {
            const float4 dy4 = *(const float4 *) (blk + ch * CH_LDS + 8 * FRAG + 16 * gq);                                     // d of tokens 4 gq .. 4 gq + 3 of chunk ch
            dy[ch][0] = dy4.x; dy[ch][1] = dy4.y; dy[ch][2] = dy4.z; dy[ch][3] = dy4.w;
#pragma unroll
            for (int i = 0; i < 4; i++) d[ch][i] = dy[ch][i] * dx;                                         // y[i].d * GGML_FP16_TO_FP32(x[i].d)
        }
        const pi_f32x4 z = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            const uint32_t q = qdword(W, kk), qh = q >> 8;
            uint4 bw;
            if constexpr (Q5) {
                // fifth bits: bit 2gq (low-nibble sub-block) and 2gq+1 (high) of the row's qh bytes 4k..4k+3; 1024 + n + 16 b keeps b at mantissa bit 4 (low
                // nibbles) and 1024 + 16 (n + 16 b) at bit 8 (high): <= 63 * 31 = 1953, still exact in f16
                const uint32_t t = hdword(W, kk) >> (2 * gq), th = t >> 8;
                bw.x = pk_fma((q & 0x000F000Fu) | magic | ((t & 0x00010001u) << 4), S1, C1);
                bw.y = pk_fma((qh & 0x000F000Fu) | magic | ((th & 0x00010001u) << 4), S1, C1);
                bw.z = pk_fma((q & 0x00F000F0u) | magic | ((t & 0x00020002u) << 7), S2, C2);
                bw.w = pk_fma((qh & 0x00F000F0u) | magic | ((th & 0x00020002u) << 7), S2, C2);
            } else {
            bw.x = pk_fma((q & 0x000F000Fu) | magic, S1, C1);                                              // sub-block 2gq:   bytes (i0, i2)
            bw.y = pk_fma((qh & 0x000F000Fu) | magic, S1, C1);                                             //                  bytes (i1, i3)
            bw.z = pk_fma((q & 0x00F000F0u) | magic, S2, C2);                                              // sub-block 2gq+1
            bw.w = pk_fma((qh & 0x00F000F0u) | magic, S2, C2);
            }
#pragma unroll
            for (int ch = 0; ch < NC; ch++) {
                const uint4 af = *(const uint4 *) (blk + ch * CH_LDS + (s * KS + kk) * FRAG + (NT8 ? (gq * 8 + (r & 7)) * 16 : lane * 16));   // tokens' fragment of accumulator lane k
                const pi_f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pi_half8, af), __builtin_bit_cast(pi_half8, bw), z, 0, 0, 0);
                if (dbg & 8) { acc[ch][kk][0] += c[0] + c[1] + c[2] + c[3]; continue; }
#pragma unroll
                for (int i = 0; i < 4; i++) acc[ch][kk][i] = fmaf(d[ch][i], c[i], acc[ch][kk][i]);         // (float) sumi[k] of row r, token 4 gq + i
            }
        }
        if (dbg & 8) return;
        // mins: acc_m[c] = fma(-y.d * dmin, (float) (m[2c]*(bs[4c]+bs[4c+1]) + m[2c+1]*(bs[4c+2]+bs[4c+3])), acc_m[c])   (ggml-quants.c:5339-5345)
        const uint32_t M0 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04020402u), 0xE400E400u);   // (m[2gq], m[2gq])
        const uint32_t M1 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04030403u), 0xE400E400u);   // (m[2gq+1], m[2gq+1])
#pragma unroll
        for (int ch = 0; ch < NC; ch++) {
        const uint2 bs = *(const uint2 *) (blk + ch * CH_LDS + 8 * FRAG + 64 + (NT8 ? (gq * 16 + (r & 7)) : lane) * 8);  // f16 (bs[4gq] .. bs[4gq+3]) of token r
        const uint2 am = make_uint2(bs.x, bs.y);
        float nd[4];
#pragma unroll
        for (int i = 0; i < 4; i++) nd[i] = -dy[ch][i] * dminx;
        if constexpr (Q5) {
            // Q5_K (ggml-quants.c:5998-6002): ONE scalar chain, summs += dmin * (float) sum_j m_j * (bs[2j] + bs[2j+1]) — a multiply, then an add; every K-slice
            // contributes to the one product, so B is unmasked.  Wave 0 of the row group carries it.
            if (s == 0) {
                const uint2 bm = make_uint2(M0, M1);
                const pi_f32x4 cm = mq_mfma_k16(am, bm, z);
#pragma unroll
                for (int i = 0; i < 4; i++) { const float prod = nd[i] * cm[i]; accm[ch][0][i] = accm[ch][0][i] + prod; }
            }
        } else {
#pragma unroll
        for (int cc = 0; cc < CS; cc++) {
            const int c = s * CS + cc;
            const uint2 bm = make_uint2(M0 & own[c], M1 & own[c]);                                 // only K-slice c contributes to chain c
            const pi_f32x4 cm = mq_mfma_k16(am, bm, z);
#pragma unroll
            for (int i = 0; i < 4; i++) accm[ch][cc][i] = fmaf(nd[i], cm[i], accm[ch][cc][i]);             // the products are integers below 2^24: exact in any form
        }
        }
        }
    };

    // ---- pipelined form.  A short matrix has few row groups, so its waves are few (two per SIMD) and a wave's block step is one chain of latencies: LDS reads -> operand
    //      build -> MFMA -> result latency (s_nop) -> chain fma, per accumulator lane.  Here the step is cut in two stages that overlap across blocks inside ONE basic block:
    //      the fragments / d / block sums of block b+1 are read from LDS while block b is computed (same tile: they have landed), and the chain updates of block b are issued
    //      behind the MFMAs of block b+1 (they only need block b's MFMA results, kept in registers).  The chains still run in block order, one fma per block: same bits.
    //      Blocks past the end of the row (the last tile's padding) run on the clamped last block with d = dmin = 0: fma(+-0, c, acc) leaves acc as it is (acc is never -0:
    //      it starts at +0, and x + (-x) is +0), so they need no branch — a branch would split the basic block and with it the overlap. ----
    struct mq_act { uint4 af[KS]; float4 dy; uint2 bs; };
    struct mq_pend { pi_f32x4 c[KS]; pi_f32x4 cm[CS]; float d[4], nd[4]; };
    auto fetch = [&](const uint8_t * blk, mq_act & A) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < KS; kk++) A.af[kk] = *(const uint4 *) (blk + (s * KS + kk) * FRAG + (NT8 ? (gq * 8 + (r & 7)) * 16 : lane * 16));
        A.dy = *(const float4 *) (blk + 8 * FRAG + 16 * gq);
        A.bs = *(const uint2 *) (blk + 8 * FRAG + 64 + (NT8 ? (gq * 16 + (r & 7)) : lane) * 8);
    };
    auto front = [&](const mq_w<TYPE, S> & W, const mq_act & A, bool live, mq_pend & P) __attribute__((always_inline)) {
        const uint32_t ddl = live ? W.dd : 0u;
        const pi_f32x2 dP = mq_pair(pi_h2f((uint16_t) (ddl & 0xffff)), pi_h2f((uint16_t) (ddl >> 16))); const float dx = dP.x, dminx = dP.y;
        const uint32_t S1 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04000400u), 0xE400E400u);
        const uint32_t S2 = pk_fma(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04010401u), 0x2C002C00u, 0xD400D400u);
        const uint32_t C1 = pk_mul(S1, 0xE400E400u), C2 = pk_mul(S2, 0xE400E400u);
        const float dy[4] = { A.dy.x, A.dy.y, A.dy.z, A.dy.w };
#pragma unroll
        for (int i = 0; i < 4; i++) { P.d[i] = dy[i] * dx; P.nd[i] = -dy[i] * dminx; }
        const pi_f32x4 z = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            const uint32_t q = qdword(W, kk), qh = q >> 8;
            uint4 bw;
            if constexpr (Q5) {
                const uint32_t t = hdword(W, kk) >> (2 * gq), th = t >> 8;
                bw.x = pk_fma((q & 0x000F000Fu) | magic | ((t & 0x00010001u) << 4), S1, C1);
                bw.y = pk_fma((qh & 0x000F000Fu) | magic | ((th & 0x00010001u) << 4), S1, C1);
                bw.z = pk_fma((q & 0x00F000F0u) | magic | ((t & 0x00020002u) << 7), S2, C2);
                bw.w = pk_fma((qh & 0x00F000F0u) | magic | ((th & 0x00020002u) << 7), S2, C2);
            } else {
                bw.x = pk_fma((q & 0x000F000Fu) | magic, S1, C1);
                bw.y = pk_fma((qh & 0x000F000Fu) | magic, S1, C1);
                bw.z = pk_fma((q & 0x00F000F0u) | magic, S2, C2);
                bw.w = pk_fma((qh & 0x00F000F0u) | magic, S2, C2);
            }
            P.c[kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pi_half8, A.af[kk]), __builtin_bit_cast(pi_half8, bw), z, 0, 0, 0);
        }
        const uint32_t M0 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04020402u), 0xE400E400u);
        const uint32_t M1 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04030403u), 0xE400E400u);
        const uint2 am = make_uint2(A.bs.x, A.bs.y);
        if constexpr (Q5) {
            if (s == 0) {
                const uint2 bm = make_uint2(M0, M1);
                P.cm[0] = mq_mfma_k16(am, bm, z);
            } else P.cm[0] = z;
        } else {
#pragma unroll
            for (int cc = 0; cc < CS; cc++) {
                const int c = s * CS + cc;
                const uint2 bm = make_uint2(M0 & own[c], M1 & own[c]);
                P.cm[cc] = mq_mfma_k16(am, bm, z);
            }
        }
    };
    auto back = [&](const mq_pend & P) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[0][kk][i] = fmaf(P.d[i], P.c[kk][i], acc[0][kk][i]);
        if constexpr (Q5) {
#pragma unroll
            for (int i = 0; i < 4; i++) { const float prod = P.nd[i] * P.cm[0][i]; accm[0][0][i] = accm[0][0][i] + prod; }      // (waves s != 0 add nd * 0: their accm is not read)
        } else {
#pragma unroll
            for (int cc = 0; cc < CS; cc++)
#pragma unroll
                for (int i = 0; i < 4; i++) accm[0][cc][i] = fmaf(P.nd[i], P.cm[cc][i], accm[0][cc][i]);
        }
    };

    // ---- main loop: NB LDS buffers in rotation, the activations of tile T + NB - 1 are requested while tile T is computed (an LDS-DMA takes
    //      about 1.1 us from issue to landing — longer than a tile of two or three blocks computes, so one tile of look-ahead stalls every
    //      tile); ring slot u is refilled for block b + D right after block b is consumed ----
    constexpr int NB = 3;
    static_assert(D == 2 * TB, "ring = two tiles");
    mq_w<TYPE, S> ring[D];
    const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) lds;
    if constexpr (!STG) {
#pragma unroll
        for (int u = 0; u < D; u++) load_w(u, ring[u]);
#pragma unroll
        for (int j = 0; j < NB - 1; j++) dma_tile(j, lds0 + j * TB * MQ_BLK_LDS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
    }
    // after the DMAs of tile T + NB - 1 a wave issues TB * LW refills in tile T; when tile T ends, the DMAs of tile T + 1 are older than
    // (NB - 2) tiles of refills + this tile's refills + at least floor(9 TB / NW) DMAs of each younger tile
    constexpr int DMA_MIN = (NP * TB) / NW, WAIT_N = (NB - 1) * TB * LW + (NB - 2) * DMA_MIN;
    // 2 NB tiles per trip: buffer (T % NB) and ring half (T % 2) are compile-time.  The uniform branches around a tile and around a block's
    // compute are deliberate: they are basic-block boundaries for the scheduler, which otherwise interleaves several blocks, doubles the
    // registers in use and spills (measured: 90 -> 124 VGPRs at S = 4, 170 -> 256 + 340 bytes of scratch at S = 1 without them)
    if constexpr (STG) {
        // ---- register-staged form: block b is computed from LDS buffer b & 1 while block b + 1's image travels global -> registers; it is written to the other buffer behind the
        //      compute and one barrier ends the step (the buffer being written was last read in step b - 1, which every wave has left through the barrier that ended it).
        //      No branch anywhere near a load: the source addresses are clamped (chunks past the launch's last one re-read that one: their results are never stored). ----
        constexpr int NTHR = 64 * S * R, UNITS = NC * 576, NL = (UNITS + NTHR - 1) / NTHR;          // 16-byte units of one block image, per thread
        const int nch = (nt + 15) >> 4;
        // (addresses are recomputed per use: a handful of VALU per 16-byte unit, against ten registers held through the whole loop)
        pi_u32x4 st[NL];                                                          // (a native vector type: an array of HIP's uint4 structs stayed in scratch memory)
        auto stage_load = [&](int b) __attribute__((always_inline)) {
            const uint32_t bb = (uint32_t) (b < nb ? b : nb - 1);
#pragma unroll
            for (int i = 0; i < NL; i++) {
                const int u = i * NTHR + (int) threadIdx.x < UNITS ? i * NTHR + (int) threadIdx.x : UNITS - 1;
                const int ch = u / 576, w = u - ch * 576, chs = ch < nch ? ch : nch - 1;
                const size_t cb = (size_t) chs * nb + bb;
                const uint8_t * src = w < 512 ? g.frag + cb * 8192 + (size_t) w * 16 : g.aux + cb * 1024 + (size_t) (w - 512) * 16;
                st[i] = *(const pi_u32x4 *) src;
            }
        };
        auto stage_store = [&](uint8_t * buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NL; i++) {
                const int u = i * NTHR + (int) threadIdx.x < UNITS ? i * NTHR + (int) threadIdx.x : UNITS - 1;
                const int ch = u / 576, w = u - ch * 576;
                *(pi_u32x4 *) (buf + ch * CH_LDS + w * 16) = st[i];
            }
        };
        mq_w<TYPE, S> W0, W1;
        stage_load(0);
        load_w(0, W0); load_w(1, W1);
        stage_store(lds);
        __syncthreads();
        for (int b0 = 0; b0 < nb; b0 += 2) {
            stage_load(b0 + 1);
            compute(W0, lds, true);
            load_w(b0 + 2, W0);
            stage_store(lds + MQ_BLK_LDS);
            __syncthreads();
            if (b0 + 1 >= nb) break;
            stage_load(b0 + 2);
            compute(W1, lds + MQ_BLK_LDS, true);
            load_w(b0 + 3, W1);
            stage_store(lds);
            __syncthreads();
        }
        __syncthreads();
    } else
    if constexpr (PIPE) {
        static_assert(TB % 2 == 0, "the pending sets alternate with the block parity");
        mq_act A0, A1;
        mq_pend P0, P1;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) P1.c[kk] = pi_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int cc = 0; cc < CS; cc++) P1.cm[cc] = pi_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int i = 0; i < 4; i++) { P1.d[i] = 0.0f; P1.nd[i] = 0.0f; }
        auto pipe_tile = [&](const int T, const int j) __attribute__((always_inline)) {
            dma_tile(T + NB - 1, lds0 + ((j + NB - 1) % NB) * TB * MQ_BLK_LDS);
            const uint8_t * buf = lds + (j % NB) * TB * MQ_BLK_LDS;
            fetch(buf, A0);                                            // (the tile landed before the barrier that ended the previous tile)
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int b = T * TB + u;
                mq_w<TYPE, S> & W = ring[(j % 2) * TB + u];
                mq_act & Ac = (u & 1) ? A1 : A0;
                mq_act & An = (u & 1) ? A0 : A1;
                if (u + 1 < TB) fetch(buf + (u + 1) * MQ_BLK_LDS, An);
                // (the uniform branch is deliberate, as in the plain form: a basic-block boundary per block step — without it the steps of a tile are one scheduling
                //  region, the scheduler overlaps them all and spills.  b == nb still enters: it retires block nb - 1 beside a dead block.)
                // (two pending sets used alternately — TB is even, so a block's parity is its u's — instead of a copy per block)
                mq_pend & Pc = (u & 1) ? P1 : P0;
                mq_pend & Pp = (u & 1) ? P0 : P1;
                if (b <= nb) {
                    front(W, Ac, b < nb, Pc);
                    back(Pp);
                }
                load_w(b + D, W);
            }
            if (T + NB - 1 < ntile) MQ_WAIT_VM(WAIT_N); else MQ_WAIT_VM((NB - 1) * TB * LW);
            wg_barrier();
        };
        // whole trips in a loop with ONE exit, the last ntile % (2 NB) tiles as straight-line code behind it (see the plain form below: an exit in the middle of a trip
        // makes the wait-count pass drain the ring at the top of every trip)
        const int nfull = ntile / (2 * NB) * (2 * NB);
        for (int T0 = 0; T0 < nfull; T0 += 2 * NB) {
#pragma unroll
            for (int j = 0; j < 2 * NB; j++) pipe_tile(T0 + j, j);
        }
#pragma unroll
        for (int j = 0; j < 2 * NB - 1; j++) {
            if (nfull + j >= ntile) goto pipe_done;
            pipe_tile(nfull + j, j);
        }
pipe_done:
        // the last block's chains: block step nb retired them when the row's last tile had room for it (what is pending then is a dead set); a row of whole tiles ends on an
        // odd block, whose set is still pending
        if (nb % TB == 0) back(P1);
    } else {
        // One tile: DMA request for tile T + NB - 1, TB block steps (compute, then the slot's refill for block b + D), the wave's wait for tile T + 1's DMAs, the barrier.
        auto tile_step = [&](const int T, const int j) __attribute__((always_inline)) {
            dma_tile(T + NB - 1, lds0 + ((j + NB - 1) % NB) * TB * MQ_BLK_LDS);
            const uint8_t * buf = lds + (j % NB) * TB * MQ_BLK_LDS;
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int b = T * TB + u;
                mq_w<TYPE, S> & W = ring[(j % 2) * TB + u];
                if (dbg & 1) acc[0][0][0] += __builtin_bit_cast(float, W.dd ^ W.rec ^ qdword(W, 0) ^ qdword(W, KS - 1));
                else if (b < nb) compute(W, buf + u * MQ_BLK_LDS, true);
                load_w(b + D, W);
            }
            if (!(dbg & 2)) {
                if (T + NB - 1 < ntile) MQ_WAIT_VM(WAIT_N); else MQ_WAIT_VM((NB - 1) * TB * LW);      // (the last tiles request nothing new)
            }
            wg_barrier();
        };
        if constexpr (NC == 1) {
            // One chunk.  The compiler's wait-count pass works on merged states: every edge INTO a block contributes what may still be in flight there.  A trip of 2 NB tiles
            // that can be left in the middle (`if (T >= ntile) goto done`, which the structuriser routes through the loop header) handed the header the ring as the allocator
            // had it named at that exit — other registers than at the bottom of the trip — and the first blocks of every trip then waited for "pending" loads into registers
            // they use as temporaries: vmcnt(1), (4), (7), (10) ... in front of their LDS reads, a drain of the 8-block ring per trip (ISA of round 5; tools_dev/scan_vmcnt.py
            // lists such waits).  So: whole trips in a loop that has ONE exit, the bottom, and the row's last ntile % (2 NB) tiles as straight-line code behind it.
            const int nfull = ntile / (2 * NB) * (2 * NB);
            for (int T0 = 0; T0 < nfull; T0 += 2 * NB) {
#pragma unroll
                for (int j = 0; j < 2 * NB; j++) tile_step(T0 + j, j);
            }
#pragma unroll
            for (int j = 0; j < 2 * NB - 1; j++) {
                if (nfull + j >= ntile) goto tiles_done;
                tile_step(nfull + j, j);
            }
        } else {
            // Two chunks keep the skip form: without the join the scheduler pipelines across their one-block tiles and runs out of registers (256 + spills).
            for (int T0 = 0; T0 < ntile; T0 += 2 * NB) {
#pragma unroll
                for (int j = 0; j < 2 * NB; j++) {
                    const int T = T0 + j;
                    if (T < ntile) tile_step(T, j);
                }
            }
        }
    }
tiles_done:
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (no DMA may still be writing LDS when the fold reuses it)
    __syncthreads();

    // ---- fold: the eight chains of every (row, token) like hsum_float_8, the four mins chains like the reference's two hadds; chunk by chunk through the same LDS ----
    float * F = (float *) lds + (size_t) rgl * (8 * 16 * 17 + 4 * 16 * 17 + 256);       // [8 k][16 rows][17]
    float * FM = F + 8 * 16 * 17;                                                         // [4 c][16 rows][17]
    float * X = FM + 4 * 16 * 17;                                                         // [16 rows][16 tokens] results (RoPE epilogue)
    const int tid_g = s * 64 + lane;                                                      // thread inside the row group
#pragma unroll
    for (int ch = 0; ch < NC; ch++) {
        const int ntc = nt - 16 * ch < 16 ? nt - 16 * ch : 16, tokc = g.tok0 + 16 * ch;  // tokens of this chunk, index of its first token in the pass
        if (ch > 0) __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
#pragma unroll
            for (int i = 0; i < 4; i++) F[((s * KS + kk) * 16 + r) * 17 + 4 * gq + i] = acc[ch][kk][i];
#pragma unroll
        for (int cc = 0; cc < CS; cc++)
#pragma unroll
            for (int i = 0; i < 4; i++) FM[((s * CS + cc) * 16 + r) * 17 + 4 * gq + i] = accm[ch][cc][i];
        __syncthreads();
        if (active) {
            for (int idx = tid_g; idx < 256; idx += 64 * S) {
                const int rr = idx >> 4, t = idx & 15;
                float a[8];
#pragma unroll
                for (int k = 0; k < 8; k++) a[k] = F[(k * 16 + rr) * 17 + t];
                float v = ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
                const float m0 = FM[(0 * 16 + rr) * 17 + t], m1 = FM[(1 * 16 + rr) * 17 + t], m2 = FM[(2 * 16 + rr) * 17 + t], m3 = FM[(3 * 16 + rr) * 17 + t];
                v = (Q6 || Q3) ? v : Q5 ? v + m0 : v + ((m0 + m2) + (m1 + m3));                                   // Q5_K: hsum_float_8(acc) + summs
                if (rr < rows && t < ntc) {
                    if (sg.store) X[rr * 16 + t] = v;
                    else {
                        const int64_t o = (int64_t) (16 * ch + t) * sg.M + row0 + rr;
                        sg.out[o] = sg.res ? sg.res[o] + v : v;
                    }
                }
            }
        }
        if (g.seg[0].store) {                                                             // (a launch either stores all its segments through the epilogue or none)
            // llm_build_kv_store / ggml_rope mode 0 (llama.cpp:3546-3577, ggml.c:11062-11163) as the epilogue of the projection, exactly as in
            // k_mmvq_kq: rows 2i, 2i+1 of a head are neighbours in this row group; the rounding points are those of k_rope_kv
            __syncthreads();
            if (active && ntc > 0) {
                if (sg.store == 3) {
                    for (int j = tid_g; j < rows * ntc; j += 64 * S) {
                        const int rr = j / ntc, t = j % ntc;
                        sg.dst16[(int64_t) pi_cell_of(g.dyn, g.kv_head, tokc + t) * sg.stride16 + row0 + rr] = pi_f2h(X[rr * 16 + t]);
                    }
                } else {
                    for (int j = tid_g; j < (rows / 2) * ntc; j += 64 * S) {
                        const int pr = j / ntc, t = j % ntc;
                        const float x0 = X[(2 * pr) * 16 + t], x1 = X[(2 * pr + 1) * 16 + t];
                        const int64_t rw = row0 + 2 * pr;
                        const int i = (int) (rw % g.hd) >> 1;
                        const float2 cs = *(const float2 *) (g.rope_cs + ((int64_t) g.pos[tokc + t] * (g.hd / 2) + i) * 2);
                        uint16_t * dd = sg.dst16 + (int64_t) (sg.store == 2 ? pi_cell_of(g.dyn, g.kv_head, tokc + t) : tokc + t) * sg.stride16 + rw;
                        dd[0] = pi_f2h(x0 * cs.x - x1 * cs.y);
                        dd[1] = pi_f2h(x0 * cs.y + x1 * cs.x);
                    }
                }
            }
        }
    }
}

template <int TYPE, int S, int R, int TB, bool NT8, int NC = 1, bool PIPE = false>
__global__ void __launch_bounds__(64 * S * R) k_mmq_kq(const mmvq_args g, const int n_groups) {
    if (pi_aborted(g.abort_w, 1)) return;
    mmq_kq_body<TYPE, S, R, TB, NT8, NC, PIPE, NC == 4>(g, n_groups, (int) blockIdx.x, 0);
}
// Two weight types in ONE matrix-core launch (<= 8 tokens): the Q4_K_M mixtures keep attn_q / attn_k in Q4_K and put attn_v into Q5_K or Q6_K (llama.cpp:7923-7933).  At one
// or two tokens k_mmvq_kq2 carries all three; from three tokens attn_v used to be a launch of its own behind q + k (64 row groups of the 70B model: 12-15 us on the VALU
// kernel, all of it launch latency).  Here the first `wgs_a` workgroups walk the TA segments (row groups [0, groups_a)) and the rest the TB segments (the launch's LAST
// segments) — a workgroup is of one type, so each keeps its own tile format, blocks per tile and barriers; both parts share S and R (the block size).  The v part runs in the
// shadow of q + k: 192 + 22 workgroups of the 70B model are still one round.  Same bits as the single-type launches (the body is the same code).
// RB: row groups per workgroup of the SECOND type's part.  The part is small (64 row groups of the 70B model) and, in Q6_K, slow per block: three row groups to a workgroup made
// it the launch's long pole (27-29 us against 21 for q + k alone); with RB = 1 it spreads over 64 CUs, one wave per SIMD (the workgroup's other waves leave at once — a
// terminated wave does not take part in barriers), and the launch is still one round: 192 + 64 workgroups.
// NC = 4: the same for prompt passes (33..64 tokens, register-staged activations): attn_v's 64 row groups were a 37-41 us launch of their own per 64 tokens.
template <int TA, int TB_, int S, int R, int RB, int TBA, int TBB, int NC = 1>
__global__ void __launch_bounds__(64 * S * R) k_mmq_kq2t(const mmvq_args g, const int groups_a, const int groups_all, const int wgs_a) {
    static_assert(NC == 1 || (NC == 4 && TBA == 1 && TBB == 1), "verify widths (<= 8 tokens) or four chunks");
    if (pi_aborted(g.abort_w, 1)) return;
    if ((int) blockIdx.x < wgs_a) mmq_kq_body<TA, S, R, TBA, NC == 1, NC, false, NC == 4>(g, groups_a, (int) blockIdx.x, 0);
    else {
        if constexpr (RB < R) { if ((int) threadIdx.x >= 64 * S * RB) return; }
        mmq_kq_body<TB_, S, RB, TBB, NC == 1, NC, false, NC == 4>(g, groups_all, (int) blockIdx.x - wgs_a, groups_a);
    }
}

// =================================================================================================================
// k_mmq_g2<R, TB>: Q4_K, <= 8 tokens, tall launches (the fused up + gate of a layer: 3 584 row groups = 14 per CU) — two row groups per wave, packed accumulators.
//
// Why: the counters (tools_dev/exp_mmq_pmc.sh) show k_mmq_kq's one-wave row groups at two waves per SIMD issuing in 32 % of their cycles, parked on a wait in 42 % and
// stalled on a dependency in 26 %: the VALU is busy 40-50 % of the time while the one-token kernel, with MORE vector instructions for the same matrix, finishes 25 % earlier
// at seven waves per SIMD.  The kernel is latency-bound; 208 VGPRs leave no room for a third wave, and 3 584 row groups on 2 048 wave slots are two rounds.
// Here a wave walks TWO row groups at once: their block steps are independent instruction streams inside one basic block (the scheduler fills one row group's MFMA and LDS
// latencies with the other's operand build), the activation fragments are read from LDS once for both, and seven such waves per CU take the launch in ONE round.
//
// Packed accumulators make the registers fit: with <= 8 tokens only rows 0..7 of the 16-token MFMA tile are real.  Accumulator lanes k and k + 4 share one tile — the A
// operand of the first MFMA holds the tokens' fragment of lane k in rows 0..7 and ZEROS in rows 8..15, the second (accumulating) MFMA holds zeros in rows 0..7 and the
// fragment of lane k + 4 in rows 8..15 — so C rows 0..7 are lane k's integer sums and rows 8..15 lane k + 4's, each exact (the other product adds exact zeros), and one chain
// fma per tile register extends both chains: 16 + 8 chain registers per row group instead of 32 + 16, half the chain fmas.  The zero rows come from a 16-byte zero slot in the
// block's LDS image (bytes 576..591 of the aux piece, written by the quantisers): per-lane LDS addresses pick the fragment or the slot, no VALU.  The mins use ONE MFMA per
// pair of chains (c, c + 2): A rows 0..7 carry the block sums only in K-slice c, rows 8..15 only in K-slice c + 2, B carries the mins of both slices.
// The fold needs no LDS: all eight chains of a (row, token) sit in two lanes 32 apart (v_permlane32_swap), summed in hsum_float_8's order.
// Same bits as k_mmq_kq, the VALU kernel and the reference (test_mul_mat_vs_oracle runs every shape through it).
// =================================================================================================================
struct g2_w { uint32_t dd, rec; pi_u32x4 q0, q1; };      // one lane's share of one Q4_K block of one row group: (d, dmin), its scales / mins record, 8 quant dwords
template <int R, int TB>
__global__ void __launch_bounds__(64 * R) k_mmq_g2(const mmvq_args g, const int n_groups) {
    constexpr uint32_t TILE = PI_MQ_TILE, QOFF = 320;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if (pi_aborted(g.abort_w, 1)) return;
    constexpr int NW = R, D = 2 * TB, LW = 4, NB = 3;
    constexpr int FRAG = 512, MQ_BLK_LDS = 8 * FRAG + 1024, NP = 5, ZSLOT = 8 * FRAG + 576;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, gq = lane >> 4;
    const int nb = g.nb, nt = g.nt, ntile = (nb + TB - 1) / TB;
    // the wave's two row groups (the second may not exist: it then walks the launch's last row group again and stores nothing)
    const int rg0 = ((int) blockIdx.x * R + wave) * 2;
    uint32_t magic = 0x64006400u;
    asm volatile("" : "+v"(magic));

    struct rowgroup { const uint8_t * wbase; const mmvq_seg * sg; int64_t row0; int rows; bool active; };
    rowgroup G[2];
#pragma unroll
    for (int x = 0; x < 2; x++) {
        const int raw = rg0 + x;
        G[x].active = raw < n_groups;
        const int rg = G[x].active ? raw : n_groups - 1;
        const int si = (rg >= g.seg[1].wave0) + (rg >= g.seg[2].wave0);
        const mmvq_seg & sg = g.seg[si];
        G[x].sg = &sg;
        G[x].row0 = (int64_t) (rg - sg.wave0) * 16;
        G[x].rows = sg.M - G[x].row0 < 16 ? (int) (sg.M - G[x].row0) : 16;
        G[x].wbase = sg.mq + (size_t) (rg - sg.wave0) * nb * TILE;
    }
    // The weight ring is loaded and waited for by hand.  With compiler-visible loads the wait-count pass meets, behind `if (b < nb) compute`, a path on which a slot's pending
    // load was never waited for, and drains the ring in front of the slot's refill (vmcnt(8) with 32 loads in flight); without the branch the block steps of a tile become one
    // scheduling region and spill.  Loads the compiler cannot see leave the counting to us: a block's four loads per row group are issued by one asm statement, every block is
    // consumed behind `s_waitcnt vmcnt(G2_WAIT)` naming the slot's registers (consumers cannot be scheduled above it; the slot stays live — unused by anything else — from its
    // load to that wait on EVERY path), and the ring is drained the same way before the fold reuses its registers.  vmcnt retires in order; LDS-DMAs issued in between only
    // make the wait more conservative.
    const uint32_t dd_off = 4 * r, rec_off = 64 + 4 * lane, q_off = QOFF + 16 * lane;
    auto load_w = [&](int x, int b, g2_w & W) __attribute__((always_inline)) {
        const uint8_t * wb = G[x].wbase + (size_t) (b < nb ? b : nb - 1) * TILE;          // wave-uniform: an SGPR pair
        asm volatile("global_load_dword %0, %4, %7\n\tglobal_load_dword %1, %5, %7\n\tglobal_load_dwordx4 %2, %6, %7\n\tglobal_load_dwordx4 %3, %6, %7 offset:1024"
                     : "=&v"(W.dd), "=&v"(W.rec), "=&v"(W.q0), "=&v"(W.q1) : "v"(dd_off), "v"(rec_off), "v"(q_off), "s"(wb));
    };
    // the loads of slot pair (Wa, Wb) have landed when at most `N` younger vector-memory operations are outstanding
    #define G2_LANDED(N, Wa, Wb) asm volatile("s_waitcnt vmcnt(%8) ; landed: %0 %1 %2 %3 %4 %5 %6 %7" : "+v"((Wa).dd), "+v"((Wa).rec), "+v"((Wa).q0), "+v"((Wa).q1), "+v"((Wb).dd), "+v"((Wb).rec), "+v"((Wb).q0), "+v"((Wb).q1) : "n"(N))
    auto dma_tile = [&](int T, uint32_t buf) __attribute__((always_inline)) {
        if (T < ntile) {
#pragma unroll
            for (int i = 0; i < (NP * TB + NW - 1) / NW; i++) {
                const int piece = i * NW + wave;
                if (piece < NP * TB) {
                    const int u = piece / NP, part = piece % NP;
                    const int b = T * TB + u < nb ? T * TB + u : nb - 1;
                    const uint8_t * src;
                    if (part == NP - 1) src = g.aux + (size_t) b * 1024 + lane * 16;
                    else src = g.frag + (((size_t) b * 8 + 2 * part + (lane >> 5)) * 64 + (lane & 7) + 16 * ((lane & 31) >> 3)) * 16;       // two fragments per piece, tokens 0..7 only
                    const uint32_t dst = buf + u * MQ_BLK_LDS + part * 1024;
                    uint32_t keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                }
            }
        }
    };
    auto wg_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // per-lane LDS offsets inside a block image.  Pair p = accumulator lanes (p, p + 4): operand 1 = lane p's fragment in token rows 0..7, the zero slot in rows 8..15;
    // operand 2 = the zero slot in rows 0..7, lane p + 4's fragment (token r - 8) in rows 8..15.  Mins pair m = chains (m, m + 2): the block sums of K-slice m in rows 0..7,
    // of K-slice m + 2 in rows 8..15, zeros elsewhere.
    const bool lo = r < 8;
    const uint32_t fr_off = (uint32_t) (gq * 8 + (r & 7)) * 16;
    uint32_t a1[4], a2[4], am_off[2];
#pragma unroll
    for (int p = 0; p < 4; p++) { a1[p] = lo ? (uint32_t) p * FRAG + fr_off : (uint32_t) ZSLOT; a2[p] = lo ? (uint32_t) ZSLOT : (uint32_t) (p + 4) * FRAG + fr_off; }
#pragma unroll
    for (int m = 0; m < 2; m++) am_off[m] = (lo ? gq == m : gq == m + 2) ? (uint32_t) (8 * FRAG + 64 + (gq * 16 + (r & 7)) * 8) : (uint32_t) ZSLOT;
    const uint32_t dy_off = 8 * FRAG + 16 * (gq & 1);                                     // d of tokens 4 (gq & 1) .. + 3: the tokens of this lane's tile registers in both halves
    const uint32_t pairmask[2] = { (gq == 0 || gq == 2) ? 0xffffffffu : 0u, (gq == 1 || gq == 3) ? 0xffffffffu : 0u };

    float acc[2][4][4], accm[2][2][4];
#pragma unroll
    for (int x = 0; x < 2; x++) {
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[x][p][i] = 0.0f;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int i = 0; i < 4; i++) accm[x][m][i] = 0.0f;
    }

    // one block of both row groups: the fragments are read once; per row group the block scalars, 8 operands, 8 + 2 MFMAs and 16 + 8 chain fmas
    auto compute2 = [&](const g2_w & Wa, const g2_w & Wb, const uint8_t * blk) __attribute__((always_inline)) {
        const float4 dy4 = *(const float4 *) (blk + dy_off);
        const float dy[4] = { dy4.x, dy4.y, dy4.z, dy4.w };
        uint4 A1[4], A2[4];
#pragma unroll
        for (int p = 0; p < 4; p++) { A1[p] = *(const uint4 *) (blk + a1[p]); A2[p] = *(const uint4 *) (blk + a2[p]); }
        uint2 AM[2];
#pragma unroll
        for (int m = 0; m < 2; m++) AM[m] = *(const uint2 *) (blk + am_off[m]);
        const pi_f32x4 z = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int x = 0; x < 2; x++) {
            const g2_w & W = x ? Wb : Wa;
            const float dx = pi_h2f((uint16_t) (W.dd & 0xffff)), dminx = pi_h2f((uint16_t) (W.dd >> 16));       // (not through mq_pair: this kernel's registers are hand-counted)
            const uint32_t S1 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04000400u), 0xE400E400u);
            const uint32_t S2 = pk_fma(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04010401u), 0x2C002C00u, 0xD400D400u);
            const uint32_t C1 = pk_mul(S1, 0xE400E400u), C2 = pk_mul(S2, 0xE400E400u);
            float d[4], nd[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { d[i] = dy[i] * dx; nd[i] = -dy[i] * dminx; }
            auto operand = [&](uint32_t q) __attribute__((always_inline)) -> uint4 {
                const uint32_t qh = q >> 8;
                uint4 bw;
                bw.x = pk_fma((q & 0x000F000Fu) | magic, S1, C1);
                bw.y = pk_fma((qh & 0x000F000Fu) | magic, S1, C1);
                bw.z = pk_fma((q & 0x00F000F0u) | magic, S2, C2);
                bw.w = pk_fma((qh & 0x00F000F0u) | magic, S2, C2);
                return bw;
            };
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t qa = W.q0[p], qb = W.q1[p];                                            // accumulator lanes p and p + 4
                const uint4 ba = operand(qa), bb = operand(qb);
                pi_f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pi_half8, A1[p]), __builtin_bit_cast(pi_half8, ba), z, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pi_half8, A2[p]), __builtin_bit_cast(pi_half8, bb), c, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) acc[x][p][i] = fmaf(d[i], c[i], acc[x][p][i]);
            }
            const uint32_t M0 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04020402u), 0xE400E400u);
            const uint32_t M1 = pk_add(__builtin_amdgcn_perm(0x64646464u, W.rec, 0x04030403u), 0xE400E400u);
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const uint2 am = make_uint2(AM[m].x, AM[m].y);
                const uint2 bm = make_uint2(M0 & pairmask[m], M1 & pairmask[m]);
                const pi_f32x4 cm = mq_mfma_k16(am, bm, z);
#pragma unroll
                for (int i = 0; i < 4; i++) accm[x][m][i] = fmaf(nd[i], cm[i], accm[x][m][i]);
            }
        }
    };

    static_assert(D == 2 * TB, "ring = two tiles");
    g2_w ring[2][D];
#pragma unroll
    for (int u = 0; u < D; u++) { load_w(0, u, ring[0][u]); load_w(1, u, ring[1][u]); }
    const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) lds;
#pragma unroll
    for (int j = 0; j < NB - 1; j++) dma_tile(j, lds0 + j * TB * MQ_BLK_LDS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    constexpr int DMA_MIN = (NP * TB) / NW, WAIT_N = (NB - 1) * TB * 2 * LW + (NB - 2) * DMA_MIN;
    for (int T0 = 0; T0 < ntile; T0 += 2 * NB) {
#pragma unroll
        for (int j = 0; j < 2 * NB; j++) {
            const int T = T0 + j;
            if (T >= ntile) goto tiles_done;                   // (leave, do not skip: see k_mmq_kq)
            dma_tile(T + NB - 1, lds0 + ((j + NB - 1) % NB) * TB * MQ_BLK_LDS);
            const uint8_t * buf = lds + (j % NB) * TB * MQ_BLK_LDS;
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int b = T * TB + u;
                g2_w & Wa = ring[0][(j % 2) * TB + u];
                g2_w & Wb = ring[1][(j % 2) * TB + u];
                G2_LANDED((D - 1) * 2 * LW, Wa, Wb);          // (the D - 1 younger blocks' loads may stay in flight)
                if (b < nb) compute2(Wa, Wb, buf + u * MQ_BLK_LDS);
                load_w(0, b + D, Wa); load_w(1, b + D, Wb);
            }
            if (T + NB - 1 < ntile) MQ_WAIT_VM(WAIT_N); else MQ_WAIT_VM((NB - 1) * TB * 2 * LW);
            wg_barrier();
        }
    }
tiles_done:
    // (the refills past the end of the row are still in flight: their registers must not be reused before they have landed)
#pragma unroll
    for (int u = 0; u < D; u++) G2_LANDED(0, ring[0][u], ring[1][u]);
    #undef G2_LANDED
    // ---- fold in registers.  Lane (r, gq < 2) holds chains p of (row r, token 4 gq + i), lane (r, gq + 2) the chains p + 4 of the same row and token: hsum_float_8 =
    //      ((a0 + a4) + (a2 + a6)) + ((a1 + a5) + (a3 + a7)) with a_p + a_{p+4} = own + the lane 32 further (the same sum in both lanes: fp addition commutes);
    //      mins ((m0 + m2) + (m1 + m3)) likewise: pair m holds chain m in the low half and chain m + 2 in the high half ----
#pragma unroll
    for (int x = 0; x < 2; x++) {
        if (!G[x].active) continue;
        const mmvq_seg & sg = *G[x].sg;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float sp[4], sm[2];
#pragma unroll
            for (int p = 0; p < 4; p++) sp[p] = acc[x][p][i] + pi_partner<32>(acc[x][p][i]);
#pragma unroll
            for (int m = 0; m < 2; m++) sm[m] = accm[x][m][i] + pi_partner<32>(accm[x][m][i]);
            float v = (sp[0] + sp[2]) + (sp[1] + sp[3]);
            v = v + (sm[0] + sm[1]);
            const int t = 4 * gq + i;
            if (gq < 2 && r < G[x].rows && t < nt) {
                const int64_t o = (int64_t) t * sg.M + G[x].row0 + r;
                sg.out[o] = sg.res ? sg.res[o] + v : v;
            }
        }
    }
}

#include <hip/hip_ext.h>

// PI_MMQ_PIPE=0 keeps every launch on the plain block step (A/B measurements)
static bool mmq_pipe_on() { static const bool on = !(getenv("PI_MMQ_PIPE") && atoi(getenv("PI_MMQ_PIPE")) == 0); return on; }
template <int TYPE, int S, int R, int TB, bool NT8, int NC = 1, bool PIPE = false>
static void mmq_go2(const mmvq_args & g, int groups, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr size_t tiles = (size_t) (NC == 4 ? 2 : 3) * TB * NC * (8 * (NT8 ? 512 : 1024) + 1024), fold = (size_t) R * (8 * 16 * 17 + 4 * 16 * 17 + 256) * 4;
    constexpr size_t lds = tiles > fold ? tiles : fold;
    static_assert(NC == 4 || tiles <= 65536, "LDS-DMA targets stay inside the first 64 KB");
    // (Two blocks per scheduling region — the block-axis analogue of k_mmq_g2's two row groups — was built and measured too: no gain on any split-row-group shape,
    //  Q6_K 8192 x 28672 62.7 -> 67.5 us; removed.)
    // the pipelined block step pays at 9..16 tokens (same-box A/B, us per launch: 8192 x 8192 18.70 -> 17.66, 8192 x 28672 53.7 -> 50.2) and is a wash at <= 8 (17.02 / 17.17, 44.7 / 44.4)
    if constexpr (!PIPE && NC == 1 && S >= 2 && (TYPE == PI_Q4_K || TYPE == PI_Q5_K) && !NT8) {
        if (!g.frag || mmq_pipe_on()) { mmq_go2<TYPE, S, R, TB, NT8, NC, true>(g, groups, stream, ea, eb); if (g.frag) return; }      // (reserve: both forms set their attribute)
    }
    if (!g.frag) { (void) hipFuncSetAttribute((const void *) k_mmq_kq<TYPE, S, R, TB, NT8, NC, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); return; }   // pi_op_mul_mat_reserve
    const unsigned grid = (unsigned) ((groups + R - 1) / R);
    if (ea) hipExtLaunchKernelGGL((k_mmq_kq<TYPE, S, R, TB, NT8, NC, PIPE>), dim3(grid), dim3(64 * S * R), lds, stream, ea, eb, 0, g, groups);
    else    hipLaunchKernelGGL((k_mmq_kq<TYPE, S, R, TB, NT8, NC, PIPE>), dim3(grid), dim3(64 * S * R), lds, stream, g, groups);
}
// batches of <= 8 tokens stage half the fragment bytes and take twice the blocks per tile (TB8) in the same LDS; 17..32 tokens are two chunks over one weight
// stream, one block per tile (the three tiles of two chunks are 54 KB)
template <int TYPE, int S, int R, int TB16, int TB8>
static void mmq_go(const mmvq_args & g, int groups, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    if (g.nt <= 8) mmq_go2<TYPE, S, R, TB8, true>(g, groups, stream, ea, eb);
    else if (g.nt <= 16) mmq_go2<TYPE, S, R, TB16, false>(g, groups, stream, ea, eb);
    else if (g.nt <= 32) mmq_go2<TYPE, S, R, 1, false, 2>(g, groups, stream, ea, eb);
    else if constexpr (S == 4) mmq_go2<TYPE, S, R, 1, false, 4>(g, groups, stream, ea, eb);   // prompt passes: four chunks over one weight stream, register-staged activations
    else if (!g.frag) mmq_go2<TYPE, S, R, 1, false, 2>(g, groups, stream, ea, eb);            // (pi_op_mul_mat_reserve walks every (cfg, width): nothing is launched)
    else pi_set_error("mmq: a launch of %d tokens reached a %d-wave configuration (only four-wave row groups walk four chunks)", g.nt, S);   // not reached: mmq_dispatch_t sends launches of > 32 tokens to 41 / 42 / 43; an error, never half a result
}

template <int TYPE>
static void mmq_dispatch_t(const mmvq_args & g, int groups, int cfg, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int T8 = TYPE == PI_Q4_K ? 4 : 2;                 // blocks per tile at <= 8 tokens (Q5_K: the ring carries the fifth bits too)
    // Two chunks double the chains a wave carries (96 accumulators on a one-wave row group).  Round 3 sent such launches to two-wave row groups (it spilled with the 8-block ring);
    // with the one-block tiles of the two-chunk form (a 2-block ring) the one-wave instance fits — 254 VGPRs, no spill — and the tall launches stop repeating the block scalars in
    // two waves and take two rounds of workgroups instead of 3.5.  Same-box A/B at 24 / 32 tokens, us per launch: 57344 x 8192 139.3 / 138.8 -> 102.2 / 100.7, 28672 x 8192
    // 72.1 / 71.6 -> 54.6 / 53.3, 44032 x 4096 59.1 / 58.7 -> 56.2 / 55.8, Q6_K 32000 x 8192 78.0 / 75.2 -> 77.9 / 75.8; 70B prompts 987 -> 1 098 tok/s.  PI_MMQ_S1NC2=0: round 3's choice.
    static const bool s1nc2 = !(getenv("PI_MMQ_S1NC2") && atoi(getenv("PI_MMQ_S1NC2")) == 0);
    if (g.nt > 16 && (cfg == 17 || cfg == 18 || cfg == 14) && !s1nc2) cfg = 24;
    // four chunks (33..64 tokens): a one-wave row group would carry 4 x 48 chain registers; two-wave row groups (4 x 24) keep them in registers
    // four chunks: four-wave row groups (48 chain registers per wave; two-wave row groups need 96 + the staged block image: 256 VGPRs and ~90 spilled inside the loop).
    // Tall launches (one-wave row groups) do not come here: pi_mmq_step keeps them at 32 tokens per launch.  PI_MMQ_PP_CFG overrides (A/B).
    // (only the four-wave configurations walk four chunks: any other value would compute the first 32 tokens of a 33..64-token launch and leave the rest unwritten — refused, loudly, once)
    static const int pp_cfg = [] {
        const int v = getenv("PI_MMQ_PP_CFG") ? atoi(getenv("PI_MMQ_PP_CFG")) : 0;
        if (v && v != 41 && v != 42 && v != 43) { fprintf(stderr, "PI_MMQ_PP_CFG=%d ignored: launches of more than 32 tokens need a four-wave configuration (41, 42, 43)\n", v); return 0; }
        return v;
    }();
    if (g.nt > 32) { if (pp_cfg) cfg = pp_cfg; else if (cfg != 41 && cfg != 42 && cfg != 43) cfg = 43; }
    switch (cfg) {                                             // (S waves per row group)(R row groups per workgroup); blocks per tile for 16 / <= 8 tokens
        case 17: mmq_go<TYPE, 1, 7, 2, T8>(g, groups, stream, ea, eb); break;
        case 18: mmq_go<TYPE, 1, 8, 2, T8>(g, groups, stream, ea, eb); break;
        case 14: mmq_go<TYPE, 1, 4, 2, T8>(g, groups, stream, ea, eb); break;
        case 22: mmq_go<TYPE, 2, 2, 2, T8>(g, groups, stream, ea, eb); break;
        case 24: mmq_go<TYPE, 2, 4, 2, T8>(g, groups, stream, ea, eb); break;
        case 41: mmq_go<TYPE, 4, 1, 2, 4>(g, groups, stream, ea, eb); break;
        case 43: mmq_go<TYPE, 4, 3, 2, 4>(g, groups, stream, ea, eb); break;
        default: mmq_go<TYPE, 4, 2, 2, 4>(g, groups, stream, ea, eb); break;
    }
}
// two row groups per wave (k_mmq_g2): R waves per workgroup so that the launch is ONE round of <= 256 workgroups
template <int R>
static void mmq_g2_go(const mmvq_args & g, int groups, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int TB = 2;
    constexpr size_t lds = (size_t) 3 * TB * (8 * 512 + 1024);
    const unsigned grid = (unsigned) ((groups + 2 * R - 1) / (2 * R));
    if (ea) hipExtLaunchKernelGGL((k_mmq_g2<R, TB>), dim3(grid), dim3(64 * R), lds, stream, ea, eb, 0, g, groups);
    else    hipLaunchKernelGGL((k_mmq_g2<R, TB>), dim3(grid), dim3(64 * R), lds, stream, g, groups);
}
// Q4_K, <= 8 tokens, plain fp32 output (no RoPE / KV-store epilogue), tall launches.  PI_MMQ_G2=0 switches it off, PI_MMQ_G2_MIN moves the bound.
static bool mmq_g2_wants(int type, const mmvq_args & g, int nseg, int groups) {
    static const bool on = !(getenv("PI_MMQ_G2") && atoi(getenv("PI_MMQ_G2")) == 0);
    // every launch of the one-wave-per-row-group range (> 1152 row groups).  Same-box A/B against k_mmq_kq, us per launch at 4 / 6 / 8 tokens: 57344 x 8192 (the 70B up + gate,
    // 3 584 row groups: two rounds before, one now) 71.5 / 71.8 / 71.8 -> 55.1 / 55.5 / 55.9 (one token: 52.8); 44032 x 4096 (7B up + gate) 42.2 -> 27.9; the
    // launches that were one round already gain little: 28672 x 8192 39.0 / 37.1 / 37.7 -> 38.4 / 37.1 / 36.8, 32768 x 8192 41.4 / 39.4 / 39.9 -> 39.9 / 38.3 / 38.4,
    // 22016 x 4096 22.4 / 21.8 / 21.9 -> 21.5 / 21.3 / 21.2
    static const int min_groups = getenv("PI_MMQ_G2_MIN") ? atoi(getenv("PI_MMQ_G2_MIN")) : 1153;
    if (!on || type != PI_Q4_K || g.nt > 8 || (int64_t) groups * 256 < (int64_t) min_groups * pi_cu_count() || groups > 8 * 2 * pi_cu_count()) return false;
    for (int i = 0; i < nseg; i++) if (g.seg[i].store) return false;
    return true;
}
static void mmq_g2_dispatch(const mmvq_args & g, int groups, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    const int ncu = pi_cu_count();
    const int rr = (groups + 2 * ncu - 1) / (2 * ncu);          // waves per workgroup that put the launch into one round of one workgroup per available CU (256 without a CU mask)
    switch (rr < 4 ? 4 : rr) {
        case 4: mmq_g2_go<4>(g, groups, stream, ea, eb); break;
        case 5: mmq_g2_go<5>(g, groups, stream, ea, eb); break;
        case 6: mmq_g2_go<6>(g, groups, stream, ea, eb); break;
        case 7: mmq_g2_go<7>(g, groups, stream, ea, eb); break;
        default: mmq_g2_go<8>(g, groups, stream, ea, eb); break;
    }
}
static void mmq_dispatch(int type, const mmvq_args & g, int groups, int cfg, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    if (type == PI_Q6_K) mmq_dispatch_t<PI_Q6_K>(g, groups, cfg, stream, ea, eb);
    else if (type == PI_Q3_K) mmq_dispatch_t<PI_Q3_K>(g, groups, cfg, stream, ea, eb);
    else if (type == PI_Q5_K) mmq_dispatch_t<PI_Q5_K>(g, groups, cfg, stream, ea, eb);
    else mmq_dispatch_t<PI_Q4_K>(g, groups, cfg, stream, ea, eb);
}
// ---- two weight types in one launch (k_mmq_kq2t): Q4_K segments followed by Q5_K or Q6_K segments, 3..8 tokens, S = 4 (the short-matrix configurations) ----
template <int TB_, int R, int RB>
static void mmq_2t_go(const mmvq_args & g, int groups_a, int groups_all, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int TBA = 4, TBB = 4;                                    // blocks per tile at <= 8 tokens, as in mmq_dispatch_t's S = 4 rows
    constexpr size_t tiles = (size_t) 3 * (TBA > TBB ? TBA : TBB) * (8 * 512 + 1024), fold = (size_t) R * (8 * 16 * 17 + 4 * 16 * 17 + 256) * 4;
    constexpr size_t lds = tiles > fold ? tiles : fold;
    static_assert(tiles <= 65536, "LDS-DMA targets stay inside the first 64 KB");
    if (!g.frag) { (void) hipFuncSetAttribute((const void *) k_mmq_kq2t<PI_Q4_K, TB_, 4, R, RB, TBA, TBB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); return; }   // pi_op_mul_mat_reserve
    const int wgs_a = (groups_a + R - 1) / R, wgs_b = (groups_all - groups_a + RB - 1) / RB;
    if (ea) hipExtLaunchKernelGGL((k_mmq_kq2t<PI_Q4_K, TB_, 4, R, RB, TBA, TBB>), dim3((unsigned) (wgs_a + wgs_b)), dim3(64 * 4 * R), lds, stream, ea, eb, 0, g, groups_a, groups_all, wgs_a);
    else    hipLaunchKernelGGL((k_mmq_kq2t<PI_Q4_K, TB_, 4, R, RB, TBA, TBB>), dim3((unsigned) (wgs_a + wgs_b)), dim3(64 * 4 * R), lds, stream, g, groups_a, groups_all, wgs_a);
}
// four chunks.  The kernel's register budget is that of its block size (R = 3: 168 VGPRs): a second part with ONE row group per workgroup stages nine 16-byte units per
// thread instead of three and spills (Q5_K: 97 registers; the fused launch then took 120 us against 48 + 37 apart), so the second part keeps R row groups per workgroup
// (measured, q + k + v of a 70B layer per 64 tokens: Q6_K 54 us fused against 117 apart).  A prompt launch is many rounds long anyway.
template <int TB_, int R>
static void mmq_2t_go4(const mmvq_args & g, int groups_a, int groups_all, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int RB = R;
    constexpr size_t tiles = (size_t) 2 * 4 * (8 * 1024 + 1024), fold = (size_t) R * (8 * 16 * 17 + 4 * 16 * 17 + 256) * 4;
    constexpr size_t lds = tiles > fold ? tiles : fold;
    if (!g.frag) { (void) hipFuncSetAttribute((const void *) k_mmq_kq2t<PI_Q4_K, TB_, 4, R, RB, 1, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); return; }
    const int wgs_a = (groups_a + R - 1) / R, wgs_b = (groups_all - groups_a + RB - 1) / RB;
    if (ea) hipExtLaunchKernelGGL((k_mmq_kq2t<PI_Q4_K, TB_, 4, R, RB, 1, 1, 4>), dim3((unsigned) (wgs_a + wgs_b)), dim3(64 * 4 * R), lds, stream, ea, eb, 0, g, groups_a, groups_all, wgs_a);
    else    hipLaunchKernelGGL((k_mmq_kq2t<PI_Q4_K, TB_, 4, R, RB, 1, 1, 4>), dim3((unsigned) (wgs_a + wgs_b)), dim3(64 * 4 * R), lds, stream, g, groups_a, groups_all, wgs_a);
}
template <int TB_>
static void mmq_2t_dispatch_t(int r, int rb, bool four, const mmvq_args & g, int groups_a, int groups_all, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    if (four) {                                                                             // (launches of a prompt pass: up to 64 tokens, four chunks)
        if (r == 1) mmq_2t_go4<TB_, 1>(g, groups_a, groups_all, stream, ea, eb); else if (r == 2) mmq_2t_go4<TB_, 2>(g, groups_a, groups_all, stream, ea, eb); else mmq_2t_go4<TB_, 3>(g, groups_a, groups_all, stream, ea, eb);
        return;
    }
    if (r == 1) mmq_2t_go<TB_, 1, 1>(g, groups_a, groups_all, stream, ea, eb);
    else if (r == 2) { if (rb == 1) mmq_2t_go<TB_, 2, 1>(g, groups_a, groups_all, stream, ea, eb); else mmq_2t_go<TB_, 2, 2>(g, groups_a, groups_all, stream, ea, eb); }
    else             { if (rb == 1) mmq_2t_go<TB_, 3, 1>(g, groups_a, groups_all, stream, ea, eb); else mmq_2t_go<TB_, 3, 3>(g, groups_a, groups_all, stream, ea, eb); }
}
static void mmq_2t_dispatch(int type_b, int r, int rb, bool four, const mmvq_args & g, int groups_a, int groups_all, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    if (type_b == PI_Q6_K) mmq_2t_dispatch_t<PI_Q6_K>(r, rb, four, g, groups_a, groups_all, stream, ea, eb);
    else                   mmq_2t_dispatch_t<PI_Q5_K>(r, rb, four, g, groups_a, groups_all, stream, ea, eb);
}
// row groups per workgroup (S = 4) the single-type dispatcher would give the Q4_K part; 0: that part is not one of the short-matrix configurations (no two-type launch)
static int mmq_2t_rows_per_wg(int64_t rows_a) {
    const int groups = (int) ((rows_a + 15) / 16 * 256 / pi_cu_count());
    return groups > 768 ? 0 : groups > 512 ? 3 : groups >= 384 ? 2 : 1;
}
// row groups per workgroup of the second type's part: one when the launch then still is one round of workgroups, else the first part's; 0: more than one round either way
static int mmq_2t_rows_per_wg_b(int r, int64_t rows_a, int64_t rows_b) {
    const int64_t wgs_a = ((rows_a + 15) / 16 + r - 1) / r, gb = (rows_b + 15) / 16;
    if (wgs_a + gb <= pi_cu_count()) return 1;
    if (wgs_a + (gb + r - 1) / r <= pi_cu_count()) return r;
    return 0;
}
// may a launch of n_a Q4_K matrices (rows_a rows in all) followed by matrices of type_b (rows_b rows) go out as ONE matrix-core launch?  PI_MMQ_2T=0 switches it off.
// (Only while both parts fit one round of workgroups: the 7B model's attn_v has as many rows as attn_k — fused, q + k + v would be a round and a half.)
// n_tok: the PASS's tokens.  Verify widths (<= 8) and prompt passes (> 32: launches of 64 tokens, four chunks; a tail of <= 32 tokens goes out in launches of 8).
bool pi_mmq2_applies(int type_b, int n_tok, int64_t rows_a, int64_t rows_b) {
    static const bool on = !(getenv("PI_MMQ_2T") && atoi(getenv("PI_MMQ_2T")) == 0);
    if (!on || (type_b != PI_Q5_K && type_b != PI_Q6_K) || (n_tok > 8 && n_tok <= 32)) return false;
    const int r = mmq_2t_rows_per_wg(rows_a);
    if (n_tok > 32) return r > 0 && pi_mmq_max_nt() >= 64;
    return pi_mmq_wants(PI_Q4_K, n_tok, rows_a, 0) && r > 0 && mmq_2t_rows_per_wg_b(r, rows_a, rows_b) > 0;
}
// L: segments [0, n_a) Q4_K, [n_a, nseg) of type_b, all with tile-major copies; g.frag / g.aux / g.nt set.  four: a launch of a prompt pass (g.nt <= 64 tokens, a multiple
// of 64 tokens into the pass: the four-chunk instance, whatever is left of the pass); else g.nt <= 8
int pi_mmq_launch2(int type_b, int n_a, bool four, mmvq_launch & L, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    if (pi_op_mul_mat_reserve() != 0) return -1;
    mmvq_args & g = L.g;
    int groups = 0, groups_a = 0; int64_t rows_a = 0, rows_b = 0;
    for (int i = 0; i < MMVQ_MAX_SEG; i++) {
        if (i < L.nseg) { g.seg[i].wave0 = groups; groups += (g.seg[i].M + 15) / 16; if (i < n_a) { groups_a = groups; rows_a += g.seg[i].M; } else rows_b += g.seg[i].M; }
        else { g.seg[i] = g.seg[0]; g.seg[i].wave0 = 0x7fffffff; }
    }
    // (the second part's row groups are counted per matrix: each of its matrices starts a row group, like the first part's)
    int r = mmq_2t_rows_per_wg(rows_a);
    if (four && type_b == PI_Q5_K && r == 3) r = 2;                                          // (the Q5_K body of the four-chunk form spills 62 registers under the 168 of a 12-wave workgroup; 8 waves: 229, none)
    int rb = four ? r : r > 0 ? mmq_2t_rows_per_wg_b(r, rows_a, rows_b) : 0;
    if (r > 0 && rb <= 0) rb = r;                                                           // (more than one round: still correct, pi_mmq2_applies does not choose it)
    if (r <= 0 || n_a <= 0 || n_a >= L.nseg || g.nt > (four ? 64 : 8)) { pi_set_error("pi_mmq_launch2: not a two-type launch"); return -1; }
    g.dbg = 0;
    mmq_2t_dispatch(type_b, r, rb, four, g, groups_a, groups, stream, ea, eb);
    PI_CHECK_LAUNCH();
    return 0;
}

static const int mmq_cfgs[] = { 17, 18, 14, 22, 24, 41, 42, 43 };
// dynamic-LDS limits of every instance, set once and outside any stream capture (a launch inside a captured graph must not change function attributes)
extern "C" int pi_op_mul_mat_reserve(void) {
    static bool done = false;
    if (done) return 0;
    mmvq_args g; memset(&g, 0, sizeof g);                      // g.frag == nullptr: the instances only set their attribute
    for (int type : { PI_Q3_K, PI_Q4_K, PI_Q5_K, PI_Q6_K }) for (int cfg : mmq_cfgs) for (int nt : { 8, 16, 32, 64 }) { g.nt = nt; mmq_dispatch(type, g, 0, cfg, nullptr, nullptr, nullptr); }
    for (int type_b : { PI_Q5_K, PI_Q6_K }) for (int r : { 1, 2, 3 }) for (int rb : { 1, r }) for (bool four : { false, true }) mmq_2t_dispatch(type_b, r, rb, four, g, 0, 0, nullptr, nullptr, nullptr);
    done = true;
    PI_CHECK_LAUNCH();
    return 0;
}

int pi_mmq_launch(int type, mmvq_launch & L, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    if (pi_op_mul_mat_reserve() != 0) return -1;
    mmvq_args & g = L.g;
    int groups = 0;
    for (int i = 0; i < MMVQ_MAX_SEG; i++) {
        if (i < L.nseg) { g.seg[i].wave0 = groups; groups += (g.seg[i].M + 15) / 16; }
        else { g.seg[i] = g.seg[0]; g.seg[i].wave0 = 0x7fffffff; }
    }
    if (!pi_mmq_supported(type)) { pi_set_error("pi_mmq_launch: weight type %d has no MFMA kernel", type); return -1; }
    static const int dbg_env = getenv("PI_MMQ_DBG") ? atoi(getenv("PI_MMQ_DBG")) : 0;
    g.dbg = dbg_env;
    // (waves per row group, row groups per workgroup).  One wave per row group needs the fewest instructions per weight byte (the block scalars
    // are decoded once), but a launch wants >= ~6 waves per CU: S = 1 for tall matrices, 2 / 4 for shorter ones; R fills workgroups of up to
    // 8 waves that share the staged activations, while keeping >= 256 workgroups.  PI_MMQ_CFG=SR overrides (tuning), e.g. 42.
    static const int cfg_env = getenv("PI_MMQ_CFG") ? atoi(getenv("PI_MMQ_CFG")) : 0;
    int cfg;
    // (the bounds below were measured on the whole chip; a context confined to fewer CUs scales them: what matters is row groups per available CU)
    const int groups_chip = groups;
    groups = (int) ((int64_t) groups * 256 / pi_cu_count());
    if (groups > 1152) cfg = groups <= 7 * 256 ? 17 : 18;            // (measured crossover against S = 2: 1024 row groups 32.5 vs 34.0 us, 1280: 42.3 vs 36.3, 1376 at K = 4096: 25.5 vs 22.3)           // S = 1: the fewest row groups per workgroup (7 or 8) that still fit the launch into one round of 256 workgroups
    else if (groups > 768) cfg = 24;
    else if (groups > 512) cfg = 43;                                    // 513..768 row groups: three per workgroup keep the launch in ONE round of <= 256 workgroups (two per workgroup: 288-384
                                                                        // workgroups, a second round at the kernel's latency floor): 9216 x 8192 23.9 -> 19.9 us, 12288 x 4096 17.1 -> 13.6, 11008 x 4096 15.6 -> 13.8
    else if (groups >= 384) cfg = 42;
    else cfg = 41;
    if (cfg_env) cfg = cfg_env;
    groups = groups_chip;
    if (!cfg_env && mmq_g2_wants(type, g, L.nseg, groups)) { mmq_g2_dispatch(g, groups, stream, ea, eb); PI_CHECK_LAUNCH(); return 0; }
    mmq_dispatch(type, g, groups, cfg, stream, ea, eb);
    PI_CHECK_LAUNCH();
    return 0;
}
