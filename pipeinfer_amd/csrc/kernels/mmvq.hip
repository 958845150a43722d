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

#include <cstring>

#define MMVQ_TILE 16            // K-quant super-blocks (or groups of four 32-blocks) staged in LDS per step

struct mmvq_args {
    pi_weight w;
    const uint8_t * aq;         // activation planes (pi_q8_layout)
    const uint8_t * ad;
    const uint8_t * abs_;
    float *         out;        // [n_tok][M]
    const float *   res;        // nullable
    int M, K, nb;               // nb: super-blocks (K-quants) or groups of 4 blocks (Q4_0/Q8_0) per row
};

static __device__ __forceinline__ int bytes_sub(int v, int c) {            // per-byte v - c for bytes whose result fits int8 (no borrow across bytes)
    return (int) ((((uint32_t) v | 0x80808080u) - (uint32_t) c) ^ 0x80808080u);
}
static __device__ __forceinline__ int ub(uint32_t w, int i) { return (int) ((w >> (8 * i)) & 0xff); }

// folds the 8 lanes of a row exactly like hsum_float_8: (a4+a0, a5+a1, a6+a2, a7+a3) -> (r0+r2, r1+r3) -> s0+s1
static __device__ __forceinline__ float hsum8_lanes(float a) {
    a = a + __shfl_xor(a, 4, 64);
    a = a + __shfl_xor(a, 2, 64);
    a = a + __shfl_xor(a, 1, 64);
    return a;
}

// ================================================================================================================
// K-quants x Q8_K
// ================================================================================================================
template <int TYPE, int NT>
__global__ void __launch_bounds__(256) k_mmvq_kq(const mmvq_args g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // LDS tile: qs [NT][TILE][256] | d [NT][TILE] f32 | bsums [NT][TILE][16] i16
    uint8_t * l_qs = lds;
    float   * l_d  = (float *) (lds + (size_t) NT * MMVQ_TILE * 256);
    int16_t * l_bs = (int16_t *) ((uint8_t *) l_d + (size_t) NT * MMVQ_TILE * 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 3, k = lane & 7;
    const int64_t row = (int64_t) blockIdx.x * (blockDim.x >> 3) + wave * 8 + r;
    const bool live = row < g.M;
    const int64_t rowc = live ? row : g.M - 1;
    const int nb = g.nb;

    float acc[NT], accm[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { acc[t] = 0.0f; accm[t] = 0.0f; }

    for (int b0 = 0; b0 < nb; b0 += MMVQ_TILE) {
        const int nbt = min(MMVQ_TILE, nb - b0);
        // issue every weight load of the tile before the first use: with only M/8 waves in a launch the bytes in flight per
        // wave, not occupancy, are what keeps HBM busy (16 blocks x 8 rows x 144 B = 18 KB per wave)
        uint4 w_hdr[MMVQ_TILE], w_q[MMVQ_TILE]; uint2 w_h[MMVQ_TILE]; uint16_t w_d[MMVQ_TILE];
#pragma unroll
        for (int bi = 0; bi < MMVQ_TILE; bi++) {
            {   // unconditional (indices clamped): arrays filled under divergent control flow would be demoted to scratch
                const int64_t blk = rowc * nb + b0 + (bi < nbt ? bi : nbt - 1);
                if constexpr (TYPE == PI_Q4_K || TYPE == PI_Q5_K) {
                    constexpr int BS = TYPE == PI_Q4_K ? 144 : 176, QOFF = TYPE == PI_Q4_K ? 16 : 48;
                    const uint8_t * p = g.w.p0 + blk * BS;
                    w_hdr[bi] = *(const uint4 *) p;
                    w_q[bi] = *(const uint4 *) (p + QOFF + 16 * k);
                    if constexpr (TYPE == PI_Q5_K) w_h[bi].x = *(const uint32_t *) (p + 16 + 4 * k);
                } else {
                    const uint8_t * p = g.w.p0 + blk * 192;
                    w_q[bi] = *(const uint4 *) (p + 16 * k);
                    w_h[bi] = *(const uint2 *) (p + 128 + 8 * k);
                    w_hdr[bi] = *(const uint4 *) (g.w.p1 + blk * 16);
                    w_d[bi] = ((const uint16_t *) g.w.p2)[blk];
                }
            }
        }
        __syncthreads();
        // ---- stage the activations of blocks [b0, b0+nbt) for all NT tokens (overlaps the weight loads in flight) ----
        for (int u = tid; u < NT * nbt * 16; u += blockDim.x) {                 // 16-byte units of qs
            const int t = u / (nbt * 16), rem = u % (nbt * 16);
            *(uint4 *) (l_qs + ((size_t) t * MMVQ_TILE * 256) + rem * 16) = *(const uint4 *) (g.aq + ((size_t) t * nb + b0) * 256 + (size_t) rem * 16);
        }
        for (int u = tid; u < NT * nbt; u += blockDim.x) {
            const int t = u / nbt, b = u % nbt;
            l_d[t * MMVQ_TILE + b] = ((const float *) g.ad)[(size_t) t * nb + b0 + b];
        }
        for (int u = tid; u < NT * nbt * 2; u += blockDim.x) {                  // 16-byte units of bsums (32 B per block)
            const int t = u / (nbt * 2), rem = u % (nbt * 2);
            *(uint4 *) ((uint8_t *) l_bs + ((size_t) t * MMVQ_TILE * 32) + rem * 16) = *(const uint4 *) (g.abs_ + ((size_t) t * nb + b0) * 32 + (size_t) rem * 16);
        }
        __syncthreads();
        if (!live) continue;

#pragma unroll
        for (int bi = 0; bi < MMVQ_TILE; bi++) {
          if (bi < nbt) {
            float dx, dminx = 0.0f;
            uint32_t scA = 0, scB = 0, mA = 0, mB = 0;
            int q[4], qx[2] = { 0, 0 };
            uint4 sc6 = { 0, 0, 0, 0 };
            q[0] = w_q[bi].x; q[1] = w_q[bi].y; q[2] = w_q[bi].z; q[3] = w_q[bi].w;
            if constexpr (TYPE == PI_Q4_K || TYPE == PI_Q5_K) {
                const uint4 hdr = w_hdr[bi];
                if constexpr (TYPE == PI_Q5_K) qx[0] = (int) w_h[bi].x;
                dx = pi_h2f((uint16_t) (hdr.x & 0xffff)); dminx = pi_h2f((uint16_t) (hdr.x >> 16));
                // 6-bit scales/mins (get_scale_min_k4, ggml-quants.c:1446-1453) unpacked four at a time like the reference's utmp[]
                scA = hdr.y & 0x3f3f3f3fu;
                mA  = hdr.z & 0x3f3f3f3fu;
                scB = (hdr.w & 0x0f0f0f0fu) | (((hdr.y >> 6) & 0x03030303u) << 4);
                mB  = ((hdr.w >> 4) & 0x0f0f0f0fu) | (((hdr.z >> 6) & 0x03030303u) << 4);
            } else {
                qx[0] = (int) w_h[bi].x; qx[1] = (int) w_h[bi].y;
                sc6 = w_hdr[bi];
                dx = pi_h2f(w_d[bi]);
            }
            // token-independent unpacking of this lane's 32 weights (8 dwords, in the order the activations are stored)
            int wq[8];
            if constexpr (TYPE == PI_Q4_K) {
#pragma unroll
                for (int j = 0; j < 4; j++) { wq[2 * j] = q[j] & 0x0F0F0F0F; wq[2 * j + 1] = (q[j] >> 4) & 0x0F0F0F0F; }
            } else if constexpr (TYPE == PI_Q5_K) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t h = (uint32_t) qx[0] >> (2 * j);
                    wq[2 * j]     = (q[j] & 0x0F0F0F0F)        | (int) ((h & 0x01010101u) << 4);
                    wq[2 * j + 1] = ((q[j] >> 4) & 0x0F0F0F0F) | (int) (((h >> 1) & 0x01010101u) << 4);
                }
            } else {
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    const uint32_t h = (uint32_t) qx[n];
                    const int la = q[2 * n], lb = q[2 * n + 1];
                    wq[4 * n + 0] = bytes_sub((la & 0x0F0F0F0F)        | (int) ((h & 0x03030303u) << 4),        0x20202020);
                    wq[4 * n + 1] = bytes_sub((lb & 0x0F0F0F0F)        | (int) (((h >> 2) & 0x03030303u) << 4), 0x20202020);
                    wq[4 * n + 2] = bytes_sub(((la >> 4) & 0x0F0F0F0F) | (int) (((h >> 4) & 0x03030303u) << 4), 0x20202020);
                    wq[4 * n + 3] = bytes_sub(((lb >> 4) & 0x0F0F0F0F) | (int) (((h >> 6) & 0x03030303u) << 4), 0x20202020);
                }
            }
            int sc[8];
            if constexpr (TYPE == PI_Q6_K) {
                const uint32_t s4[4] = { sc6.x, sc6.y, sc6.z, sc6.w };
                const int hi = k >> 2;
#pragma unroll
                for (int i = 0; i < 8; i++) {                                   // i = 4n + g: scale index 8n + 2g + hi
                    const int idx = 8 * (i >> 2) + 2 * (i & 3);
                    sc[i] = (int) (int8_t) ((s4[idx >> 2] >> (8 * ((idx & 3) + hi))) & 0xff);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) { sc[i] = ub(scA, i); sc[4 + i] = ub(scB, i); }
            }
            int m0 = 0, m1 = 0;
            if constexpr (TYPE != PI_Q6_K) {
                const uint32_t mw = (k & 2) ? mB : mA;
                const int sh = (k & 1) * 16;
                m0 = (int) ((mw >> sh) & 0xff); m1 = (int) ((mw >> (sh + 8)) & 0xff);
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const uint4 y0 = *(const uint4 *) (l_qs + ((size_t) t * MMVQ_TILE + bi) * 256 + 32 * k);
                const uint4 y1 = *(const uint4 *) (l_qs + ((size_t) t * MMVQ_TILE + bi) * 256 + 32 * k + 16);
                const int y[8] = { (int) y0.x, (int) y0.y, (int) y0.z, (int) y0.w, (int) y1.x, (int) y1.y, (int) y1.z, (int) y1.w };
                const float dy = l_d[t * MMVQ_TILE + bi];
                int sumi = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) sumi += sc[i] * pi_dot4(wq[i], y[i], 0);
                acc[t] = fmaf(dy * dx, (float) sumi, acc[t]);
                if constexpr (TYPE != PI_Q6_K) {
                    // mins: lanes 0..3 own prod[k] = m[2k]*(bsums[4k]+bsums[4k+1]) + m[2k+1]*(bsums[4k+2]+bsums[4k+3])   (int16 pair sums like _mm_hadd_epi16)
                    const short4 bs = *(const short4 *) (l_bs + ((size_t) t * MMVQ_TILE + bi) * 16 + 4 * (k & 3));
                    int prod = m0 * (int) (short) (bs.x + bs.y) + m1 * (int) (short) (bs.z + bs.w);
                    const float dmin = -dy * dminx;
                    if constexpr (TYPE == PI_Q4_K) {
                        accm[t] = fmaf(dmin, (float) prod, accm[t]);           // used from lanes 0..3 only
                    } else {
                        prod += __shfl_xor(prod, 1, 64); prod += __shfl_xor(prod, 2, 64);      // lanes 0..3 of the row hold the total
                        accm[t] = accm[t] + dmin * (float) prod;               // summs += dmin * hsum (mul, then add)
                    }
                }
            }
          }
        }
    }
    // ---- fold the eight chains ----
#pragma unroll
    for (int t = 0; t < NT; t++) {
        float v = hsum8_lanes(acc[t]);
        if constexpr (TYPE == PI_Q4_K) {
            float mv = (k < 4) ? accm[t] : 0.0f;
            mv = mv + __shfl_xor(mv, 2, 64);                                    // (m0+m2, m1+m3)
            mv = mv + __shfl_xor(mv, 1, 64);                                    // (m0+m2) + (m1+m3)
            v = v + mv;
        } else if constexpr (TYPE == PI_Q5_K) {
            v = v + accm[t];
        }
        if (live && k == 0) {
            const int64_t o = (int64_t) t * g.M + row;
            g.out[o] = g.res ? g.res[o] + v : v;
        }
    }
}

// ================================================================================================================
// Q4_0 / Q8_0 x Q8_0: lane (r, k) walks groups of four 32-blocks
// ================================================================================================================
template <int TYPE, int NT>
__global__ void __launch_bounds__(256) k_mmvq_q0(const mmvq_args g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // LDS tile: qs [NT][TILE groups][128] | d [NT][TILE][4] f16
    uint8_t * l_qs = lds;
    uint16_t * l_d = (uint16_t *) (lds + (size_t) NT * MMVQ_TILE * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 3, k = lane & 7;
    const int64_t row = (int64_t) blockIdx.x * (blockDim.x >> 3) + wave * 8 + r;
    const bool live = row < g.M;
    const int64_t rowc = live ? row : g.M - 1;
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
                if constexpr (TYPE == PI_Q4_0) w_x[gi] = *(const uint4 *) (g.w.p0 + grp * 64 + 16 * (k & 3));
                else                           w_x[gi] = *(const uint4 *) (g.w.p0 + grp * 128 + 16 * k);
                w_dv[gi] = *(const uint2 *) (g.w.p1 + grp * 8);
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
            const int64_t o = (int64_t) t * g.M + row;
            g.out[o] = g.res ? g.res[o] + v : v;
        }
    }
}

// ---- sampled per-launch timing for the roofline report (bench.py): HIP events on the launch stream around every Nth
//      launch of each kernel instantiation.  Eager launches: samples resolved by pi_kernel_timing_report after a sync.
//      Captured launches (HIP graphs): the event pair becomes part of the graph and is re-recorded by every replay; the
//      engine owns the samples of a graph (pi_kt_sink) and folds their elapsed times in after each replay has finished. ----
#include <mutex>
#include <vector>
#include "kernel_timing.h"
namespace {
std::mutex g_kt_mu;
bool g_kt_on = false; int g_kt_every = 8;
std::vector<pi_kt_sample> g_kt;                      // eager samples
uint64_t g_kt_launches[16][9];
double g_kt_us[16][9], g_kt_bytes[16][9]; uint64_t g_kt_n[16][9];
std::vector<hipEvent_t> g_kt_pool;
thread_local pi_kt_sink * g_kt_sink = nullptr;
hipEvent_t kt_event() {
    if (!g_kt_pool.empty()) { hipEvent_t e = g_kt_pool.back(); g_kt_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void) hipEventCreate(&e); return e;
}
void kt_fold(const pi_kt_sample & s) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) return;
    const int t = s.type & 15, n = s.nt <= 8 ? s.nt : 8;
    g_kt_us[t][n] += 1e3 * ms; g_kt_bytes[t][n] += s.bytes; g_kt_n[t][n]++;
}
}
void pi_kt_set_sink(pi_kt_sink * sink) { g_kt_sink = sink; }
bool pi_kt_enabled() { return g_kt_on; }
void pi_kt_accumulate(pi_kt_sink * sink, uint64_t replays_launches) {
    std::lock_guard<std::mutex> lk(g_kt_mu);
    if (!g_kt_on) return;
    for (auto & s : sink->samples) kt_fold(s);
    for (auto & c : sink->launches) g_kt_launches[c.type & 15][c.nt <= 8 ? c.nt : 8] += c.count * replays_launches;
}
extern "C" void pi_kernel_timing(int on, int sample_every) {
    std::lock_guard<std::mutex> lk(g_kt_mu);
    g_kt_on = on != 0; g_kt_every = sample_every > 0 ? sample_every : 1;
    for (auto & s : g_kt) { g_kt_pool.push_back(s.a); g_kt_pool.push_back(s.b); }
    g_kt.clear();
    memset(g_kt_launches, 0, sizeof g_kt_launches); memset(g_kt_us, 0, sizeof g_kt_us); memset(g_kt_bytes, 0, sizeof g_kt_bytes); memset(g_kt_n, 0, sizeof g_kt_n);
}
// mean duration (us) and mean algorithmic bytes of the sampled launches of k_mmvq<type, nt>; call after a device sync
extern "C" int pi_kernel_timing_report(int type, int nt, double * avg_us, double * avg_bytes, uint64_t * n_sampled, uint64_t * n_launches) {
    std::lock_guard<std::mutex> lk(g_kt_mu);
    for (auto & s : g_kt) { kt_fold(s); g_kt_pool.push_back(s.a); g_kt_pool.push_back(s.b); }
    g_kt.clear();
    const int t = type & 15, n = nt <= 8 ? nt : 8;
    *avg_us = g_kt_n[t][n] ? g_kt_us[t][n] / g_kt_n[t][n] : 0.0; *avg_bytes = g_kt_n[t][n] ? g_kt_bytes[t][n] / g_kt_n[t][n] : 0.0;
    *n_sampled = g_kt_n[t][n]; *n_launches = g_kt_launches[t][n];
    return 0;
}

template <int TYPE, int NT>
static int launch_mmvq(const mmvq_args & g, hipStream_t stream) {
    bool sample = false; hipEvent_t ea = nullptr, eb = nullptr;
    pi_kt_sink * sink = g_kt_sink;
    if (sink) {                                            // being captured into a graph: count per graph, sample a fixed subset
        bool found = false;
        for (auto & c : sink->launches) if (c.type == TYPE && c.nt == NT) { sample = (c.count++ % sink->every) == 0; found = true; break; }
        if (!found) { sink->launches.push_back(pi_kt_count{ TYPE, NT, 1 }); sample = true; }
        if (sample) { std::lock_guard<std::mutex> lk(g_kt_mu); ea = kt_event(); eb = kt_event(); }
    } else if (g_kt_on) {
        std::lock_guard<std::mutex> lk(g_kt_mu);
        sample = (g_kt_launches[TYPE & 15][NT]++ % g_kt_every) == 0;
        if (sample) { ea = kt_event(); eb = kt_event(); }
    }
    if (sample) (void) hipEventRecord(ea, stream);
    // waves per workgroup: keep >= ~512 workgroups for the 256 CUs when the matrix is tall enough
    int wpb = 1;
    while (wpb < 4 && (int64_t) g.M / (8 * wpb * 2) >= 512) wpb *= 2;
    const unsigned grid = (unsigned) ((g.M + 8 * wpb - 1) / (8 * wpb));
    const int tile = g.nb < MMVQ_TILE ? g.nb : MMVQ_TILE;
    (void) tile;
    if constexpr (TYPE == PI_Q4_0 || TYPE == PI_Q8_0) {
        const size_t lds = (size_t) NT * MMVQ_TILE * (128 + 8);
        hipLaunchKernelGGL((k_mmvq_q0<TYPE, NT>), dim3(grid), dim3(64 * wpb), lds, stream, g);
    } else {
        const size_t lds = (size_t) NT * MMVQ_TILE * (256 + 4 + 32);
        hipLaunchKernelGGL((k_mmvq_kq<TYPE, NT>), dim3(grid), dim3(64 * wpb), lds, stream, g);
    }
    if (sample) {
        (void) hipEventRecord(eb, stream);
        // algorithmic bytes of this launch: the weight matrix once + quantised activations + fp32 output (+ residual)
        const double wbytes = (double) g.w.nbytes;
        const double abytes = (double) NT * ((TYPE == PI_Q4_0 || TYPE == PI_Q8_0) ? (double) g.K / 32 * 34 : (double) g.K / 256 * 292);
        const double obytes = (double) NT * g.M * 4 * (g.res ? 2 : 1);
        if (sink) sink->samples.push_back(pi_kt_sample{ ea, eb, TYPE, NT, wbytes + abytes + obytes });
        else { std::lock_guard<std::mutex> lk(g_kt_mu); g_kt.push_back(pi_kt_sample{ ea, eb, TYPE, NT, wbytes + abytes + obytes }); }
    }
    PI_CHECK_LAUNCH();
    return 0;
}

template <int TYPE>
static int dispatch_nt(const mmvq_args & g, int nt, hipStream_t stream) {
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
    const int64_t nbq = vdt == PI_Q8_K ? K / 256 : w->ng;
    for (int t0 = 0; t0 < n_tok; t0 += 8) {                   // verify batches are <= 8 wide; prompts loop over chunks of 8
        const int nt = n_tok - t0 < 8 ? n_tok - t0 : 8;
        mmvq_args g;
        g.w = *w;
        g.M = (int) M; g.K = (int) K; g.nb = (int) nbq;
        if (vdt == PI_Q8_K) {
            g.aq = base + oq + (size_t) t0 * nbq * 256; g.ad = base + od + (size_t) t0 * nbq * 4; g.abs_ = base + ob + (size_t) t0 * nbq * 32;
        } else {
            g.aq = base + oq + (size_t) t0 * nbq * 128; g.ad = base + od + (size_t) t0 * nbq * 8; g.abs_ = nullptr;
        }
        g.out = out + (int64_t) t0 * M;
        g.res = res ? res + (int64_t) t0 * M : nullptr;
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
