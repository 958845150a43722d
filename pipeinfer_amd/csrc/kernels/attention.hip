// KV-cache attention for the decode / verify path, BIT-EXACT with the reference's x86 AVX+F16C arm.
//
// Reference: llm_build_kqv (llama.cpp:3698-3764) = mul_mat(K, Q) -> ggml_scale -> ggml_add(mask) -> ggml_soft_max ->
// mul_mat(V, P) on the CPU:
//   * both mat-muls run ggml_vec_dot_f16 (ggml.c:1193-1228, macros :751-850): 32 fp32 accumulators (4 registers x 8 lanes),
//     element e of every 32-element step goes to accumulator e % 32 by fma, the registers are folded (0+2, 1+3, then 0+1),
//     the 8 lanes are folded as (l, l+4) -> hadd -> hadd, leftover elements (n % 32) are added in double;
//     Q and the soft-max output P are rounded to f16 first (ggml.c:9493-9506), K/V are f16 in the cache;
//   * soft_max (ggml.c:10551-10620): row max, exp through the fp16 table indexed by f16(x - max), sum in double, scale by
//     (float)(1/sum); masked cells are exactly 0.  Every table value is a multiple of 2^-24 and <= 1, so the reference's
//     double sum is exact and an integer accumulation reproduces it bit for bit in any order.
// f16 rounding of P and the Q8 quantisation of the attention output turn a 1-ulp score difference into 1e-2 logit
// differences, so these accumulation orders are part of the contract.
//
// One workgroup (256 threads) per (kv head, chunk of RW query rows); a row is one (token, query head of the GQA group), so
// K/V are read once per chunk for all query heads and verify tokens that share them.  RW is chosen so that a launch has
// enough workgroups (decode has few rows: one workgroup per (kv head, row)).
//   phase 1  16 lanes per cell: lane a owns accumulators 2a and 2a+1 of the reference's 32 (one dword per 32-element step; its q values
//            stay in registers), folds them in the reference's order (DPP row shifts 8, 4, 2, in-lane add, shift 1), scale, mask -> S
//   phase 2  all threads: row max, table exp, exact integer sum, P = f16-rounded probability (kept as f32 in LDS)
//   phase 3  lane (l, ev) owns cells c % 8 == l and dims 8ev..8ev+7: four accumulator sets j = (c / 8) % 4 per row, folded
//            in the reference's order with lane shuffles; tail cells in double by lane l == 0
#include "ops_common.h"
#include "kernel_timing.h"

#include <cstdlib>

#define ATT_ROWS 8
#ifndef PI_ATT_NO_JSPLIT
#define PI_ATT_NO_JSPLIT 0
#endif

struct att_args {
    const uint16_t * q16; const uint16_t * kc; const uint16_t * vc; const uint32_t * mask;
    float * out;
    int n_head, n_head_kv, n_tok, n_kv, mask_words;        // n_kv / mask_words: LDS row stride and mask row stride (static)
    const uint16_t * exp_tab;
    float scale;
    const int32_t * dyn;                                    // optional {kv_head, n_kv} in device memory: lets one captured graph serve every call
    const int32_t * abort_w;                                // abort word of the run (pi_device.h), nullable
    const float * alibi_m;                                  // ALiBi slope of every query head (ggml_alibi, ggml.c:10736-10793), nullable
    int per_token;                                          // the pass may carry several runs: dyn = {.., n_kv of the widest, n, has_table, cell[n], n_kv[n]} and a workgroup's rows belong to ONE token
};

// x[i] + x[i + n] inside a row of 16 lanes (lanes whose partner is outside read 0 and are not used)
template <int CTRL> static __device__ __forceinline__ float dpp_add(float x) {
#ifdef PI_ATT_NO_DPP
    const int lane = threadIdx.x & 63;
    const int src = CTRL == 0x108 ? lane + 8 : CTRL == 0x104 ? lane + 4 : CTRL == 0x102 ? lane + 2 : CTRL == 0x101 ? lane + 1 : CTRL == 0xB1 ? (lane ^ 1) : (lane ^ 2);
    return x + __shfl(x, src & 63, 64);
#else
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
#endif
}

template <int HD, int RW, bool ALIBI = false>
__global__ void __launch_bounds__(256) k_attention(const att_args ga) {
    extern __shared__ float smem[];
    __shared__ float redf[RW][4];
    __shared__ unsigned long long redu[RW][4];
    att_args g = ga;
    const int s_stride = ga.n_kv;                           // LDS rows keep the static stride
    const int hk = blockIdx.x, chunk = blockIdx.y;
    const int r = g.n_head / g.n_head_kv;                   // query heads per kv head (broadcast rule i02 = i12 / r2, ggml.c:9425)
    // kv_self.n of the run this workgroup's token belongs to (llama.cpp:5571): the reference's vec_dot_f16 sums the last n % 32 cells apart, in double, so every token
    // must see exactly the n its own run was started with — a pass that carries several runs brings one n per token (the launcher keeps a workgroup inside one token)
    if (ga.dyn) g.n_kv = (ga.per_token && ga.dyn[3]) ? ga.dyn[4 + ga.dyn[2] + (chunk * RW) / r] : ga.dyn[1];
    const int n_rows_total = g.n_tok * r;
    const int row0 = chunk * RW;
    const int n_rows = min(RW, n_rows_total - row0);
    const int n_gqa = g.n_head_kv * HD, n_embd = g.n_head * HD;
    float * S = smem;                                       // [RW][n_kv]
    uint32_t * M = (uint32_t *) (smem + (size_t) RW * s_stride);   // [RW][mask_words]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // phase-3 lane mapping, and its first V rows requested now: they depend on nothing, so their HBM latency hides behind
    // phases 1 and 2 instead of starting a third dependent round trip
    constexpr int TPC = HD / 8;                              // lanes along the head dimension
    constexpr int NRG = 256 / (8 * TPC);                     // row groups in the workgroup: 4 (HD 64) or 2 (HD 128)
    constexpr int RPT = (RW + NRG - 1) / NRG;                // rows per thread
    const int l = tid & 7, ev = (tid >> 3) % TPC, rg = tid / (8 * TPC);
    const int np = g.n_kv & ~31;
    const uint16_t * vbase = g.vc + hk * HD + 8 * ev;
    constexpr int PD = RW <= 2 ? 4 : 2;                      // 32-cell steps of V in flight
    // with a single row per workgroup (decode) the row-group slots would idle in phase 3, and the kernel is bound per CU: they take
    // accumulator sets instead — slot rg owns j = rg*JPT .. rg*JPT + JPT - 1 of the four; the sets meet in LDS for the reference's fold
    constexpr bool SPLIT = RW == 1 && NRG > 1 && !PI_ATT_NO_JSPLIT;
    constexpr int JS = SPLIT ? NRG : 1, JPT = 4 / JS;
    const int j0 = SPLIT ? rg * JPT : 0;
    uint4 vb[PD][JPT];
#pragma unroll
    for (int d = 0; d < PD; d++)
#pragma unroll
        for (int jj = 0; jj < JPT; jj++) { const int c = 32 * d + 8 * (j0 + jj) + l; vb[d][jj] = *(const uint4 *) (vbase + (int64_t) (c < np ? c : 0) * n_gqa); }

    // the n_kv % 32 leftover cells of phase 3 are summed sequentially in double (ggml_vec_dot_f16's tail loop): one dim per lane,
    // all of them requested now so that the tail costs arithmetic only
    constexpr int NTAIL = 31;
    const int n_tail = g.n_kv - np;
    uint16_t vt[NTAIL];
    if (n_tail > 0 && (SPLIT ? rg == 0 : rg < n_rows)) {
        const uint16_t * vtb = g.vc + hk * HD + 8 * ev + l;
#pragma unroll
        for (int i = 0; i < NTAIL; i++) vt[i] = vtb[(int64_t) min(np + i, g.n_kv - 1) * n_gqa];
    }

    int aborted = 0;
    // ---- phase 1: scores, ggml_vec_dot_f16 order ----
    {
        constexpr int NS = HD / 32;                         // 32-element steps
        const int a = lane & 15, cg = lane >> 4;
        // lane a owns accumulators 2a and 2a+1 of the reference's 32 (one dword = two neighbouring f16 of every 32-element step:
        // half the load instructions of the (a, a+16) mapping, and phase 1 is bound by their issue); q stays in registers
        float q[RW][2 * NS];                                // rows are token-major: row = t * r + gq -> query head hk * r + gq
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            const int row = row0 + (rr < n_rows ? rr : 0), t = row / r, gq = row % r;
            const uint32_t * qp = (const uint32_t *) (g.q16 + (int64_t) t * n_embd + (hk * r + gq) * HD);
#pragma unroll
            for (int st = 0; st < NS; st++) { const uint32_t w = qp[16 * st + a]; q[rr][2 * st] = pi_h2f((uint16_t) (w & 0xffff)); q[rr][2 * st + 1] = pi_h2f((uint16_t) (w >> 16)); }
        }
        float am[RW];
#pragma unroll
        for (int rr = 0; rr < RW; rr++) am[rr] = ALIBI ? ga.alibi_m[hk * r + (row0 + (rr < n_rows ? rr : 0)) % r] : 0.0f;
        constexpr int U = 4;                                // cells per 16-lane group and step; two steps are in flight
        uint32_t kr[2][U][NS];
        auto load_k = [&](int c0, uint32_t (&dst)[U][NS]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int c = c0 + 16 * u + wv * 4 + cg, cc = c < g.n_kv ? c : g.n_kv - 1;
                const uint32_t * kp = (const uint32_t *) (g.kc + (int64_t) cc * n_gqa + hk * HD);
#pragma unroll
                for (int st = 0; st < NS; st++) dst[u][st] = kp[16 * st + a];
            }
        };
        auto score = [&](int c0, const uint32_t (&src)[U][NS]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int c = c0 + 16 * u + wv * 4 + cg;
                float kf[2 * NS];
#pragma unroll
                for (int st = 0; st < NS; st++) { kf[2 * st] = pi_h2f((uint16_t) (src[u][st] & 0xffff)); kf[2 * st + 1] = pi_h2f((uint16_t) (src[u][st] >> 16)); }
#pragma unroll
                for (int rr = 0; rr < RW; rr++) {            // rows past n_rows are computed on row 0's q and not stored: no branch in the loop
                    float x0 = 0.0f, x1 = 0.0f;             // s[2a], s[2a+1]
#pragma unroll
                    for (int st = 0; st < NS; st++) { x0 = fmaf(kf[2 * st], q[rr][2 * st], x0); x1 = fmaf(kf[2 * st + 1], q[rr][2 * st + 1], x1); }
                    x0 = dpp_add<0x108>(x0); x1 = dpp_add<0x108>(x1);      // s[l] += s[16+l]: lanes a, a+8        row_shl:8
                    x0 = dpp_add<0x104>(x0); x1 = dpp_add<0x104>(x1);      // s[l] += s[8+l]:  lanes a, a+4        row_shl:4
                    x0 = dpp_add<0x102>(x0); x1 = dpp_add<0x102>(x1);      // t[l] = s[l] + s[l+4]: lanes a, a+2   row_shl:2  (lane 0: t0, t1; lane 1: t2, t3)
                    float x = x0 + x1;                                     // t0+t1 (lane 0), t2+t3 (lane 1)
                    x = dpp_add<0x101>(x);                                 // (t0+t1) + (t2+t3)                    row_shl:1
                    if (a == 0 && c < g.n_kv && rr < n_rows) {
                        const bool vis = (M[rr * g.mask_words + (c >> 5)] >> (c & 31)) & 1;
                        float sc = x * g.scale;                                  // ggml_scale
                        if constexpr (ALIBI) sc = (float) c * am[rr] + sc;      // ggml_alibi: pdst = i * m_k + src, i = the cell's index in the view (ggml.c:10789), then + mask
                        S[(size_t) rr * s_stride + c] = vis ? sc : -INFINITY;
                    }
                }
            }
        };
        load_k(0, kr[0]);
        // the run's abort word (pi_device.h) is requested behind the first K loads and the V prefetch — in front of them its round trip would gate them (loads retire in
        // order) — and looked at before anything is stored: the kernel writes only its output rows, at the very end
        aborted = pi_abort_peek(ga.abort_w);
        // visibility words of this workgroup's rows go to LDS once (a global load inside the scoring loop would put its latency
        // on every step); the K loads above are already in flight
        for (int i = tid; i < n_rows * g.mask_words; i += 256) {
            const int rr = i / g.mask_words, w = i - rr * g.mask_words;
            M[rr * g.mask_words + w] = g.mask[(int64_t) ((row0 + rr) / r) * g.mask_words + w];
        }
        __syncthreads();
        int c0 = 0;
        for (; c0 + 32 * U <= g.n_kv; c0 += 32 * U) {        // straight-line body (branches would make the wait counts drain every load);
            load_k(c0 + 16 * U, kr[1]);                      // a load past the end re-reads the last cell
            score(c0, kr[0]);
            load_k(c0 + 32 * U, kr[0]);
            score(c0 + 16 * U, kr[1]);
        }
        if (c0 < g.n_kv) {                                   // at most two more steps
            const bool two = c0 + 16 * U < g.n_kv;
            if (two) load_k(c0 + 16 * U, kr[1]);
            score(c0, kr[0]);
            if (two) score(c0 + 16 * U, kr[1]);
        }
    }
    __syncthreads();
#ifdef ATT_STOP_AFTER_P1                                    // tools_dev/README.md: phase timing builds
    return;
#endif
    // ---- phase 2: soft max ----
    for (int rr = 0; rr < n_rows; rr++) {
        float mx = -INFINITY;
        for (int c = tid; c < g.n_kv; c += 256) mx = fmaxf(mx, S[(size_t) rr * s_stride + c]);
        mx = pi_wave_max(mx);
        if (lane == 0) redf[rr][wv] = mx;
    }
    __syncthreads();
    for (int rr = 0; rr < n_rows; rr++) {
        float * s = S + (size_t) rr * s_stride;
        const float mx = fmaxf(fmaxf(redf[rr][0], redf[rr][1]), fmaxf(redf[rr][2], redf[rr][3]));
        unsigned long long isum = 0;                          // in units of 2^-24: exact
        for (int c = tid; c < g.n_kv; c += 256) {
            const float x = s[c];
            float v = 0.0f;
            if (x != -INFINITY) { v = pi_h2f(g.exp_tab[pi_f2h(x - mx)]); isum += (unsigned long long) (v * 16777216.0f); }
            s[c] = v;
        }
        pi_butterfly<32>([&](auto o) { isum += pi_partner<decltype(o)::value>(isum); });
        if (lane == 0) redu[rr][wv] = isum;
    }
    __syncthreads();
    for (int rr = 0; rr < n_rows; rr++) {
        float * s = S + (size_t) rr * s_stride;
        const unsigned long long isum = (redu[rr][0] + redu[rr][1]) + (redu[rr][2] + redu[rr][3]);
        const double sum = (double) isum * (1.0 / 16777216.0);
        const float inv = (float) (1.0 / sum);
        for (int c = tid; c < g.n_kv; c += 256) s[c] = pi_h2f(pi_f2h(s[c] * inv));          // P rounded to f16 as the V.P mul_mat does
    }
    __syncthreads();

#ifdef ATT_STOP_AFTER_P2
    return;
#endif
    // ---- phase 3: out[row][e] = ggml_vec_dot_f16(n_kv, V[:, e], P[row, :]) ----
    float acc[RPT][JPT][8];
#pragma unroll
    for (int i = 0; i < RPT; i++)
#pragma unroll
        for (int jj = 0; jj < JPT; jj++)
#pragma unroll
            for (int e = 0; e < 8; e++) acc[i][jj][e] = 0.0f;
    // ring of PD steps with a straight-line loop body: with branches inside, the compiler's wait counts drain every load before
    // each step; without, PD - 1 steps of loads stay in flight. Loads past np re-read cell 0 and are not used
    int srow[RPT];
#pragma unroll
    for (int i = 0; i < RPT; i++) { const int rr = SPLIT ? 0 : rg + i * NRG; srow[i] = (rr < n_rows ? rr : 0) * s_stride; }
    auto pv_step = [&](int cb, const uint4 (&vv)[JPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < JPT; jj++) {
            const int c = cb + 8 * (j0 + jj) + l;
            const uint32_t w4[4] = { vv[jj].x, vv[jj].y, vv[jj].z, vv[jj].w };
            float vf[8];
#pragma unroll
            for (int u = 0; u < 4; u++) { vf[2 * u] = pi_h2f((uint16_t) (w4[u] & 0xffff)); vf[2 * u + 1] = pi_h2f((uint16_t) (w4[u] >> 16)); }
#pragma unroll
            for (int i = 0; i < RPT; i++) {
                const float p = S[srow[i] + c];
#pragma unroll
                for (int e = 0; e < 8; e++) acc[i][jj][e] = fmaf(vf[e], p, acc[i][jj][e]);
            }
        }
    };
    int c0 = 0;
    for (; c0 + 32 * PD <= np; c0 += 32 * PD) {
#pragma unroll
        for (int d = 0; d < PD; d++) {
            pv_step(c0 + 32 * d, vb[d]);
#pragma unroll
            for (int jj = 0; jj < JPT; jj++) { const int cn = c0 + 32 * (d + PD) + 8 * (j0 + jj) + l; vb[d][jj] = *(const uint4 *) (vbase + (int64_t) (cn < np ? cn : 0) * n_gqa); }
        }
    }
#pragma unroll
    for (int d = 0; d < PD - 1; d++)                          // the last, partial group: its rows are already in the ring
        if (c0 + 32 * d < np) pv_step(c0 + 32 * d, vb[d]);
    // decode: the accumulator sets of slots 1.. go to slot 0 through LDS ([slot - 1][set * 8 + dim][lane of the slot])
    constexpr int LPR = 8 * TPC;                              // lanes of a row-group slot
    __shared__ float xch[SPLIT ? (JS - 1) * JPT * 8 * LPR : 1];
    if constexpr (SPLIT) {
        const int ls = tid % LPR;
        if (rg > 0) {
#pragma unroll
            for (int jj = 0; jj < JPT; jj++)
#pragma unroll
                for (int e = 0; e < 8; e++) xch[((rg - 1) * JPT * 8 + jj * 8 + e) * LPR + ls] = acc[0][jj][e];
        }
        __syncthreads();
        if (rg > 0) return;
    }
    if (aborted & 1) pi_abort_note(ga.abort_w, 5);                                      // (counted once per launch, where the value is first looked at: in front of the stores)
#pragma unroll
    for (int i = 0; i < RPT; i++) {
        const int rr = SPLIT ? 0 : rg + i * NRG;
        float res[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float a4[4];                                                                // the four accumulator sets of this lane and dim
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if constexpr (SPLIT) a4[j] = j < JPT ? acc[0][j % JPT][e] : xch[((j / JPT - 1) * JPT * 8 + (j % JPT) * 8 + e) * LPR + tid % LPR];
                else a4[j] = acc[i][j % JPT][e];
            }
            float s = (a4[0] + a4[2]) + (a4[1] + a4[3]);                                  // registers: 0+=2, 1+=3, 0+=1  (per lane l)
            s = s + pi_xor4(s);                                                         // t[l] = x[l] + x[l+4]
            s = s + pi_partner<1>(s);                                                   // hadd
            s = s + pi_partner<2>(s);                                                   // hadd
            res[e] = s;
        }
        float mine = res[0];                                                            // every lane of the 8 holds all 8 sums: lane l keeps dim l
#pragma unroll
        for (int e = 1; e < 8; e++) mine = l == e ? res[e] : mine;
        if (rr < n_rows && !(aborted & 1)) {
            const int row = row0 + rr, t = row / r, gq = row % r;
            if (n_tail > 0) {
                double sd = (double) mine;
                const float * ps = S + (size_t) rr * s_stride + np;
#pragma unroll
                for (int i = 0; i < NTAIL; i++) {                                       // straight-line: a select per cell, no register indexing
                    const double pr = (double) (pi_h2f(vt[i]) * ps[min(i, n_tail - 1)]);
                    sd = i < n_tail ? sd + pr : sd;
                }
                mine = (float) sd;
            }
            g.out[(int64_t) t * n_embd + (hk * r + gq) * HD + 8 * ev + l] = mine;
        }
    }
}

// =================================================================================================================
// k_attention_any — every OTHER head width (even, <= 256; e.g. StableLM-3B's 80): the reference takes any n_embd_head (llama.cpp:3698-3778).  The same arithmetic written
// plainly, not tuned: one workgroup per (kv head, query row); phase 1 one thread per cell, phase 3 one thread per output dimension, each running ggml_vec_dot_f16's 32
// accumulators itself (4 registers x 8 lanes, fma in element order, folds (0 += 2, 1 += 3, 0 += 1), lanes l + (l + 4), two hadds) and — what the 64 / 128 instances never
// need in phase 1 — the n % 32 leftover ELEMENTS of a head added in double (ggml.c:1221-1224); phase 2 is the tuned kernel's.
// =================================================================================================================
static __device__ __forceinline__ double att_fold32(const float (&sum)[4][8]) {        // the reduced sum as the double the leftovers are added to, one by one
    float s0[8];
#pragma unroll
    for (int l = 0; l < 8; l++) s0[l] = (sum[0][l] + sum[2][l]) + (sum[1][l] + sum[3][l]);
    const float t0 = s0[0] + s0[4], t1 = s0[1] + s0[5], t2 = s0[2] + s0[6], t3 = s0[3] + s0[7];
    return (double) ((t0 + t1) + (t2 + t3));
}
template <bool ALIBI>
__global__ void __launch_bounds__(256) k_attention_any(const att_args ga, const int hd) {
    extern __shared__ float smem[];
    __shared__ float redf[4];
    __shared__ unsigned long long redu[4];
    att_args g = ga;
    const int s_stride = ga.n_kv;
    const int hk = blockIdx.x, row = blockIdx.y;
    const int r = g.n_head / g.n_head_kv, t = row / r, gq = row % r;
    if (ga.dyn) g.n_kv = (ga.per_token && ga.dyn[3]) ? ga.dyn[4 + ga.dyn[2] + t] : ga.dyn[1];
    const int n_gqa = g.n_head_kv * hd, n_embd = g.n_head * hd;
    float * S = smem;                                                                // [n_kv] scores, then probabilities
    uint32_t * M = (uint32_t *) (smem + s_stride);                                   // [mask_words]
    float * Q = smem + s_stride + g.mask_words;                                      // [hd] the query row, f16-rounded
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int aborted = pi_abort_peek(ga.abort_w);
    for (int i = tid; i < g.mask_words; i += 256) M[i] = g.mask[(int64_t) t * g.mask_words + i];
    for (int i = tid; i < hd; i += 256) Q[i] = pi_h2f(g.q16[(int64_t) t * n_embd + (hk * r + gq) * hd + i]);
    __syncthreads();
    const float am = ALIBI ? ga.alibi_m[hk * r + gq] : 0.0f;
    const int nph = hd & ~31;
    // ---- phase 1: S[c] = scale * ggml_vec_dot_f16(hd, K[c], q) (+ ALiBi), masked cells -inf ----
    for (int c = tid; c < g.n_kv; c += 256) {
        const uint16_t * kp = g.kc + (int64_t) c * n_gqa + hk * hd;
        float sum[4][8];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int l = 0; l < 8; l++) sum[j][l] = 0.0f;
        for (int i = 0; i < nph; i += 32) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t * kd = (const uint32_t *) (kp + i + 8 * j);           // (an even head width keeps dwords aligned; 16-byte vectors would need hd % 8 == 0)
                const uint32_t w4[4] = { kd[0], kd[1], kd[2], kd[3] };
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    sum[j][2 * u] = fmaf(pi_h2f((uint16_t) (w4[u] & 0xffff)), Q[i + 8 * j + 2 * u], sum[j][2 * u]);
                    sum[j][2 * u + 1] = fmaf(pi_h2f((uint16_t) (w4[u] >> 16)), Q[i + 8 * j + 2 * u + 1], sum[j][2 * u + 1]);
                }
            }
        }
        double sd = att_fold32(sum);
        for (int i = nph; i < hd; i++) sd += (double) (pi_h2f(kp[i]) * Q[i]);
        const float x = (float) sd;
        const bool vis = (M[c >> 5] >> (c & 31)) & 1;
        float sc = x * g.scale;
        if constexpr (ALIBI) sc = (float) c * am + sc;
        S[c] = vis ? sc : -INFINITY;
    }
    __syncthreads();
    // ---- phase 2: soft max (row max, fp16 exp table, exact integer sum, P rounded to f16) ----
    {
        float mx = -INFINITY;
        for (int c = tid; c < g.n_kv; c += 256) mx = fmaxf(mx, S[c]);
        mx = pi_wave_max(mx);
        if (lane == 0) redf[wv] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
        unsigned long long isum = 0;
        for (int c = tid; c < g.n_kv; c += 256) {
            const float x = S[c];
            float v = 0.0f;
            if (x != -INFINITY) { v = pi_h2f(g.exp_tab[pi_f2h(x - mx)]); isum += (unsigned long long) (v * 16777216.0f); }
            S[c] = v;
        }
        pi_butterfly<32>([&](auto o) { isum += pi_partner<decltype(o)::value>(isum); });
        if (lane == 0) redu[wv] = isum;
        __syncthreads();
        const double sum = (double) ((redu[0] + redu[1]) + (redu[2] + redu[3])) * (1.0 / 16777216.0);
        const float inv = (float) (1.0 / sum);
        for (int c = tid; c < g.n_kv; c += 256) S[c] = pi_h2f(pi_f2h(S[c] * inv));
        __syncthreads();
    }
    // ---- phase 3: out[e] = ggml_vec_dot_f16(n_kv, V[:, e], P) ----
    if (aborted & 1) { pi_abort_note(ga.abort_w, 5); return; }
    for (int e = tid; e < hd; e += 256) {
        const uint16_t * vp = g.vc + hk * hd + e;
        const int np = g.n_kv & ~31;
        float sum[4][8];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int l = 0; l < 8; l++) sum[j][l] = 0.0f;
        for (int c0 = 0; c0 < np; c0 += 32) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int l = 0; l < 8; l++) { const int c = c0 + 8 * j + l; sum[j][l] = fmaf(pi_h2f(vp[(int64_t) c * n_gqa]), S[c], sum[j][l]); }
        }
        double sd = att_fold32(sum);
        for (int c = np; c < g.n_kv; c++) sd += (double) (pi_h2f(vp[(int64_t) c * n_gqa]) * S[c]);
        g.out[(int64_t) t * n_embd + (hk * r + gq) * hd + e] = (float) sd;
    }
}

// raises the dynamic-LDS limit of the kernel for contexts longer than ~2000 cells; call once outside any graph capture
template <int HD, int RW> static int att_set_lds(size_t lds) {
    PI_CHECK_HIP(hipFuncSetAttribute((const void *) k_attention<HD, RW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    PI_CHECK_HIP(hipFuncSetAttribute((const void *) k_attention<HD, RW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    return 0;
}
// LDS a workgroup of rw rows needs for contexts of n_kv cells (score rows + visibility words); the single-row instance also has 8 KiB of static LDS
static size_t att_lds_bytes(int rw, int n_kv) { return (size_t) rw * ((size_t) n_kv + ((size_t) n_kv + 31) / 32) * sizeof(float); }
static size_t att_lds_cap(int rw) { return (size_t) (rw == 1 ? 146 : 156) * 1024; }
// Long contexts: the score rows of a workgroup live in LDS, so the rows per workgroup shrink as the context grows — 8 rows up to 4 840 cells, 4 up to 9 680, 2 up to
// 19 360, one row up to 36 200 (the reference has no bound, llama.cpp:8729-8748; K/V are then re-read from L2 by more workgroups, nothing else changes).
extern "C" int pi_op_attention_max_cells(void) { int n = 36200; while (att_lds_bytes(1, n) > att_lds_cap(1)) n--; return n; }
extern "C" int pi_op_attention_reserve(int hd, int n_kv_max) {
    if (att_lds_bytes(1, n_kv_max) > att_lds_cap(1)) { pi_set_error("pi_op_attention_reserve: n_kv=%d needs %zu B of LDS for ONE score row (> 146 KiB): contexts beyond %d cells are not supported", n_kv_max, att_lds_bytes(1, n_kv_max), pi_op_attention_max_cells()); return -1; }
    if (hd != 64 && hd != 128) {                                          // the plain kernel of the other head widths: one score row + the mask words + the query row
        const size_t b = att_lds_bytes(1, n_kv_max) + (size_t) hd * sizeof(float);
        if (b <= 60 * 1024) return 0;
        PI_CHECK_HIP(hipFuncSetAttribute((const void *) k_attention_any<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) (b < att_lds_cap(1) ? b : att_lds_cap(1))));
        PI_CHECK_HIP(hipFuncSetAttribute((const void *) k_attention_any<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) (b < att_lds_cap(1) ? b : att_lds_cap(1))));
        return 0;
    }
    if (att_lds_bytes(ATT_ROWS, n_kv_max) <= 60 * 1024) return 0;
    int rc = 0;
    auto want = [&](int rw) { const size_t b = att_lds_bytes(rw, n_kv_max), cap = att_lds_cap(rw); return b < cap ? b : cap; };
    if (hd == 128) rc = att_set_lds<128, 1>(want(1)) | att_set_lds<128, 2>(want(2)) | att_set_lds<128, 4>(want(4)) | att_set_lds<128, 8>(want(8));
    else           rc = att_set_lds<64, 1>(want(1)) | att_set_lds<64, 2>(want(2)) | att_set_lds<64, 4>(want(4)) | att_set_lds<64, 8>(want(8));
    return rc;
}

template <int HD>
static void att_launch(const att_args & g, int rows, int rw, hipStream_t stream) {
    const dim3 grid(g.n_head_kv, (rows + rw - 1) / rw);
    const size_t lds = (size_t) rw * (g.n_kv + g.mask_words) * sizeof(float);
    // algorithmic bytes: K and V of n_kv cells (f16) + the query rows + the fp32 output.  (A captured graph is launched with n_kv = n_ctx and reads the real
    // n_kv from device memory: its count carries the upper bound, bench.py prices attention from the call statistics instead.)
    const int nt = g.n_tok > 0 ? g.n_tok : 1;
    const int n_kv_priced = pi_account_n_kv() > 0 && pi_account_n_kv() < g.n_kv ? pi_account_n_kv() : g.n_kv;      // (eager launches of the engine: the pass's own n_kv)
    const double bytes = (double) n_kv_priced * g.n_head_kv * HD * 2 * 2 + (double) nt * g.n_head * HD * (2 + 4);
    if (g.alibi_m) {
        switch (rw) {
            case 1: PI_KT_LAUNCH(PI_KT_ATT, 0, nt, bytes, (k_attention<HD, 1, true>), grid, dim3(256), lds, stream, g); break;
            case 2: PI_KT_LAUNCH(PI_KT_ATT, 0, nt, bytes, (k_attention<HD, 2, true>), grid, dim3(256), lds, stream, g); break;
            case 4: PI_KT_LAUNCH(PI_KT_ATT, 0, nt, bytes, (k_attention<HD, 4, true>), grid, dim3(256), lds, stream, g); break;
            default: PI_KT_LAUNCH(PI_KT_ATT, 0, nt, bytes, (k_attention<HD, 8, true>), grid, dim3(256), lds, stream, g); break;
        }
        return;
    }
    switch (rw) {
        case 1: PI_KT_LAUNCH(PI_KT_ATT, 0, nt, bytes, (k_attention<HD, 1>), grid, dim3(256), lds, stream, g); break;
        case 2: PI_KT_LAUNCH(PI_KT_ATT, 0, nt, bytes, (k_attention<HD, 2>), grid, dim3(256), lds, stream, g); break;
        case 4: PI_KT_LAUNCH(PI_KT_ATT, 0, nt, bytes, (k_attention<HD, 4>), grid, dim3(256), lds, stream, g); break;
        default: PI_KT_LAUNCH(PI_KT_ATT, 0, nt, bytes, (k_attention<HD, 8>), grid, dim3(256), lds, stream, g); break;
    }
}

extern "C" int pi_op_attention(const uint16_t * q16, const uint16_t * kc, const uint16_t * vc, const uint32_t * mask, int hd, int n_head, int n_head_kv,
                               int n_tok, int n_kv, float * out, const int32_t * dyn_dev, void * stream) {
    return pi_op_attention_alibi(q16, kc, vc, mask, hd, n_head, n_head_kv, n_tok, n_kv, out, dyn_dev, 0, nullptr, stream);
}
extern "C" int pi_op_attention_ex(const uint16_t * q16, const uint16_t * kc, const uint16_t * vc, const uint32_t * mask, int hd, int n_head, int n_head_kv,
                                  int n_tok, int n_kv, float * out, const int32_t * dyn_dev, int per_token_kv, void * stream) {
    return pi_op_attention_alibi(q16, kc, vc, mask, hd, n_head, n_head_kv, n_tok, n_kv, out, dyn_dev, per_token_kv, nullptr, stream);
}
// ALiBi slopes of the query heads exactly as ggml_compute_forward_alibi_f32 forms them (ggml.c:10768-10787): m_k = powf(m0, k + 1) for the first
// 2^floor(log2 n_head) heads, powf(m1, 2 (k - that) + 1) for the rest (host libm, the values the reference adds)
extern "C" void pi_alibi_slopes(int n_head, float max_bias, float * out) {
    const int n_heads_log2_floor = 1 << (int) floor(log2(n_head));
    const float m0 = powf(2.0f, -(max_bias) / n_heads_log2_floor);
    const float m1 = powf(2.0f, -(max_bias / 2.0f) / n_heads_log2_floor);
    for (int k = 0; k < n_head; k++) out[k] = k < n_heads_log2_floor ? powf(m0, k + 1) : powf(m1, 2 * (k - n_heads_log2_floor) + 1);
}
extern "C" int pi_op_attention_alibi(const uint16_t * q16, const uint16_t * kc, const uint16_t * vc, const uint32_t * mask, int hd, int n_head, int n_head_kv,
                                     int n_tok, int n_kv, float * out, const int32_t * dyn_dev, int per_token_kv, const float * alibi_slopes_dev, void * stream) {
    if (n_tok <= 0) return 0;
    att_args g;
    g.alibi_m = alibi_slopes_dev;
    g.q16 = q16; g.kc = kc; g.vc = vc; g.mask = mask; g.out = out;
    g.n_head = n_head; g.n_head_kv = n_head_kv; g.n_tok = n_tok; g.n_kv = n_kv; g.mask_words = (n_kv + 31) / 32;
    g.exp_tab = pi_get_tables().exp_f16;
    g.scale = 1.0f / sqrtf((float) hd);                      // KQ_scale, llama.cpp:5189-5193
    g.dyn = dyn_dev; g.abort_w = pi_abort_word(); g.per_token = per_token_kv && dyn_dev ? 1 : 0;
    const int rows = n_tok * (n_head / n_head_kv);
    // rows per workgroup.  K/V are re-read (from L2) by every workgroup of a kv head, so more rows per workgroup mean fewer bytes through L2 — but a workgroup walks its rows'
    // phases one after the other, and at the contexts of a generation (a few hundred cells) it is that chain, not L2, that a launch waits for.  Round 6, 70B (8 kv heads x 8 query
    // heads), n_kv ~ 250, us per launch, one row per workgroup against the old rule (<= 256 workgroups): 7 tokens (448 workgroups) 12.8 against 16.1, 6 tokens 12.6 / 15.9, a
    // 127-token prompt pass (8 128 workgroups) 75.5 / 88.6; accepted tok/s 110.6 / 110.9 against 107.2 / 109.7 (profiles/r06_attention_rows_per_workgroup.log).  So: ONE row per
    // workgroup for as long as the launch's K/V re-reads stay under a budget (1.5 GB through L2 ~ 50 us: a 127-token prompt at 250 cells is 1 GB), rows doubled beyond it —
    // long contexts and wide prompts keep their large chunks.  PI_ATT_L2_MB moves the budget (0: the old rule, <= 256 workgroups).
    static const int64_t l2_budget = (getenv("PI_ATT_L2_MB") ? atoll(getenv("PI_ATT_L2_MB")) : 1536) << 20;
    constexpr int wg_max = 256;          // (the old rule; 128 -> 256 in round 3: a 4-token verify batch of the 70B model gets one row per workgroup, 16.4 -> ~11 us per layer)
    int rw = 1;
    if (l2_budget > 0) { while (rw < ATT_ROWS && (int64_t) n_head_kv * ((rows + rw - 1) / rw) * n_kv * hd * 4 > l2_budget) rw *= 2; }
    else while (rw < ATT_ROWS && (int64_t) n_head_kv * ((rows + rw - 1) / rw) > wg_max) rw *= 2;
    static const int rw_env = getenv("PI_ATT_RW") ? atoi(getenv("PI_ATT_RW")) : 0;      // tuning / tests
    if (rw_env == 1 || rw_env == 2 || rw_env == 4 || rw_env == 8) rw = rw_env;
    if (g.per_token) { const int r = n_head / n_head_kv; while (rw > 1 && r % rw) rw >>= 1; }      // a workgroup's rows stay inside one token (rows are token-major)
    while (rw > 1 && att_lds_bytes(rw, n_kv) > att_lds_cap(rw)) rw >>= 1;               // long contexts: fewer score rows per workgroup (they live in LDS)
    if ((size_t) rw * (n_kv + g.mask_words) * sizeof(float) > att_lds_cap(rw)) { pi_set_error("pi_op_attention: n_kv=%d needs more than %d KiB of LDS", n_kv, rw == 1 ? 146 : 156); return -1; }
    if (hd == 128) att_launch<128>(g, rows, rw, (hipStream_t) stream);
    else if (hd == 64) att_launch<64>(g, rows, rw, (hipStream_t) stream);
    else if (hd > 0 && hd <= 256 && hd % 2 == 0) {                       // every other head width: the plain kernel, one workgroup per (kv head, query row)
        const size_t lds = ((size_t) n_kv + g.mask_words + (size_t) hd) * sizeof(float);
        if (lds > att_lds_cap(1)) { pi_set_error("pi_op_attention: n_kv=%d needs more than 146 KiB of LDS", n_kv); return -1; }
        const int nt = n_tok > 0 ? n_tok : 1;
        const int n_kv_priced = pi_account_n_kv() > 0 && pi_account_n_kv() < n_kv ? pi_account_n_kv() : n_kv;
        const double bytes = (double) n_kv_priced * n_head_kv * hd * 2 * 2 + (double) nt * n_head * hd * (2 + 4);
        if (alibi_slopes_dev) PI_KT_LAUNCH(PI_KT_ATT, 0, nt, bytes, (k_attention_any<true>), dim3(n_head_kv, rows), dim3(256), lds, (hipStream_t) stream, g, hd);
        else                  PI_KT_LAUNCH(PI_KT_ATT, 0, nt, bytes, (k_attention_any<false>), dim3(n_head_kv, rows), dim3(256), lds, (hipStream_t) stream, g, hd);
    }
    else { pi_set_error("pi_op_attention: head_dim %d not supported (even, <= 256)", hd); return -1; }
    PI_CHECK_LAUNCH();
    return 0;
}
