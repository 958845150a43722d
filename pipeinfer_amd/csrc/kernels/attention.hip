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
// One workgroup (256 threads) per (kv head, chunk of <= 8 query rows); a row is one (token, query head of the GQA group),
// so K/V are read once per chunk for all query heads and verify tokens that share them.
//   phase 1  thread per cell: K row -> registers, the 32-accumulator dot with every query row (q in LDS), scale, mask -> S in LDS
//   phase 2  wave per row: max, table exp, exact integer sum, P = f16-rounded probability (kept as f32 in LDS)
//   phase 3  lane (l, ev) owns cells c % 8 == l and dims 8ev..8ev+7: four accumulator sets j = (c / 8) % 4 per row, folded
//            in the reference's order with lane shuffles; tail cells in double by lane l == 0
#include "ops_common.h"

#define ATT_ROWS 8

struct att_args {
    const uint16_t * q16; const uint16_t * kc; const uint16_t * vc; const uint32_t * mask;
    float * out;
    int n_head, n_head_kv, n_tok, n_kv, mask_words;        // n_kv / mask_words: LDS row stride and mask row stride (static)
    const uint16_t * exp_tab;
    float scale;
    const int32_t * dyn;                                    // optional {kv_head, n_kv} in device memory: lets one captured graph serve every call
};

template <int HD>
__global__ void __launch_bounds__(256) k_attention(const att_args ga) {
    extern __shared__ float smem[];
    att_args g = ga;
    const int s_stride = ga.n_kv;                           // LDS rows keep the static stride
    if (ga.dyn) g.n_kv = ga.dyn[1];
    const int hk = blockIdx.x, chunk = blockIdx.y;
    const int r = g.n_head / g.n_head_kv;                   // query heads per kv head (broadcast rule i02 = i12 / r2, ggml.c:9425)
    const int n_rows_total = g.n_tok * r;
    const int row0 = chunk * ATT_ROWS;
    const int n_rows = min(ATT_ROWS, n_rows_total - row0);
    const int n_gqa = g.n_head_kv * HD, n_embd = g.n_head * HD;
    float * S  = smem;                                      // [ATT_ROWS][n_kv]
    float * qf = S + (size_t) ATT_ROWS * s_stride;           // [ATT_ROWS][HD]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    for (int i = tid; i < n_rows * HD; i += 256) {          // rows are token-major: row = t * r + gq -> query head hk * r + gq
        const int rr = i / HD, e = i % HD, row = row0 + rr, t = row / r, gq = row % r;
        qf[rr * HD + e] = pi_h2f(g.q16[(int64_t) t * n_embd + (hk * r + gq) * HD + e]);
    }
    __syncthreads();

    // ---- phase 1: scores, ggml_vec_dot_f16 order ----
    for (int c = tid; c < g.n_kv; c += 256) {
        float kf[HD];
        const uint4 * kp = (const uint4 *) (g.kc + (int64_t) c * n_gqa + hk * HD);
#pragma unroll
        for (int i = 0; i < HD / 8; i++) {
            const uint4 v = kp[i];
            const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int j = 0; j < 4; j++) { kf[8 * i + 2 * j] = pi_h2f((uint16_t) (w4[j] & 0xffff)); kf[8 * i + 2 * j + 1] = pi_h2f((uint16_t) (w4[j] >> 16)); }
        }
        for (int rr = 0; rr < n_rows; rr++) {
            const int t = (row0 + rr) / r;
            const bool vis = (g.mask[(int64_t) t * g.mask_words + (c >> 5)] >> (c & 31)) & 1;
            const float * q = qf + rr * HD;
            float s[32];
#pragma unroll
            for (int a = 0; a < 32; a++) s[a] = 0.0f;
#pragma unroll
            for (int st = 0; st < HD / 32; st++)
#pragma unroll
                for (int a = 0; a < 32; a++) s[a] = fmaf(kf[32 * st + a], q[32 * st + a], s[a]);
#pragma unroll
            for (int l = 0; l < 8; l++) { s[l] = s[l] + s[16 + l]; s[8 + l] = s[8 + l] + s[24 + l]; }
#pragma unroll
            for (int l = 0; l < 8; l++) s[l] = s[l] + s[8 + l];
            const float t0 = s[0] + s[4], t1 = s[1] + s[5], t2 = s[2] + s[6], t3 = s[3] + s[7];
            const float dot = (t0 + t1) + (t2 + t3);
            S[(size_t) rr * s_stride + c] = vis ? dot * g.scale : -INFINITY;
        }
    }
    __syncthreads();

    // ---- phase 2: soft max, one wave per row ----
    for (int rr = wv; rr < n_rows; rr += 4) {
        float * s = S + (size_t) rr * s_stride;
        float mx = -INFINITY;
        for (int c = lane; c < g.n_kv; c += 64) mx = fmaxf(mx, s[c]);
        mx = pi_wave_max(mx);
        unsigned long long isum = 0;                          // in units of 2^-24: exact
        for (int c = lane; c < g.n_kv; c += 64) {
            const float x = s[c];
            float v = 0.0f;
            if (x != -INFINITY) { v = pi_h2f(g.exp_tab[pi_f2h(x - mx)]); isum += (unsigned long long) (v * 16777216.0f); }
            s[c] = v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) isum += __shfl_xor(isum, o, 64);
        const double sum = (double) isum * (1.0 / 16777216.0);
        const float inv = (float) (1.0 / sum);
        for (int c = lane; c < g.n_kv; c += 64) s[c] = pi_h2f(pi_f2h(s[c] * inv));        // P rounded to f16 as the V.P mul_mat does
    }
    __syncthreads();

    // ---- phase 3: out[row][e] = ggml_vec_dot_f16(n_kv, V[:, e], P[row, :]) ----
    constexpr int TPC = HD / 8;                              // lanes along the head dimension
    constexpr int NRG = 256 / (8 * TPC);                     // row groups in the workgroup: 4 (HD 64) or 2 (HD 128)
    constexpr int RPT = ATT_ROWS / NRG;                      // rows per thread
    const int l = tid & 7, ev = (tid >> 3) % TPC, rg = tid / (8 * TPC);
    const int np = g.n_kv & ~31;
    float acc[RPT][4][8];
#pragma unroll
    for (int i = 0; i < RPT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 8; e++) acc[i][j][e] = 0.0f;
    for (int c0 = 0; c0 < np; c0 += 32) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = c0 + 8 * j + l;
            const uint4 v = *(const uint4 *) (g.vc + (int64_t) c * n_gqa + hk * HD + 8 * ev);
            const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
            float vf[8];
#pragma unroll
            for (int u = 0; u < 4; u++) { vf[2 * u] = pi_h2f((uint16_t) (w4[u] & 0xffff)); vf[2 * u + 1] = pi_h2f((uint16_t) (w4[u] >> 16)); }
#pragma unroll
            for (int i = 0; i < RPT; i++) {
                const int rr = rg + i * NRG;
                if (rr < n_rows) {
                    const float p = S[(size_t) rr * s_stride + c];
#pragma unroll
                    for (int e = 0; e < 8; e++) acc[i][j][e] = fmaf(vf[e], p, acc[i][j][e]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RPT; i++) {
        const int rr = rg + i * NRG;
        float res[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float s = (acc[i][0][e] + acc[i][2][e]) + (acc[i][1][e] + acc[i][3][e]);     // registers: 0+=2, 1+=3, 0+=1  (per lane l)
            s = s + __shfl_xor(s, 4, 64);                                               // t[l] = x[l] + x[l+4]
            s = s + __shfl_xor(s, 1, 64);                                               // hadd
            s = s + __shfl_xor(s, 2, 64);                                               // hadd
            res[e] = s;
        }
        if (rr < n_rows && l == 0) {
            const int row = row0 + rr, t = row / r, gq = row % r;
            float * o = g.out + (int64_t) t * n_embd + (hk * r + gq) * HD + 8 * ev;
            if (np == g.n_kv) {
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = res[e];
            } else {
                double sd[8];
#pragma unroll
                for (int e = 0; e < 8; e++) sd[e] = (double) res[e];
                for (int c = np; c < g.n_kv; c++) {                                     // leftovers, sequential, in double
                    const float p = S[(size_t) rr * s_stride + c];
                    const uint4 v = *(const uint4 *) (g.vc + (int64_t) c * n_gqa + hk * HD + 8 * ev);
                    const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        sd[2 * u]     += (double) (pi_h2f((uint16_t) (w4[u] & 0xffff)) * p);
                        sd[2 * u + 1] += (double) (pi_h2f((uint16_t) (w4[u] >> 16)) * p);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = (float) sd[e];
            }
        }
    }
}

// raises the dynamic-LDS limit of the kernel for contexts longer than ~2000 cells; call once outside any graph capture
extern "C" int pi_op_attention_reserve(int hd, int n_kv_max) {
    const size_t lds = ((size_t) ATT_ROWS * n_kv_max + (size_t) ATT_ROWS * hd) * sizeof(float);
    if (lds > 160 * 1024) { pi_set_error("pi_op_attention_reserve: n_kv=%d needs %zu B of LDS (> 160 KiB)", n_kv_max, lds); return -1; }
    if (lds <= 64 * 1024) return 0;
    if (hd == 128) PI_CHECK_HIP(hipFuncSetAttribute((const void *) k_attention<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    else           PI_CHECK_HIP(hipFuncSetAttribute((const void *) k_attention<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    return 0;
}

extern "C" int pi_op_attention(const uint16_t * q16, const uint16_t * kc, const uint16_t * vc, const uint32_t * mask, int hd, int n_head, int n_head_kv,
                               int n_tok, int n_kv, float * out, const int32_t * dyn_dev, void * stream) {
    if (n_tok <= 0) return 0;
    att_args g;
    g.q16 = q16; g.kc = kc; g.vc = vc; g.mask = mask; g.out = out;
    g.n_head = n_head; g.n_head_kv = n_head_kv; g.n_tok = n_tok; g.n_kv = n_kv; g.mask_words = (n_kv + 31) / 32;
    g.exp_tab = pi_get_tables().exp_f16;
    g.scale = 1.0f / sqrtf((float) hd);                      // KQ_scale, llama.cpp:5189-5193
    g.dyn = dyn_dev;
    const int rows = n_tok * (n_head / n_head_kv);
    const dim3 grid(n_head_kv, (rows + ATT_ROWS - 1) / ATT_ROWS);
    const size_t lds = ((size_t) ATT_ROWS * n_kv + (size_t) ATT_ROWS * hd) * sizeof(float);
    if (lds > 160 * 1024) { pi_set_error("pi_op_attention: n_kv=%d needs %zu B of LDS (> 160 KiB)", n_kv, lds); return -1; }
    if (hd == 128) {
        hipLaunchKernelGGL(k_attention<128>, grid, dim3(256), lds, (hipStream_t) stream, g);
    } else if (hd == 64) {
        hipLaunchKernelGGL(k_attention<64>, grid, dim3(256), lds, (hipStream_t) stream, g);
    } else { pi_set_error("pi_op_attention: head_dim %d not supported (64 or 128)", hd); return -1; }
    PI_CHECK_LAUNCH();
    return 0;
}
