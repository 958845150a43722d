// KV-cache attention for the decode / verify path.
//
// Reference: llm_build_kqv (llama.cpp:3698-3764) = mul_mat(K, Q) -> ggml_scale -> ggml_add(mask) -> ggml_soft_max ->
// mul_mat(V, P), with the CPU path's rounding points:
//   * Q and the soft-max output P are rounded to f16 before their dot products (ggml.c:9493-9506), K/V are f16 in the cache;
//   * products are accumulated in fp32 (ggml_vec_dot_f16, ggml.c:1193-1228);
//   * soft_max: row max, exp through the fp16 table indexed by f16(x - max), sum in double, scale by (float)(1/sum),
//     masked cells contribute exactly 0 (ggml.c:10551-10620).
//
// One workgroup (256 threads) per (kv head, chunk of <= 8 query rows); a "row" is one (token, query head of the group).
// All query heads that share a KV head (GQA/MQA) and all tokens of the verify batch read K/V once per chunk.
//   phase 1  thread per cell: K row -> registers, dot with every query row (q in LDS as f32), scale, mask -> S[row][cell] in LDS
//   phase 2  wave per row:    max, table exp, double sum, P = f16-rounded probability (kept as f32 in LDS)
//   phase 3  16/8 threads per cell (8 halfs of V each), fp32 accumulators per (row, dim), reduced through shuffles + LDS
#include "ops_common.h"

#define ATT_ROWS 8

struct att_args {
    const uint16_t * q16; const uint16_t * kc; const uint16_t * vc; const uint32_t * mask;
    float * out;
    int n_head, n_head_kv, n_tok, n_kv, mask_words;
    const uint16_t * exp_tab;
    float scale;
};

template <int HD>
__global__ void __launch_bounds__(256) k_attention(const att_args g) {
    extern __shared__ float smem[];
    const int hk = blockIdx.x, chunk = blockIdx.y;
    const int r = g.n_head / g.n_head_kv;                   // query heads per kv head
    const int n_rows_total = g.n_tok * r;
    const int row0 = chunk * ATT_ROWS;
    const int n_rows = min(ATT_ROWS, n_rows_total - row0);
    const int n_gqa = g.n_head_kv * HD, n_embd = g.n_head * HD;
    float * S   = smem;                                     // [ATT_ROWS][n_kv]
    float * qf  = S + (size_t) ATT_ROWS * g.n_kv;            // [ATT_ROWS][HD]
    float * red = qf + ATT_ROWS * HD;                        // [4 waves][ATT_ROWS][HD]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // rows are ordered token-major: row = t * r + g  ->  query head h = hk * r + g  (broadcast rule i02 = i12 / r2, ggml.c:9425)
    for (int i = tid; i < n_rows * HD; i += 256) {
        const int rr = i / HD, e = i % HD, row = row0 + rr, t = row / r, gq = row % r;
        qf[rr * HD + e] = pi_h2f(g.q16[(int64_t) t * n_embd + (hk * r + gq) * HD + e]);
    }
    __syncthreads();

    // ---- phase 1: scores ----
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
            float acc = 0.0f;
            const float4 * q4 = (const float4 *) (qf + rr * HD);
#pragma unroll
            for (int i = 0; i < HD / 4; i++) {
                const float4 qv = q4[i];
                acc = fmaf(kf[4 * i], qv.x, acc); acc = fmaf(kf[4 * i + 1], qv.y, acc);
                acc = fmaf(kf[4 * i + 2], qv.z, acc); acc = fmaf(kf[4 * i + 3], qv.w, acc);
            }
            S[(size_t) rr * g.n_kv + c] = vis ? acc * g.scale : -INFINITY;
        }
    }
    __syncthreads();

    // ---- phase 2: soft max, one wave per row ----
    for (int rr = wv; rr < n_rows; rr += 4) {
        float * s = S + (size_t) rr * g.n_kv;
        float mx = -INFINITY;
        for (int c = lane; c < g.n_kv; c += 64) mx = fmaxf(mx, s[c]);
        mx = pi_wave_max(mx);
        double sum = 0.0;
        for (int c = lane; c < g.n_kv; c += 64) {
            const float x = s[c];
            float v = 0.0f;
            if (x != -INFINITY) { v = pi_h2f(g.exp_tab[pi_f2h(x - mx)]); sum += (double) v; }
            s[c] = v;
        }
        sum = pi_wave_sum_f64(sum);
        const float inv = (float) (1.0 / sum);
        for (int c = lane; c < g.n_kv; c += 64) s[c] = pi_h2f(pi_f2h(s[c] * inv));        // P rounded to f16 as the V.P mul_mat does
    }
    __syncthreads();

    // ---- phase 3: out[row][e] = sum_c V[c][e] * P[row][c] ----
    constexpr int TPC = HD / 8;                            // threads per cell
    constexpr int CPW = 64 / TPC;                          // cells per wave step
    const int ev = lane % TPC, cg = wv * CPW + lane / TPC; // dims 8*ev..8*ev+7 ; cell group 0..(4*CPW-1)
    float acc[ATT_ROWS][8];
#pragma unroll
    for (int rr = 0; rr < ATT_ROWS; rr++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[rr][j] = 0.0f;
    for (int c = cg; c < g.n_kv; c += 4 * CPW) {
        const uint4 v = *(const uint4 *) (g.vc + (int64_t) c * n_gqa + hk * HD + 8 * ev);
        const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
        float vf[8];
#pragma unroll
        for (int j = 0; j < 4; j++) { vf[2 * j] = pi_h2f((uint16_t) (w4[j] & 0xffff)); vf[2 * j + 1] = pi_h2f((uint16_t) (w4[j] >> 16)); }
#pragma unroll
        for (int rr = 0; rr < ATT_ROWS; rr++) {
            if (rr < n_rows) {
                const float p = S[(size_t) rr * g.n_kv + c];
#pragma unroll
                for (int j = 0; j < 8; j++) acc[rr][j] = fmaf(vf[j], p, acc[rr][j]);
            }
        }
    }
    // reduce the CPW cell groups inside the wave
#pragma unroll
    for (int rr = 0; rr < ATT_ROWS; rr++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float v = acc[rr][j];
#pragma unroll
            for (int o = TPC; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            acc[rr][j] = v;
        }
    if (lane < TPC) {
#pragma unroll
        for (int rr = 0; rr < ATT_ROWS; rr++)
#pragma unroll
            for (int j = 0; j < 8; j++) red[(wv * ATT_ROWS + rr) * HD + 8 * lane + j] = acc[rr][j];
    }
    __syncthreads();
    for (int i = tid; i < n_rows * HD; i += 256) {
        const int rr = i / HD, e = i % HD, row = row0 + rr, t = row / r, gq = row % r;
        const float v = (red[(0 * ATT_ROWS + rr) * HD + e] + red[(1 * ATT_ROWS + rr) * HD + e]) + (red[(2 * ATT_ROWS + rr) * HD + e] + red[(3 * ATT_ROWS + rr) * HD + e]);
        g.out[(int64_t) t * n_embd + (hk * r + gq) * HD + e] = v;
    }
}

extern "C" int pi_op_attention(const uint16_t * q16, const uint16_t * kc, const uint16_t * vc, const uint32_t * mask, int hd, int n_head, int n_head_kv,
                               int n_tok, int n_kv, float * out, void * stream) {
    if (n_tok <= 0) return 0;
    att_args g;
    g.q16 = q16; g.kc = kc; g.vc = vc; g.mask = mask; g.out = out;
    g.n_head = n_head; g.n_head_kv = n_head_kv; g.n_tok = n_tok; g.n_kv = n_kv; g.mask_words = (n_kv + 31) / 32;
    g.exp_tab = pi_get_tables().exp_f16;
    g.scale = 1.0f / sqrtf((float) hd);                      // KQ_scale, llama.cpp:5189-5193
    const int rows = n_tok * (n_head / n_head_kv);
    const dim3 grid(n_head_kv, (rows + ATT_ROWS - 1) / ATT_ROWS);
    const size_t lds = ((size_t) ATT_ROWS * n_kv + (size_t) ATT_ROWS * hd + (size_t) 4 * ATT_ROWS * hd) * sizeof(float);
    if (lds > 160 * 1024) { pi_set_error("pi_op_attention: n_kv=%d needs %zu B of LDS (> 160 KiB)", n_kv, lds); return -1; }
    if (hd == 128) {
        if (lds > 64 * 1024) PI_CHECK_HIP(hipFuncSetAttribute((const void *) k_attention<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        hipLaunchKernelGGL(k_attention<128>, grid, dim3(256), lds, (hipStream_t) stream, g);
    } else if (hd == 64) {
        if (lds > 64 * 1024) PI_CHECK_HIP(hipFuncSetAttribute((const void *) k_attention<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        hipLaunchKernelGGL(k_attention<64>, grid, dim3(256), lds, (hipStream_t) stream, g);
    } else { pi_set_error("pi_op_attention: head_dim %d not supported (64 or 128)", hd); return -1; }
    PI_CHECK_LAUNCH();
    return 0;
}
