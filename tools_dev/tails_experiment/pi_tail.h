// Tails: a small row operator that CONSUMES what a launch produced, run by the launch's last workgroup instead of as a launch of its own.
//
// At decode widths a layer is nine dependent launches, four of them tiny (two norm + quantise, the attention output's quantiser, the activation + quantise): alone each
// costs the ~5 us floor of a dependent launch, and beside another context's kernels — the draft of a one-GPU PipeInfer run next to the 70B target — each waits
// for a dispatch slot between the other queue's kernels, which is most of what a draft step costs there.  A tail removes the launch: every workgroup of the
// producer checks in when its (write-through) output stores are out; workgroup 0 waits for all of them and runs the row operator over the whole result.  Arithmetic: the operator's own (the quantiser core is shared through pi_device.h; the norm's double sums are formed per
// 256-value chunk and folded in chunk order, which is what k_norm_quant's one-chunk-per-wave layout computes for rows of up to 16 chunks), so the bytes do not move.
// One workgroup walks the rows: tails are for small rows and decode widths (pi_tail_fits), everything else keeps its launch.
#pragma once
#include "pi_device.h"

enum { PI_TAIL_NONE = 0, PI_TAIL_QUANT = 1, PI_TAIL_NORM_QUANT = 2, PI_TAIL_ACT_QUANT = 3 };

struct pi_tail {
    int kind = PI_TAIL_NONE;
    int n = 0, n_tok = 0, rms = 0;          // row width (multiple of 256: Q8_K activations only), rows
    float eps = 0.0f;
    const float * x = nullptr;              // QUANT / NORM_QUANT: the rows; ACT_QUANT: gate
    const float * u = nullptr;              // ACT_QUANT: up
    const float * w = nullptr, * b = nullptr;   // NORM_QUANT: weight, bias (nullable)
    int8_t * qs = nullptr; float * dd = nullptr; int16_t * bsums = nullptr;     // Q8_K planes of the consumer (pi_q8_layout)
    const uint16_t * tab = nullptr;         // ACT_QUANT: silu table (f16 -> f16)
    int * counter = nullptr;                // PI_TAIL_SUBS counters PI_TAIL_STRIDE ints apart (4 KB), one set per stream of launches, zero between launches
};

// rows one workgroup can walk without costing more than the launch it replaces
static inline bool pi_tail_fits(int kind, int64_t n, int n_tok) {
    if (n % 256 || n_tok < 1 || n_tok > 2) return false;
    const int64_t chunks = n / 256 * n_tok;
    return kind == PI_TAIL_ACT_QUANT ? chunks <= 48 : chunks <= 16;
}

// Coherence without fences.  A device-scope fence (buffer_wbl2 / buffer_inv) per workgroup writes back and invalidates a whole XCD's L2 — measured: a TinyLlama step went
// from 1.3 to 6.1 ms and the 70B target beside it from 12 to 32 ms per call.  Instead the PRODUCER's output goes out write-through (pi_store_coherent: sc1, past the
// XCD's L2) when its launch carries a tail, the ticket is taken once those stores have been acknowledged (s_waitcnt vmcnt(0) in front of the workgroup barrier), and the
// tail READS the rows with sc1 loads that do not look at its own L2 (which may hold lines of the same buffers from an earlier layer).  What the tail writes is ordinary:
// the next kernel sees it like any kernel's output.
static __device__ __forceinline__ void pi_store_coherent(float * p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// eight 16-byte loads in flight and the wait for them in ONE asm statement: the compiler takes an asm's outputs as ready when the statement ends
typedef float pi_f4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void pi_load4x8_coherent(pi_f4 (&v)[8], const float * const (&p)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %9, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc0 sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc0 sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc0 sc1\n\t"
                 "global_load_dwordx4 %5, %13, off sc0 sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc0 sc1\n\t"
                 "global_load_dwordx4 %7, %15, off sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
}

// Who runs the tail: workgroup 0 of the launch, once every workgroup has checked in.  Checking in is ONE fire-and-forget device-scope add (no returned value to wait
// for) on one of PI_TAIL_SUBS counters a cache line apart — thousands of workgroups taking a TICKET from one address serialise at the memory side (measured: +16 us per
// launch on TinyLlama's 1 000-5 600-workgroup mat-vecs, a step 1.3 -> 2.7 ms), and waiting for the ticket's value keeps every workgroup's slot ~1.5 us longer.
// Workgroup 0 (the first one dispatched) polls the counters' sum with one coalesced read per try and zeroes them for the next launch on the stream.
// Every workgroup of the launch must call this exactly once, with all its threads (aborted ones too: the count must come out).  True only in workgroup 0.
constexpr int PI_TAIL_SUBS = 32, PI_TAIL_STRIDE = 32;                             // 32 counters, 128 bytes apart: 4 KB
static __device__ __forceinline__ bool pi_last_workgroup(int * counter, unsigned n_wg, unsigned wg_id) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // this wave's (write-through) output stores are out
    __syncthreads();
    if (threadIdx.x == 0) (void) __hip_atomic_fetch_add(counter + (wg_id % PI_TAIL_SUBS) * PI_TAIL_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wg_id != 0) return false;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        for (;;) {
            int v = lane < PI_TAIL_SUBS ? __hip_atomic_load(counter + lane * PI_TAIL_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((unsigned) v >= n_wg) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane < PI_TAIL_SUBS) __hip_atomic_store(counter + lane * PI_TAIL_STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // every add has landed: ready for the next launch
    }
    __syncthreads();
    return true;
}

// the row operator, by all threads of ONE workgroup (blockDim.x a multiple of 64, at least two waves)
static __device__ __forceinline__ void pi_run_tail(const pi_tail & T) {
    __shared__ double t_sh[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int chunks = T.n >> 8;
    constexpr int CAP = 8;                                                       // chunks a wave holds in registers at once (one round trip per batch)
    for (int t = 0; t < T.n_tok; t++) {
        const float * xr = T.x + (int64_t) t * T.n;
        if (T.kind == PI_TAIL_NORM_QUANT) {
            // rows of <= 16 chunks on >= 2 waves: a wave's chunks (wave, wave + nw, ...) all fit its registers; loaded ONCE, used by the statistics and by the quantiser
            pi_f4 xv[CAP];
            {   // (chunks past the row's end re-read its last chunk: every wave issues the same eight loads)
                const float * ptr[CAP];
#pragma unroll
                for (int i = 0; i < CAP; i++) { const int c = wave + i * nw; ptr[i] = xr + (int64_t) (c < chunks ? c : chunks - 1) * 256 + lane * 4; }
                pi_load4x8_coherent(xv, ptr);
            }
            float mean = 0.0f;
            // k_norm_quant's statistics for rows of <= 16 chunks: the partial sum of chunk c is the wave sum of the lanes' four-value sums, the total is the partial sums
            // added in chunk order (llm_build_norm: ggml_rms_norm / ggml_norm accumulate in double, ggml.c:9028-9074, 8959-9008)
            auto chunk_sums = [&](bool centred) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < CAP; i++) {
                    const int c = wave + i * nw;
                    if (c < chunks) {
                        const pi_f4 v = xv[i];
                        double acc = 0.0;
                        if (centred) { const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean; acc += (double) (a0 * a0); acc += (double) (a1 * a1); acc += (double) (a2 * a2); acc += (double) (a3 * a3); }
                        else if (T.rms) { acc += (double) (v.x * v.x); acc += (double) (v.y * v.y); acc += (double) (v.z * v.z); acc += (double) (v.w * v.w); }
                        else { acc += (double) v.x; acc += (double) v.y; acc += (double) v.z; acc += (double) v.w; }
                        acc = pi_wave_sum_f64(acc);
                        if (lane == 0) t_sh[c] = acc;
                    }
                }
                __syncthreads();
                double s = 0.0;
                for (int c = 0; c < chunks; c++) s += t_sh[c];
                return s;
            };
            double s = chunk_sums(false);
            if (!T.rms) { mean = (float) (s / T.n); s = chunk_sums(true); }
            const float var = (float) (s / T.n);
            const float scale = 1.0f / sqrtf(var + T.eps);
#pragma unroll
            for (int i = 0; i < CAP; i++) {
                const int c = wave + i * nw;
                if (c < chunks) {
                    const int64_t e0 = (int64_t) c * 256 + lane * 4;
                    const float4 wv = T.w ? *(const float4 *) (T.w + e0) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                    const float4 bv = T.b ? *(const float4 *) (T.b + e0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    const float xs[4] = { xv[i].x, xv[i].y, xv[i].z, xv[i].w }, ww[4] = { wv.x, wv.y, wv.z, wv.w }, bb[4] = { bv.x, bv.y, bv.z, bv.w };
                    float ys[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float y = (xs[j] - mean) * scale;
                        if (T.w) y = y * ww[j];
                        if (T.b) y = y + bb[j];
                        ys[j] = y;
                    }
                    int q[4], s16;
                    const float d = pi_wave_q8K(ys, lane, q, s16);
                    pi_wave_q8K_store(q, s16, d, lane, (int64_t) t * chunks + c, T.qs, T.dd, T.bsums);
                }
            }
        } else {
            const float * ur = T.u ? T.u + (int64_t) t * T.n : nullptr;
            for (int c0 = wave; c0 < chunks; c0 += CAP * nw) {                   // batches of CAP chunks per wave: one round trip each
                pi_f4 xv[CAP], uv[CAP];
                {
                    const float * ptr[CAP];
#pragma unroll
                    for (int i = 0; i < CAP; i++) { const int c = c0 + i * nw; ptr[i] = xr + (int64_t) (c < chunks ? c : chunks - 1) * 256 + lane * 4; }
                    pi_load4x8_coherent(xv, ptr);
                    if (T.kind == PI_TAIL_ACT_QUANT) {
#pragma unroll
                        for (int i = 0; i < CAP; i++) { const int c = c0 + i * nw; ptr[i] = ur + (int64_t) (c < chunks ? c : chunks - 1) * 256 + lane * 4; }
                        pi_load4x8_coherent(uv, ptr);
                    }
                }
#pragma unroll
                for (int i = 0; i < CAP; i++) {
                    const int c = c0 + i * nw;
                    if (c < chunks) {
                        const float xs[4] = { xv[i].x, xv[i].y, xv[i].z, xv[i].w };
                        float ys[4];
                        if (T.kind == PI_TAIL_ACT_QUANT) {
                            const float us[4] = { uv[i].x, uv[i].y, uv[i].z, uv[i].w };
#pragma unroll
                            for (int j = 0; j < 4; j++) ys[j] = pi_h2f(T.tab[pi_f2h(xs[j])]) * us[j];            // ggml_silu through the f16 table, then ggml_mul (k_act_quant)
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++) ys[j] = xs[j];
                        }
                        int q[4], s16;
                        const float d = pi_wave_q8K(ys, lane, q, s16);
                        pi_wave_q8K_store(q, s16, d, lane, (int64_t) t * chunks + c, T.qs, T.dd, T.bsums);
                    }
                }
            }
        }
    }
}
