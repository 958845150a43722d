// Device runtime glue: error text, fp16 tables, memory helpers, weight upload + one-time re-layout.
#include "ops_common.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static thread_local char g_err[512] = "";

void pi_set_error(const char * fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char * pi_last_error(void) { return g_err; }

extern "C" int pi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- fp16 lookup tables (ggml.c:2190-2201; formulas ggml.c:1418-1421, 1479-1481) ---------------------------------
// Built on the host with the host libm so that they are the very tables the reference CPU path uses on this machine.
static uint16_t host_f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t) ((x >> 16) & 0x8000u);
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (uint16_t) (sign | 0x7e00u | ((x >> 13) & 0x3ffu));
    if (x >= 0x47800000u) return (uint16_t) (sign | 0x7c00u);
    const uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u;
    if (e < 113) {
        if (e < 102) return sign;
        const int sh = 126 - (int) e;
        uint32_t q = m >> sh;
        const uint32_t rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) q++;
        return (uint16_t) (sign | q);
    }
    uint32_t h = (e - 112) << 10 | (m & 0x7fffffu) >> 13;
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t) (sign | h);
}

static float host_f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t) (h & 0x8000u) << 16, e = (h >> 10) & 0x1f;
    uint32_t m = h & 0x3ffu, bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { int s = -1; do { s++; m <<= 1; } while (!(m & 0x400u)); bits = sign | (uint32_t) (112 - s) << 23 | (m & 0x3ffu) << 13; }
    } else if (e == 31) bits = sign | 0x7f800000u | m << 13;
    else bits = sign | (e + 112) << 23 | m << 13;
    float f; memcpy(&f, &bits, 4);
    return f;
}

static pi_tables g_tables = { nullptr, nullptr, nullptr };
static int g_device = -1;

const pi_tables & pi_get_tables() { return g_tables; }

extern "C" int pi_device_init(int device) {
    if (pi_device_count() <= 0) { pi_set_error("pi_device_init: no HIP device visible (the MI355X path has no CPU fallback)"); return -1; }
    PI_CHECK_HIP(hipSetDevice(device));
    if (g_tables.exp_f16 && g_device == device) return 0;
    g_device = device;
    std::vector<uint16_t> t(3u << 16);
    for (int i = 0; i < (1 << 16); i++) {
        const float f = host_f16_to_f32((uint16_t) i);
        t[i]             = host_f32_to_f16(expf(f));
        t[(1 << 16) + i] = host_f32_to_f16(f / (1.0f + expf(-f)));
        const float inner = 0.79788456080286535587989211986876f * f * (1.0f + 0.044715f * f * f);
        t[(2 << 16) + i] = host_f32_to_f16(0.5f * f * (1.0f + tanhf(inner)));
    }
    uint16_t * dev = nullptr;
    PI_CHECK_HIP(hipMalloc((void **) &dev, t.size() * 2));
    PI_CHECK_HIP(hipMemcpy(dev, t.data(), t.size() * 2, hipMemcpyHostToDevice));
    g_tables.exp_f16 = dev; g_tables.silu_f16 = dev + (1 << 16); g_tables.gelu_f16 = dev + (2 << 16);
    return 0;
}

// ---- memory helpers ---------------------------------------------------------------------------------------------
extern "C" void * pi_malloc(size_t bytes) {
    void * p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { pi_set_error("hipMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
extern "C" void pi_free(void * p) { if (p) (void) hipFree(p); }
extern "C" int pi_memcpy_h2d(void * dst, const void * src, size_t n, void * stream) {
    PI_CHECK_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (hipStream_t) stream));
    return 0;
}
extern "C" int pi_memcpy_d2h(void * dst, const void * src, size_t n, void * stream) {
    PI_CHECK_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, (hipStream_t) stream));
    PI_CHECK_HIP(hipStreamSynchronize((hipStream_t) stream));
    return 0;
}
extern "C" int pi_memset(void * dst, int v, size_t n, void * stream) { PI_CHECK_HIP(hipMemsetAsync(dst, v, n, (hipStream_t) stream)); return 0; }
extern "C" int pi_device_sync(void) { PI_CHECK_HIP(hipDeviceSynchronize()); return 0; }
extern "C" int pi_stream_sync(void * stream) { PI_CHECK_HIP(hipStreamSynchronize((hipStream_t) stream)); return 0; }

// ---- abort word of a run (pi_device.h: pi_aborted) --------------------------------------------------------------------------------
static thread_local const int32_t * g_abort_word = nullptr;
const int32_t * pi_abort_word() { return g_abort_word; }
extern "C" void pi_set_abort_word(const int32_t * word_dev) { g_abort_word = word_dev; }
// cells the attention launches of this thread are PRICED with (launch accounting only): a forward pass is launched with n_kv = n_ctx and reads the real n_kv on the device
static thread_local int g_account_n_kv = 0;
int pi_account_n_kv() { return g_account_n_kv; }
extern "C" void pi_set_account_n_kv(int n_kv) { g_account_n_kv = n_kv > 0 ? n_kv : 0; }
static thread_local int g_cu_count = 256;
int pi_cu_count() { return g_cu_count; }
extern "C" void pi_set_cu_count(int n) { g_cu_count = n > 0 && n <= 256 ? n : 256; }
// device-scope atomics only: the exchange at a run's start and the compare-and-swap of a cancel never tear, whichever stream runs first
__global__ void k_abort_begin(int32_t * word, int32_t run_id, const int32_t * host_flag) {
    const int32_t f = host_flag ? __hip_atomic_load(host_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
    // (a later segment of the SAME run keeps an abort the word already carries: one inherited from the upstream stage has no host flag behind it)
    const int32_t old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int32_t keep = (old >> 1) == run_id ? (old & 1) : 0;
    __hip_atomic_exchange(word, (run_id << 1) | (f != 0) | keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- arrival flags of the stage transport (transport.cpp): the activations of a run cross a stage boundary with no host thread on the critical path.
// A flag lives in page-locked host memory that both processes map: {ticket << 1 | aborted, -, realtime stamp}.  The producing stage's stream writes it right behind its peer
// store (k_flag_publish); the consuming stage has enqueued its run already, behind k_abort_begin_wait, which polls the flag from the device and arms the run's abort word —
// with the abort the producer's run ended with, if any, so that a run cut short upstream costs nothing downstream either.  The poll is bounded (a dead producer must not
// hang the consumer's GPU): on time-out the run is marked aborted and *timeout_flag is raised.
struct pi_xflag { uint32_t value, pad; unsigned long long stamp; };
__global__ void k_flag_publish(pi_xflag * flag, uint32_t ticket, const int32_t * abort_word, int force_abort) {
    const int32_t w = abort_word ? __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    flag->stamp = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(&flag->value, (ticket << 1) | (uint32_t) ((w & 1) | (force_abort != 0)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_abort_begin_wait(int32_t * word, int32_t run_id, const int32_t * host_flag, const pi_xflag * flag, uint32_t ticket, unsigned long long limit_ticks,
                                   int32_t * timeout_flag, unsigned long long * hop_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t v = 0; bool timed_out = false, given_up = false;
    for (uint32_t it = 0; ; it++) {
        v = __hip_atomic_load(&flag->value, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((v >> 1) == ticket) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit_ticks) { timed_out = true; break; }
        // the run's host flag (llama_cancel_run / stage_abort_run / a failed launch of this very run): whoever raises it ends the wait — the run is aborted, not timed out
        if (host_flag && (it & 15) == 15 && __hip_atomic_load(host_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) { given_up = true; break; }
        __builtin_amdgcn_s_sleep(4);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (timeout_flag) *timeout_flag = timed_out ? 1 : 0;                           // (rewritten by every run: a time-out is that run's, not sticky)
    if (timed_out || given_up) v = 1;
    else if (hop_ticks) { hop_ticks[0] += t1 - flag->stamp; hop_ticks[1] += t1 - t0; hop_ticks[2] += 1; }     // flag written -> seen (one clock: stages on one device); time spent waiting; runs
    const int32_t f = host_flag ? __hip_atomic_load(host_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
    __hip_atomic_exchange(word, (run_id << 1) | (f != 0) | (int32_t) (v & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
extern "C" int pi_op_flag_publish(void * flag_dev, uint32_t ticket, const int32_t * abort_word_dev, int force_abort, void * stream) {
    hipLaunchKernelGGL(k_flag_publish, dim3(1), dim3(1), 0, (hipStream_t) stream, (pi_xflag *) flag_dev, ticket & 0x7fffffffu, abort_word_dev, force_abort);
    PI_CHECK_LAUNCH();
    return 0;
}
extern "C" int pi_op_abort_begin_wait(int32_t * word_dev, int32_t run_id, const int32_t * host_flag, const void * flag_dev, uint32_t ticket, int64_t timeout_us,
                                      int32_t * timeout_flag_dev, uint64_t * hop_ticks_dev, void * stream) {
    hipLaunchKernelGGL(k_abort_begin_wait, dim3(1), dim3(1), 0, (hipStream_t) stream, word_dev, run_id & 0x3fffffff, host_flag, (const pi_xflag *) flag_dev, ticket & 0x7fffffffu,
                       (unsigned long long) timeout_us * 100ull, timeout_flag_dev, (unsigned long long *) hop_ticks_dev);
    PI_CHECK_LAUNCH();
    return 0;
}
__global__ void k_abort_cancel(int32_t * word, int32_t run_id) {
    int32_t expect = run_id << 1;
    __hip_atomic_compare_exchange_strong(word, &expect, (run_id << 1) | 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
extern "C" int pi_op_abort_begin(int32_t * word_dev, int32_t run_id, const int32_t * host_flag, void * stream) {
    hipLaunchKernelGGL(k_abort_begin, dim3(1), dim3(1), 0, (hipStream_t) stream, word_dev, run_id & 0x3fffffff, host_flag);
    PI_CHECK_LAUNCH();
    return 0;
}
extern "C" int pi_op_abort_cancel(int32_t * word_dev, int32_t run_id, void * stream) {
    hipLaunchKernelGGL(k_abort_cancel, dim3(1), dim3(1), 0, (hipStream_t) stream, word_dev, run_id & 0x3fffffff);
    PI_CHECK_LAUNCH();
    return 0;
}

// ---- results leave the device through a kernel ---------------------------------------------------------------------------------------------------
// hipMemcpyAsync(device -> page-locked host) of a run's logits goes through a copy engine on this runtime, and that copy does not complete while ANOTHER stream's kernels
// keep the device busy: a draft step of 1.2 ms waited for the whole 11 ms target run beside it (tools_dev/wrap_beside.py, profiles/r04_d2h_copy_stall.log; the same step
// with its results written by a kernel: 1.3 ms beside the target).  Page-locked host memory is device-addressable, so the last kernel of a run writes the caller's
// buffers itself; the event recorded behind it releases them to the host.
typedef uint32_t rt_u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy_out(const float * __restrict__ logits, float * __restrict__ logits_h, size_t n_logits, const int32_t * __restrict__ ids, int32_t * __restrict__ ids_h,
                                                  const float * __restrict__ vals, float * __restrict__ vals_h, const float * __restrict__ sum, float * __restrict__ sum_h, int n_tok) {
    const size_t tid = (size_t) blockIdx.x * 256 + threadIdx.x, nthr = (size_t) gridDim.x * 256;
    if ((((uintptr_t) logits | (uintptr_t) logits_h) & 15) == 0) {
        const size_t n4 = n_logits / 4;
        for (size_t i = tid; i < n4; i += nthr) __builtin_nontemporal_store(((const rt_u32x4 *) logits)[i], (rt_u32x4 *) logits_h + i);
        for (size_t i = n4 * 4 + tid; i < n_logits; i += nthr) logits_h[i] = logits[i];
    } else for (size_t i = tid; i < n_logits; i += nthr) logits_h[i] = logits[i];
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < 8 * n_tok; i += 256) { ids_h[i] = ids[i]; vals_h[i] = vals[i]; }
        for (int i = threadIdx.x; i < n_tok; i += 256) sum_h[i] = sum[i];
    }
}
extern "C" int pi_op_copy_out(const float * logits_dev, float * logits_host, int64_t n_logits, const int32_t * ids_dev, int32_t * ids_host, const float * vals_dev, float * vals_host,
                              const float * sum_dev, float * sum_host, int n_tok, void * stream) {
    const int64_t per_wg = 256 * 4 * 4;                                       // four 16-byte stores per thread
    const unsigned grid = (unsigned) std::max<int64_t>(1, std::min<int64_t>(256, (n_logits + per_wg - 1) / per_wg));
    hipLaunchKernelGGL(k_copy_out, dim3(grid), dim3(256), 0, (hipStream_t) stream, logits_dev, logits_host, (size_t) n_logits, ids_dev, ids_host, vals_dev, vals_host, sum_dev, sum_host, n_tok);
    PI_CHECK_LAUNCH();
    return 0;
}

static thread_local bool g_mq_copies = true;      // pi_weight_set_mq_copies: whether uploads of this thread also make the tile-major copy of the matrix-core kernel
// ---- weight upload + re-layout ------------------------------------------------------------------------------------
// one thread per source block; bytes are moved into the k-major layouts of ops_common.h
__global__ void k_repack_k45(const uint8_t * __restrict__ src, uint8_t * __restrict__ dst, int64_t nblocks, int bsize, int qs_off) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    const uint8_t * s = src + i * bsize; uint8_t * d = dst + i * bsize;
    for (int b = 0; b < qs_off; b++) d[b] = s[b];
    const uint32_t * sq = (const uint32_t *) (s + qs_off); uint32_t * dq = (uint32_t *) (d + qs_off);
    // "p-half" order: dword (k, m) — bytes 4k..4k+3 of the m-th 32-byte qs group — goes to [p = m >> 1][k][e = m & 1].  The decode kernel's quad
    // lane c still reads two 16-byte vectors (p = 0, 1) that hold all four dwords of its accumulator lanes 2c, 2c+1; the MFMA kernel's lane
    // (row, p, h) reads the 64 contiguous bytes of half p and takes nibble-half h of every dword (mmq_mfma.hip)
    for (int k = 0; k < 8; k++) for (int m = 0; m < 4; m++) dq[(m >> 1) * 16 + k * 2 + (m & 1)] = sq[m * 8 + k];
}
// Q4_K, tile-major copy for the MFMA kernel (mmq_mfma.hip; k_repack_mq_q45_K): tile (row group g, block b) = PI_MQ_TILE = 2368 contiguous bytes
//   [  0,   64)  16 rows x (d, dmin) as the block's two f16
//   [ 64,  320)  64 records of 4 bytes, record (r + 16 c) = s(2c), s(2c+1), m(2c), m(2c+1) of row r: the 6-bit scales / mins of sub-blocks
//                2c, 2c+1 already unpacked (get_scale_min_k4, ggml-quants.c:1446-1453) — lane (r, c) of the kernel needs exactly these four
//   [320, 1344)  part 0: lane (r + 16 c) x 16 bytes = dwords k = 0..3 of qs group c (dword k = bytes 4k..4k+3 of the c-th 32-byte group:
//                low nibbles = sub-block 2c, high nibbles = sub-block 2c+1, accumulator lane k)
//   [1344, 2368) part 1: k = 4..7
// Rows past the end of the matrix are zero (d = 0: they add nothing).
// Q5_K: the same tile with the block's 32 qh bytes of every row after the records ([320, 832): 16 rows x 32 B; bit j of byte l = fifth bit of weight l of
// sub-block j), quants from 832: PI_MQ_TILE_Q5 = 2880 bytes.
__global__ void k_repack_mq_q45_K(const uint8_t * __restrict__ src, uint8_t * __restrict__ dst, int64_t rows, int64_t nb, int q5) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;                  // (row group, block, row in group)
    const int64_t n_rg = (rows + 15) / 16;
    if (i >= n_rg * nb * 16) return;
    const int r = (int) (i & 15); const int64_t gb = i >> 4, g = gb / nb, b = gb % nb, row = g * 16 + r;
    const int tile_sz = q5 ? PI_MQ_TILE_Q5 : PI_MQ_TILE, qoff = q5 ? 832 : 320;
    uint8_t * tile = dst + gb * tile_sz;
    uint32_t * dd = (uint32_t *) tile, * rec = (uint32_t *) (tile + 64), * qh = (uint32_t *) (tile + 320), * q0 = (uint32_t *) (tile + qoff), * q1 = (uint32_t *) (tile + qoff + 1024);
    if (row >= rows) {
        dd[r] = 0;
        if (q5) for (int k = 0; k < 8; k++) qh[r * 8 + k] = 0;
        for (int c = 0; c < 4; c++) { rec[r + 16 * c] = 0; for (int k = 0; k < 4; k++) { q0[(r + 16 * c) * 4 + k] = 0; q1[(r + 16 * c) * 4 + k] = 0; } }
        return;
    }
    const uint8_t * blk = src + (row * nb + b) * (q5 ? 176 : 144);
    const uint32_t * s = (const uint32_t *) blk;
    const uint8_t * sc = blk + 4;
    dd[r] = s[0];
    uint32_t scl[8], mn[8];
    for (int j = 0; j < 8; j++) {
        if (j < 4) { scl[j] = sc[j] & 63; mn[j] = sc[j + 4] & 63; }
        else { scl[j] = (sc[j + 4] & 0xF) | ((sc[j - 4] >> 6) << 4); mn[j] = (sc[j + 4] >> 4) | ((sc[j] >> 6) << 4); }
    }
    if (q5) for (int k = 0; k < 8; k++) qh[r * 8 + k] = s[4 + k];
    const uint32_t * sq = s + (q5 ? 12 : 4);
    for (int c = 0; c < 4; c++) {
        rec[r + 16 * c] = scl[2 * c] | scl[2 * c + 1] << 8 | mn[2 * c] << 16 | mn[2 * c + 1] << 24;
        for (int k = 0; k < 4; k++) { q0[(r + 16 * c) * 4 + k] = sq[c * 8 + k]; q1[(r + 16 * c) * 4 + k] = sq[c * 8 + 4 + k]; }
    }
}
// Q6_K, tile-major copy for the MFMA kernel: tile (row group, block) = PI_MQ_TILE_Q6 = 3648 bytes.  K-slice c = (half n, hh) of a row covers the two
// 32-groups that share ql bytes: g = hh (low nibbles of ql[64n + 32hh ..]) and g = hh + 2 (high nibbles); qh[32n ..] carries both groups' top bits.
//   [   0,   64)  16 rows x d (f16 in a dword)
//   [  64,  576)  64 records of 8 bytes, record (r + 16 c): the four int8 scales of the slice — sub-blocks 8n+2hh, 8n+2hh+1 (accumulator lanes k < 4, k >= 4 of the
//                 low-nibble group) and 8n+2hh+4, 8n+2hh+5 (high-nibble group) — split so that both parts are exact as f16 weights: bytes 0..3 = sc & 15,
//                 bytes 4..7 = (sc >> 4) + 8   (sc = 16 * (sc >> 4) + (sc & 15), arithmetic shift)
//   [ 576, 1600)  16 rows x 2 halves x 32 bytes of qh
//   [1600, 3648)  part 0 / part 1: lane (r + 16 c) x 16 bytes = dwords k = 0..3 / 4..7 of ql[64n + 32hh + 4k ..]
__global__ void k_repack_mq_q6_K(const uint8_t * __restrict__ src, uint8_t * __restrict__ dst, int64_t rows, int64_t nb) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;                  // (row group, block, row in group)
    const int64_t n_rg = (rows + 15) / 16;
    if (i >= n_rg * nb * 16) return;
    const int r = (int) (i & 15); const int64_t gb = i >> 4, g = gb / nb, b = gb % nb, row = g * 16 + r;
    uint8_t * tile = dst + gb * PI_MQ_TILE_Q6;
    uint32_t * dd = (uint32_t *) tile, * rec = (uint32_t *) (tile + 64), * qh = (uint32_t *) (tile + 576), * q0 = (uint32_t *) (tile + 1600), * q1 = (uint32_t *) (tile + 2624);
    if (row >= rows) {
        dd[r] = 0;
        for (int k = 0; k < 16; k++) qh[r * 16 + k] = 0;
        for (int c = 0; c < 4; c++) { rec[(r + 16 * c) * 2] = 0; rec[(r + 16 * c) * 2 + 1] = 0x08080808u; for (int k = 0; k < 4; k++) { q0[(r + 16 * c) * 4 + k] = 0; q1[(r + 16 * c) * 4 + k] = 0; } }
        return;
    }
    const uint8_t * blk = src + (row * nb + b) * 210;                                    // 2-byte aligned only: assemble dwords from bytes
    auto ld = [&](int off) { return (uint32_t) blk[off] | (uint32_t) blk[off + 1] << 8 | (uint32_t) blk[off + 2] << 16 | (uint32_t) blk[off + 3] << 24; };
    dd[r] = (uint32_t) blk[208] | (uint32_t) blk[209] << 8;
    for (int k = 0; k < 16; k++) qh[r * 16 + k] = ld(128 + 4 * k);
    const int8_t * sc = (const int8_t *) (blk + 192);
    for (int c = 0; c < 4; c++) {
        const int n = c >> 1, hh = c & 1;
        const int idx[4] = { 8 * n + 2 * hh, 8 * n + 2 * hh + 1, 8 * n + 2 * hh + 4, 8 * n + 2 * hh + 5 };
        uint32_t lo = 0, hi = 0;
        for (int j = 0; j < 4; j++) { const int v = sc[idx[j]]; lo |= (uint32_t) (v & 15) << (8 * j); hi |= (uint32_t) ((v >> 4) + 8) << (8 * j); }
        rec[(r + 16 * c) * 2] = lo; rec[(r + 16 * c) * 2 + 1] = hi;
        for (int k = 0; k < 4; k++) { q0[(r + 16 * c) * 4 + k] = ld(64 * n + 32 * hh + 4 * k); q1[(r + 16 * c) * 4 + k] = ld(64 * n + 32 * hh + 16 + 4 * k); }
    }
}
// Q3_K, tile-major copy for the MFMA kernel: tile (row group, block) = PI_MQ_TILE_Q3 = 1856 bytes (block: hmask 32 | qs 64 | scales 12 | d, ggml-quants.h).
//   [   0,   64)  16 rows x d (f16 in a dword)
//   [  64,  320)  64 records of 4 bytes, record (r + 16 c): the four 6-bit scales of K-slice c as stored (value + 32, decoded like ggml-quants.c:1731-1736): groups 2c and
//                 2c+1, their first / second 16 weights (sub-blocks 4c .. 4c+3)
//   [ 320,  832)  16 rows x 32 bytes of hmask (bit g of byte l = high bit of weight l of 32-group g)
//   [ 832, 1856)  part 0 / part 1: (row r, half h) x 16 bytes = dwords k = 0..3 / 4..7 of qs[32h + 4k ..] (a byte holds groups 4h .. 4h+3 in its four bit pairs)
__global__ void k_repack_mq_q3_K(const uint8_t * __restrict__ src, uint8_t * __restrict__ dst, int64_t rows, int64_t nb) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;                  // (row group, block, row in group)
    const int64_t n_rg = (rows + 15) / 16;
    if (i >= n_rg * nb * 16) return;
    const int r = (int) (i & 15); const int64_t gb = i >> 4, g = gb / nb, b = gb % nb, row = g * 16 + r;
    uint8_t * tile = dst + gb * PI_MQ_TILE_Q3;
    uint32_t * dd = (uint32_t *) tile, * rec = (uint32_t *) (tile + 64), * hm = (uint32_t *) (tile + 320), * q0 = (uint32_t *) (tile + 832), * q1 = (uint32_t *) (tile + 1344);
    if (row >= rows) {
        dd[r] = 0;
        for (int k = 0; k < 8; k++) hm[r * 8 + k] = 0;
        for (int c = 0; c < 4; c++) rec[r + 16 * c] = 0x20202020u;
        for (int h = 0; h < 2; h++) for (int k = 0; k < 4; k++) { q0[(r * 2 + h) * 4 + k] = 0; q1[(r * 2 + h) * 4 + k] = 0; }
        return;
    }
    const uint8_t * blk = src + (row * nb + b) * 110;                                    // 2-byte aligned only: assemble dwords from bytes
    auto ld = [&](int off) { return (uint32_t) blk[off] | (uint32_t) blk[off + 1] << 8 | (uint32_t) blk[off + 2] << 16 | (uint32_t) blk[off + 3] << 24; };
    dd[r] = (uint32_t) blk[108] | (uint32_t) blk[109] << 8;
    for (int k = 0; k < 8; k++) hm[r * 8 + k] = ld(4 * k);
    const uint32_t a0 = ld(96), a1 = ld(100), a2 = ld(104), kmask1 = 0x03030303u, kmask2 = 0x0f0f0f0fu;
    const uint32_t aux[4] = { (a0 & kmask2) | (((a2 >> 0) & kmask1) << 4), (a1 & kmask2) | (((a2 >> 2) & kmask1) << 4),
                              ((a0 >> 4) & kmask2) | (((a2 >> 4) & kmask1) << 4), ((a1 >> 4) & kmask2) | (((a2 >> 6) & kmask1) << 4) };
    for (int c = 0; c < 4; c++) rec[r + 16 * c] = aux[c];                                 // scales 4c .. 4c+3 = sub-blocks (2 (2c) + 0, + 1), (2 (2c+1) + 0, + 1)
    for (int h = 0; h < 2; h++) for (int k = 0; k < 4; k++) { q0[(r * 2 + h) * 4 + k] = ld(32 + 32 * h + 4 * k); q1[(r * 2 + h) * 4 + k] = ld(32 + 32 * h + 16 + 4 * k); }
}
__global__ void k_repack_q6_K(const uint8_t * __restrict__ src, uint8_t * a, uint8_t * sc, uint16_t * d, int64_t nblocks) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    const uint8_t * b = src + i * 210;                       // 2-byte aligned only: assemble dwords from bytes
    auto ld = [&](int off) { return (uint32_t) b[off] | (uint32_t) b[off + 1] << 8 | (uint32_t) b[off + 2] << 16 | (uint32_t) b[off + 3] << 24; };
    uint32_t * o = (uint32_t *) (a + i * 192);                // ql: [2 halves n][8 lanes k][2 dwords x] (p-half order, like Q4_K), then qh: [8 lanes][2 dwords n]
    for (int k = 0; k < 8; k++) {
        for (int n = 0; n < 2; n++) { o[n * 16 + k * 2 + 0] = ld(64 * n + 4 * k); o[n * 16 + k * 2 + 1] = ld(64 * n + 32 + 4 * k); }
        o[32 + k * 2 + 0] = ld(128 + 4 * k); o[32 + k * 2 + 1] = ld(160 + 4 * k);
    }
    for (int k = 0; k < 16; k++) sc[i * 16 + k] = b[192 + k];
    d[i] = (uint16_t) (b[208] | (b[209] << 8));
}
// Q3_K (110 bytes: hmask 32 | qs 64 | scales 12 | d): plane a = 96 bytes per block, k-major for quad lane c (elements 8c..8c+7 of every
// 32-group): [c][half j][dword h] of qs at 16c + 8j + 4h, then hmask dwords at 64 + 8c + 4h; the sixteen 6-bit scales are decoded
// once (ggml-quants.c:1731-1736) and stored as int8 (value - 32) in plane sc; d in plane d
__global__ void k_repack_q3_K(const uint8_t * __restrict__ src, uint8_t * a, int8_t * sc, uint16_t * d, int64_t nblocks) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    const uint8_t * b = src + i * 110;                       // 2-byte aligned only: assemble dwords from bytes
    auto ld = [&](int off) { return (uint32_t) b[off] | (uint32_t) b[off + 1] << 8 | (uint32_t) b[off + 2] << 16 | (uint32_t) b[off + 3] << 24; };
    uint32_t * o = (uint32_t *) (a + i * 96);
    for (int c = 0; c < 4; c++) {
        for (int j = 0; j < 2; j++) for (int h = 0; h < 2; h++) o[c * 4 + j * 2 + h] = ld(32 + 32 * j + 8 * c + 4 * h);      // qs[32j + 8c + 4h ..]
        for (int h = 0; h < 2; h++) o[16 + c * 2 + h] = ld(8 * c + 4 * h);                                                    // hmask[8c + 4h ..]
    }
    const uint32_t a0 = ld(96), a1 = ld(100), a2 = ld(104), kmask1 = 0x03030303u, kmask2 = 0x0f0f0f0fu;
    const uint32_t aux[4] = { (a0 & kmask2) | (((a2 >> 0) & kmask1) << 4), (a1 & kmask2) | (((a2 >> 2) & kmask1) << 4),
                              ((a0 >> 4) & kmask2) | (((a2 >> 4) & kmask1) << 4), ((a1 >> 4) & kmask2) | (((a2 >> 6) & kmask1) << 4) };
    for (int k = 0; k < 16; k++) sc[i * 16 + k] = (int8_t) ((int) ((aux[k >> 2] >> (8 * (k & 3))) & 0xff) - 32);
    d[i] = (uint16_t) (b[108] | (b[109] << 8));
}
// Q2_K (84 bytes: scales 16 | qs 64 | d | dmin): plane a = 64 bytes per block, qs k-major like Q3_K ([c][half j][dword h] at 16c + 8j + 4h);
// plane sc = the sixteen (scale | min << 4) bytes; plane d = d | dmin << 16
__global__ void k_repack_q2_K(const uint8_t * __restrict__ src, uint8_t * a, uint8_t * sc, uint32_t * d, int64_t nblocks) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    const uint8_t * b = src + i * 84;                        // 4-byte aligned (84 = 21 dwords)
    const uint32_t * bw = (const uint32_t *) b;
    uint32_t * o = (uint32_t *) (a + i * 64);
    for (int c = 0; c < 4; c++)
        for (int j = 0; j < 2; j++) for (int h = 0; h < 2; h++) o[c * 4 + j * 2 + h] = bw[4 + 8 * j + 2 * c + h];             // qs[32j + 8c + 4h ..]
    for (int k = 0; k < 4; k++) ((uint32_t *) (sc + i * 16))[k] = bw[k];
    d[i] = bw[20];
}
__global__ void k_repack_q4_0(const uint8_t * __restrict__ src, uint32_t * qs, uint16_t * d, int64_t rows, int64_t nb, int64_t ng) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nb) return;
    const int64_t row = i / nb, blk = i % nb, g = blk >> 2, bi = blk & 3;
    const uint8_t * b = src + i * 18;
    d[(row * ng + g) * 4 + bi] = (uint16_t) (b[0] | (b[1] << 8));
    for (int c = 0; c < 4; c++)
        qs[((row * ng + g) * 4 + c) * 4 + bi] = (uint32_t) b[2 + 4 * c] | (uint32_t) b[3 + 4 * c] << 8 | (uint32_t) b[4 + 4 * c] << 16 | (uint32_t) b[5 + 4 * c] << 24;
}
__global__ void k_repack_q8_0(const uint8_t * __restrict__ src, uint32_t * qs, uint16_t * d, int64_t rows, int64_t nb, int64_t ng) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nb) return;
    const int64_t row = i / nb, blk = i % nb, g = blk >> 2, bi = blk & 3;
    const uint8_t * b = src + i * 34;
    d[(row * ng + g) * 4 + bi] = (uint16_t) (b[0] | (b[1] << 8));
    for (int k = 0; k < 8; k++)
        qs[((row * ng + g) * 8 + k) * 4 + bi] = (uint32_t) b[2 + 4 * k] | (uint32_t) b[3 + 4 * k] << 8 | (uint32_t) b[4 + 4 * k] << 16 | (uint32_t) b[5 + 4 * k] << 24;
}

// Q4_1 / Q5_0 / Q5_1: nibbles exactly like Q4_0; the per-block scalars in planes of their own (hdr = bytes before qh / qs: 4 / 2 / 4, qh present for Q5)
__global__ void k_repack_legacy(const uint8_t * __restrict__ src, int type, uint32_t * qs, uint16_t * dm, uint32_t * qh, int64_t rows, int64_t nb, int64_t ng) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nb) return;
    const int64_t row = i / nb, blk = i % nb, g = blk >> 2, bi = blk & 3;
    const int tsz = type == PI_Q4_1 ? 20 : type == PI_Q5_0 ? 22 : 24;
    const uint8_t * b = src + i * tsz;
    const bool has_m = type != PI_Q5_0, has_h = type != PI_Q4_1;
    dm[(row * ng + g) * 8 + bi] = (uint16_t) (b[0] | (b[1] << 8));
    dm[(row * ng + g) * 8 + 4 + bi] = has_m ? (uint16_t) (b[2] | (b[3] << 8)) : (uint16_t) 0;
    const uint8_t * h = b + (has_m ? 4 : 2);
    qh[(row * ng + g) * 4 + bi] = has_h ? ((uint32_t) h[0] | (uint32_t) h[1] << 8 | (uint32_t) h[2] << 16 | (uint32_t) h[3] << 24) : 0u;
    const uint8_t * q = h + (has_h ? 4 : 0);
    for (int c = 0; c < 4; c++)
        qs[((row * ng + g) * 4 + c) * 4 + bi] = (uint32_t) q[4 * c] | (uint32_t) q[4 * c + 1] << 8 | (uint32_t) q[4 * c + 2] << 16 | (uint32_t) q[4 * c + 3] << 24;
}

// the tile-major copy the matrix-core kernel reads (mmq_mfma.hip), from the tensor's GGUF bytes in `staging` (device); enqueued on `stream`.  Without HBM for it the weight
// simply has none: mul_mat then takes the (bit-identical) VALU kernel at every width — a model that fits without the copies must still load
static void make_mq_copy(pi_weight * w, const uint8_t * staging, hipStream_t stream) {
    static const bool mq_on = !(getenv("PI_MMQ_COPIES") && atoi(getenv("PI_MMQ_COPIES")) == 0);
    const int type = w->type;
    if (!mq_on || w->mq || !(type == PI_Q3_K || type == PI_Q4_K || type == PI_Q5_K || type == PI_Q6_K)) return;
    const int threads = 256;
    const int64_t n_rg = (w->ne1 + 15) / 16;
    const size_t tile_sz = type == PI_Q6_K ? PI_MQ_TILE_Q6 : type == PI_Q5_K ? PI_MQ_TILE_Q5 : type == PI_Q3_K ? PI_MQ_TILE_Q3 : PI_MQ_TILE;
    const size_t mq_bytes = (size_t) n_rg * w->nb * tile_sz + 512;
    if (hipMalloc(&w->mq_base, mq_bytes) != hipSuccess) {
        (void) hipGetLastError();
        w->mq_base = nullptr;
        static bool said = false;
        if (!said) { said = true; fprintf(stderr, "pi_weight_upload: no HBM left for the tile-major copies of the matrix-core kernel (%zu bytes asked): batches fall back to the VALU kernel\n", mq_bytes); }
        return;
    }
    w->mq = (const uint8_t *) w->mq_base; w->mq_nbytes = mq_bytes;
    const dim3 mgrid((unsigned) ((n_rg * w->nb * 16 + threads - 1) / threads));
    if (type == PI_Q6_K) hipLaunchKernelGGL(k_repack_mq_q6_K, mgrid, dim3(threads), 0, stream, staging, (uint8_t *) w->mq_base, w->ne1, w->nb);
    else if (type == PI_Q3_K) hipLaunchKernelGGL(k_repack_mq_q3_K, mgrid, dim3(threads), 0, stream, staging, (uint8_t *) w->mq_base, w->ne1, w->nb);
    else hipLaunchKernelGGL(k_repack_mq_q45_K, mgrid, dim3(threads), 0, stream, staging, (uint8_t *) w->mq_base, w->ne1, w->nb, type == PI_Q5_K ? 1 : 0);
}

extern "C" pi_weight * pi_weight_upload(int type, int64_t ne0, int64_t ne1, const void * host, void * stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    const int blck = pi_type_blck(type), tsz = pi_type_size(type);
    if (!tsz || type == PI_Q8_K || type == PI_Q8_1) { pi_set_error("pi_weight_upload: unsupported tensor type %d", type); return nullptr; }
    if (ne0 % blck) { pi_set_error("pi_weight_upload: ne0=%lld not a multiple of the block size %d", (long long) ne0, blck); return nullptr; }
    pi_weight * w = new pi_weight();
    w->type = type; w->ne0 = ne0; w->ne1 = ne1; w->nb = ne0 / blck; w->ng = (w->nb + 3) / 4;
    const int64_t nblocks = w->nb * ne1;
    w->nbytes = (size_t) nblocks * tsz;
    const bool legacy = pi_type_is_legacy(type);
    const bool small = type == PI_Q4_0 || type == PI_Q8_0 || legacy;
    const size_t alloc = small ? (size_t) ne1 * w->ng * 4 * (type == PI_Q8_0 ? 32 : 16) + (size_t) ne1 * w->ng * (legacy ? 32 : 8) + 1024
                       : type == PI_Q3_K ? (size_t) nblocks * (96 + 16 + 2) + 512 : w->nbytes + 512;        // (Q2_K: 64 + 16 + 4 = its 84 bytes)
    if (hipMalloc(&w->base, alloc) != hipSuccess) { pi_set_error("pi_weight_upload: hipMalloc(%zu) failed", alloc); delete w; return nullptr; }
    uint8_t * base = (uint8_t *) w->base;
    w->p0 = base; w->p1 = w->p2 = w->p3 = nullptr; w->mq = nullptr; w->mq_base = nullptr;
    auto fail = [&](const char * what) { pi_set_error("pi_weight_upload: %s", what); (void) hipFree(w->base); delete w; return (pi_weight *) nullptr; };
    if (type == PI_F32 || type == PI_F16) {
        if (hipMemcpyAsync(base, host, w->nbytes, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return fail("H2D copy failed");
        return w;
    }
    uint8_t * staging = nullptr;
    if (hipMalloc((void **) &staging, w->nbytes) != hipSuccess) return fail("staging hipMalloc failed");
    (void) hipMemcpyAsync(staging, host, w->nbytes, hipMemcpyHostToDevice, stream);
    const int threads = 256;
    const unsigned grid = (unsigned) ((nblocks + threads - 1) / threads);
    if (type == PI_Q4_K) hipLaunchKernelGGL(k_repack_k45, dim3(grid), dim3(threads), 0, stream, staging, base, nblocks, 144, 16);
    else if (type == PI_Q5_K) hipLaunchKernelGGL(k_repack_k45, dim3(grid), dim3(threads), 0, stream, staging, base, nblocks, 176, 48);
    else if (type == PI_Q2_K) {
        w->p1 = base + nblocks * 64; w->p2 = w->p1 + nblocks * 16;
        hipLaunchKernelGGL(k_repack_q2_K, dim3(grid), dim3(threads), 0, stream, staging, base, (uint8_t *) w->p1, (uint32_t *) w->p2, nblocks);
    } else if (type == PI_Q3_K) {
        w->p1 = base + nblocks * 96; w->p2 = w->p1 + nblocks * 16;
        hipLaunchKernelGGL(k_repack_q3_K, dim3(grid), dim3(threads), 0, stream, staging, base, (int8_t *) w->p1, (uint16_t *) w->p2, nblocks);
    } else if (type == PI_Q6_K) {
        w->p1 = base + nblocks * 192; w->p2 = w->p1 + nblocks * 16;
        hipLaunchKernelGGL(k_repack_q6_K, dim3(grid), dim3(threads), 0, stream, staging, base, (uint8_t *) w->p1, (uint16_t *) w->p2, nblocks);
    } else {
        const size_t qbytes = (size_t) ne1 * w->ng * 4 * (type == PI_Q8_0 ? 32 : 16);
        w->p1 = base + ((qbytes + 255) & ~(size_t) 255);
        (void) hipMemsetAsync(base, 0, alloc, stream);                    // zero blocks pad the last group of every row
        if (legacy) {
            w->p2 = w->p1 + (((size_t) ne1 * w->ng * 16 + 255) & ~(size_t) 255);
            hipLaunchKernelGGL(k_repack_legacy, dim3(grid), dim3(threads), 0, stream, staging, type, (uint32_t *) base, (uint16_t *) w->p1, (uint32_t *) w->p2, ne1, w->nb, w->ng);
        } else if (type == PI_Q4_0) hipLaunchKernelGGL(k_repack_q4_0, dim3(grid), dim3(threads), 0, stream, staging, (uint32_t *) base, (uint16_t *) w->p1, ne1, w->nb, w->ng);
        else                 hipLaunchKernelGGL(k_repack_q8_0, dim3(grid), dim3(threads), 0, stream, staging, (uint32_t *) base, (uint16_t *) w->p1, ne1, w->nb, w->ng);
    }
    w->mq = nullptr; w->mq_base = nullptr; w->mq_nbytes = 0;
    if (g_mq_copies) make_mq_copy(w, staging, stream);
    const hipError_t e = hipStreamSynchronize(stream);
    (void) hipFree(staging);
    if (e != hipSuccess) { if (w->mq_base) (void) hipFree(w->mq_base); return fail(hipGetErrorString(e)); }
    return w;
}

// a weight uploaded without its tile-major copy (pi_weight_set_mq_copies(0): no pass of its first context was wide enough) gets it now, from the same GGUF bytes
extern "C" int pi_weight_ensure_mq_copy(pi_weight * w, const void * host, void * stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    if (!w || !host) { pi_set_error("pi_weight_ensure_mq_copy: null argument"); return -1; }
    if (w->mq || !(w->type == PI_Q4_K || w->type == PI_Q5_K || w->type == PI_Q6_K)) return 0;
    uint8_t * staging = nullptr;
    if (hipMalloc((void **) &staging, w->nbytes) != hipSuccess) { (void) hipGetLastError(); pi_set_error("pi_weight_ensure_mq_copy: staging hipMalloc(%zu) failed", w->nbytes); return -1; }
    hipError_t e = hipMemcpyAsync(staging, host, w->nbytes, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) { make_mq_copy(w, staging, stream); e = hipStreamSynchronize(stream); }
    (void) hipFree(staging);
    if (e != hipSuccess) { if (w->mq_base) { (void) hipFree(w->mq_base); w->mq_base = nullptr; w->mq = nullptr; w->mq_nbytes = 0; } pi_set_error("pi_weight_ensure_mq_copy: %s", hipGetErrorString(e)); return -1; }
    return 0;
}

extern "C" void pi_weight_free(pi_weight * w) { if (w) { (void) hipFree(w->base); if (w->mq_base) (void) hipFree(w->mq_base); delete w; } }
extern "C" size_t pi_weight_nbytes(const pi_weight * w) { return w->nbytes; }
extern "C" size_t pi_weight_copy_nbytes(const pi_weight * w) { return w ? w->mq_nbytes : 0; }
extern "C" void pi_weight_set_mq_copies(int on) { g_mq_copies = on != 0; }

// ---- Infinity-Cache warm-up: read [p, p + bytes) once so that the bytes sit in the 256 MB memory-side cache when the mat-vec that owns them starts.
// Runs on a side stream next to the latency-bound launches of a decode layer (norm / quantise / attention keep one to 64 CUs busy and leave HBM idle).
__global__ void __launch_bounds__(256) k_prefetch(const uint4 * __restrict__ p, size_t n16, uint32_t * __restrict__ sink) {
    const size_t stride = (size_t) gridDim.x * 256;
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u].x ^ v[u].w;
    }
    for (; i < n16; i += stride) { const uint4 v = p[i]; acc ^= v.x ^ v.w; }
    if (acc == 0x9e3779b9u && sink) *sink = acc;                                    // (keeps the loads alive; practically never taken)
}
extern "C" int pi_op_prefetch(const void * p, size_t bytes, int workgroups, void * stream) {
    if (!p || bytes < 16) return 0;
    static uint32_t * sink = nullptr;
    if (!sink) PI_CHECK_HIP(hipMalloc((void **) &sink, 64));
    const size_t n16 = bytes / 16;
    size_t wg = (n16 + 256 * 8 - 1) / (256 * 8);
    if (workgroups > 0 && wg > (size_t) workgroups) wg = (size_t) workgroups;
    hipLaunchKernelGGL(k_prefetch, dim3((unsigned) wg), dim3(256), 0, (hipStream_t) stream, (const uint4 *) p, n16, sink);
    PI_CHECK_LAUNCH();
    return 0;
}
extern "C" const void * pi_weight_data(const pi_weight * w) { return w->base; }
extern "C" int pi_weight_type(const pi_weight * w) { return w->type; }

extern "C" int pi_vec_dot_type(int t) {
    switch (t) {
        case PI_Q4_0: case PI_Q5_0: case PI_Q8_0: return PI_Q8_0;
        case PI_Q4_1: case PI_Q5_1: return PI_Q8_1;                   // type_traits[].vec_dot_type (ggml.c:436-470)
        case PI_Q2_K: case PI_Q3_K: case PI_Q4_K: case PI_Q5_K: case PI_Q6_K: return PI_Q8_K;
        case PI_F16: return PI_F16; case PI_F32: return PI_F32;
    }
    return -1;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t) 255; }

// Device layout of quantised activations (k-major, mirrors the weight layouts):
//   Q8_K: qs [n_tok][K/256][8 lanes k][8 dwords = (j, lo/hi)]   byte of value e: k*32 + ((e>>6)*2 + ((e>>5)&1))*4 + (e&3), k = (e&31)>>2
//         d  [n_tok][K/256] f32       bsums [n_tok][K/256][16] int16
//   Q8_0: qs [n_tok][groups of 4 blocks][8 lanes k][4 blocks b] dwords (zero padded)     d [n_tok][groups][4] f16
//   Q8_1: qs as Q8_0     d [n_tok][groups][4] f32     s [n_tok][groups][4] f32 (= d * sum of the block's quants, the third plane)
extern "C" void pi_q8_layout(int vdt, int64_t k, int n_tok, size_t * off_qs, size_t * off_d, size_t * off_bsums) {
    size_t qs, d;
    if (vdt == PI_Q8_K) { qs = align256((size_t) n_tok * k); d = align256((size_t) n_tok * (k / 256) * 4); }
    else { const size_t ng = (size_t) (k / 32 + 3) / 4; qs = align256((size_t) n_tok * ng * 128); d = align256((size_t) n_tok * ng * (vdt == PI_Q8_1 ? 16 : 8)); }
    if (off_qs) *off_qs = 0;
    if (off_d) *off_d = qs;
    if (off_bsums) *off_bsums = qs + d;
}

// Q8_K only — operand planes of the multi-token MFMA kernel (mmq_mfma.hip), written by the quantisers beside the planes above when
// n_tok >= PI_MMQ_MIN_NT, in chunks of 16 tokens:
//   frag_a, frag_b [chunk][K/256][8 k][64 lanes][8 x f16]   the int8 values as f16, already in the B-operand lane order of
//                  v_mfma_f32_16x16x32_f16: lane = (token % 16) + 16 g holds the eight values (sub-block j, bytes 4k..4k+3) of the two
//                  sub-blocks of K-slice g, in the order (i0, i2, i1, i3).  frag_a: slice g = (p, h) = sub-blocks {4p+h, 4p+2+h}
//                  (Q4_K / Q5_K nibble pairing), frag_b: slice g = (n, hh) = {4n+2hh, 4n+2hh+1} (Q6_K pairing)
//   aux            [chunk][K/256][1 KB]: d of the 16 tokens (16 floats), then for c = 0..3 the 16 tokens' float2
//                  { bsums[4c]+bsums[4c+1], bsums[4c+2]+bsums[4c+3] } (the int16 pair sums of _mm_hadd_epi16, as floats) at 64 + 128 c; rest unused
extern "C" void pi_q8_frag_layout(int64_t k, int n_tok, size_t * off_fa, size_t * off_fb, size_t * off_aux) {
    size_t oq, od, ob;
    pi_q8_layout(PI_Q8_K, k, n_tok, &oq, &od, &ob);
    const size_t chunks = (size_t) (n_tok + 15) / 16, nb = (size_t) (k / 256);
    const size_t fa = ob + align256((size_t) n_tok * (k / 16) * 2), frag = chunks * nb * 8 * 1024;
    if (off_fa) *off_fa = fa;
    if (off_fb) *off_fb = fa;                                   // (one pairing only: every MFMA kernel so far pairs sub-blocks 2c, 2c+1)
    if (off_aux) *off_aux = fa + frag;
}

extern "C" size_t pi_q8_nbytes(int vdt, int64_t k, int n_tok) {
    size_t oq, od, ob;
    pi_q8_layout(vdt, k, n_tok, &oq, &od, &ob);
    if (vdt == PI_Q8_1) return ob + (ob - od);                   // the s plane has the size of the d plane
    if (vdt != PI_Q8_K) return ob;
    size_t oa;
    pi_q8_frag_layout(k, n_tok, nullptr, nullptr, &oa);
    return oa + (size_t) ((n_tok + 15) / 16) * (size_t) (k / 256) * 1024;
}

// =================================================================================================================
// launch accounting (kernel_timing.h): per-instance launch counts + algorithmic bytes, sampled durations of eager launches
// =================================================================================================================
#include "kernel_timing.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
namespace {
// us / sbytes: every timed launch of the instance, kept one by one — launches that found their run aborted return on entry (a few microseconds whatever the shape) and must not
// be averaged with the launches that did the work (kt_rows drops the samples below a third of the instance's median and reports how many)
struct kt_stat { uint64_t launches = 0; double bytes = 0.0; std::vector<float> us; std::vector<double> sbytes; };
std::mutex g_kt_mu;
bool g_kt_on = false; int g_kt_every = 8;
std::map<std::tuple<int, int, int, uint64_t>, kt_stat> g_kt_tab;       // (family, type, tokens, launch shape) -> eager launches
uint64_t kt_shape(int fam, double bytes) { return fam <= PI_KT_MMQ ? (uint64_t) bytes : 0; }   // mat launches: one row per distinct launch (its algorithmic bytes name it: wo, q+k, up+gate, ...)
std::map<std::tuple<int, int, int, uint64_t>, kt_stat> g_kt_life;      // ... since the process started, never cleared (what a rocprofv3 --pmc pass of the same process sees)
std::vector<pi_kt_sample> g_kt_pending;                      // timed launches whose events have not been read yet
std::vector<hipEvent_t> g_kt_pool;
thread_local pi_kt_sink * g_kt_sink = nullptr;
int kt_nt(int nt) { return nt < 1 ? 1 : nt < PI_KT_MAXNT ? nt : PI_KT_MAXNT - 1; }
hipEvent_t kt_event() {
    if (!g_kt_pool.empty()) { hipEvent_t e = g_kt_pool.back(); g_kt_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void) hipEventCreate(&e); return e;
}
void kt_fold_pending() {                                     // (caller holds g_kt_mu; the device has been synchronised)
    for (auto & s : g_kt_pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { kt_stat & st = g_kt_tab[std::make_tuple(s.fam, s.type, s.nt, kt_shape(s.fam, s.bytes))]; st.us.push_back(1e3f * ms); st.sbytes.push_back(s.bytes); }
        g_kt_pool.push_back(s.a); g_kt_pool.push_back(s.b);
    }
    g_kt_pending.clear();
}
}
void pi_kt_set_sink(pi_kt_sink * sink) { g_kt_sink = sink; }
bool pi_kt_enabled() { return g_kt_on; }
bool pi_kt_begin(int fam, int type, int nt, double bytes, hipEvent_t & ea, hipEvent_t & eb) {
    const int ntc = kt_nt(nt);
    if (pi_kt_sink * sink = g_kt_sink) {                     // being captured: the graph's own list (the engine multiplies by its replays); nothing is timed
        for (auto & c : sink->launches) if (c.fam == fam && c.type == type && c.nt == ntc && kt_shape(fam, c.bytes) == kt_shape(fam, bytes)) { c.count++; c.bytes_sum += bytes; return false; }
        sink->launches.push_back(pi_kt_count{ fam, type, ntc, 1, bytes, bytes });
        return false;
    }
    std::lock_guard<std::mutex> lk(g_kt_mu);
    { kt_stat & life = g_kt_life[std::make_tuple(fam, type, ntc, kt_shape(fam, bytes))]; life.launches++; life.bytes += bytes; }      // every eager launch of the process (PMC passes: no graphs)
    if (!g_kt_on) return false;                              // (eager launches outside a timing window are not part of its table)
    kt_stat & st = g_kt_tab[std::make_tuple(fam, type, ntc, kt_shape(fam, bytes))];
    const bool sample = (st.launches % (uint64_t) g_kt_every) == 0;
    st.launches++; st.bytes += bytes;
    if (sample) { ea = kt_event(); eb = kt_event(); }
    return sample;
}
void pi_kt_end(int fam, int type, int nt, double bytes, hipEvent_t ea, hipEvent_t eb) {
    std::lock_guard<std::mutex> lk(g_kt_mu);
    g_kt_pending.push_back(pi_kt_sample{ ea, eb, fam, type, kt_nt(nt), bytes });
}
extern "C" void pi_kernel_timing(int on, int sample_every) {
    std::lock_guard<std::mutex> lk(g_kt_mu);
    g_kt_on = on != 0; g_kt_every = sample_every > 0 ? sample_every : 1;
    for (auto & s : g_kt_pending) { g_kt_pool.push_back(s.a); g_kt_pool.push_back(s.b); }
    g_kt_pending.clear();
    g_kt_tab.clear();
}
// one row per instance launched since pi_kernel_timing(1, N) — or, lifetime != 0, per instance launched eagerly since the process started (no durations): call after a
// device sync.  Returns the number of rows (<= cap)
static int kt_rows(std::map<std::tuple<int, int, int, uint64_t>, kt_stat> & tab, struct pi_kt_row * out, int cap);
extern "C" int pi_kernel_timing_rows(struct pi_kt_row * out, int cap) {
    std::lock_guard<std::mutex> lk(g_kt_mu);
    kt_fold_pending();
    return kt_rows(g_kt_tab, out, cap);
}
extern "C" int pi_kernel_lifetime_rows(struct pi_kt_row * out, int cap) {
    std::lock_guard<std::mutex> lk(g_kt_mu);
    return kt_rows(g_kt_life, out, cap);
}
static int kt_rows(std::map<std::tuple<int, int, int, uint64_t>, kt_stat> & tab, struct pi_kt_row * out, int cap) {
    int n = 0;
    for (auto & kv : tab) {
        if (n >= cap) break;
        const kt_stat & st = kv.second;
        pi_kt_row & r = out[n++];
        r.family = std::get<0>(kv.first); r.type = std::get<1>(kv.first); r.n_tok = std::get<2>(kv.first); r.shape_bytes = std::get<3>(kv.first);
        r.launches = st.launches; r.bytes_total = st.bytes;
        // the launches that did the work: samples of at least a third of the instance's median duration (an aborted launch costs its dispatch, 2-5 us)
        double us = 0.0, sb = 0.0; uint64_t kept = 0;
        if (!st.us.empty()) {
            std::vector<float> sorted(st.us);
            std::nth_element(sorted.begin(), sorted.begin() + sorted.size() / 2, sorted.end());
            const float floor_us = sorted[sorted.size() / 2] / 3.0f;
            for (size_t i = 0; i < st.us.size(); i++) if (st.us[i] >= floor_us) { us += st.us[i]; sb += st.sbytes[i]; kept++; }
        }
        r.sampled = kept; r.sampled_dropped = (int32_t) (st.us.size() - kept);
        r.avg_us = kept ? us / (double) kept : 0.0; r.avg_sampled_bytes = kept ? sb / (double) kept : 0.0;
    }
    return n;
}
