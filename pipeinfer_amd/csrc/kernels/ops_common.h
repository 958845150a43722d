// Host-side declarations shared by the kernel translation units (not part of the public C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../../include/pipeinfer_ops.h"
#include "pi_device.h"

// A weight matrix in HBM.  Bytes are only re-ordered at load time (never changed) so that lane k of an 8-lane row group —
// the lane that owns bytes 4k..4k+3 of every 32-value group, exactly like lane k of the reference's AVX2 accumulators —
// reads its share of a block with one or two 16-byte loads ("k-major" layouts):
//   Q4_K (144 B)  [d,dmin,scales 16 B][qs: 2 halves p x 8 lanes k x 2 dwords e]   dword (p, k, e) = original qs[32(2p+e) + 4k .. +3]
//   Q5_K (176 B)  [hdr 16 B][qh 32 B natural][qs as for Q4_K]          lane k also reads qh[4k .. +3]
//   Q6_K (210 B)  p0: [blocks][192 B] = [2 halves n][8 lanes k][8 B: ql[64n+4k], ql[64n+32+4k]] + [8 lanes][8 B: qh[4k], qh[32+4k]]   p1: scales [blocks][16]   p2: d [blocks] f16
//   Q4_0 (18 B)   groups of 4 blocks, rows padded with zero blocks:  p0: [groups][4 dwords c][4 blocks b]   p1: d [groups][4] f16
//   Q8_0 (34 B)   groups of 4 blocks, rows padded with zero blocks:  p0: [groups][8 lanes k][4 blocks b]    p1: d [groups][4] f16
//   Q4_1 / Q5_0 / Q5_1 (20 / 22 / 24 B)  as Q4_0:  p0: [groups][4 dwords c][4 blocks b] nibbles   p1: [groups][d x 4, m x 4] f16 (m = 0 for Q5_0)   p2: qh [groups][4] u32 (0 for Q4_1)
//   F32 / F16     native rows
#define PI_MQ_TILE 2368      /* bytes of one (16 rows x 256 values) tile of the tile-major Q4_K copy */
#define PI_MQ_TILE_Q5 2880   /* Q5_K: + 16 rows x 32 bytes of fifth bits */
#define PI_MQ_TILE_Q3 1856   /* Q3_K: 64 d | 256 scale records | 512 hmask | 1024 qs */
#define PI_MQ_TILE_Q6 3648   /* Q6_K: 64 d | 512 scale records | 1024 qh | 2048 ql */

struct pi_weight {
    int      type;
    int64_t  ne0, ne1;     // K (contiguous), rows
    int64_t  nb;           // blocks per row
    int64_t  ng;           // Q4_0/Q8_0: groups of 4 blocks per row (ceil(nb/4))
    void *   base;         // allocation
    size_t   nbytes;       // == ggml_nbytes of the tensor (padding of Q4_0/Q8_0 tails not counted)
    const uint8_t * p0, * p1, * p2, * p3;
    // second, TILE-MAJOR copy for the multi-token MFMA kernel (mmq_mfma.hip), null when the type has none or copies are off (PI_MMQ_COPIES=0):
    // that kernel reads 16 rows x one block per step, which streams at 1.3-1.6 TB/s from the row-major planes above and at ~5 TB/s when the
    // 16 blocks are contiguous (tools_dev/pattern.hip).  [row group of 16][block][PI_MQ_TILE (Q4_K) / PI_MQ_TILE_Q5 (Q5_K) bytes]; scales / mins unpacked per lane, quants re-ordered (layout in runtime.hip)
    const uint8_t * mq;
    void *   mq_base;
    size_t   mq_nbytes;    // HBM bytes of that copy (0 without one)
};

void pi_set_error(const char * fmt, ...);
// abort word the calling thread's launches (and captures) hand to their kernels (pi_set_abort_word; nullptr: no check)
const int32_t * pi_abort_word();
int pi_account_n_kv();                                       // > 0: the cells an attention launch of this thread really reads (pi_set_account_n_kv)
// compute units the launching context's stream may use (pipeinfer_context_set_cu_range -> hipExtStreamCreateWithCUMask; 256 without a mask): what "one round of workgroups" means
int pi_cu_count();
const pi_tables & pi_get_tables();

#define PI_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t err__ = (expr);                                                           \
        if (err__ != hipSuccess) {                                                           \
            pi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), __FILE__, __LINE__); \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)

#define PI_CHECK_LAUNCH()                                                                    \
    do {                                                                                     \
        hipError_t err__ = hipGetLastError();                                                \
        if (err__ != hipSuccess) {                                                           \
            pi_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(err__), __FILE__, __LINE__); \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)

// smallest batch width that takes the MFMA kernel (mmq_mfma.hip) instead of the VALU mat-vec; the quantisers write its operand
// planes from the same width on.  PI_MMQ_MIN_NT overrides (tuning / A-B tests; read once).
int pi_mmq_min_nt();
int pi_mmq_max_nt();
int pi_mmq_step(int type, int n_tok, int64_t rows);             // tokens per matrix-core launch of a pass of n_tok tokens over matrices of `rows` rows in all

static inline bool pi_type_is_legacy(int t) { return t == PI_Q4_1 || t == PI_Q5_0 || t == PI_Q5_1; }
static inline int pi_type_blck(int t) {
    switch (t) { case PI_F32: case PI_F16: return 1; case PI_Q4_0: case PI_Q4_1: case PI_Q5_0: case PI_Q5_1: case PI_Q8_0: case PI_Q8_1: return 32; default: return 256; }
}
static inline int pi_type_size(int t) {
    switch (t) {
        case PI_F32: return 4; case PI_F16: return 2; case PI_Q4_0: return 18; case PI_Q8_0: return 34;
        case PI_Q4_1: return 20; case PI_Q5_0: return 22; case PI_Q5_1: return 24; case PI_Q8_1: return 40;
        case PI_Q2_K: return 84; case PI_Q3_K: return 110; case PI_Q4_K: return 144; case PI_Q5_K: return 176; case PI_Q6_K: return 210; case PI_Q8_K: return 292;
    }
    return 0;
}
