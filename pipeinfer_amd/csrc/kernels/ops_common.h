// Host-side declarations shared by the kernel translation units (not part of the public C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../../include/pipeinfer_ops.h"
#include "pi_device.h"

// A weight matrix in HBM.  Layouts (all planes 16-byte aligned, total bytes == ggml_nbytes):
//   Q4_K / Q5_K / F32 / F16 : native ggml rows (144 B / 176 B super-blocks are multiples of 16 B already)
//   Q6_K (210 B blocks, 2-byte aligned in GGUF) : four planes  ql[rows][nb*128] | qh[rows][nb*64] | scales[rows][nb*16] | d[rows][nb] f16
//   Q4_0 (18 B) : qs[rows][nb*16] | d[rows][nb] f16          Q8_0 (34 B) : qs[rows][nb*32] | d[rows][nb] f16
struct pi_weight {
    int      type;
    int64_t  ne0, ne1;     // K (contiguous), rows
    int64_t  nb;           // blocks per row
    void *   base;         // allocation
    size_t   nbytes;
    const uint8_t * p0, * p1, * p2, * p3;
};

void pi_set_error(const char * fmt, ...);
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

static inline int pi_type_blck(int t) {
    switch (t) { case PI_F32: case PI_F16: return 1; case PI_Q4_0: case PI_Q8_0: return 32; default: return 256; }
}
static inline int pi_type_size(int t) {
    switch (t) {
        case PI_F32: return 4; case PI_F16: return 2; case PI_Q4_0: return 18; case PI_Q8_0: return 34;
        case PI_Q4_K: return 144; case PI_Q5_K: return 176; case PI_Q6_K: return 210; case PI_Q8_K: return 292;
    }
    return 0;
}
