// Launch accounting shared by every kernel launcher and the engine's graph cache: what bench.py's roofline is made of.
//   * every launch is COUNTED per instance = (family, weight type, tokens of the launch) with its algorithmic bytes — into the sink of the graph being captured (the engine
//     multiplies by the graph's replays: the timed region's own launch counts), or into the process-wide table for eager launches;
//   * with pi_kernel_timing(1, N) every Nth eager launch of an instance is TIMED: it goes through hipExtLaunchKernelGGL, whose start / stop events carry the kernel's own
//     begin / end timestamps (what rocprofv3 reports as the duration).  Events recorded inside a captured graph cannot be read back on ROCm 7.2, so durations come from a
//     graphs-off replay of the same schedule; counts and bytes come from the timed region itself.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <vector>

enum pi_kt_family {
    PI_KT_MATVEC = 0,    // VALU mat-vec k_mmvq_* (one token; also the 2..8-token launches the matrix cores do not take)
    PI_KT_MMQ    = 1,    // matrix-core mat-mat k_mmq_kq (3..32 tokens)
    PI_KT_NORMQ  = 2,    // k_norm_quant
    PI_KT_QUANT  = 3,    // k_quantize_q8_K / _q8_0
    PI_KT_ACTQ   = 4,    // k_act_quant
    PI_KT_ATT    = 5,    // k_attention
    PI_KT_OTHER  = 6,    // get_rows, rope + KV store, norm, add, top-k, ...
    PI_KT_NFAM   = 7
};
#define PI_KT_MAXNT 33                    // widths 1..32 have a bucket of their own; wider launches share the last one

struct pi_kt_sample { hipEvent_t a, b; int fam, type, nt; double bytes; };
struct pi_kt_count  { int fam, type, nt; uint64_t count; double bytes, bytes_sum; };     // bytes: algorithmic bytes of the first such launch (mat launches: it names the launch shape); bytes_sum over `count`
struct pi_kt_sink {                       // what one captured graph launches
    std::vector<pi_kt_count> launches;
};
void pi_kt_set_sink(pi_kt_sink * sink);   // thread-local: launches on this thread are being captured into a graph that owns `sink` (nullptr: eager)
bool pi_kt_enabled();
// counts the launch; true when it is one of the timed ones (the caller then launches through hipExtLaunchKernelGGL with ea / eb and calls pi_kt_end)
bool pi_kt_begin(int fam, int type, int nt, double bytes, hipEvent_t & ea, hipEvent_t & eb);
void pi_kt_end(int fam, int type, int nt, double bytes, hipEvent_t ea, hipEvent_t eb);

// launch `kernel` counted as (fam, type, nt) with `bytes` algorithmic bytes; timed when it is this instance's turn
#define PI_KT_LAUNCH(fam, type, nt, bytes, kernel, grid, block, lds, stream, ...) do { \
        hipEvent_t ea__ = nullptr, eb__ = nullptr; \
        if (pi_kt_begin((fam), (type), (nt), (double) (bytes), ea__, eb__)) { \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ea__, eb__, 0, __VA_ARGS__); \
            pi_kt_end((fam), (type), (nt), (double) (bytes), ea__, eb__); \
        } else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); \
    } while (0)
