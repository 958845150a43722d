/* ggml.h — the handful of ggml.h names that callers of the llama.h boundary use (the reference's llama.h includes ggml.h, and its
 * examples/speculative + common/ sources rely on that): assertion macros, the timers, the log callback types, the CPU-feature queries
 * that common.cpp prints, and the opaque tensor / graph types that only appear as pointers.  Nothing of ggml's tensor API is part of
 * the boundary: the engine behind llama.h is libpipeinfer_amd.so (HIP), not ggml.
 * Reference: ggml.h:201-260 (GGML_ASSERT, GGML_UNREACHABLE), :1904-1912 (timers), :2103-2125 (ggml_cpu_has_*), ggml.h log types.
 */
#ifndef PIPEINFER_GGML_H
#define PIPEINFER_GGML_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef GGML_API
#define GGML_API __attribute__((visibility("default")))
#endif

#define GGML_ASSERT(x) \
    do { \
        if (!(x)) { \
            fprintf(stderr, "GGML_ASSERT: %s:%d: %s\n", __FILE__, __LINE__, #x); \
            abort(); \
        } \
    } while (0)
#define GGML_UNREACHABLE() __builtin_unreachable()
#define GGML_UNUSED(x) (void)(x)
#define GGML_PAD(x, n) (((x) + (n) - 1) & ~((n) - 1))

struct ggml_tensor;                 /* opaque: only pointers cross the boundary */
struct ggml_cgraph;                 /* opaque ticket of an in-flight run */

enum ggml_log_level { GGML_LOG_LEVEL_ERROR = 2, GGML_LOG_LEVEL_WARN = 3, GGML_LOG_LEVEL_INFO = 4 };
typedef void (*ggml_log_callback)(enum ggml_log_level level, const char * text, void * user_data);

GGML_API void    ggml_time_init(void);
GGML_API int64_t ggml_time_ms(void);
GGML_API int64_t ggml_time_us(void);
GGML_API int64_t ggml_cycles(void);
GGML_API int64_t ggml_cycles_per_ms(void);

/* what common.cpp's get_system_info() / llama_print_system_info() report: properties of the HOST the caller runs on */
GGML_API int ggml_cpu_has_avx(void);
GGML_API int ggml_cpu_has_avx2(void);
GGML_API int ggml_cpu_has_avx512(void);
GGML_API int ggml_cpu_has_avx512_vbmi(void);
GGML_API int ggml_cpu_has_avx512_vnni(void);
GGML_API int ggml_cpu_has_fma(void);
GGML_API int ggml_cpu_has_neon(void);
GGML_API int ggml_cpu_has_arm_fma(void);
GGML_API int ggml_cpu_has_metal(void);
GGML_API int ggml_cpu_has_f16c(void);
GGML_API int ggml_cpu_has_fp16_va(void);
GGML_API int ggml_cpu_has_wasm_simd(void);
GGML_API int ggml_cpu_has_blas(void);
GGML_API int ggml_cpu_has_cublas(void);
GGML_API int ggml_cpu_has_clblast(void);
GGML_API int ggml_cpu_has_gpublas(void);
GGML_API int ggml_cpu_has_sse3(void);
GGML_API int ggml_cpu_has_ssse3(void);
GGML_API int ggml_cpu_has_vsx(void);

#ifdef __cplusplus
}
#endif
#endif
