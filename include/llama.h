/* llama.h — the drop-in boundary of the MI355X pipelined speculative-decoding engine (libpipeinfer_amd.so).
 *
 * This is OUR header: it declares, with the same names, argument meaning, struct layouts and return codes, the subset
 * of the reference's llama.h that its speculative drivers use (reference llama.h), so that
 * examples/speculative/speculative.cpp-style callers bind unchanged.  Each group cites the reference declaration
 * (llama.h:line) and implementation (llama.cpp:line) it replaces.  The three reference signatures that take C++
 * references (llama.h:285-290, :583) are declared here with pointers: identical machine ABI on x86-64 SysV.
 *
 * Semantics that differ by design on one MI355X node (documented in DESIGN.md):
 *   - llama_start_async_decode() only enqueues work on the context's HIP stream and returns a ticket; the reference
 *     computes the head's layer slice synchronously (llama.cpp:5850-5855).
 *   - llama_mpi_iprobe() is hipEventQuery on the oldest in-flight run instead of MPI_Iprobe (llama.cpp:5457-5459).
 *   - llama_sync_token(_data), llama_swap_comm and llama_split_comm keep their signatures; inside one process they only do the
 *     bookkeeping the driver can observe (llama_node_id).
 */
#ifndef PIPEINFER_LLAMA_H
#define PIPEINFER_LLAMA_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLAMA_API __attribute__((visibility("default")))
#define LLAMA_DEFAULT_SEED 0xFFFFFFFF

struct llama_model;
struct llama_context;
struct ggml_cgraph;                 /* opaque ticket of an in-flight run (reference returns its graph pointer) */

typedef int32_t llama_pos;
typedef int32_t llama_token;
typedef int32_t llama_seq_id;

enum llama_vocab_type { LLAMA_VOCAB_TYPE_SPM = 0, LLAMA_VOCAB_TYPE_BPE = 1 };                        /* llama.h:68-71 */
enum llama_token_type {                                                                              /* llama.h:73-81 */
    LLAMA_TOKEN_TYPE_UNDEFINED = 0, LLAMA_TOKEN_TYPE_NORMAL = 1, LLAMA_TOKEN_TYPE_UNKNOWN = 2, LLAMA_TOKEN_TYPE_CONTROL = 3,
    LLAMA_TOKEN_TYPE_USER_DEFINED = 4, LLAMA_TOKEN_TYPE_UNUSED = 5, LLAMA_TOKEN_TYPE_BYTE = 6,
};

typedef struct llama_token_data { llama_token id; float logit; float p; } llama_token_data;           /* llama.h:117-121 */
typedef struct llama_token_data_array { llama_token_data * data; size_t size; bool sorted; } llama_token_data_array;

typedef void (*llama_progress_callback)(float progress, void * ctx);

/* llama.h:142-162 — including PipeInfer's batch_id / max_n_seq */
typedef struct llama_batch {
    int32_t n_tokens;
    llama_token  *  token;
    float        *  embd;
    llama_pos    *  pos;
    int32_t      *  n_seq_id;
    llama_seq_id ** seq_id;
    int8_t       *  logits;
    llama_pos    all_pos_0;
    llama_pos    all_pos_1;
    llama_seq_id all_seq_id;
    int32_t      batch_id;
    int32_t      max_n_seq;
} llama_batch;

struct llama_model_params {                                                                         /* llama.h:164-178 */
    int32_t n_gpu_layers;
    int32_t main_gpu;              /* HIP device that holds this model's stage(s) */
    const float * tensor_split;
    llama_progress_callback progress_callback;
    void * progress_callback_user_data;
    bool vocab_only;
    bool use_mmap;
    bool use_mlock;
};

struct llama_context_params {                                                                       /* llama.h:180-203 */
    uint32_t seed;
    uint32_t n_ctx;
    uint32_t n_batch;
    uint32_t n_threads;
    uint32_t n_threads_batch;
    int8_t   rope_scaling_type;
    float    rope_freq_base;
    float    rope_freq_scale;
    float    yarn_ext_factor;
    float    yarn_attn_factor;
    float    yarn_beta_fast;
    float    yarn_beta_slow;
    uint32_t yarn_orig_ctx;
    bool mul_mat_q;
    bool f16_kv;
    bool logits_all;
    bool embedding;
};

struct llama_timings {                                                                              /* llama.h:247-258 */
    double t_start_ms, t_end_ms, t_load_ms, t_sample_ms, t_p_eval_ms, t_eval_ms;
    int32_t n_sample, n_p_eval, n_eval;
};

struct llama_kv_cache_view_cell { llama_pos pos; };                                                 /* llama.h:402-404 */
struct llama_kv_cache_view {                                                                        /* llama.h:407-437 */
    int32_t n_cells, n_max_seq, token_count, used_cells, max_contiguous, max_contiguous_idx;
    struct llama_kv_cache_view_cell * cells;
    llama_seq_id * cells_sequences;
};

/* ---- lifecycle (llama.h:261-300; llama.cpp:8566-8946) ---- */
LLAMA_API struct llama_model_params   llama_model_default_params(void);
LLAMA_API struct llama_context_params llama_context_default_params(void);
LLAMA_API void llama_backend_init(bool numa);
LLAMA_API void llama_backend_free(void);
LLAMA_API struct llama_model * llama_load_model_from_file(const char * path_model, struct llama_model_params params);
LLAMA_API void llama_free_model(struct llama_model * model);
LLAMA_API struct llama_context * llama_new_context_with_model(struct llama_model * model, struct llama_context_params params);
LLAMA_API void llama_free(struct llama_context * ctx);
LLAMA_API int64_t llama_time_us(void);
LLAMA_API int64_t ggml_time_us(void);                                                               /* ggml.h; used by the drivers for timing */

/* ---- PipeInfer pipeline control (llama.h:279-313; llama.cpp:8901-8938, 9981-9993, 5457-5459) ---- */
LLAMA_API void llama_split_layers_weighted(struct llama_context * ctx, float device_weights[], size_t num_weights);
LLAMA_API void llama_swap_comm(struct llama_context * ctx);
LLAMA_API void llama_split_comm(struct llama_context * ctx, int color);
LLAMA_API void llama_sync_token(struct llama_context * ctx, llama_token * token, int root);
LLAMA_API void llama_sync_token_data(struct llama_context * ctx, llama_token_data * data, int root);
LLAMA_API int  llama_node_id(struct llama_context * ctx);
LLAMA_API bool llama_mpi_iprobe(struct llama_context * lctx);
LLAMA_API void llama_cancel_run(struct llama_context * ctx, int32_t * canceled, int count);
LLAMA_API struct ggml_cgraph * llama_start_async_decode(struct llama_context * lctx, struct llama_batch batch);          /* reference: llama_context & */
LLAMA_API int  llama_finish_async_decode(struct llama_context * lctx, struct llama_batch batch, struct ggml_cgraph * cgraph); /* returns 1 on success, -1 on null ticket */

/* ---- model / context queries (llama.h:315-345) ---- */
LLAMA_API const struct llama_model * llama_get_model(const struct llama_context * ctx);
LLAMA_API int llama_n_ctx(const struct llama_context * ctx);
LLAMA_API enum llama_vocab_type llama_vocab_type(const struct llama_model * model);
LLAMA_API int llama_n_vocab(const struct llama_model * model);
LLAMA_API int llama_n_ctx_train(const struct llama_model * model);
LLAMA_API int llama_n_embd(const struct llama_model * model);
LLAMA_API int llama_model_desc(const struct llama_model * model, char * buf, size_t buf_size);
LLAMA_API uint64_t llama_model_size(const struct llama_model * model);
LLAMA_API uint64_t llama_model_n_params(const struct llama_model * model);

/* ---- KV cache (llama.h:439-500; llama.cpp:1593-1765, 9238-9359) ---- */
LLAMA_API struct llama_kv_cache_view llama_kv_cache_view_init(const struct llama_context * ctx, int32_t n_max_seq);
LLAMA_API void llama_kv_cache_view_free(struct llama_kv_cache_view * view);
LLAMA_API void llama_kv_cache_view_update(const struct llama_context * ctx, struct llama_kv_cache_view * view);
LLAMA_API int  llama_get_kv_cache_token_count(const struct llama_context * ctx);
LLAMA_API int  llama_get_kv_cache_used_cells(const struct llama_context * ctx);
LLAMA_API void llama_kv_cache_clear(struct llama_context * ctx);
LLAMA_API void llama_kv_cache_seq_rm(struct llama_context * ctx, llama_seq_id seq_id, llama_pos p0, llama_pos p1);
LLAMA_API void llama_kv_cache_seq_cp(struct llama_context * ctx, llama_seq_id seq_id_src, llama_seq_id seq_id_dst, llama_pos p0, llama_pos p1);
LLAMA_API void llama_kv_cache_seq_keep(struct llama_context * ctx, llama_seq_id seq_id);

/* ---- decoding (llama.h:560-610; llama.cpp:5461-5872, 9839-10030) ---- */
LLAMA_API struct llama_batch llama_batch_get_one(llama_token * tokens, int32_t n_tokens, llama_pos pos_0, llama_seq_id seq_id);
LLAMA_API struct llama_batch llama_batch_init(int32_t n_tokens, int32_t embd, int32_t n_seq_max);
LLAMA_API void llama_batch_free(struct llama_batch * batch);                                         /* reference: llama_batch & */
LLAMA_API int  llama_decode(struct llama_context * ctx, struct llama_batch batch);                   /* 0 ok, 1 no KV slot, <0 error */
LLAMA_API void llama_set_n_threads(struct llama_context * ctx, uint32_t n_threads, uint32_t n_threads_batch);
LLAMA_API float * llama_get_logits(struct llama_context * ctx);
LLAMA_API float * llama_get_logits_ith(struct llama_context * ctx, int32_t i);

/* ---- vocabulary / tokenizer (llama.h:612-672; llama.cpp:5878-6605, 10039-10168) ---- */
LLAMA_API const char * llama_token_get_text(const struct llama_model * model, llama_token token);
LLAMA_API float llama_token_get_score(const struct llama_model * model, llama_token token);
LLAMA_API enum llama_token_type llama_token_get_type(const struct llama_model * model, llama_token token);
LLAMA_API llama_token llama_token_bos(const struct llama_model * model);
LLAMA_API llama_token llama_token_eos(const struct llama_model * model);
LLAMA_API llama_token llama_token_nl(const struct llama_model * model);
LLAMA_API int llama_add_bos_token(const struct llama_model * model);
LLAMA_API int llama_add_eos_token(const struct llama_model * model);
LLAMA_API int llama_tokenize(const struct llama_model * model, const char * text, int text_len, llama_token * tokens, int n_max_tokens, bool add_bos, bool special);
LLAMA_API int llama_token_to_piece(const struct llama_model * model, llama_token token, char * buf, int length);

/* ---- sampling (llama.h:700-790; llama.cpp:7019-7046, 7496-7510) ---- */
LLAMA_API void llama_sample_softmax(struct llama_context * ctx, llama_token_data_array * candidates);
LLAMA_API llama_token llama_sample_token_greedy(struct llama_context * ctx, llama_token_data_array * candidates);

/* ---- timings (llama.h:836-842; llama.cpp:10170-10206) ---- */
LLAMA_API struct llama_timings llama_get_timings(struct llama_context * ctx);
LLAMA_API void llama_print_timings(struct llama_context * ctx);
LLAMA_API void llama_reset_timings(struct llama_context * ctx);
LLAMA_API const char * llama_print_system_info(void);

#ifdef __cplusplus
}
/* The reference declares these with C++ references (llama.h:285-290); callers written against it keep compiling.
 * Same machine ABI: a reference is passed as the object's address. */
static inline struct ggml_cgraph * llama_start_async_decode(struct llama_context & lctx, struct llama_batch batch) { return llama_start_async_decode(&lctx, batch); }
static inline int llama_finish_async_decode(struct llama_context & lctx, struct llama_batch batch, struct ggml_cgraph * cgraph) { return llama_finish_async_decode(&lctx, batch, cgraph); }
#endif
#endif
