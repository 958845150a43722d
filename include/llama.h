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

#include "ggml.h"                   /* as the reference's llama.h does (llama.h:4): assertion macros, timers, log types */

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
#include <string>                   /* llama.h:15 — for dump_kv_cache_view_seqs */
extern "C" {
#endif

#define LLAMA_API __attribute__((visibility("default")))
#define DEPRECATED(func, hint) func __attribute__((deprecated(hint)))
#define LLAMA_MAX_DEVICES 1                                                                           /* llama.h:9 (no GGML_USE_CUBLAS) */
#define LLAMA_DEFAULT_SEED 0xFFFFFFFF
#define LLAMA_MAX_RNG_STATE (64*1024)
#define LLAMA_FILE_MAGIC_GGSN 0x6767736eu
#define LLAMA_SESSION_MAGIC   LLAMA_FILE_MAGIC_GGSN
#define LLAMA_SESSION_VERSION 2

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

enum llama_ftype {                                                                                   /* llama.h:85-108 */
    LLAMA_FTYPE_ALL_F32 = 0, LLAMA_FTYPE_MOSTLY_F16 = 1, LLAMA_FTYPE_MOSTLY_Q4_0 = 2, LLAMA_FTYPE_MOSTLY_Q4_1 = 3, LLAMA_FTYPE_MOSTLY_Q4_1_SOME_F16 = 4,
    LLAMA_FTYPE_MOSTLY_Q8_0 = 7, LLAMA_FTYPE_MOSTLY_Q5_0 = 8, LLAMA_FTYPE_MOSTLY_Q5_1 = 9, LLAMA_FTYPE_MOSTLY_Q2_K = 10, LLAMA_FTYPE_MOSTLY_Q3_K_S = 11,
    LLAMA_FTYPE_MOSTLY_Q3_K_M = 12, LLAMA_FTYPE_MOSTLY_Q3_K_L = 13, LLAMA_FTYPE_MOSTLY_Q4_K_S = 14, LLAMA_FTYPE_MOSTLY_Q4_K_M = 15, LLAMA_FTYPE_MOSTLY_Q5_K_S = 16,
    LLAMA_FTYPE_MOSTLY_Q5_K_M = 17, LLAMA_FTYPE_MOSTLY_Q6_K = 18,
    LLAMA_FTYPE_GUESSED = 1024,
};
enum llama_rope_scaling_type {                                                                       /* llama.h:110-116 */
    LLAMA_ROPE_SCALING_UNSPECIFIED = -1, LLAMA_ROPE_SCALING_NONE = 0, LLAMA_ROPE_SCALING_LINEAR = 1, LLAMA_ROPE_SCALING_YARN = 2,
    LLAMA_ROPE_SCALING_MAX_VALUE = LLAMA_ROPE_SCALING_YARN,
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

typedef struct llama_model_quantize_params {                                                        /* llama.h:205-212 */
    int nthread;
    enum llama_ftype ftype;
    bool allow_requantize;
    bool quantize_output_tensor;
    bool only_copy;
    bool pure;
} llama_model_quantize_params;

struct llama_grammar;                                                                               /* llama.h:214-245 — GBNF grammars: types only, see below */
enum llama_gretype {
    LLAMA_GRETYPE_END = 0, LLAMA_GRETYPE_ALT = 1, LLAMA_GRETYPE_RULE_REF = 2, LLAMA_GRETYPE_CHAR = 3, LLAMA_GRETYPE_CHAR_NOT = 4,
    LLAMA_GRETYPE_CHAR_RNG_UPPER = 5, LLAMA_GRETYPE_CHAR_ALT = 6,
};
typedef struct llama_grammar_element { enum llama_gretype type; uint32_t value; } llama_grammar_element;

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
LLAMA_API struct llama_model_quantize_params llama_model_quantize_default_params(void);
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

LLAMA_API int  llama_max_devices(void);
LLAMA_API bool llama_mmap_supported(void);                                                          /* false: weights are read into HBM, not mapped */
LLAMA_API bool llama_mlock_supported(void);

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
LLAMA_API float llama_rope_freq_scale_train(const struct llama_model * model);
LLAMA_API int llama_model_meta_val_str(const struct llama_model * model, const char * key, char * buf, size_t buf_size);     /* llama.h:335-345: GGUF metadata as strings */
LLAMA_API int llama_model_meta_count(const struct llama_model * model);
LLAMA_API int llama_model_meta_key_by_index(const struct llama_model * model, int i, char * buf, size_t buf_size);
LLAMA_API int llama_model_meta_val_str_by_index(const struct llama_model * model, int i, char * buf, size_t buf_size);

/* ---- present for source compatibility, OUTSIDE the decode hot path: each prints why and fails (llama.h:357-390, 503-560) ---- */
LLAMA_API struct ggml_tensor * llama_get_model_tensor(struct llama_model * model, const char * name);
LLAMA_API int llama_model_quantize(const char * fname_inp, const char * fname_out, const llama_model_quantize_params * params);
LLAMA_API DEPRECATED(int llama_apply_lora_from_file(struct llama_context * ctx, const char * path_lora, float scale, const char * path_base_model, int n_threads),
                     "use llama_model_apply_lora_from_file instead");
LLAMA_API int llama_model_apply_lora_from_file(const struct llama_model * model, const char * path_lora, float scale, const char * path_base_model, int n_threads);
LLAMA_API size_t llama_get_state_size(const struct llama_context * ctx);
LLAMA_API size_t llama_copy_state_data(struct llama_context * ctx, uint8_t * dst);
LLAMA_API size_t llama_set_state_data(struct llama_context * ctx, uint8_t * src);
LLAMA_API bool llama_load_session_file(struct llama_context * ctx, const char * path_session, llama_token * tokens_out, size_t n_token_capacity, size_t * n_token_count_out);
LLAMA_API bool llama_save_session_file(struct llama_context * ctx, const char * path_session, const llama_token * tokens, size_t n_token_count);
LLAMA_API DEPRECATED(int llama_eval(struct llama_context * ctx, llama_token * tokens, int32_t n_tokens, int n_past), "use llama_decode() instead");
LLAMA_API DEPRECATED(int llama_eval_embd(struct llama_context * ctx, float * embd, int32_t n_tokens, int n_past), "use llama_decode() instead");
LLAMA_API float * llama_get_embeddings(struct llama_context * ctx);

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
LLAMA_API void llama_kv_cache_seq_cp_back(struct llama_context * ctx, llama_seq_id seq_id_src, llama_seq_id seq_id_dst, llama_pos p0, llama_pos p1);      /* llama.h:478-490: same cell update */
LLAMA_API void llama_kv_cache_seq_cp_sync_bi(struct llama_context * ctx, llama_seq_id seq_id_src, llama_seq_id seq_id_dst, llama_pos p0, llama_pos p1);
LLAMA_API void llama_kv_cache_seq_shift(struct llama_context * ctx, llama_seq_id seq_id, llama_pos p0, llama_pos p1, llama_pos delta);                   /* aborts: not on this path (nor on the reference's MPI path) */

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
LLAMA_API llama_token llama_token_prefix(const struct llama_model * model);
LLAMA_API llama_token llama_token_middle(const struct llama_model * model);
LLAMA_API llama_token llama_token_suffix(const struct llama_model * model);
LLAMA_API llama_token llama_token_eot(const struct llama_model * model);
LLAMA_API int llama_tokenize(const struct llama_model * model, const char * text, int text_len, llama_token * tokens, int n_max_tokens, bool add_bos, bool special);
LLAMA_API int llama_token_to_piece(const struct llama_model * model, llama_token token, char * buf, int length);

/* ---- grammar (llama.h:674-686): GBNF is outside the decode hot path — init returns NULL after saying so, sample/accept abort ---- */
LLAMA_API struct llama_grammar * llama_grammar_init(const llama_grammar_element ** rules, size_t n_rules, size_t start_rule_index);
LLAMA_API void llama_grammar_free(struct llama_grammar * grammar);
LLAMA_API struct llama_grammar * llama_grammar_copy(const struct llama_grammar * grammar);

/* ---- sampling (llama.h:688-815; llama.cpp:7011-7545): host-side samplers over a candidate array, same arithmetic as the reference ---- */
LLAMA_API void llama_set_rng_seed(struct llama_context * ctx, uint32_t seed);
LLAMA_API void llama_sample_repetition_penalties(struct llama_context * ctx, llama_token_data_array * candidates, const llama_token * last_tokens, size_t penalty_last_n,
                                                 float penalty_repeat, float penalty_freq, float penalty_present);
LLAMA_API void llama_sample_classifier_free_guidance(struct llama_context * ctx, llama_token_data_array * candidates, struct llama_context * guidance_ctx, float scale);
LLAMA_API void llama_sample_softmax(struct llama_context * ctx, llama_token_data_array * candidates);
LLAMA_API void llama_sample_top_k(struct llama_context * ctx, llama_token_data_array * candidates, int k, size_t min_keep);
LLAMA_API void llama_sample_top_p(struct llama_context * ctx, llama_token_data_array * candidates, float p, size_t min_keep);
LLAMA_API void llama_sample_min_p(struct llama_context * ctx, llama_token_data_array * candidates, float p, size_t min_keep);
LLAMA_API void llama_sample_tail_free(struct llama_context * ctx, llama_token_data_array * candidates, float z, size_t min_keep);
LLAMA_API void llama_sample_typical(struct llama_context * ctx, llama_token_data_array * candidates, float p, size_t min_keep);
LLAMA_API void llama_sample_temp(struct llama_context * ctx, llama_token_data_array * candidates, float temp);
LLAMA_API DEPRECATED(void llama_sample_temperature(struct llama_context * ctx, llama_token_data_array * candidates, float temp), "use llama_sample_temp instead");
LLAMA_API void llama_sample_grammar(struct llama_context * ctx, llama_token_data_array * candidates, const struct llama_grammar * grammar);
LLAMA_API llama_token llama_sample_token_mirostat(struct llama_context * ctx, llama_token_data_array * candidates, float tau, float eta, int m, float * mu);
LLAMA_API llama_token llama_sample_token_mirostat_v2(struct llama_context * ctx, llama_token_data_array * candidates, float tau, float eta, float * mu);
LLAMA_API llama_token llama_sample_token_greedy(struct llama_context * ctx, llama_token_data_array * candidates);
LLAMA_API llama_token llama_sample_token(struct llama_context * ctx, llama_token_data_array * candidates);
LLAMA_API void llama_grammar_accept_token(struct llama_context * ctx, struct llama_grammar * grammar, llama_token token);

/* ---- beam search (llama.h:817-858): outside the hot path, says so and returns ---- */
struct llama_beam_view { const llama_token * tokens; size_t n_tokens; float p; bool eob; };
struct llama_beams_state { struct llama_beam_view * beam_views; size_t n_beams; size_t common_prefix_length; bool last_call; };
typedef void (*llama_beam_search_callback_fn_t)(void * callback_data, struct llama_beams_state);
LLAMA_API void llama_beam_search(struct llama_context * ctx, llama_beam_search_callback_fn_t callback, void * callback_data, size_t n_beams, int n_past, int n_predict);

/* ---- timings (llama.h:836-842; llama.cpp:10170-10206) ---- */
LLAMA_API struct llama_timings llama_get_timings(struct llama_context * ctx);
LLAMA_API void llama_print_timings(struct llama_context * ctx);
LLAMA_API void llama_reset_timings(struct llama_context * ctx);
LLAMA_API const char * llama_print_system_info(void);
LLAMA_API void llama_log_set(ggml_log_callback log_callback, void * user_data);
LLAMA_API void llama_dump_timing_info_yaml(FILE * stream, const struct llama_context * ctx);

#ifdef __cplusplus
}
/* llama.h:670-673 */
__attribute__((visibility("default"))) std::string dump_kv_cache_view_seqs(const llama_kv_cache_view & view, int row_size = 40);   /* C++ linkage, as in the reference */
/* llama.h:585: the reference frees a batch through a C++ reference */
static inline void llama_batch_free(struct llama_batch & batch) { llama_batch_free(&batch); }
/* The reference declares these with C++ references (llama.h:285-290); callers written against it keep compiling.
 * Same machine ABI: a reference is passed as the object's address. */
static inline struct ggml_cgraph * llama_start_async_decode(struct llama_context & lctx, struct llama_batch batch) { return llama_start_async_decode(&lctx, batch); }
static inline int llama_finish_async_decode(struct llama_context & lctx, struct llama_batch batch, struct ggml_cgraph * cgraph) { return llama_finish_async_decode(&lctx, batch, cgraph); }
#endif
#endif
