// TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference's per-token hot path.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
// (pipeinfer_amd/) never links, imports or calls it.  Every function cites the reference file:line it follows
// (paths relative to /root/reference).  Parity status: PINNED for the tensor arithmetic (quantisers, dot
// products, norms, RoPE, softmax, activations, attention, whole-model logits) against the real reference
// library built by `make ref` (oracle/_ref/, tests/test_oracle_vs_ref.py) and against committed golden vectors
// generated from it (tests/golden/); PINNED for KV-cell bookkeeping (oracle/kvcache.c) against the reference's own
// llama_kv_cache_* functions, compiled from llama.cpp by ref_kv_harness.cpp (tests/test_kv_cells.py).  Restated only: the
// decode-level head rules and the mask loop, which sit inside llama_decode_internal / the graph callback.
//
// Where the reference has several SIMD arms, the oracle follows the x86 AVX2/FMA/F16C arm (the one `make ref`
// compiles and the one the paper's x86 CPU clusters ran), *including its floating-point summation order*
// (8 fp32 lanes + the hsum tree), so that oracle == reference bit-for-bit on this class of host.
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_API __attribute__((visibility("default")))

// ggml type ids (ggml.h enum ggml_type)
enum { OT_F32 = 0, OT_F16 = 1, OT_Q4_0 = 2, OT_Q4_1 = 3, OT_Q5_0 = 6, OT_Q5_1 = 7, OT_Q8_0 = 8, OT_Q8_1 = 9, OT_Q2_K = 10, OT_Q3_K = 11, OT_Q4_K = 12, OT_Q5_K = 13, OT_Q6_K = 14, OT_Q8_K = 15 };

#define OQK   32
#define OQK_K 256

#pragma pack(push, 1)
typedef struct { uint16_t d; uint8_t qs[16]; }                                     oblk_q4_0;  // ggml-quants.h:10-15
typedef struct { uint16_t d; int8_t qs[32]; }                                      oblk_q8_0;  // :42-47
typedef struct { uint16_t d, m; uint8_t qs[16]; }                                  oblk_q4_1;  // ggml-quants.h:18-22 (20 bytes)
typedef struct { uint16_t d; uint8_t qh[4]; uint8_t qs[16]; }                      oblk_q5_0;  // :26-30 (22 bytes)
typedef struct { uint16_t d, m; uint8_t qh[4]; uint8_t qs[16]; }                   oblk_q5_1;  // :34-39 (24 bytes)
typedef struct { float d; float s; int8_t qs[32]; }                                oblk_q8_1;  // :50-54 (40 bytes): s = d * sum(qs)
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qs[128]; }          oblk_q4_K;  // :116-122
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qh[32]; uint8_t qs[128]; } oblk_q5_K; // :138-145
typedef struct { uint8_t ql[128]; uint8_t qh[64]; int8_t scales[16]; uint16_t d; } oblk_q6_K;  // :152-158
typedef struct { uint8_t scales[16]; uint8_t qs[64]; uint16_t d; uint16_t dmin; } oblk_q2_K;                // ggml-quants.h:74-79 (84 bytes)
typedef struct { uint8_t hmask[32]; uint8_t qs[64]; uint8_t scales[12]; uint16_t d; } oblk_q3_K;   // ggml-quants.h:95-100 (110 bytes)
typedef struct { float d; int8_t qs[256]; int16_t bsums[16]; }                     oblk_q8_K;  // :161-166
#pragma pack(pop)

// ---- quants.c -------------------------------------------------------------------------------------------
ORACLE_API float    o_fp16_to_fp32(uint16_t h);
ORACLE_API uint16_t o_fp32_to_fp16(float f);
ORACLE_API int      o_blck_size(int type);
ORACLE_API int      o_type_size(int type);
ORACLE_API size_t   o_row_size(int type, int64_t n);
ORACLE_API int      o_vec_dot_type(int type);
ORACLE_API void     o_quantize_row_q8_K(const float * x, void * y, int k);
ORACLE_API void     o_quantize_row_q8_0(const float * x, void * y, int k);
ORACLE_API void     o_quantize_act(int vec_dot_type, const float * x, void * y, int k);
ORACLE_API void     o_dequantize_row(int type, const void * x, float * y, int k);
ORACLE_API float    o_vec_dot(int type, int n, const void * x, const void * y);
// exact integer partial sums of one K-quant super-block (or one 32-block for q4_0/q8_0): used by GPU parity tests
ORACLE_API void     o_block_isums(int type, const void * xblk, const void * yblk, int32_t * isum /*[8]*/, int32_t * msum /*[1]*/);

// ---- ops.c ----------------------------------------------------------------------------------------------
ORACLE_API void     o_tables_init(void);
ORACLE_API const uint16_t * o_table_exp_f16(void);
ORACLE_API const uint16_t * o_table_silu_f16(void);
ORACLE_API const uint16_t * o_table_gelu_f16(void);
ORACLE_API void     o_rms_norm(const float * x, float * y, int n, int rows, float eps);
ORACLE_API void     o_norm(const float * x, float * y, int n, int rows, float eps);
ORACLE_API void     o_rope(const float * x, float * y, int head_dim, int n_head, int n_tok, const int32_t * pos,
                           int mode, float freq_base, float freq_scale);
ORACLE_API void     o_rope_ex(const float * x, float * y, int head_dim, int n_dims, int n_head, int n_tok, const int32_t * pos,
                              int mode, float freq_base, float freq_scale);       // n_dims < head_dim: NeoX blocks (ggml.c:11133-11160)
ORACLE_API void     o_soft_max(const float * x, float * y, int nc, int nr);
ORACLE_API void     o_silu(const float * x, float * y, int n);
ORACLE_API void     o_gelu(const float * x, float * y, int n);
ORACLE_API float    o_vec_dot_f16(int n, const uint16_t * x, const uint16_t * y);
ORACLE_API void     o_mul_mat(int type, const void * w, int ne00, int ne01, const float * act, int n_tok, float * out);
ORACLE_API void     o_get_rows(int type, const void * w, int ne00, const int32_t * ids, int n, float * out);
ORACLE_API void     o_attention(uint16_t * k_cache, uint16_t * v_cache, int n_ctx,
                                const float * q_cur, const float * k_cur, const float * v_cur, const float * mask,
                                int head_dim, int n_head, int n_head_kv, int n_tok, int kv_head, int n_kv, float * out);
ORACLE_API void     o_attention_alibi(uint16_t * k_cache, uint16_t * v_cache, int n_ctx,
                                const float * q_cur, const float * k_cur, const float * v_cur, const float * mask,
                                int head_dim, int n_head, int n_head_kv, int n_tok, int kv_head, int n_kv, float max_alibi_bias, float * out);

// ---- kvcache.c ------------------------------------------------------------------------------------------
typedef struct o_kv_cache o_kv_cache;
ORACLE_API o_kv_cache * o_kv_new(int n_ctx, int max_seq);
ORACLE_API void     o_kv_free(o_kv_cache * c);
ORACLE_API int      o_kv_find_slot(o_kv_cache * c, int n_tokens, const int32_t * pos, const int32_t * n_seq_id,
                                   const int32_t * seq_ids_flat, int seq_stride);   // returns head or -1
ORACLE_API void     o_kv_pre_decode(o_kv_cache * c, int n_tokens);     // llama.cpp:5548-5550
ORACLE_API void     o_kv_post_decode(o_kv_cache * c, int n_tokens);    // llama.cpp:5708-5713
ORACLE_API int      o_kv_cell_max(const o_kv_cache * c);
ORACLE_API int      o_kv_n(const o_kv_cache * c);                      // min(n_ctx, max(32, cell_max))
ORACLE_API void     o_kv_clear(o_kv_cache * c);
ORACLE_API void     o_kv_seq_rm(o_kv_cache * c, int seq_id, int p0, int p1);
ORACLE_API void     o_kv_seq_cp(o_kv_cache * c, int src, int dst, int p0, int p1);
ORACLE_API void     o_kv_seq_keep(o_kv_cache * c, int seq_id);
ORACLE_API int      o_kv_head(const o_kv_cache * c);
ORACLE_API int      o_kv_used(const o_kv_cache * c);
ORACLE_API int      o_kv_cell_pos(const o_kv_cache * c, int i);
ORACLE_API int      o_kv_cell_has(const o_kv_cache * c, int i, int seq_id);
ORACLE_API int      o_kv_cell_nseq(const o_kv_cache * c, int i);
ORACLE_API void     o_kv_build_mask(const o_kv_cache * c, int n_kv, int n_tokens, const int32_t * pos, const int32_t * n_seq_id,
                                    const int32_t * seq_ids_flat, int seq_stride, float * mask /*[n_tok][n_kv]*/);

// ---- model.c --------------------------------------------------------------------------------------------
typedef struct o_model o_model;
ORACLE_API o_model * o_model_load(const char * gguf_path, int n_ctx, int max_seq);
ORACLE_API void     o_model_free(o_model * m);
ORACLE_API int      o_model_info(const o_model * m, int what);         // 0 n_vocab,1 n_embd,2 n_layer,3 n_head,4 n_head_kv,5 n_ff,6 is_falcon,7 n_ctx
ORACLE_API o_kv_cache * o_model_kv(o_model * m);
// full decode as llama_decode does it (find_slot + mask + graph + head advance); logits for every token. 0 ok, 1 no slot.
ORACLE_API int      o_model_decode(o_model * m, int n_tok, const int32_t * tokens, const int32_t * pos, const int32_t * n_seq_id,
                                   const int32_t * seq_ids_flat, int seq_stride, float * logits);
// compute only, with caller-supplied placement and mask (mirrors oracle/ref_harness.c ref_model_decode)
ORACLE_API int      o_model_forward(o_model * m, const int32_t * tokens, const int32_t * pos, int n_tok, int kv_head, int n_kv,
                                    const float * mask, float * logits);
ORACLE_API int      o_argmax(const float * x, int n);                  // llama.cpp:7496-7510 (first max wins)
ORACLE_API void     o_softmax_topk(const float * logits, int n, int k, int32_t * ids, float * logit, float * p); // llama.cpp:7019-7046
ORACLE_API void     o_set_threads(int n);

#ifdef __cplusplus
}
#endif
