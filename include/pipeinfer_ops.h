/* pipeinfer_ops.h — C ABI of the MI355X hot-path operators (libpipeinfer_amd.so).
 *
 * One entry point per reference operator on the per-token path.  Every pointer named *_dev is a device
 * (HBM) pointer; `stream` is a hipStream_t passed as void* (NULL = the default stream).  All functions return 0 on
 * success and a negative value on error (pi_last_error() has the text).  There is NO CPU fallback: without a GPU the
 * compute entry points fail loudly.
 *
 * Reference interface each one replaces (file:line under the reference tree):
 *   pi_weight_upload        ggml tensor data as laid out by the GGUF loader             llama.cpp:1805-2134, ggml-quants.h:10-166
 *   pi_op_get_rows          ggml_compute_forward_get_rows_q                             ggml.c:10161-10188
 *   pi_op_norm              ggml_compute_forward_rms_norm_f32 / _norm_f32 + ggml_mul/add ggml.c:9028-9074, 8959-9008; llama.cpp:3579-3609
 *   pi_op_quantize          quantize_row_q8_K / quantize_row_q8_0 (mul_mat INIT phase)   ggml-quants.c:2329-2364, 702-760; ggml.c:9493-9506
 *   pi_op_mul_mat           ggml_compute_forward_mul_mat + ggml_vec_dot_{q4_0,q8_0,q4_K,q5_K,q6_K}_q8_*  ggml.c:9388-9597; ggml-quants.c:2426,3521,5216,5864,6594
 *   pi_op_rope_kv_store     ggml_compute_forward_rope_f32 + llm_build_kv_store           ggml.c:11006-11168; llama.cpp:3546-3577
 *   pi_op_attention         llm_build_kqv (KQ, scale, mask, soft_max, KQV)               llama.cpp:3698-3764; ggml.c:10551-10620, 1193-1228
 *   pi_op_silu_mul/_gelu/_add  ggml_silu / ggml_gelu / ggml_mul / ggml_add               ggml.c:1418-1498, 7583, 6803
 *   pi_op_topk              llama_sample_softmax / llama_sample_token_greedy             llama.cpp:7019-7046, 7496-7510
 */
#ifndef PIPEINFER_OPS_H
#define PIPEINFER_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PI_API __attribute__((visibility("default")))

/* ggml tensor type ids (ggml.h enum ggml_type) */
enum pi_type { PI_F32 = 0, PI_F16 = 1, PI_Q4_0 = 2, PI_Q4_1 = 3, PI_Q5_0 = 6, PI_Q5_1 = 7, PI_Q8_0 = 8, PI_Q8_1 = 9, PI_Q2_K = 10, PI_Q3_K = 11, PI_Q4_K = 12, PI_Q5_K = 13, PI_Q6_K = 14, PI_Q8_K = 15 };

typedef struct pi_weight pi_weight;    /* a weight matrix resident in HBM in the kernel-friendly layout */

PI_API const char * pi_last_error(void);
PI_API int     pi_device_count(void);
PI_API int     pi_device_init(int device);            /* selects the GPU, builds + uploads the fp16 lookup tables */

PI_API void *  pi_malloc(size_t bytes);
PI_API void    pi_free(void * p_dev);
PI_API int     pi_memcpy_h2d(void * dst_dev, const void * src, size_t bytes, void * stream);
PI_API int     pi_memcpy_d2h(void * dst, const void * src_dev, size_t bytes, void * stream);
PI_API int     pi_memset(void * dst_dev, int value, size_t bytes, void * stream);
PI_API int     pi_stream_sync(void * stream);
PI_API int     pi_device_sync(void);

/* Kernel-granular early cancellation (reference: the abort callback polled between graph nodes, ggml.c:15748-15811, llama.cpp:5614-5628, 5665-5687).
 * word_dev: the first int32 of a 64-byte block in device memory (zeroed by the owner) holding (run id << 1 | aborted); words 8..14 of the block count, per kernel family
 * (enum order of pi_kt_row.family), the launches that found the run aborted and returned on entry.  Every hot-path kernel launched — or captured into a
 * graph — by the calling thread after pi_set_abort_word(word_dev) reads it on entry and returns at once when bit 0 is set (NULL: no check).
 *   pi_op_abort_begin   (in stream order, ahead of a run's kernels)  word = run_id << 1 | (*host_flag != 0); host_flag: pinned host memory, nullable
 *   pi_op_abort_cancel  (any stream, any time)                        if word == run_id << 1 then word |= 1: only the run that is executing NOW is hit */
PI_API void    pi_set_abort_word(const int32_t * word_dev);
/* launch accounting only: the attention launches of the calling thread are priced with n_kv cells from now on (0: with the n_kv they are launched with) */
PI_API void    pi_set_account_n_kv(int n_kv);
/* launches of the calling thread from now on go to a stream restricted to `n` compute units (hipExtStreamCreateWithCUMask): the launchers size "one round of workgroups" for it (0 / 256: the whole chip) */
PI_API void    pi_set_cu_count(int n);
PI_API int     pi_op_abort_begin(int32_t * word_dev, int32_t run_id, const int32_t * host_flag, void * stream);
PI_API int     pi_op_abort_cancel(int32_t * word_dev, int32_t run_id, void * stream);
/* Arrival flags of the stage transport (reference: the blocking MPI_Recv of the activations, ggml-mpi.c:451-487): a 16-byte record {ticket << 1 | aborted, -, realtime stamp}
 * in page-locked host memory mapped by the producing and the consuming process.
 *   pi_op_flag_publish      (producer's stream, behind its peer store)  flag = ticket << 1 | (abort word & 1 | force_abort), stamped with the device's 100 MHz clock
 *   pi_op_abort_begin_wait  (consumer's stream, ahead of the run)       polls the flag from the device until it carries `ticket` (at most timeout_us: then the run is marked
 *                           aborted and *timeout_flag_dev = 1), then arms the abort word like pi_op_abort_begin, with the producer's abort bit or-ed in;
 *                           hop_ticks_dev (nullable) accumulates {flag written -> seen, time waited, runs} in 10 ns ticks */
/* A run's results (logits rows, top-8 ids / values, soft-max sums) written into the caller's PAGE-LOCKED host buffers by a kernel on the run's stream, instead of
 * hipMemcpyAsync: on this runtime a device -> host copy of the logits' size does not complete while another stream's kernels keep the device busy (runtime.hip). */
PI_API int     pi_op_copy_out(const float * logits_dev, float * logits_host, int64_t n_logits, const int32_t * ids_dev, int32_t * ids_host, const float * vals_dev, float * vals_host,
                              const float * sum_dev, float * sum_host, int n_tok, void * stream);
PI_API int     pi_op_flag_publish(void * flag_dev, uint32_t ticket, const int32_t * abort_word_dev, int force_abort, void * stream);
PI_API int     pi_op_abort_begin_wait(int32_t * word_dev, int32_t run_id, const int32_t * host_flag, const void * flag_dev, uint32_t ticket, int64_t timeout_us,
                                      int32_t * timeout_flag_dev, uint64_t * hop_ticks_dev, void * stream);

/* Launch accounting for the roofline report.  An instance = (family, weight type, tokens of the launch); families: 0 VALU mat-vec (k_mmvq_*), 1 matrix-core mat-mat
 * (k_mmq_kq), 2 norm + quantise, 3 quantise, 4 activation + quantise, 5 attention, 6 everything else (get_rows, RoPE + KV store, norms, adds, top-k).
 * pi_kernel_timing(1, N) opens a window: every eager launch is counted with its algorithmic bytes and every Nth launch of an instance is timed (HIP events of
 * hipExtLaunchKernelGGL = the kernel's own begin / end timestamps); pi_kernel_timing_rows reads the window after a device sync.  Launches captured into a HIP graph are
 * counted per graph instead (pipeinfer_get_launch_rows in pipeinfer.h: graph contents x replays). */
struct pi_kt_row { int32_t family, type, n_tok, sampled_dropped; /* timed launches left out of avg_us: below a third of the median (they found their run aborted) */ uint64_t shape_bytes; /* mat launches: algorithmic bytes of this launch shape (0 for the other families) */
                   uint64_t launches; double bytes_total; uint64_t sampled; double avg_us, avg_sampled_bytes; };
PI_API void    pi_kernel_timing(int on, int sample_every);
PI_API int     pi_kernel_timing_rows(struct pi_kt_row * out, int cap);
PI_API int     pi_kernel_lifetime_rows(struct pi_kt_row * out, int cap);   /* every eager launch since the process started (counts and bytes, never reset): what a PMC pass of the process sees */

/* host_blocks: ne1 rows of ne0 elements in the GGUF/ggml block format of `type` (row-major, rows contiguous) */
PI_API pi_weight * pi_weight_upload(int type, int64_t ne0, int64_t ne1, const void * host_blocks, void * stream);
PI_API void    pi_weight_free(pi_weight * w);
PI_API size_t  pi_weight_nbytes(const pi_weight * w);  /* HBM bytes == ggml_nbytes of the tensor */
PI_API size_t  pi_weight_copy_nbytes(const pi_weight * w); /* HBM bytes of the second, tile-major copy the matrix-core kernel reads (0 without one) */
/* uploads of the calling thread from now on do (1, default) / do not (0) make that copy: a context that never decodes more than two tokens at once has no use for it */
PI_API void    pi_weight_set_mq_copies(int on);
/* gives a weight that was uploaded without the copy its tile-major copy now, from the same GGUF bytes (no-op when it has one or its type has no matrix-core kernel); 0 ok
   (also when HBM is short: the weight then stays on the VALU kernel), < 0 on a failed copy */
PI_API int     pi_weight_ensure_mq_copy(pi_weight * w, const void * host_blocks, void * stream);
PI_API const void * pi_weight_data(const pi_weight * w); /* device address of the (repacked) tensor bytes */
/* reads [p, p + bytes) once with up to `workgroups` workgroups (0: as many as the bytes need): warms the memory-side Infinity Cache */
PI_API int     pi_op_prefetch(const void * p, size_t bytes, int workgroups, void * stream);
PI_API int     pi_weight_type(const pi_weight * w);

/* type the activations must be quantised to for this weight type (type_traits[].vec_dot_type, ggml.c:415-600) */
PI_API int     pi_vec_dot_type(int weight_type);
/* bytes of the device activation buffer for n_tok rows of k values quantised to vdt (PI_Q8_K, PI_Q8_0 or PI_Q8_1) */
PI_API size_t  pi_q8_nbytes(int vdt, int64_t k, int n_tok);
/* byte offsets of the planes inside that buffer: qs (int8 [n_tok][k]), d (f32 [n_tok][k/256] | f16 [n_tok][k/32] | Q8_1: f32 [n_tok][k/32]), third plane: bsums (int16 [n_tok][k/16], Q8_K) or s = d * sum(qs) (f32 [n_tok][k/32], Q8_1) */
PI_API void    pi_q8_layout(int vdt, int64_t k, int n_tok, size_t * off_qs, size_t * off_d, size_t * off_bsums);
/* Q8_K, n_tok >= 2: byte offsets of the MFMA operand planes the quantisers also write (f16 copies of qs in the lane order of
 * v_mfma_f32_16x16x32_f16, two sub-block pairings, and per-token block sums / scales as floats); layout in runtime.hip */
PI_API void    pi_q8_frag_layout(int64_t k, int n_tok, size_t * off_frag_a, size_t * off_frag_b, size_t * off_aux);

PI_API int pi_op_get_rows(const pi_weight * w, const int32_t * tokens_dev, int n_tok, float * out_dev, void * stream);
PI_API int pi_op_norm(const float * x_dev, const float * w_dev, const float * b_dev, float * y_dev, int n, int n_tok, float eps, int rms, void * stream);
PI_API int pi_op_quantize(int vdt, const float * x_dev, int64_t k, int n_tok, void * q8_dev, void * stream);
/* fusions of the above (same arithmetic, same rounding points): norm -> quantised planes (y_f32_dev nullable: also keep the fp32 result);
 * which = 0: silu(gate) * up, 1: gelu(gate), 2: relu(gate)^2 (LLM_FFN_RELU_SQR) -> quantised planes */
PI_API int pi_op_norm_quantize(const float * x_dev, const float * w_dev, const float * b_dev, float * y_f32_dev, int n, int n_tok, float eps, int rms, int vdt, void * q8_dev, void * stream);
PI_API int pi_op_act_quantize(const float * gate_dev, const float * up_dev, int which, int64_t n, int n_tok, int vdt, void * q8_dev, void * stream);
/* out[t][row] = (residual ? residual[t][row] : 0) + W[row] . act[t]      out/residual: f32 [n_tok][ne1] */
PI_API int pi_op_mul_mat(const pi_weight * w, const void * q8_dev, int n_tok, float * out_dev, const float * residual_dev, void * stream);
/* up to 3 weights of one type and width applied to the same activations in ONE launch (wq/wk/wv, ffn_up/ffn_gate); same results as separate calls */
/* attention projections with RoPE and the KV-cache writes as the mat-vec's epilogue (Llama-style RoPE, K-quant weights):
 *   q16[t][n_head*hd] = f16(rope(Wq.act)),  kc[(kv_head+t)][n_head_kv*hd] = f16(rope(Wk.act)),  vc[(kv_head+t)][..] = f16(Wv.act)
 * == pi_op_mul_mat x3 + pi_op_rope_kv_store, same bytes.  dyn_dev (optional): {kv_head, n_kv} in device memory, as for pi_op_rope_kv_store. */
PI_API int pi_op_mul_mat_qkv_rope_ok(const pi_weight * wq, const pi_weight * wk, const pi_weight * wv, int neox, int head_dim);
PI_API int pi_op_mul_mat_qkv_rope(const pi_weight * wq, const pi_weight * wk, const pi_weight * wv, const void * q8_dev, int n_tok, const int32_t * pos_dev, const void * rope_table,
                                  int head_dim, int n_head, int n_head_kv, int kv_head, const int32_t * dyn_dev, uint16_t * q16_dev, uint16_t * k_cache_dev, uint16_t * v_cache_dev, void * stream);
PI_API int pi_op_mul_mat_multi(int n_w, const pi_weight * const * ws, const void * q8_dev, int n_tok, float * const * outs_dev, const float * const * residuals_dev, void * stream);

/* RoPE tables as the reference iterates them: cos/sin [n_pos][head_dim/2] f32, built on the host (theta *= theta_scale) */
PI_API void *  pi_rope_table_create(int n_pos, int head_dim, float freq_base, float freq_scale, int neox);
/* NeoX mode with n_rot < head_dim (StableLM, llama.cpp:4860-4871): the reference walks head_dim / n_rot blocks of n_rot values, pairs (i, i + n_rot/2) inside a block,
 * theta_scale = freq_base^(-2/n_rot) and theta keeps advancing across the blocks (ggml.c:11133-11160) */
PI_API void *  pi_rope_table_create_ex(int n_pos, int head_dim, int n_rot, float freq_base, float freq_scale, int neox);
PI_API void    pi_rope_table_free(void * table);
/* q/k/v: f32 rows of n_head*hd / n_head_kv*hd values per token, *_stride floats apart (a fused QKV output is passed as three
 * views); writes roped Q as f16 [n_tok][n_head*hd]; K and V as f16 rows of the layer's cache at cells kv_head..
 * (k_cache/v_cache: [n_ctx][n_head_kv*hd] f16 each) */
PI_API int pi_op_rope_kv_store(const float * q_dev, const float * k_dev, const float * v_dev, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                               const int32_t * pos_dev, const void * rope_table,
                               int neox, int head_dim, int n_head, int n_head_kv, int n_tok, int kv_head,
                               uint16_t * q16_dev, uint16_t * k_cache_dev, uint16_t * v_cache_dev, const int32_t * dyn_dev, void * stream);
/* build_persimmon's attention inputs (llama.cpp:4372-4460): qkv_dev rows are [head][q | k | v][64] (fused projection + bias); per-head LayerNorm of Q and K
 * (attn_q_norm / attn_k_norm), NeoX RoPE over the first 32 values of a head (rope_table: pi_rope_table_create_ex(n_pos, 32, 32, ...)), f16 Q and cache rows */
PI_API int pi_op_persimmon_qkv_store(const float * qkv_dev, int n_head, int head_dim, const float * q_norm_w, const float * q_norm_b, const float * k_norm_w, const float * k_norm_b,
                                     float eps, const int32_t * pos_dev, const void * rope_table, int n_tok, int kv_head,
                                     uint16_t * q16_dev, uint16_t * k_cache_dev, uint16_t * v_cache_dev, const int32_t * dyn_dev, void * stream);
/* mask_dev: uint32 bitmap [n_tok][ceil(n_kv/32)], bit i set <=> cell i visible to the token; out f32 [n_tok][n_head*hd].
 * dyn_dev (nullable): int32 {kv_head, n_kv} in device memory that overrides the scalar arguments at run time, so that a
 * captured HIP graph can be replayed for every call (n_kv then only sizes LDS and the mask row stride). */
PI_API int pi_op_attention(const uint16_t * q16_dev, const uint16_t * k_cache_dev, const uint16_t * v_cache_dev, const uint32_t * mask_dev,
                           int head_dim, int n_head, int n_head_kv, int n_tok, int n_kv, float * out_dev, const int32_t * dyn_dev, void * stream);
/* the same for a pass that may carry several runs (per_token_kv != 0): dyn_dev = {kv_head, n_kv of the widest run, n_tok, has_table, cell[n_tok], n_kv[n_tok]}; with
 * has_table set every token attends over the n_kv of ITS run (kv_self.n at that run's start, llama.cpp:5571) */
PI_API int pi_op_attention_ex(const uint16_t * q16_dev, const uint16_t * k_cache_dev, const uint16_t * v_cache_dev, const uint32_t * mask_dev,
                              int head_dim, int n_head, int n_head_kv, int n_tok, int n_kv, float * out_dev, const int32_t * dyn_dev, int per_token_kv, void * stream);
/* ... with ALiBi (llm_build_kqv with max_alibi_bias > 0, llama.cpp:3735-3740; ggml_compute_forward_alibi_f32, ggml.c:10736-10793): score = i * m_k + kq * scale, i the cell's index in
 * the n_kv view, m_k the slope of query head k.  alibi_slopes_dev: f32 [n_head] (pi_alibi_slopes writes the host values), nullable = no bias */
PI_API void pi_alibi_slopes(int n_head, float max_bias, float * out_host);
PI_API int pi_op_attention_alibi(const uint16_t * q16_dev, const uint16_t * k_cache_dev, const uint16_t * v_cache_dev, const uint32_t * mask_dev,
                                 int head_dim, int n_head, int n_head_kv, int n_tok, int n_kv, float * out_dev, const int32_t * dyn_dev, int per_token_kv,
                                 const float * alibi_slopes_dev, void * stream);
PI_API int pi_op_mmq_min_tokens(void);                             /* narrowest batch that pi_op_mul_mat sends to the matrix-core kernel (and its tile-major weight copies) */
PI_API int pi_op_mul_mat_reserve(void);                            /* once per process, outside graph capture: dynamic-LDS limits of the matrix-core kernels */
PI_API int pi_op_attention_max_cells(void);                       /* longest context the attention kernel serves (one score row per workgroup in LDS): 36 200 cells */
PI_API int pi_op_attention_reserve(int head_dim, int n_kv_max);   /* once per context, outside graph capture: LDS limit for long contexts */
PI_API int pi_op_silu_mul(const float * gate_dev, const float * up_dev, float * out_dev, int64_t n, void * stream);
PI_API int pi_op_gelu(const float * x_dev, float * out_dev, int64_t n, void * stream);
PI_API int pi_op_add(const float * a_dev, const float * b_dev, float * out_dev, int64_t n, void * stream);
/* ggml_add with a row broadcast (the bias of a projection, llama.cpp:3628-3630, 3770-3772): x[t][i] += b[i] */
PI_API int pi_op_add_bias(float * x_dev, const float * b_dev, int64_t n, int n_tok, void * stream);
/* ggml_clamp (ggml.c:10893-10931: MAX(MIN(x, max), min); MPT's clamp_kqv, llama.cpp:4755-4757) in place */
PI_API int pi_op_clamp(float * x_dev, float lo, float hi, int64_t n, void * stream);
/* per token: ids/logits of the k largest (logit desc, id asc), max logit and sum_i expf(l_i - max) over the full vocab */
PI_API size_t pi_op_topk_scratch_bytes(int n_tok);                  /* caller-owned scratch, one per stream */
PI_API int pi_op_topk(const float * logits_dev, int n_vocab, int n_tok, int k, int32_t * ids_dev, float * vals_dev, float * sumexp_dev, void * scratch_dev, void * stream);

#ifdef __cplusplus
}
#endif
#endif
