/* pipeinfer.h — extensions beside the llama.h boundary: closed-form KV range operations, the PipeInfer driver as a
 * library call, engine statistics for the roofline report, and a stand-alone handle on the host KV-cell table so its
 * semantics can be tested without a GPU.
 */
#ifndef PIPEINFER_H
#define PIPEINFER_H

#include "llama.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- closed forms of the loops speculative.cpp runs over every sequence id (speculative.cpp:603-612, 1252-1268).
 *      Each equals issuing the single llama_kv_cache_seq_* calls in order (same cells, used count and head). ---- */
LLAMA_API void pipeinfer_kv_cache_seq_rm_range(struct llama_context * ctx, llama_seq_id first, llama_seq_id last, llama_pos p0, llama_pos p1);
LLAMA_API void pipeinfer_kv_cache_seq_cp_range(struct llama_context * ctx, llama_seq_id src, llama_seq_id first, llama_seq_id last, llama_pos p0, llama_pos p1);
LLAMA_API void pipeinfer_kv_cache_seq_rm_cp_range(struct llama_context * ctx, llama_seq_id src, llama_seq_id first, llama_seq_id last, llama_pos p0, llama_pos p1);

/* uploads this process's stage weights and allocates KV cache + scratch now (otherwise done by the first decode) */
LLAMA_API int  pipeinfer_context_prepare(struct llama_context * ctx);
/* queue priority of the context's HIP stream (> 0 high, 0 default, < 0 low); call before the first decode.  The draft context of a
 * single-GPU run wants high: its tokens gate speculation while the target's runs keep the GPU full. */
LLAMA_API int  pipeinfer_context_set_priority(struct llama_context * ctx, int priority);
/* restricts the context's stream to compute units [cu_lo, cu_hi) of the 256 (hipExtStreamCreateWithCUMask); before the first decode.
 * Draft and target contexts on one GPU get disjoint slices so that their kernels overlap instead of alternating. */
LLAMA_API int  pipeinfer_context_set_cu_range(struct llama_context * ctx, int cu_lo, int cu_hi);
/* before the first decode: capture this context's stage as segments of `layers_per_segment` layers (0 = one graph for the whole stage, the
 * default) and launch them from a feeder thread, at most two ahead of the GPU.  A run cancelled by llama_cancel_run while it is in flight
 * then stops at the next segment boundary (the reference aborts between graph nodes, llama.cpp:5665-5687), and contexts with a higher
 * stream priority (pipeinfer_context_set_priority) on the same GPU are served between segments instead of after the whole run. */
LLAMA_API int  pipeinfer_context_set_segments(struct llama_context * ctx, int layers_per_segment);
/* Run coalescing for a segmented context (any time between generations; max_tokens <= n_batch, 0 or 1 = off, the default): its feeder takes every live run waiting in its queue
 * — up to max_tokens tokens — into ONE forward pass.  A pass streams the stage's weights once whatever its width, so two queued runs cost ~1.4x one instead of 2x; every
 * token's logits and KV entries are the bits it would get alone (the kernels are exact at every batch width; a token attends to the cells of its own run's mask; a later
 * run's reads of an earlier run's cells are satisfied inside the pass, like causal attention inside any batch).  The feeder then does not run ahead of the GPU across
 * passes and waits up to PIPEINFER_REACTION_US (10000 = 10 ms) for the caller to react to a finished pass before it composes the next one.  For one GPU that serves both models:
 * the PipeInfer driver keeps a non-speculative and one or more speculative runs queued behind the run in flight (speculative.cpp:1182-1275). */
LLAMA_API int  pipeinfer_context_set_coalescing(struct llama_context * ctx, int max_tokens);
/* 0: launch the decode kernels one by one instead of replaying the captured HIP graph (profiling / per-kernel event timing) */
LLAMA_API void pipeinfer_context_set_graphs(struct llama_context * ctx, int enable);
LLAMA_API const char * pipeinfer_context_error(struct llama_context * ctx);
LLAMA_API int32_t pipeinfer_model_n_layer(const struct llama_model * model);
/* first KV cell find_slot allotted to a run in flight (ticket of llama_start_async_decode); tests */
LLAMA_API int32_t pipeinfer_run_first_cell(const struct ggml_cgraph * ticket);
/* launches of this context's captured graphs since pipeinfer_reset_stats — graph contents x replays, one row per instance (struct pi_kt_row, pipeinfer_ops.h; durations
 * are not part of it: `sampled`, `avg_us` are 0) — i.e. the launch counts and algorithmic bytes of exactly what ran in between.  Returns the number of rows. */
struct pi_kt_row;
LLAMA_API int pipeinfer_get_launch_rows(struct llama_context * ctx, struct pi_kt_row * out, int cap);
/* Run tap (tests: per-run oracle comparison of whatever schedule the context runs — segments, feeder, coalesced passes).  Called on the CALLER's thread, twice per run:
 * from llama_start_async_decode with logits == NULL, and from llama_finish_async_decode with the run's rows [n_tokens][n_vocab] (NULL again when the run produced none:
 * cancelled before or after its start, or a failed launch).  meta = the run's own metadata block (tokens | pos | {kv_head, n_kv, n_tokens, 0} | mask rows), valid during the call. */
typedef void (*pipeinfer_run_tap_fn)(void * user, int32_t run_id, int32_t batch_id, int32_t n_tokens, const void * meta, size_t meta_bytes, const float * logits);
LLAMA_API void pipeinfer_context_set_run_tap(struct llama_context * ctx, pipeinfer_run_tap_fn fn, void * user);

/* device-side top-8 of row i of the last finished run: ids/logits sorted (logit desc, id asc), sumexp = sum_j expf(l_j - max) */
/* ---- multi-process pipeline (one process per GPU) -------------------------------------------------------------------------
 * Replaces the reference's MPI plumbing (ggml-mpi.c:186-310 pipelined sync, :541-557 layer ranges; llama.cpp:5588-5760 phases):
 * the HEAD process owns the driver and the KV-cell bookkeeping of the target context; it ships one metadata block per run
 *     int32 tokens[n] | int32 pos[n] | int32 {kv_head, n_kv, n_tokens, 0} | uint32 mask[n][ceil(n_ctx/32)]
 * through `start`; STAGE processes own layer ranges and run pipeinfer_stage_run on that block, passing activations
 * [n_tokens][n_embd] fp32 from stage to stage.  Any transport can sit behind the callbacks below; the library's own (shared-memory rings +
 * peer stores, see pipeinfer_transport_* further down) is what pipeinfer_amd/pipeline.py and bench.py --gpus N use. */
struct pipeinfer_remote_ops {
    void * user;
    int  (*start)(void * user, int32_t run_seq, int32_t batch_id, int32_t n_tokens, const void * meta, size_t meta_bytes);
    int  (*poll)(void * user);                      /* 1 when the oldest outstanding run's result has arrived (llama_mpi_iprobe) */
    /* blocks for run_seq (runs finish in start order): 1 = computed, 0 = skipped (cancelled), < 0 error */
    int  (*wait)(void * user, int32_t run_seq, float * logits, int32_t * topk_ids, float * topk_vals, float * topk_sum);
    void (*cancel)(void * user, int32_t batch_id);  /* llama_cancel_run; batch_id -2: a new generation starts, batch ids restart */
};
/* turns ctx into the head-side proxy of a remote pipeline: no weights, no device memory */
LLAMA_API void pipeinfer_context_set_remote(struct llama_context * ctx, const struct pipeinfer_remote_ops * ops);
/* before the first decode: this process owns layers [l0, l1), the embedding if is_first, final norm + output if is_last */
LLAMA_API int  pipeinfer_context_set_stage(struct llama_context * ctx, int32_t l0, int32_t l1, int32_t is_first, int32_t is_last);
/* layer ranges of the reference's --mpi-layer-split (ggml-mpi.c:541-557): ranges[2*i], ranges[2*i+1] = [start, end) of stage i */
LLAMA_API void pipeinfer_split_layers(int32_t n_layer, const float * weights, int32_t n_stage, int32_t * ranges);
/* llama_split_layers_weighted (llama.cpp:8929-8938) for ONE stage process: the rule above on the model's layer count, then set_stage for
 * stage `stage` of `n_stage` (first stage: embedding, last: norm + output).  l0_out / l1_out (optional) receive the range. */
LLAMA_API int  pipeinfer_context_set_stage_weighted(struct llama_context * ctx, const float * weights, int32_t n_stage, int32_t stage, int32_t * l0_out, int32_t * l1_out);
/* one run on this stage; x_in_dev / x_out_dev: device fp32 [n_tokens][n_embd] (ignored on the first / last stage); the last
 * stage fills the host buffers.  Returns after the stage's stream is idle. */
LLAMA_API int  pipeinfer_stage_run(struct llama_context * ctx, int32_t n_tokens, const void * meta, size_t meta_bytes, const float * x_in_dev, float * x_out_dev,
                                   float * logits_host, int32_t * topk_ids, float * topk_vals, float * topk_sum);

/* pipeinfer_stage_run for a segmented stage context (pipeinfer_context_set_segments): should_abort(user) is asked before every segment after
 * the first; when it answers 1 the rest of the run is dropped and the call returns 1 (nothing is written to x_out / the host buffers). */
LLAMA_API int  pipeinfer_stage_run_abortable(struct llama_context * ctx, int32_t n_tokens, const void * meta, size_t meta_bytes, const float * x_in_dev, float * x_out_dev,
                                             float * logits_host, int32_t * topk_ids, float * topk_vals, float * topk_sum, int (*should_abort)(void * user), void * user);

/* ---- in-library transport of a one-node pipeline (csrc/engine/transport.cpp) ------------------------------------------------------------
 * rank 0 = head (driver + proxy target context), ranks 1..world-1 = target stages in layer order.  One POSIX shared-memory segment per
 * job carries the run metadata table, the header rings, the cancel table and the result ring; activations travel as peer stores into a
 * ring in the next stage's HBM (hipIpc), or through rings in the segment when `device` < 0 or the peer mapping is not available.
 * Replaces ggml_mpi_sync_pipelined / ggml_mpi_sync_pipelined_back / ggml_mpi_cancel_run (ggml-mpi.c:186-310) for the speculative path. */
typedef struct pipeinfer_transport pipeinfer_transport;
struct pipeinfer_transport_params {
    const char * name;           /* shared-memory name, unique per job (every rank passes the same) */
    int32_t rank, world;
    int32_t n_embd, n_vocab;     /* of the target model */
    int32_t n_batch, n_ctx;      /* of the target contexts (sizes of the metadata blocks, activation and result slots) */
    int32_t device;              /* HIP device of this process, or -1: activations in host memory (processes without a GPU) */
    int32_t timeout_ms;          /* set-up and result waits (0: 120 s) */
};
struct pipeinfer_transport_stats {
    uint64_t runs, skipped, cut_short, cancel_notes;   /* stage: computed / forwarded as skipped (cut_short included) / stopped inside the run / notes seen; head: started / came back skipped / - / notes written */
    double   wait_in_ms, compute_ms, send_ms;          /* stage: waiting for a header, inside the backend, between the end of a run and its header being published downstream (host side of the hop) */
    uint64_t received;                                 /* stage: headers taken from the incoming ring == runs + skipped; head: results consumed (== started once everything is back) */
    double   wait_out_ms;                              /* stage: back-pressure — waiting for a free slot in the next stage's ring (or the result ring) before a run */
    double   hop_gpu_ms;                               /* stage: GPU time of the in-stream store of the activations into the next stage's ring (last stage: of the logits into the result slot) */
    int32_t  peer_access;                              /* stage: what hipDeviceCanAccessPeer said for (this device -> next stage's device); -1: not asked (same device, host rings, last stage) */
    int32_t  device_flags;                             /* stage: 1 = runs cross the stage boundaries on device-side arrival flags (announced early, enqueued behind a device wait); 0 = headers published by the host after the run */
    uint64_t flag_runs;                                /* stage: runs that waited for their activations on the device */
    double   flag_wait_gpu_ms;                         /* stage: GPU time those runs' wait kernels spent waiting for their input (pipeline fill + the upstream stage's compute) */
    double   flag_seen_us;                             /* stage: sum over those runs of (flag seen by the waiting kernel - flag stamped by the producing stream), device clock: valid between stages on ONE device */
    int32_t  hop_store_kernel;                         /* stage: 1 = the activations are stored by a kernel of the producing stream (device rings), 0 = by hipMemcpyAsync (host rings, PIPEINFER_HOP_STORE=memcpy); decided and logged at set-up */
    int32_t  ring_alloc;                               /* stage: this stage's activation ring: 0 none (first stage / host rings), 1 coarse-grained hipMalloc, 2 fine-grained, 3 uncached device memory */
    uint64_t passes_of_several_runs;                   /* stage that is the whole target, run coalescing on: forward passes that carried more than one run */
};
/* stage compute behind the transport when it is not a llama_context of this library (the CPU tests plug the oracle in here):
 * 0 = computed, 1 = dropped, < 0 error; x_in / x_out are host pointers when the transport runs on host rings */
struct pipeinfer_stage_backend {
    void * user;
    int (*run)(void * user, int32_t n_tokens, const void * meta, size_t meta_bytes, const float * x_in, float * x_out,
               float * logits, int32_t * topk_ids, float * topk_vals, float * topk_sum);
};
/* collective over the job's ranks: returns when every rank has joined and the activation path has been agreed on; NULL on failure */
LLAMA_API pipeinfer_transport * pipeinfer_transport_open(const struct pipeinfer_transport_params * params);
LLAMA_API void pipeinfer_transport_close(pipeinfer_transport * tp);
LLAMA_API const char * pipeinfer_transport_error(pipeinfer_transport * tp);
LLAMA_API int  pipeinfer_transport_device_mode(pipeinfer_transport * tp);          /* 1: peer stores into HBM rings, 0: host rings */
/* rank 0: makes ctx the head-side proxy of the pipeline (pipeinfer_context_set_remote with the transport's own callbacks) */
LLAMA_API int  pipeinfer_transport_attach_head(pipeinfer_transport * tp, struct llama_context * ctx);
/* rank 0, once every started run has been waited for: ends the stages' serve loops */
LLAMA_API int  pipeinfer_transport_shutdown(pipeinfer_transport * tp);
LLAMA_API void pipeinfer_transport_get_stats(pipeinfer_transport * tp, struct pipeinfer_transport_stats * out);      /* head-side counters */
/* ranks >= 1: serves runs until the head shuts the pipeline down.  Compute = pipeinfer_stage_run_abortable on stage_ctx (backend NULL),
 * or the backend's run().  A stage stores its activations, in stream order behind its last kernel, straight into a slot of the NEXT stage's ring (reserved before the
 * run starts) and then publishes the 32-byte header: one device-to-device copy and one flag per hop, no staging buffer, no sender thread.  A watcher thread follows the cancel
 * table while a run computes and stops it at the next kernel (pi_op_abort_cancel). */
LLAMA_API int  pipeinfer_transport_serve(pipeinfer_transport * tp, struct llama_context * stage_ctx, const struct pipeinfer_stage_backend * backend, struct pipeinfer_transport_stats * out);

/* Offline quantisation of one run of fp32 values into `type` (ggml type id: 0 f32, 1 f16, 2 Q4_0, 3 Q4_1, 6 Q5_0, 7 Q5_1, 8 Q8_0, 10..14 Q2_K..Q6_K) with the
 * reference's deterministic quantize_row_*_reference arithmetic (ggml-quants.c:428-640, 1458-2316) — what llama_model_quantize applies to every row.
 * k: number of values, a multiple of the block size.  Returns the bytes written, -1 on a bad type / length. */
LLAMA_API int64_t pipeinfer_quantize_row(int32_t type, const float * x, void * y, int64_t k);
/* the inverse direction (dequantize_row_*, ggml-quants.c:1074-1188, 1535-2310): k stored values of `type` as fp32, bit-identical to the reference.  Returns k or -1. */
LLAMA_API int64_t pipeinfer_dequantize_row(int32_t type, const void * x, float * y, int64_t k);
LLAMA_API int  pipeinfer_get_topk(struct llama_context * ctx, int32_t i, llama_token * ids, float * logits, float * sumexp);

struct pipeinfer_stats {
    uint64_t n_calls;            /* decode calls issued on this context */
    double   gpu_ms;             /* sum over calls of (done event - start event) on the context's stream */
    uint64_t alg_bytes;          /* algorithmic HBM bytes of those calls (SURVEY 8d formula) */
    uint64_t weight_bytes;       /* bytes of this stage's weights resident in HBM (== ggml_nbytes: what one call streams) */
    uint64_t kv_bytes_per_cell;  /* K+V bytes per cached token for this stage's layers */
    uint64_t n_calls_1;          /* the same three counters restricted to single-token calls (prompt batches excluded) */
    double   gpu_ms_1;
    uint64_t alg_bytes_1;
    uint64_t n_segments, n_segments_skipped;   /* segmented launch: segments launched / dropped by a mid-run cancel */
    uint64_t n_cancel_runs;      /* runs finished as cancelled (before or after their start) */
    uint64_t n_cancel_runs_started; /* ... of which llama_cancel_run arrived after llama_start_async_decode: their kernels return on entry from then on (abort word) */
    double   cancel_gpu_ms;      /* GPU time (first to last event) those started-then-cancelled runs still took */
    uint64_t n_passes, n_pass_runs, n_pass_tokens, n_coalesced_passes;   /* feeder contexts: forward passes computed in full, the runs / tokens they carried, passes of more than one run */
    uint64_t weight_copy_bytes;  /* HBM held by the tile-major copies of the matrix-core kernel (0 for contexts with n_batch <= 2: they are not made) */
    uint64_t n_reaction_windows;          /* feeder contexts with run coalescing: reaction windows opened behind a pass (only passes that put work on the GPU open one; cancelled runs at the queue head are completed on the spot) */
};
LLAMA_API void pipeinfer_get_stats(struct llama_context * ctx, struct pipeinfer_stats * out);
LLAMA_API void pipeinfer_reset_stats(struct llama_context * ctx);    /* call with no run in flight (it also zeroes the device-side running sums) */
/* launches since pipeinfer_reset_stats that found their run cancelled (llama_cancel_run after the start, llama.cpp:9981-9993) and returned on entry — what a cancelled run
 * stops doing (llama.cpp:5665-5687) — per kernel family (index = pi_kt_row.family in pipeinfer_ops.h).  Returns the families written (<= cap), -1 on error */
LLAMA_API int  pipeinfer_get_aborted_launches(struct llama_context * ctx, uint64_t * out, int cap);
/* schedule timeline (a measurement aid; recording is on only when the process was started with PI_TIMELINE set): host-clock events of the caller's thread and the target's
 * feeder — start / probe / finish / cancel of runs, what each forward pass carried, when it left the GPU, why the feeder's reaction window ended — written as
 * "t_us,tag,a,b,c" lines; the events are forgotten afterwards.  Returns the number of events (path NULL: only forgets them), -1 when the file cannot be written */
LLAMA_API int  pipeinfer_timeline_dump(const char * path);

/* ---- host KV-cell table on its own (no GPU needed): used by the CPU tests ---- */
typedef struct pi_kvcells pi_kvcells;
LLAMA_API pi_kvcells * pi_kvcells_new(int n_ctx);
LLAMA_API void pi_kvcells_free(pi_kvcells * c);
LLAMA_API int  pi_kvcells_decode(pi_kvcells * c, int n_tokens, const llama_pos * pos, const int32_t * n_seq_id, const llama_seq_id * seq_ids_flat, int stride); /* pre_decode + find_slot + post_decode; returns slot head or -1 */
LLAMA_API void pi_kvcells_seq_rm(pi_kvcells * c, int seq, int p0, int p1);
LLAMA_API void pi_kvcells_seq_cp(pi_kvcells * c, int src, int dst, int p0, int p1);
LLAMA_API void pi_kvcells_seq_keep(pi_kvcells * c, int seq);
LLAMA_API void pi_kvcells_clear(pi_kvcells * c);
LLAMA_API void pi_kvcells_seq_rm_range(pi_kvcells * c, int first, int last, int p0, int p1);
LLAMA_API void pi_kvcells_seq_cp_range(pi_kvcells * c, int src, int first, int last, int p0, int p1);
LLAMA_API void pi_kvcells_seq_rm_cp_range(pi_kvcells * c, int src, int first, int last, int p0, int p1);
LLAMA_API int  pi_kvcells_head(const pi_kvcells * c);
LLAMA_API int  pi_kvcells_used(const pi_kvcells * c);
LLAMA_API int  pi_kvcells_n_attend(const pi_kvcells * c);
LLAMA_API int  pi_kvcells_pos(const pi_kvcells * c, int i);
LLAMA_API int  pi_kvcells_has(const pi_kvcells * c, int i, int seq);
LLAMA_API int  pi_kvcells_nseq(const pi_kvcells * c, int i);
LLAMA_API void pi_kvcells_mask(const pi_kvcells * c, int n_kv, int n_tokens, const llama_pos * pos, const int32_t * n_seq_id, const llama_seq_id * seq_ids_flat, int stride, uint32_t * out_bits);

/* ---- the PipeInfer driver (examples/speculative/speculative.cpp re-stated as a library call) ---- */
struct pipeinfer_spec_params {
    int32_t n_predict;       /* -n   tokens to generate */
    int32_t n_draft;         /* --draft */
    int32_t n_parallel;      /* -np  tree branches per speculative run */
    float   p_accept;        /* -pa */
    float   p_split;         /* -ps */
    float   p_recovery;      /* -pr */
    float   p_decay;         /* -pd */
    float   temp;            /* --temp (<= 0: greedy; < 0 also computes probabilities) */
    int32_t ignore_eos;      /* --ignore-eos */
    int32_t mode;            /* 0 = PipeInfer (async, continuous speculation: examples/speculative), 1 = synchronous tree speculation
                              * (examples/speculative_orig: draft up to n_draft steps, verify the tree in one target batch),
                              * 2 = target only (greedy, no draft model) */
    int32_t max_inflight;    /* cap on target runs in flight (0 = unlimited, the reference behaviour) */
    int32_t min_spec_iters;  /* speculation iterations to run even when the target is already back (0 = reference behaviour) */
    int32_t verbose;
    /* our addition (0 = off = the reference's behaviour): when `adaptive_window` accepted tokens in a row passed without a single
     * speculative run being launched *because the target run was already back when the speculation loop started* (the draft never got
     * ahead of the target — the normal state when both share one GPU; an unsure draft or a lack of free partitions does not count), the
     * draft model is left alone for `adaptive_pause` tokens (doubling up to 16x while that keeps happening) and then caught up with
     * one batched decode before speculation is tried again.  Output tokens are unaffected (greedy verification is lossless).
     * adaptive_window < 0 (tests): pause after every |window| tokens whether or not speculation is happening. */
    int32_t adaptive_window, adaptive_pause;
};
struct pipeinfer_spec_result {
    int32_t n_input, n_predict, n_drafted, n_accept;       /* the reference's counters (speculative.cpp:681-718) */
    int32_t n_accept_clean;                                 /* drafted tokens that ended up in the output */
    int32_t n_drafted_tokens;                               /* drafted tokens sent to the target for verification */
    int32_t n_runs_spec, n_runs_nonspec, n_runs_canceled;
    int32_t n_draft_paused, n_draft_catchups;              /* adaptive pause: tokens the draft sat out, batched catch-up decodes */
    double  t_enc_s, t_dec_s, ttft_s, itl_s;
    double  decode_tps, encode_tps;
};
LLAMA_API struct pipeinfer_spec_params pipeinfer_spec_default_params(void);
LLAMA_API int pipeinfer_speculative(struct llama_context * ctx_tgt, struct llama_context * ctx_dft, const struct pipeinfer_spec_params * params,
                                    const llama_token * prompt, int32_t n_prompt, llama_token * out_tokens, int32_t out_cap, struct pipeinfer_spec_result * result);

#ifdef __cplusplus
}
#endif
#endif
