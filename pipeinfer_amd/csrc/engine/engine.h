// Engine internals behind include/llama.h: model (GGUF -> HBM), context (KV cache, streams, in-flight runs) and the
// per-stage forward pass that strings the HIP operators together in the order the reference graphs use
// (build_llama llama.cpp:3872-3983, build_falcon llama.cpp:4106-4227).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/llama.h"
#include "../../../include/pipeinfer.h"
#include "../../../include/pipeinfer_ops.h"
#include "gguf.h"
#include "kv_cells.h"
#include "vocab.h"
#include "../kernels/kernel_timing.h"

// LLAMA and FALCON have their own blocks in forward(); the others are one generic sequential block (norm -> attention -> + -> norm -> MLP -> +) described by pi_block below
enum pi_arch { PI_ARCH_LLAMA = 0, PI_ARCH_FALCON = 1, PI_ARCH_BAICHUAN = 2, PI_ARCH_STARCODER = 3, PI_ARCH_REFACT = 4, PI_ARCH_BLOOM = 5, PI_ARCH_MPT = 6, PI_ARCH_STABLELM = 7, PI_ARCH_PERSIMMON = 8 };

// what the reference's build_* functions differ in (llama.cpp:3985-4104 baichuan, 4229-4327 starcoder, 4540-4630 refact, 4632-4725 bloom, 4727-4825 mpt, 4827-4939 stablelm)
struct pi_block {
    bool rms = false;            // LLM_NORM_RMS or LLM_NORM (weights, and biases where the file has them)
    bool fused_qkv = false;      // attn_qkv [q | k | v] or attn_q / attn_k / attn_v
    int rope = -1;               // -1 none, 0 pairs (2i, 2i+1), 2 NeoX over n_rot
    float alibi = 0.0f;          // max_alibi_bias (> 0: ggml_alibi on the scaled scores)
    bool gated = false;          // SiLU(gate) * up (LLM_FFN_PAR) or GELU(up) (LLM_FFN_SEQ)
    bool relu_sqr = false;       // ... or relu(up)^2 (LLM_FFN_RELU_SQR: Persimmon)
    bool qk_norm = false;        // Persimmon: fused QKV rows are [head][q | k | v], Q and K get a per-head LayerNorm, RoPE covers the first half of a head (llama.cpp:4372-4460)
    float clamp_kqv = 0.0f;
};

struct pi_hparams {
    pi_arch arch = PI_ARCH_LLAMA;
    int n_vocab = 0, n_ctx_train = 0, n_embd = 0, n_head = 0, n_head_kv = 0, n_layer = 0, n_ff = 0, n_rot = 0;
    float f_norm_eps = 0.0f, f_norm_rms_eps = 0.0f, rope_freq_base_train = 10000.0f, rope_freq_scale_train = 1.0f;
    float f_max_alibi_bias = 0.0f, f_clamp_kqv = 0.0f;
    pi_block block;
    int head_dim() const { return n_embd / n_head; }
    int n_embd_gqa() const { return head_dim() * n_head_kv; }
    bool generic() const { return arch != PI_ARCH_LLAMA && arch != PI_ARCH_FALCON; }
};

struct pi_layer {
    pi_weight * wq = nullptr, * wk = nullptr, * wv = nullptr, * wo = nullptr, * wqkv = nullptr;
    pi_weight * ffn_gate = nullptr, * ffn_down = nullptr, * ffn_up = nullptr;
    float * attn_norm = nullptr, * attn_norm_b = nullptr, * attn_norm_2 = nullptr, * attn_norm_2_b = nullptr, * ffn_norm = nullptr, * ffn_norm_b = nullptr;
    float * bqkv = nullptr, * bo = nullptr, * ffn_up_b = nullptr, * ffn_down_b = nullptr;       // biases of the projections (StarCoder, Bloom, Persimmon)
    float * attn_q_norm = nullptr, * attn_q_norm_b = nullptr, * attn_k_norm = nullptr, * attn_k_norm_b = nullptr;   // Persimmon's per-head LayerNorm of Q and K (64 values)
    bool loaded = false;
    bool has_mq = false;                      // its matrices carry the tile-major copies of the matrix-core kernel (made when a context whose passes can be >= 3 tokens wide loads the layer)
};

struct llama_model {
    pi::gguf_file file;
    pi_hparams hparams;
    pi::vocab vocab;
    std::string name, path;
    int device = 0;
    uint64_t n_elements = 0, n_bytes = 0;
    pi_weight * tok_embd = nullptr, * output = nullptr, * pos_embd = nullptr;                 // pos_embd: StarCoder's learned positions (llama.cpp:4251-4255)
    float * output_norm = nullptr, * output_norm_b = nullptr, * tok_norm = nullptr, * tok_norm_b = nullptr;   // tok_norm: Bloom's LayerNorm on the embeddings (llama.cpp:4650-4654)
    float * alibi_slopes = nullptr;                                                              // f32 [n_head] on the device when block.alibi > 0
    std::vector<pi_layer> layers;
    std::vector<void *> f32_allocs;
    struct matrix_ref { pi_weight * w; const void * host; int layer; };                           // every uploaded matrix with its GGUF bytes (layer -1: embeddings / output head)
    std::vector<matrix_ref> matrices;
    bool output_has_mq = false;
    size_t hbm_bytes = 0;                     // bytes resident in HBM for this process's stage(s)
    size_t hbm_copy_bytes = 0;                // ... plus the tile-major copies the matrix-core kernel reads (pi_weight::mq)

    ~llama_model();
    // uploads (once) token_embd if first, layers [l0,l1), final norm + output if last
    // want_mq: also make the tile-major copies of the K-quant matrices (contexts whose widest pass reaches the matrix-core kernel)
    bool ensure_loaded(int l0, int l1, bool first, bool last, bool want_mq, std::string & err);
    size_t stage_weight_bytes(int l0, int l1, bool first, bool last, int n_tok) const;
};

struct pi_run;
struct pi_pass {                               // the runs one forward pass of a feeder context carries (usually one; several when the feeder coalesces its queue)
    std::vector<pi_run *> runs;                // members not yet finished by the caller (a finished run leaves the list under fmu: its object is recycled)
    int32_t id = 0; int n_tokens = 0, n_kv = 0;
};

struct pi_run {                                // one llama_start_async_decode .. llama_finish_async_decode
    int32_t batch_id = 0;
    int n_tokens = 0;
    bool canceled = false, launched = false, finished = false, last_only = false;
    std::vector<int8_t> want_logits;
    hipEvent_t done = nullptr, begin = nullptr;
    float * logits_host = nullptr;            // pinned [n_tokens][n_vocab]
    size_t logits_cap = 0;
    int32_t * topk_ids = nullptr; float * topk_vals = nullptr; float * topk_sum = nullptr;   // pinned [n_tokens][8], [n_tokens]
    uint8_t * meta_host = nullptr; size_t meta_cap = 0;                                       // pinned staging for tokens/pos/mask
    int64_t t_start_us = 0;
    pi_kt_sink * sink = nullptr;            // kernel-timing samples of the graph this run replayed
    int32_t run_seq = 0;                    // remote contexts: sequence number of the run on the wire
    bool host_plain = false;                // buffers come from malloc (remote contexts never touch the GPU)
    // feeder contexts (segmented launch): the caller thread only places the run in the queue; the feeder thread enqueues its segments
    std::atomic<int>  state{0};             // 0 queued, 1 being launched, 2 completion event recorded (the run's fields are final)
    std::atomic<bool> cancel_req{false};    // llama_cancel_run after the start: the feeder stops launching its segments
    int segments_run = 0;                   // segments that were launched before a cancel took effect (statistics)
    std::shared_ptr<pi_pass> pass;          // feeder contexts: the pass that carries this run (assigned under fmu when the feeder takes the run)
    int row0 = 0;                           // first row of this run inside its pass (logits / top-k rows, token order)
    bool pass_leader = true;                // the pass's GPU time is accounted to its first run only
    int pass_tokens = 0, n_kv = 0;          // tokens of the whole pass; cells this run attends over (kv_self.n at its start)
    int32_t run_id = 0;                     // per-context counter: the identity the device-side abort word carries (pi_op_abort_begin / _cancel)
    bool failed = false;                    // a launch of this run failed (distinct from cancelled / cut short): finish reports -1 and gives the cells back
    int kv_head = 0;                        // first cell find_slot allotted (released again when the run failed)
    size_t meta_bytes = 0;                  // size of the run's own metadata block in meta_host
};

// The context's error text is written by the caller's thread AND by the feeder thread of a segmented context (a failed launch inside a pass): every access locks.
// Behaves like the std::string it replaces where the engine assigns, tests for emptiness and concatenates.
class pi_error_text {
    mutable std::mutex mu; std::string s;
public:
    pi_error_text & operator=(const std::string & v) { std::lock_guard<std::mutex> lk(mu); s = v; return *this; }
    pi_error_text & operator=(const char * v) { std::lock_guard<std::mutex> lk(mu); s = v; return *this; }
    pi_error_text & operator=(const pi_error_text & o) { if (this != &o) { const std::string v = o.str(); std::lock_guard<std::mutex> lk(mu); s = v; } return *this; }
    bool empty() const { std::lock_guard<std::mutex> lk(mu); return s.empty(); }
    std::string str() const { std::lock_guard<std::mutex> lk(mu); return s; }
    operator std::string() const { return str(); }
};
inline std::string operator+(const std::string & a, const pi_error_text & b) { return a + b.str(); }
inline std::string operator+(const char * a, const pi_error_text & b) { return std::string(a) + b.str(); }

struct llama_context {
    llama_model * model = nullptr;
    llama_context_params cparams;
    int n_ctx = 0, n_batch = 512;
    // pipeline stage owned by this process (single process: the whole model)
    int l0 = 0, l1 = 0; bool is_first = true, is_last = true;
    int node_id = 0; bool comm_swapped = false; int comm_color = 0, comm_color_alt = 0;
    // head of a multi-process pipeline: this context only keeps the KV-cell table and ships each run's metadata through
    // the callbacks (include/pipeinfer.h); the layers run in other processes (pipeinfer_stage_run)
    bool remote = false; pipeinfer_remote_ops remote_ops{}; int32_t remote_seq = 0;

    std::unique_ptr<pi::kv_cells> cells;
    uint16_t * k_cache = nullptr, * v_cache = nullptr;        // [n_layer_local][n_ctx][n_embd_gqa] f16 each
    void * rope = nullptr;
    hipStream_t stream = nullptr;
    int stream_priority = 0;                 // > 0 high, < 0 low (pipeinfer_context_set_priority, before the first decode)
    int cu_lo = 0, cu_hi = 0;                // compute units [cu_lo, cu_hi) reserved for this context's stream (0,0: all)

    // device scratch, sized for n_batch tokens
    float * x = nullptr, * x2 = nullptr, * nx = nullptr, * nx2 = nullptr, * qkv = nullptr, * att = nullptr, * up = nullptr, * gate = nullptr, * logits_dev = nullptr;
    uint16_t * q16 = nullptr;
    void * q8a = nullptr, * q8b = nullptr, * q8c = nullptr;
    int32_t * tokens_dev = nullptr, * pos_dev = nullptr, * dyn_dev = nullptr; uint32_t * mask_dev = nullptr;
    struct graph_entry { hipGraphExec_t exec = nullptr; pi_kt_sink sink; std::atomic<uint64_t> replays{0}; };     // sink: what the graph launches (kernel_timing.h); replays since the last reset_stats
    std::map<std::pair<int, int>, graph_entry> graphs;   // one captured graph per (n_tokens, segment) (kv_head / n_kv come from dyn_dev)
    std::mutex graphs_mu;                    // the map itself (lookups / first-use inserts by the launching thread, the statistics readers on the caller's thread); entries are stable
    bool use_graphs = true;
    // Segmented launch (pipeinfer_context_set_segments): the stage's layers are captured as segments of `seg_layers` layers and a feeder
    // thread enqueues them one or two ahead of the GPU.  Between segments it (a) drops the rest of a run that has been cancelled meanwhile
    // — the reference polls its abort callback between graph nodes (llama.cpp:5665-5687, ggml.c:15748-15811) — and (b) lets contexts of
    // higher stream priority go first: on ONE GPU a kernel of another queue is only dispatched between this queue's kernels once its
    // workgroups have all been placed, so a draft step behind a queued 70B run takes as long as that run; with at most two segments
    // queued it waits for two segments.
    int seg_layers = 0;                      // 0: the whole stage is one graph, launched by the caller (no feeder)
    std::thread feeder; std::mutex fmu; std::condition_variable fcv; std::deque<pi_run *> fqueue; bool fstop = false;
    int urgent_pending = 0;                 // runs of this (priority) context counted in the device arbiter and not finished yet (caller's thread only)
    std::atomic<uint64_t> stat_segments{0}, stat_segments_skipped{0};     // written by the feeder thread (or the stage loop), read by the caller: atomics
    hipEvent_t seg_events[2] = { nullptr, nullptr };
    // Kernel-granular early cancellation (reference: abort callback between graph nodes, ggml.c:15748-15811; llama.cpp:5614-5628, 5665-5687, 9981-9993):
    // every kernel of this context reads abort_dev = (id of the run whose kernels are executing << 1 | aborted) on entry.  A run's launches are preceded, in
    // stream order, by pi_op_abort_begin (which also picks up a cancel that arrived before: abort_host[id & 255], pinned); llama_cancel_run of a run already
    // handed to the GPU sets that flag and enqueues pi_op_abort_cancel on ctl_stream — a compare-and-swap that only hits while that very run is executing.
    // Run coalescing (pipeinfer_context_set_coalescing): the feeder of a segmented context takes EVERY live run waiting in its queue — up to coalesce_max tokens — into one
    // forward pass: a multi-token pass streams the stage's weights once, whatever the number of tokens (the matrix-core kernel is flat from 3 to 16), so two queued runs
    // cost about 1.4x one instead of 2x.  Every token's result is what it would be alone: the kernels are bit-exact at every batch width, a token attends to exactly the
    // cells of ITS run's mask, and a later run's reads of an earlier run's cells are satisfied inside the pass like causal attention inside any batch (K/V of a layer are
    // stored for all tokens before that layer's attention).  The feeder does not run ahead of the GPU across passes and gives the caller a short window to react to a
    // finished pass (cancel / start follow-up runs) before it composes the next one.
    std::atomic<int> coalesce_max{0};       // may change between generations (the feeder reads it per pass)
    std::atomic<uint64_t> caller_epoch{0}; std::atomic<int> waiters{0};
    std::atomic<int32_t> last_finished_id{0}; std::atomic<uint64_t> epoch_after_finish{0};
    hipEvent_t pass_event = nullptr; bool pass_pending = false; int32_t pass_first_id = 0; int64_t pass_fed_us = 0;      // feeder thread only
    uint8_t * pass_host[4] = { nullptr, nullptr, nullptr, nullptr }; size_t pass_host_cap = 0; int pass_slot = 0;   // pinned staging blocks of composed passes
    std::atomic<uint64_t> stat_passes{0}, stat_pass_runs{0}, stat_pass_tokens{0}, stat_coalesced_passes{0};     // feeder thread writes, caller reads
    std::atomic<uint64_t> stat_reaction_windows{0};          // reaction windows opened behind a pass (feeder thread writes, caller reads)
    int32_t * abort_dev = nullptr; int32_t * abort_host = nullptr; hipStream_t ctl_stream = nullptr; int32_t run_counter = 0;
    std::atomic<int32_t> stage_run_id{0};    // pipeline stages (stage_run): id of the run being computed, 0 between runs (read by the transport's cancel watcher)
    double stat_hop_ms = 0.0;                // pipeline stages: GPU time of the in-stream store of the activations into the next stage's ring
    // arrival flags (pi_stage_hop): device block {int32 timed_out, -, uint64 ticks[3] = flag written -> seen, waited, runs} and its pinned mirror + the abort word read back after a run
    uint8_t * hop_dev = nullptr; uint8_t * hop_host = nullptr;
    double stat_flag_seen_us = 0.0, stat_flag_wait_ms = 0.0; uint64_t stat_flag_runs = 0;        // sums over the runs that waited on a flag
    uint64_t stat_cancel_runs = 0, stat_cancel_runs_started = 0; double stat_cancel_gpu_ms = 0.0;     // cancelled runs: all / handed to the GPU before the cancel / GPU time they still took
    bool prepared = false, prepare_failed = false;      // device state fully allocated / an attempt failed half-way (never retried)
    int32_t * topk_ids_dev = nullptr; float * topk_vals_dev = nullptr, * topk_sum_dev = nullptr; void * topk_scratch_dev = nullptr;
    size_t mask_cap_words = 0;

    std::deque<pi_run *> inflight;            // oldest at front
    std::vector<pi_run *> pool;
    std::set<int32_t> canceled_batches;
    std::vector<float> logits;                // API-visible buffer (llama_get_logits)
    std::vector<int32_t> last_topk_ids; std::vector<float> last_topk_vals, last_topk_sum; int last_n_tokens = 0;

    // timings (llama.cpp:10170-10206)
    int64_t t_start_us = 0, t_load_us = 0, t_eval_us = 0, t_p_eval_us = 0, t_sample_us = 0;
    int32_t n_eval = 0, n_p_eval = 0, n_sample = 0;
    // roofline accounting
    mutable size_t alg_w_fixed = 0, alg_w_per_tok = 0; mutable std::tuple<int, int, bool, bool> alg_w_key{-1, -1, false, false};   // decode_alg_bytes: the stage's weight bytes, computed once per (l0, l1, first, last)
    uint64_t stat_calls = 0; double stat_gpu_ms = 0.0; uint64_t stat_alg_bytes = 0;
    uint64_t stat_calls_1 = 0; double stat_gpu_ms_1 = 0.0; uint64_t stat_alg_bytes_1 = 0;      // single-token calls only (the decode rate)

    pipeinfer_run_tap_fn run_tap = nullptr; void * run_tap_user = nullptr;      // tests: every run's metadata at its start and its rows at its finish (caller's thread)
    std::mt19937 rng;                          // llama_set_rng_seed / llama_sample_token (llama.cpp:1488)
    pi_error_text last_error;                 // (caller and feeder thread: see pi_error_text)
    ~llama_context();
};

namespace pi {
int64_t time_us();
// Schedule timeline (PI_TIMELINE=1: off otherwise, one predictable branch per call site): host-clock events of the caller, the feeder and the draft, read back with
// pipeinfer_timeline_dump (tools_dev/timeline.py: where the target's idle time between two passes goes).  Measurement aid, no effect on what is computed.
extern bool g_timeline_on;
void timeline_note(const char * tag, int a, int b, int c);
inline void tl(const char * tag, int a = 0, int b = 0, int c = 0) { if (g_timeline_on) timeline_note(tag, a, b, c); }
bool context_prepare(llama_context * ctx, std::string & err);          // uploads weights, allocates KV + scratch
pi_run * context_start(llama_context * ctx, const llama_batch & batch, int & rc);   // rc: 0 ok, 1 no slot, <0 error
int  context_finish(llama_context * ctx, pi_run * run);
bool context_poll_oldest(llama_context * ctx);
bool context_precapture(llama_context * ctx, int max_tokens);             // captures the graphs of passes of 1..max_tokens tokens now (idle context, calling thread)
void context_cancel_run(llama_context * ctx, pi_run * run);              // stops a started run: unlaunched segments are dropped, kernels already enqueued return on entry (abort word)
void urgent_drop_all(llama_context * ctx);                                  // the context gives up its claim on the device arbiter (destructor, priority change)
bool context_wait(llama_context * ctx, pi_run * run);                    // blocks until the run's results are in host memory
// device-side arrival flags of a stage boundary (transport.cpp, runtime.hip: k_flag_publish / k_abort_begin_wait).  in_flag: the run's activations have landed in x_in when it
// carries in_ticket — the run is enqueued at once and waits for it ON THE DEVICE; out_flag: written behind the store of x_out with out_ticket (and the abort the run ended with)
struct pi_stage_hop { const void * in_flag = nullptr; uint32_t in_ticket = 0; void * out_flag = nullptr; uint32_t out_ticket = 0; int64_t wait_timeout_us = 10 * 1000 * 1000; };
int  stage_run(llama_context * ctx, int n_tokens, const void * meta, size_t meta_bytes, const float * x_in, float * x_out,
               float * logits_host, int32_t * topk_ids, float * topk_vals, float * topk_sum,
               int (*should_abort)(void *) = nullptr, void * abort_user = nullptr, const pi_stage_hop * hop = nullptr);        // 0 done, 1 cut short (should_abort, a cancel, or upstream), < 0 error
// a stage that is the whole target carries the runs waiting in its ring in ONE forward pass (context.cpp: run coalescing for the transport's serve loop)
bool stage_pass_conflict(int n_ctx, int n_members, const void * const * metas, const int32_t * n_tokens, const void * q_meta, int q_tokens);
int  stage_run_pass(llama_context * ctx, int n_members, const void * const * metas, const int32_t * n_tokens, float * const * logits_host, int32_t * const * topk_ids,
                    float * const * topk_vals, float * const * topk_sum, int (*should_abort)(void *), void * abort_user);
void stage_abort_run(llama_context * ctx, int32_t run_id);                  // any thread: run `run_id` (ctx->stage_run_id while stage_run computes it) returns from every kernel on entry from now on; no effect on any other run
bool hop_store(void * dst, const void * src, size_t bytes, hipStream_t s);        // context.cpp: a stage's activations into the consumer's ring slot (kernel store for device memory)
void hop_store_set_mode(int mode);                                               // 1: always the kernel (device rings), 2: always hipMemcpyAsync (host rings), 0: ask the runtime per call (no transport)
int  hop_store_mode();
size_t decode_alg_bytes(const llama_context * ctx, int n_tok, int n_kv);
}
