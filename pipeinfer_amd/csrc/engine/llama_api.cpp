// The extern "C" surface declared in include/llama.h and include/pipeinfer.h.
#include "engine.h"

#include <map>
#include <tuple>

#include <algorithm>
#include <cmath>
#include <cstring>

// ---- defaults (llama.cpp:8566-8620) ---------------------------------------------------------------------------------------
extern "C" struct llama_model_params llama_model_default_params(void) {
    struct llama_model_params r;
    memset(&r, 0, sizeof r);
    r.n_gpu_layers = 0; r.main_gpu = 0; r.tensor_split = nullptr; r.progress_callback = nullptr; r.progress_callback_user_data = nullptr;
    r.vocab_only = false; r.use_mmap = true; r.use_mlock = false;
    return r;
}

extern "C" struct llama_context_params llama_context_default_params(void) {
    struct llama_context_params r;
    memset(&r, 0, sizeof r);
    r.seed = LLAMA_DEFAULT_SEED; r.n_ctx = 512; r.n_batch = 512; r.n_threads = 4; r.n_threads_batch = 4;
    r.rope_scaling_type = -1; r.rope_freq_base = 0.0f; r.rope_freq_scale = 0.0f;
    r.yarn_ext_factor = -1.0f; r.yarn_attn_factor = 1.0f; r.yarn_beta_fast = 32.0f; r.yarn_beta_slow = 1.0f; r.yarn_orig_ctx = 0;
    r.mul_mat_q = true; r.f16_kv = true; r.logits_all = false; r.embedding = false;
    return r;
}

// PIPEINFER_CRASH_TRACE=1: a fatal signal prints the C stack of the faulting thread before the default action (debugging aid for rare crashes under pytest,
// whose output capture otherwise leaves nothing behind); offsets resolve with addr2line against this library
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void crash_trace(int sig) {
    void * frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "\npipeinfer: fatal signal, C stack of the faulting thread:\n";
    (void) !write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void crash_trace_install() {
    if (!getenv("PIPEINFER_CRASH_TRACE")) return;
    for (int sig : { SIGSEGV, SIGABRT, SIGBUS, SIGFPE, SIGILL }) signal(sig, crash_trace);
}
extern "C" void llama_backend_init(bool) {}
extern "C" void llama_backend_free(void) {}
extern "C" int64_t llama_time_us(void) { return pi::time_us(); }
extern "C" int64_t ggml_time_us(void) { return pi::time_us(); }
extern "C" const char * llama_print_system_info(void) { return "MI355X (gfx950) | HIP kernels: mmvq q4_0/q8_0/q4_K/q5_K/q6_K, attention, norms, rope | wave64 | "; }

extern "C" struct llama_context * llama_new_context_with_model(struct llama_model * model, struct llama_context_params params) {
    if (!model) return nullptr;
    if (!params.f16_kv) { fprintf(stderr, "llama_new_context_with_model: only an f16 KV cache is supported\n"); return nullptr; }
    llama_context * ctx = new llama_context();
    ctx->model = model;
    ctx->cparams = params;
    ctx->n_ctx = params.n_ctx == 0 ? model->hparams.n_ctx_train : (int) params.n_ctx;       // llama.cpp:8726
    if (ctx->n_ctx > pi_op_attention_max_cells()) {   // the attention kernel keeps a workgroup's score rows in LDS (8 rows up to 4 840 cells ... one row up to 36 200): say so HERE, not at the first decode
        fprintf(stderr, "llama_new_context_with_model: n_ctx = %d is beyond this engine's attention kernel (one score row of n_ctx floats must fit 146 KiB of LDS: n_ctx <= %d)%s\n",
                ctx->n_ctx, pi_op_attention_max_cells(), params.n_ctx == 0 ? "; pass an explicit n_ctx (0 selected the model's training context)" : "");
        delete ctx;
        return nullptr;
    }
    ctx->n_batch = (int) std::max<uint32_t>(params.n_batch, 1);
    ctx->l0 = 0; ctx->l1 = model->hparams.n_layer; ctx->is_first = ctx->is_last = true;
    ctx->cells.reset(new pi::kv_cells(ctx->n_ctx));
    ctx->t_start_us = pi::time_us();
    return ctx;
}

extern "C" void llama_free(struct llama_context * ctx) { delete ctx; }
extern "C" const struct llama_model * llama_get_model(const struct llama_context * ctx) { return ctx->model; }
extern "C" int llama_n_ctx(const struct llama_context * ctx) { return ctx->n_ctx; }
extern "C" void llama_set_n_threads(struct llama_context *, uint32_t, uint32_t) {}

// ---- PipeInfer pipeline control -------------------------------------------------------------------------------------------
// One process drives every stage of its pipeline, so the calling thread is always the root of whichever communicator is
// current; the only driver-visible state is llama_node_id (llama.cpp:8915-8927).
extern "C" int llama_node_id(struct llama_context * ctx) { return ctx->node_id; }
// One process plays every MPI rank of the reference: it is the head of the target pipeline AND the root of the draft pipeline, so it stays
// a member (node 0) of whatever communicator the driver splits off — examples/speculative/speculative.cpp:118-132 hands rank 0 a negative
// colour for the draft's model communicator because there another rank computes the draft; here this process does.
extern "C" void llama_split_comm(struct llama_context * ctx, int color) { ctx->comm_color = color; ctx->node_id = 0; }
extern "C" void llama_swap_comm(struct llama_context * ctx) {
    std::swap(ctx->comm_color, ctx->comm_color_alt);
    ctx->comm_swapped = !ctx->comm_swapped;
    ctx->node_id = 0;
}
extern "C" void llama_sync_token(struct llama_context *, llama_token *, int) {}                   // MPI_Bcast among roots (llama.cpp:8901-8906): one process
extern "C" void llama_sync_token_data(struct llama_context *, llama_token_data *, int) {}

// llama_split_layers_weighted (llama.cpp:8929-8938 -> ggml_mpi_split_range, ggml-mpi.c:523-559).  In a single process the
// context owns every stage; the split only matters to the multi-process pipeline (pipeinfer_amd/pipeline.py), which sets
// the stage range through pipeinfer_context_set_stage.
extern "C" void llama_split_layers_weighted(struct llama_context *, float *, size_t) {}

extern "C" bool llama_mpi_iprobe(struct llama_context * ctx) { return pi::context_poll_oldest(ctx); }

extern "C" void llama_cancel_run(struct llama_context * ctx, int32_t * canceled, int count) {
    for (int i = 0; i < count; i++) {
        ctx->canceled_batches.insert(canceled[i]);                                                     // llama.cpp:9981-9993
        for (pi_run * r : ctx->inflight) if (r->batch_id == canceled[i]) pi::context_cancel_run(ctx, r);      // already started: stops at kernel granularity (context.cpp)
        if (ctx->remote && ctx->remote_ops.cancel) ctx->remote_ops.cancel(ctx->remote_ops.user, canceled[i]);
    }
}

extern "C" struct ggml_cgraph * llama_start_async_decode(struct llama_context * ctx, struct llama_batch batch) {
    if (ctx->node_id < 0) return nullptr;                                                           // llama.cpp:5466-5468
    int rc;
    return (struct ggml_cgraph *) pi::context_start(ctx, batch, rc);
}

extern "C" int llama_finish_async_decode(struct llama_context * ctx, struct llama_batch, struct ggml_cgraph * cgraph) {
    if (!cgraph) return -1;                                                                         // llama.cpp:5862-5871
    return pi::context_finish(ctx, (pi_run *) cgraph);
}

extern "C" int llama_decode(struct llama_context * ctx, struct llama_batch batch) {
    if (ctx->node_id < 0) return 0;                                                                 // llama.cpp:10007-10008
    int rc;
    pi_run * run = pi::context_start(ctx, batch, rc);
    if (!run) return rc;
    return pi::context_finish(ctx, run) == 1 ? 0 : -1;
}

extern "C" float * llama_get_logits(struct llama_context * ctx) { return ctx->logits.data(); }
extern "C" float * llama_get_logits_ith(struct llama_context * ctx, int32_t i) { return ctx->logits.data() + (size_t) i * ctx->model->hparams.n_vocab; }

// ---- batches (llama.cpp:9839-9898) ------------------------------------------------------------------------------------------
extern "C" struct llama_batch llama_batch_get_one(llama_token * tokens, int32_t n_tokens, llama_pos pos_0, llama_seq_id seq_id) {
    struct llama_batch b;
    memset(&b, 0, sizeof b);
    b.n_tokens = n_tokens; b.token = tokens; b.all_pos_0 = pos_0; b.all_pos_1 = 1; b.all_seq_id = seq_id; b.max_n_seq = 1;
    return b;
}

extern "C" struct llama_batch llama_batch_init(int32_t n_tokens, int32_t embd, int32_t n_seq_max) {
    struct llama_batch b;
    memset(&b, 0, sizeof b);
    b.max_n_seq = n_seq_max;
    if (embd) b.embd = (float *) malloc(sizeof(float) * n_tokens * embd);
    else      b.token = (llama_token *) malloc(sizeof(llama_token) * n_tokens);
    b.pos      = (llama_pos *) malloc(sizeof(llama_pos) * n_tokens);
    b.n_seq_id = (int32_t *) malloc(sizeof(int32_t) * n_tokens);
    b.seq_id   = (llama_seq_id **) malloc(sizeof(llama_seq_id *) * n_tokens);
    // one block for all rows (the reference mallocs n_tokens rows and frees only the first batch.n_tokens of them)
    llama_seq_id * rows = (llama_seq_id *) malloc(sizeof(llama_seq_id) * (size_t) n_tokens * (size_t) std::max(n_seq_max, 1));
    for (int i = 0; i < n_tokens; ++i) b.seq_id[i] = rows + (size_t) i * std::max(n_seq_max, 1);
    b.logits   = (int8_t *) malloc(sizeof(int8_t) * n_tokens);
    return b;
}

extern "C" void llama_batch_free(struct llama_batch * b) {
    if (b->token) free(b->token);
    if (b->embd) free(b->embd);
    if (b->pos) free(b->pos);
    if (b->n_seq_id) free(b->n_seq_id);
    if (b->seq_id) { free(b->seq_id[0]); free(b->seq_id); }
    if (b->logits) free(b->logits);
    memset(b, 0, sizeof *b);
}

// ---- KV cache API (llama.cpp:9238-9359; MPI transactions become plain host calls) --------------------------------------------
extern "C" int  llama_get_kv_cache_token_count(const struct llama_context * ctx) { return ctx->cells->token_count(); }
extern "C" int  llama_get_kv_cache_used_cells(const struct llama_context * ctx) { return ctx->cells->used(); }
extern "C" void llama_kv_cache_clear(struct llama_context * ctx) { ctx->cells->clear(); }
extern "C" void llama_kv_cache_seq_rm(struct llama_context * ctx, llama_seq_id s, llama_pos p0, llama_pos p1) { ctx->cells->seq_rm(s, p0, p1); }
extern "C" void llama_kv_cache_seq_cp(struct llama_context * ctx, llama_seq_id a, llama_seq_id b, llama_pos p0, llama_pos p1) {
    if (a == b) return;                                                                             // llama.cpp:9296-9298
    ctx->cells->seq_cp(a, b, p0, p1);
}
extern "C" void llama_kv_cache_seq_keep(struct llama_context * ctx, llama_seq_id s) { ctx->cells->seq_keep(s); }
extern "C" void pipeinfer_kv_cache_seq_rm_range(struct llama_context * ctx, llama_seq_id a, llama_seq_id b, llama_pos p0, llama_pos p1) { ctx->cells->seq_rm_range(a, b, p0, p1); }
extern "C" void pipeinfer_kv_cache_seq_cp_range(struct llama_context * ctx, llama_seq_id src, llama_seq_id a, llama_seq_id b, llama_pos p0, llama_pos p1) { ctx->cells->seq_cp_range(src, a, b, p0, p1); }
extern "C" void pipeinfer_kv_cache_seq_rm_cp_range(struct llama_context * ctx, llama_seq_id src, llama_seq_id a, llama_seq_id b, llama_pos p0, llama_pos p1) { ctx->cells->seq_rm_cp_range(src, a, b, p0, p1); }

extern "C" struct llama_kv_cache_view llama_kv_cache_view_init(const struct llama_context * ctx, int32_t n_max_seq) {     // llama.cpp:9157-9171
    struct llama_kv_cache_view v;
    memset(&v, 0, sizeof v);
    v.n_cells = 0; v.n_max_seq = n_max_seq;
    (void) ctx;
    return v;
}
extern "C" void llama_kv_cache_view_free(struct llama_kv_cache_view * v) {
    if (v->cells) { free(v->cells); v->cells = nullptr; }
    if (v->cells_sequences) { free(v->cells_sequences); v->cells_sequences = nullptr; }
}
extern "C" void llama_kv_cache_view_update(const struct llama_context * ctx, struct llama_kv_cache_view * v) {            // llama.cpp:9184-9236
    const pi::kv_cells & c = *ctx->cells;
    if (v->n_cells < c.size() || !v->cells) {
        v->n_cells = c.size();
        v->cells = (struct llama_kv_cache_view_cell *) realloc(v->cells, sizeof(struct llama_kv_cache_view_cell) * v->n_cells);
        v->cells_sequences = (llama_seq_id *) realloc(v->cells_sequences, sizeof(llama_seq_id) * v->n_max_seq * v->n_cells);
    }
    int used = 0, tokens = 0, cur_contig = 0, max_contig = 0, max_idx = -1;
    for (int i = 0; i < c.size(); i++) {
        llama_seq_id * cs = v->cells_sequences + (size_t) i * v->n_max_seq;
        const int sz = c.nseq(i);
        tokens += sz;
        v->cells[i].pos = c.pos(i);
        if (sz > 0) {
            if (cur_contig > max_contig) { max_contig = cur_contig; max_idx = i - cur_contig; }
            cur_contig = 0;
        } else cur_contig++;
        int k = 0;
        for (int s = 0; k < v->n_max_seq && k < sz && s < 1 << 20; s++) if (c.has(i, s)) cs[k++] = s;
        if (k > 0) used++;
        for (; k < v->n_max_seq; k++) cs[k] = -1;
    }
    if (cur_contig > max_contig) { max_contig = cur_contig; max_idx = c.size() - cur_contig; }
    v->max_contiguous = max_contig; v->max_contiguous_idx = max_idx; v->token_count = tokens; v->used_cells = used;
}

// ---- greedy samplers (llama.cpp:7019-7046, 7496-7510) ---------------------------------------------------------------------------
extern "C" void llama_sample_softmax(struct llama_context * ctx, llama_token_data_array * c) {
    const int64_t t0 = pi::time_us();
    if (!c->sorted) {
        std::stable_sort(c->data, c->data + c->size, [](const llama_token_data & a, const llama_token_data & b) { return a.logit > b.logit; });
        c->sorted = true;
    }
    const float max_l = c->data[0].logit;
    float cum = 0.0f;
    for (size_t i = 0; i < c->size; ++i) { const float p = expf(c->data[i].logit - max_l); c->data[i].p = p; cum += p; }
    for (size_t i = 0; i < c->size; ++i) c->data[i].p /= cum;
    if (ctx) ctx->t_sample_us += pi::time_us() - t0;
}

extern "C" llama_token llama_sample_token_greedy(struct llama_context * ctx, llama_token_data_array * c) {
    const int64_t t0 = pi::time_us();
    size_t best = 0;
    for (size_t i = 1; i < c->size; i++) if (c->data[best].logit < c->data[i].logit) best = i;      // std::max_element: first maximum
    if (ctx) { ctx->t_sample_us += pi::time_us() - t0; ctx->n_sample++; }
    return c->data[best].id;
}

// ---- timings ------------------------------------------------------------------------------------------------------------------
extern "C" struct llama_timings llama_get_timings(struct llama_context * ctx) {
    struct llama_timings r;
    r.t_start_ms = 1e-3 * ctx->t_start_us; r.t_end_ms = 1e-3 * pi::time_us(); r.t_load_ms = 1e-3 * ctx->t_load_us;
    r.t_sample_ms = 1e-3 * ctx->t_sample_us; r.t_p_eval_ms = 1e-3 * ctx->t_p_eval_us; r.t_eval_ms = 1e-3 * ctx->t_eval_us;
    r.n_sample = std::max(1, ctx->n_sample); r.n_p_eval = std::max(1, ctx->n_p_eval); r.n_eval = std::max(1, ctx->n_eval);
    return r;
}
extern "C" void llama_print_timings(struct llama_context * ctx) {
    const struct llama_timings t = llama_get_timings(ctx);
    fprintf(stderr, "\n");
    fprintf(stderr, "llama_print_timings:        load time = %10.2f ms\n", t.t_load_ms);
    fprintf(stderr, "llama_print_timings:      sample time = %10.2f ms / %5d runs   (%8.2f ms per token)\n", t.t_sample_ms, t.n_sample, t.t_sample_ms / t.n_sample);
    fprintf(stderr, "llama_print_timings: prompt eval time = %10.2f ms / %5d tokens (%8.2f ms per token)\n", t.t_p_eval_ms, t.n_p_eval, t.t_p_eval_ms / t.n_p_eval);
    fprintf(stderr, "llama_print_timings:        eval time = %10.2f ms / %5d runs   (%8.2f ms per token)\n", t.t_eval_ms, t.n_eval, t.t_eval_ms / t.n_eval);
    fprintf(stderr, "llama_print_timings:       total time = %10.2f ms\n", t.t_end_ms - t.t_start_ms);
}
extern "C" void llama_reset_timings(struct llama_context * ctx) {
    ctx->t_start_us = pi::time_us(); ctx->t_sample_us = ctx->t_eval_us = ctx->t_p_eval_us = 0; ctx->n_sample = ctx->n_eval = ctx->n_p_eval = 0;
}

// ---- extensions ------------------------------------------------------------------------------------------------------------------
extern "C" int pipeinfer_context_set_priority(struct llama_context * ctx, int priority) {
    if (ctx->stream) { ctx->last_error = "pipeinfer_context_set_priority: the stream already exists"; return -1; }
    ctx->stream_priority = priority;
    return 0;
}

extern "C" int pipeinfer_context_set_cu_range(struct llama_context * ctx, int cu_lo, int cu_hi) {
    if (ctx->stream) { ctx->last_error = "pipeinfer_context_set_cu_range: the stream already exists"; return -1; }
    if (cu_lo < 0 || cu_hi < cu_lo || cu_hi > 1024) { ctx->last_error = "pipeinfer_context_set_cu_range: bad range"; return -1; }
    ctx->cu_lo = cu_lo; ctx->cu_hi = cu_hi;
    return 0;
}

extern "C" int pipeinfer_context_set_segments(struct llama_context * ctx, int layers_per_segment) {
    if (ctx->prepared || ctx->feeder.joinable()) { ctx->last_error = "pipeinfer_context_set_segments: the context has already decoded"; return -1; }
    if (ctx->remote) { ctx->last_error = "pipeinfer_context_set_segments: a remote head owns no layers"; return -1; }
    ctx->seg_layers = layers_per_segment > 0 ? layers_per_segment : 0;
    return 0;
}

extern "C" int pipeinfer_context_set_coalescing(struct llama_context * ctx, int max_tokens) {
    if (ctx->remote) { ctx->last_error = "pipeinfer_context_set_coalescing: a remote head owns no layers"; return -1; }
    if (max_tokens > 1 && ctx->seg_layers <= 0) { ctx->last_error = "pipeinfer_context_set_coalescing: needs a segmented context (pipeinfer_context_set_segments): the feeder thread composes the passes"; return -1; }
    ctx->coalesce_max.store(max_tokens > 1 ? std::min(max_tokens, ctx->n_batch) : 0);
    if (max_tokens > 1 && !pi::context_precapture(ctx, std::min(max_tokens, ctx->n_batch))) { ctx->coalesce_max.store(0); return -1; }
    return 0;
}
// launches of this context's captured graphs since pipeinfer_reset_stats: graph contents x replays, one row per instance (family, type, tokens, launch shape)
extern "C" int pipeinfer_get_launch_rows(struct llama_context * ctx, struct pi_kt_row * out, int cap) {
    std::map<std::tuple<int, int, int, uint64_t>, std::pair<uint64_t, double>> tab;
    std::lock_guard<std::mutex> glk(ctx->graphs_mu);                          // (the feeder thread inserts a graph the first time it meets a pass width)
    for (auto & kv : ctx->graphs) {
        const uint64_t rep = kv.second.replays.load(std::memory_order_relaxed);
        if (!rep) continue;
        for (const pi_kt_count & c : kv.second.sink.launches) {
            auto & e = tab[std::make_tuple(c.fam, c.type, c.nt, c.fam <= PI_KT_MMQ ? (uint64_t) c.bytes : 0)];
            e.first += c.count * rep; e.second += c.bytes_sum * (double) rep;
        }
    }
    int n = 0;
    for (auto & kv : tab) {
        if (n >= cap) break;
        pi_kt_row & r = out[n++];
        r.family = std::get<0>(kv.first); r.type = std::get<1>(kv.first); r.n_tok = std::get<2>(kv.first); r.sampled_dropped = 0; r.shape_bytes = std::get<3>(kv.first);
        r.launches = kv.second.first; r.bytes_total = kv.second.second; r.sampled = 0; r.avg_us = 0.0; r.avg_sampled_bytes = 0.0;
    }
    return n;
}
extern "C" void pipeinfer_context_set_run_tap(struct llama_context * ctx, pipeinfer_run_tap_fn fn, void * user) { ctx->run_tap = fn; ctx->run_tap_user = user; }
extern "C" int32_t pipeinfer_model_n_layer(const struct llama_model * model) { return model ? model->hparams.n_layer : 0; }
extern "C" int32_t pipeinfer_run_first_cell(const struct ggml_cgraph * ticket) { return ticket ? ((const pi_run *) ticket)->kv_head : -1; }
extern "C" void pipeinfer_context_set_graphs(struct llama_context * ctx, int enable) { ctx->use_graphs = enable != 0 && getenv("PI_NO_GRAPHS") == nullptr; }

extern "C" void pipeinfer_context_set_remote(struct llama_context * ctx, const struct pipeinfer_remote_ops * ops) {
    ctx->remote = ops != nullptr;
    if (ops) ctx->remote_ops = *ops;
}

extern "C" int pipeinfer_context_set_stage(struct llama_context * ctx, int32_t l0, int32_t l1, int32_t is_first, int32_t is_last) {
    if (ctx->x) { ctx->last_error = "pipeinfer_context_set_stage: context already prepared"; return -1; }
    if (l0 < 0 || l1 < l0 || l1 > ctx->model->hparams.n_layer) { ctx->last_error = "pipeinfer_context_set_stage: bad layer range"; return -1; }
    ctx->l0 = l0; ctx->l1 = l1; ctx->is_first = is_first != 0; ctx->is_last = is_last != 0;
    return 0;
}

// ggml_mpi_split_layers (ggml-mpi.c:541-557): stage i gets [next, min(n_layer - 1, next + w_i * n_layer)] as a closed pair in the
// reference's encoding, consumed as the half-open [start, end) of ggml_mpi_graph_creation_post (:640-700); the last stage
// takes whatever is left
extern "C" void pipeinfer_split_layers(int32_t n_layer, const float * weights, int32_t n_stage, int32_t * ranges) {
    int next = 0;
    for (int i = 0; i < n_stage; i++) {
        int end = i == n_stage - 1 ? n_layer : std::min(n_layer, next + (int) (weights[i] * n_layer));
        if (end < next) end = next;
        ranges[2 * i] = next; ranges[2 * i + 1] = end;
        next = end;
    }
}

// llama_split_layers_weighted for ONE stage process (llama.cpp:8929-8938): the split rule above on this model's layer count, then set_stage
extern "C" int pipeinfer_context_set_stage_weighted(struct llama_context * ctx, const float * weights, int32_t n_stage, int32_t stage, int32_t * l0_out, int32_t * l1_out) {
    if (n_stage < 1 || stage < 0 || stage >= n_stage || !weights) { ctx->last_error = "pipeinfer_context_set_stage_weighted: bad stage index"; return -1; }
    std::vector<int32_t> ranges((size_t) 2 * n_stage);
    pipeinfer_split_layers(ctx->model->hparams.n_layer, weights, n_stage, ranges.data());
    if (l0_out) *l0_out = ranges[2 * stage];
    if (l1_out) *l1_out = ranges[2 * stage + 1];
    return pipeinfer_context_set_stage(ctx, ranges[2 * stage], ranges[2 * stage + 1], stage == 0, stage == n_stage - 1);
}

extern "C" int pipeinfer_stage_run_abortable(struct llama_context * ctx, int32_t n_tokens, const void * meta, size_t meta_bytes, const float * x_in_dev, float * x_out_dev,
                                             float * logits_host, int32_t * topk_ids, float * topk_vals, float * topk_sum, int (*should_abort)(void *), void * user) {
    return pi::stage_run(ctx, n_tokens, meta, meta_bytes, x_in_dev, x_out_dev, logits_host, topk_ids, topk_vals, topk_sum, should_abort, user);
}
extern "C" int pipeinfer_stage_run(struct llama_context * ctx, int32_t n_tokens, const void * meta, size_t meta_bytes, const float * x_in_dev, float * x_out_dev,
                                   float * logits_host, int32_t * topk_ids, float * topk_vals, float * topk_sum) {
    return pi::stage_run(ctx, n_tokens, meta, meta_bytes, x_in_dev, x_out_dev, logits_host, topk_ids, topk_vals, topk_sum);
}

extern "C" int pipeinfer_context_prepare(struct llama_context * ctx) {
    std::string err;
    if (!pi::context_prepare(ctx, err)) { ctx->last_error = err; return -1; }
    return 0;
}
extern "C" const char * pipeinfer_context_error(struct llama_context * ctx) {
    static thread_local std::string text;                                      // a copy taken under the text's lock: the feeder thread may be writing a new one
    text = ctx->last_error.str();
    return text.c_str();
}

extern "C" int pipeinfer_get_topk(struct llama_context * ctx, int32_t i, llama_token * ids, float * logits, float * sumexp) {
    if (i < 0 || i >= ctx->last_n_tokens) return -1;
    memcpy(ids, ctx->last_topk_ids.data() + (size_t) i * 8, 8 * sizeof(llama_token));
    memcpy(logits, ctx->last_topk_vals.data() + (size_t) i * 8, 8 * sizeof(float));
    *sumexp = ctx->last_topk_sum[i];
    return 0;
}

extern "C" void pipeinfer_get_stats(struct llama_context * ctx, struct pipeinfer_stats * out) {
    out->n_calls = ctx->stat_calls; out->gpu_ms = ctx->stat_gpu_ms; out->alg_bytes = ctx->stat_alg_bytes;
    out->n_calls_1 = ctx->stat_calls_1; out->gpu_ms_1 = ctx->stat_gpu_ms_1; out->alg_bytes_1 = ctx->stat_alg_bytes_1;
    out->weight_bytes = ctx->model->stage_weight_bytes(ctx->l0, ctx->l1, ctx->is_first, ctx->is_last, 1);
    out->kv_bytes_per_cell = (uint64_t) 4 * ctx->model->hparams.n_embd_gqa() * (uint64_t) (ctx->l1 - ctx->l0);
    out->n_segments = ctx->stat_segments; out->n_segments_skipped = ctx->stat_segments_skipped;
    out->n_cancel_runs = ctx->stat_cancel_runs; out->n_cancel_runs_started = ctx->stat_cancel_runs_started; out->cancel_gpu_ms = ctx->stat_cancel_gpu_ms;
    out->n_passes = ctx->stat_passes; out->n_pass_runs = ctx->stat_pass_runs; out->n_pass_tokens = ctx->stat_pass_tokens; out->n_coalesced_passes = ctx->stat_coalesced_passes;
    out->weight_copy_bytes = ctx->model->hbm_copy_bytes;
    out->n_reaction_windows = ctx->stat_reaction_windows;
}
extern "C" void pipeinfer_reset_stats(struct llama_context * ctx) { ctx->stat_calls = 0; ctx->stat_gpu_ms = 0.0; ctx->stat_alg_bytes = 0; ctx->stat_calls_1 = 0; ctx->stat_gpu_ms_1 = 0.0; ctx->stat_alg_bytes_1 = 0; ctx->stat_segments = 0; ctx->stat_segments_skipped = 0; ctx->stat_cancel_runs = 0; ctx->stat_cancel_runs_started = 0; ctx->stat_cancel_gpu_ms = 0.0; ctx->stat_passes = 0; ctx->stat_pass_runs = 0; ctx->stat_pass_tokens = 0; ctx->stat_coalesced_passes = 0; ctx->stat_reaction_windows = 0;
    { std::lock_guard<std::mutex> glk(ctx->graphs_mu); for (auto & kv : ctx->graphs) kv.second.replays.store(0); }
    // device-side running sums: aborted launches per kernel family (words 8..14 of the abort block), the arrival-flag clock ticks of a pipeline stage.  Call with no run in flight.
    if (ctx->abort_dev && ctx->stream) { (void) hipMemsetAsync(ctx->abort_dev + 8, 0, 8 * sizeof(int32_t), ctx->stream); (void) hipStreamSynchronize(ctx->stream); }
    if (ctx->hop_dev && ctx->stream) { (void) hipMemsetAsync(ctx->hop_dev, 0, 64, ctx->stream); (void) hipStreamSynchronize(ctx->stream); }
    ctx->stat_flag_seen_us = 0.0; ctx->stat_flag_wait_ms = 0.0; ctx->stat_flag_runs = 0;
}
// launches of this context since pipeinfer_reset_stats that found their run aborted (llama_cancel_run after the start) and returned on entry, per kernel family
// (index = pi_kt_row.family); synchronises the context's control stream.  Returns the number of families written (<= cap), -1 on error
extern "C" int pipeinfer_get_aborted_launches(struct llama_context * ctx, uint64_t * out, int cap) {
    const int n = std::min(cap, (int) PI_KT_NFAM);
    for (int i = 0; i < n; i++) out[i] = 0;
    if (!ctx->abort_dev) return n;
    int32_t h[8] = { 0 };
    if (hipMemcpy(h, ctx->abort_dev + 8, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { (void) hipGetLastError(); return -1; }
    for (int i = 0; i < n; i++) out[i] = (uint64_t) (uint32_t) h[i];
    return n;
}

// ---- stand-alone KV-cell table (host logic tests) ---------------------------------------------------------------------------------
struct pi_kvcells { pi::kv_cells cells; explicit pi_kvcells(int n) : cells(n) {} };
extern "C" pi_kvcells * pi_kvcells_new(int n_ctx) { return new pi_kvcells(n_ctx); }
extern "C" void pi_kvcells_free(pi_kvcells * c) { delete c; }
static std::vector<llama_seq_id *> row_ptrs(const llama_seq_id * flat, int n, int stride) {
    std::vector<llama_seq_id *> p((size_t) n);
    for (int i = 0; i < n; i++) p[i] = const_cast<llama_seq_id *>(flat) + (size_t) i * stride;
    return p;
}
extern "C" int pi_kvcells_decode(pi_kvcells * c, int n, const llama_pos * pos, const int32_t * n_seq_id, const llama_seq_id * flat, int stride) {
    auto p = row_ptrs(flat, n, stride);
    c->cells.pre_decode(n);
    const int head = c->cells.find_slot(n, pos, n_seq_id, p.data());
    if (head >= 0) c->cells.post_decode(n);
    return head;
}
extern "C" void pi_kvcells_seq_rm(pi_kvcells * c, int s, int p0, int p1) { c->cells.seq_rm(s, p0, p1); }
extern "C" void pi_kvcells_seq_cp(pi_kvcells * c, int a, int b, int p0, int p1) { c->cells.seq_cp(a, b, p0, p1); }
extern "C" void pi_kvcells_seq_keep(pi_kvcells * c, int s) { c->cells.seq_keep(s); }
extern "C" void pi_kvcells_clear(pi_kvcells * c) { c->cells.clear(); }
extern "C" void pi_kvcells_seq_rm_range(pi_kvcells * c, int a, int b, int p0, int p1) { c->cells.seq_rm_range(a, b, p0, p1); }
extern "C" void pi_kvcells_seq_cp_range(pi_kvcells * c, int src, int a, int b, int p0, int p1) { c->cells.seq_cp_range(src, a, b, p0, p1); }
extern "C" void pi_kvcells_seq_rm_cp_range(pi_kvcells * c, int src, int a, int b, int p0, int p1) { c->cells.seq_rm_cp_range(src, a, b, p0, p1); }
extern "C" int  pi_kvcells_head(const pi_kvcells * c) { return c->cells.head(); }
extern "C" int  pi_kvcells_used(const pi_kvcells * c) { return c->cells.used(); }
extern "C" int  pi_kvcells_n_attend(const pi_kvcells * c) { return c->cells.n_attend(); }
extern "C" int  pi_kvcells_pos(const pi_kvcells * c, int i) { return c->cells.pos(i); }
extern "C" int  pi_kvcells_has(const pi_kvcells * c, int i, int s) { return c->cells.has(i, s); }
extern "C" int  pi_kvcells_nseq(const pi_kvcells * c, int i) { return c->cells.nseq(i); }
extern "C" void pi_kvcells_mask(const pi_kvcells * c, int n_kv, int n, const llama_pos * pos, const int32_t * n_seq_id, const llama_seq_id * flat, int stride, uint32_t * out) {
    auto p = row_ptrs(flat, n, stride);
    c->cells.build_mask(n_kv, n, pos, n_seq_id, p.data(), out);
}
