// llama_context: KV cache + scratch in HBM, one HIP stream per context, asynchronous runs.
//
// Run lifecycle (reference: llama_decode_internal_phased, llama.cpp:5461-5827):
//   start  = phase 0: cell placement (find_slot) + attention mask are fixed on the host NOW, exactly as the reference fixes
//            them at graph-build time; the stage's kernels, the logits D2H copy and a completion event are enqueued on the
//            context's stream; the call returns a ticket.  Nothing blocks: draft and target contexts overlap on the GPU.
//   finish = phase 1: wait for the event, copy the requested logits rows into the context's logits buffer.
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

#define HIP_OK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(e__); return false; } } while (0)
#define OP_OK(expr)  do { if ((expr) != 0) { ctx->last_error = std::string(#expr) + ": " + pi_last_error(); return false; } } while (0)

llama_context::~llama_context() {
    pi::urgent_drop_all(this);
    if (feeder.joinable()) {
        { std::lock_guard<std::mutex> lk(fmu); fstop = true; }
        fcv.notify_all();
        feeder.join();
    }
    if (stream) (void) hipStreamSynchronize(stream);
    for (pi_run * r : inflight) pool.push_back(r);
    for (pi_run * r : pool) {
        if (r->done) (void) hipEventDestroy(r->done);
        if (r->begin) (void) hipEventDestroy(r->begin);
        void * hb[] = { r->logits_host, r->topk_ids, r->meta_host };
        for (void * b : hb) if (b) { if (r->host_plain) free(b); else (void) hipHostFree(b); }
        delete r;
    }
    for (auto & kv : graphs) if (kv.second.exec) (void) hipGraphExecDestroy(kv.second.exec);
    for (hipEvent_t e : seg_events) if (e) (void) hipEventDestroy(e);
    if (pass_event) (void) hipEventDestroy(pass_event);
    for (uint8_t * b : pass_host) if (b) (void) hipHostFree(b);
    if (ctl_stream) { (void) hipStreamSynchronize(ctl_stream); (void) hipStreamDestroy(ctl_stream); }
    if (abort_host) (void) hipHostFree(abort_host);
    if (hop_host) (void) hipHostFree(hop_host);
    pi_free(hop_dev);
    pi_free(abort_dev);
    void * bufs[] = { k_cache, v_cache, x, x2, nx, nx2, qkv, att, up, gate, logits_dev, q16, q8a, q8b, q8c, tokens_dev, pos_dev, dyn_dev, mask_dev, topk_ids_dev, topk_vals_dev, topk_sum_dev, topk_scratch_dev };
    for (void * b : bufs) pi_free(b);
    if (rope) pi_rope_table_free(rope);
    if (stream) (void) hipStreamDestroy(stream);
}

namespace pi {

static size_t kv_bytes_per_cell(const llama_context * ctx) {
    return (size_t) 2 * 2 * ctx->model->hparams.n_embd_gqa() * (size_t) (ctx->l1 - ctx->l0);
}

// weights of the stage + K/V read over n_kv cells + K/V written for n_tok + activation hops (+ logits)
size_t decode_alg_bytes(const llama_context * ctx, int n_tok, int n_kv) {
    const pi_hparams & hp = ctx->model->hparams;
    // stage_weight_bytes walks every tensor NAME of the stage (22 strings and map look-ups per layer): ~70 us for TinyLlama, ~250 us for an 80-layer stage — and this is
    // called at every finish (twice for a one-token pass), i.e. on the draft's step and in the driver's reaction to a finished pass (round 6: the timeline showed 200 us
    // between a draft step's GPU time and its host time).  It is linear in n_tok (only the embedding rows depend on it): both terms are computed once per stage.
    static const bool uncached = getenv("PIPEINFER_ALG_BYTES_UNCACHED") != nullptr;        // (A/B: the walk at every call, as until round 6)
    if (uncached || ctx->alg_w_key != std::make_tuple(ctx->l0, ctx->l1, ctx->is_first, ctx->is_last)) {
        ctx->alg_w_fixed = ctx->model->stage_weight_bytes(ctx->l0, ctx->l1, ctx->is_first, ctx->is_last, 0);
        ctx->alg_w_per_tok = ctx->model->stage_weight_bytes(ctx->l0, ctx->l1, ctx->is_first, ctx->is_last, 1) - ctx->alg_w_fixed;
        ctx->alg_w_key = std::make_tuple(ctx->l0, ctx->l1, ctx->is_first, ctx->is_last);
    }
    size_t b = ctx->alg_w_fixed + ctx->alg_w_per_tok * (size_t) n_tok;
    b += (size_t) n_kv * kv_bytes_per_cell(ctx) + (size_t) n_tok * kv_bytes_per_cell(ctx);
    b += (size_t) 8 * hp.n_embd * n_tok;
    if (ctx->is_last) b += (size_t) 4 * hp.n_vocab * n_tok;
    return b;
}

static bool context_prepare_impl(llama_context * ctx, std::string & err);
bool context_precapture(llama_context * ctx, int max_tokens);
bool context_prepare(llama_context * ctx, std::string & err) {
    if (ctx->prepared || ctx->remote) return true;                           // the head of a remote pipeline owns no device state
    if (ctx->prepare_failed) { err = "an earlier attempt to allocate this context's device state failed: " + ctx->last_error; return false; }
    if (!context_prepare_impl(ctx, err)) { ctx->prepare_failed = true; ctx->last_error = err; return false; }      // (a half-built context must never launch kernels)
    ctx->prepared = true;
    // A segmented context's graphs are replayed by its feeder thread.  Capture the narrow widths (single tokens, drafted batches) NOW, on the caller's thread and before
    // that thread exists: captured lazily they would be recorded by the feeder in the middle of a generation, beside the caller's own HIP calls (the draft's first runs,
    // pinned allocations, cancels) — the combination behind this runtime's rare capture failures, and (seen once in 36 repetitions of the segmented-target tests) a stall.
    // Prompt-sized passes are still captured on first use: the caller is then blocked in llama_decode on that very run.
    // (pipeinfer_context_set_coalescing called BEFORE the first decode: its pass widths are captured here too, not by the feeder in mid-generation)
    if (ctx->seg_layers > 0 && !context_precapture(ctx, std::max(8, ctx->coalesce_max.load()))) { ctx->last_error = "graph capture of the narrow pass widths failed: " + ctx->last_error; err = ctx->last_error; return false; }
    return true;
}
static bool context_prepare_impl(llama_context * ctx, std::string & err) {
    llama_model * m = ctx->model;
    const pi_hparams & hp = m->hparams;
    if (pi_device_init(m->device) != 0) { err = pi_last_error(); return false; }
    const int64_t t0 = time_us();
    // a context whose n_batch-bounded widest pass stays below the matrix-core kernel's first width never reads the tile-major copies: do not make them
    if (!m->ensure_loaded(ctx->l0, ctx->l1, ctx->is_first, ctx->is_last, ctx->n_batch >= pi_op_mmq_min_tokens(), err)) return false;
    ctx->t_load_us += time_us() - t0;
    const int nb = ctx->n_batch, n_embd = hp.n_embd, n_gqa = hp.n_embd_gqa(), n_ff = hp.n_ff, n_layer_local = ctx->l1 - ctx->l0;
    {   // the draft context asks for the high-priority queue: its kernels are small and on the critical path of speculation, the
        // target's verify runs fill the GPU for milliseconds — without priority a draft token waits behind a whole target run
        int lo = 0, hi = 0;
        (void) hipDeviceGetStreamPriorityRange(&lo, &hi);
        const int prio = ctx->stream_priority > 0 ? hi : ctx->stream_priority < 0 ? lo : (lo + hi) / 2;
        if (ctx->cu_hi > ctx->cu_lo) {
            // a private slice of the compute units: two contexts on one GPU then run side by side instead of taking turns kernel
            // by kernel (a small draft kernel otherwise waits for the target kernel that currently fills all 256 CUs — 242 times
            // per draft token)
            std::vector<uint32_t> mask((size_t) (ctx->cu_hi + 31) / 32, 0u);
            for (int i = ctx->cu_lo; i < ctx->cu_hi; i++) mask[(size_t) i >> 5] |= 1u << (i & 31);
            if (hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t) mask.size(), mask.data()) != hipSuccess) { err = "hipExtStreamCreateWithCUMask failed"; return false; }
        } else if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio) != hipSuccess) { err = "hipStreamCreate failed"; return false; }
    }
    auto dev = [&](size_t bytes) -> void * { void * p = pi_malloc(bytes); if (!p) err = pi_last_error(); return p; };
    const size_t kv_elems = (size_t) n_layer_local * ctx->n_ctx * n_gqa;
    if (!(ctx->k_cache = (uint16_t *) dev(kv_elems * 2)) || !(ctx->v_cache = (uint16_t *) dev(kv_elems * 2))) return false;
    pi_memset(ctx->k_cache, 0, kv_elems * 2, nullptr); pi_memset(ctx->v_cache, 0, kv_elems * 2, nullptr);   // llama.cpp:1543 memset(cache.buf, 0)
    // rotation mode of the context's table: 0 pairs (2i, 2i+1), 1 NeoX over n_rot values, -1 none (identity table: ALiBi or learned positions)
    const int neox = hp.arch == PI_ARCH_FALCON ? 1 : !hp.generic() ? 0 : hp.block.rope == 2 ? 1 : hp.block.rope == 0 ? 0 : -1;
    const float freq_base  = ctx->cparams.rope_freq_base  == 0.0f ? hp.rope_freq_base_train  : ctx->cparams.rope_freq_base;     // llama.cpp:8729-8731
    const float freq_scale = ctx->cparams.rope_freq_scale == 0.0f ? hp.rope_freq_scale_train : ctx->cparams.rope_freq_scale;
    if (hp.generic() && hp.block.qk_norm) {                                        // build_persimmon ropes a VIEW of the first head_dim / 2 values with n_dims = head_dim / 2 (llama.cpp:4331, 4415-4440)
        if (!(ctx->rope = pi_rope_table_create_ex(std::max(ctx->n_ctx, 32), hp.head_dim() / 2, hp.head_dim() / 2, freq_base, freq_scale, 1))) { err = pi_last_error(); return false; }
    } else
    if (!(ctx->rope = pi_rope_table_create_ex(std::max(ctx->n_ctx, 32), hp.head_dim(), neox == 1 ? hp.n_rot : hp.head_dim(), freq_base, freq_scale, neox))) { err = pi_last_error(); return false; }
    const size_t qkv_w = (size_t) n_embd + 2 * n_gqa;
    const size_t kmax = (size_t) std::max(n_embd, n_ff);
    const size_t q8_bytes = std::max(pi_q8_nbytes(PI_Q8_K, (int64_t) ((kmax + 255) / 256 * 256), nb), std::max(pi_q8_nbytes(PI_Q8_0, (int64_t) kmax, nb), pi_q8_nbytes(PI_Q8_1, (int64_t) kmax, nb)));
    if (!(ctx->x = (float *) dev((size_t) 4 * nb * n_embd)) || !(ctx->x2 = (float *) dev((size_t) 4 * nb * n_embd)) || !(ctx->nx = (float *) dev((size_t) 4 * nb * n_embd)) ||
        !(ctx->nx2 = (float *) dev((size_t) 4 * nb * n_embd)) || !(ctx->qkv = (float *) dev((size_t) 4 * nb * qkv_w)) || !(ctx->att = (float *) dev((size_t) 4 * nb * n_embd)) ||
        !(ctx->up = (float *) dev((size_t) 4 * nb * n_ff)) || !(ctx->gate = (float *) dev((size_t) 4 * nb * n_ff)) || !(ctx->q16 = (uint16_t *) dev((size_t) 2 * nb * n_embd)) ||
        !(ctx->q8a = dev(q8_bytes)) || !(ctx->q8b = dev(q8_bytes)) || !(ctx->q8c = dev(q8_bytes)) || !(ctx->tokens_dev = (int32_t *) dev((size_t) 4 * nb)) || !(ctx->pos_dev = (int32_t *) dev((size_t) 4 * nb)) || !(ctx->dyn_dev = (int32_t *) dev((size_t) 4 * (4 + 2 * nb))))
        return false;
    {   // abort word + cancel flags + the control stream cancels travel on (highest priority: a cancel must not queue behind the run it stops)
        int lo = 0, hi = 0;
        (void) hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (!(ctx->abort_dev = (int32_t *) dev(256))) return false;
        pi_memset(ctx->abort_dev, 0, 256, nullptr);
        if (hipHostMalloc((void **) &ctx->abort_host, 256 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) { err = "hipHostMalloc(abort flags) failed"; return false; }
        memset(ctx->abort_host, 0, 256 * sizeof(int32_t));
        if (hipStreamCreateWithPriority(&ctx->ctl_stream, hipStreamNonBlocking, hi) != hipSuccess) { err = "hipStreamCreate(control) failed"; return false; }
    }
    ctx->use_graphs = getenv("PI_NO_GRAPHS") == nullptr;
    if (pi_op_attention_reserve(hp.head_dim(), ctx->n_ctx) != 0 || pi_op_mul_mat_reserve() != 0) { err = pi_last_error(); return false; }
    ctx->mask_cap_words = (size_t) nb * ((ctx->n_ctx + 31) / 32);
    if (!(ctx->mask_dev = (uint32_t *) dev(ctx->mask_cap_words * 4))) return false;
    if (ctx->is_last) {
        if (!(ctx->logits_dev = (float *) dev((size_t) 4 * nb * hp.n_vocab)) || !(ctx->topk_ids_dev = (int32_t *) dev((size_t) 4 * nb * 8)) ||
            !(ctx->topk_vals_dev = (float *) dev((size_t) 4 * nb * 8)) || !(ctx->topk_sum_dev = (float *) dev((size_t) 4 * nb)) ||
            !(ctx->topk_scratch_dev = dev(pi_op_topk_scratch_bytes(nb))))
            return false;
    }
    if (hipDeviceSynchronize() != hipSuccess) { err = "hipDeviceSynchronize failed after allocation"; return false; }
    return true;
}

static pi_run * acquire_run(llama_context * ctx, int n_tokens, size_t meta_bytes) {
    pi_run * r = nullptr;
    if (!ctx->pool.empty()) { r = ctx->pool.back(); ctx->pool.pop_back(); }
    else {
        r = new pi_run(); r->host_plain = ctx->remote;
        if (!ctx->remote && (hipEventCreate(&r->done) != hipSuccess || hipEventCreate(&r->begin) != hipSuccess)) { delete r; return nullptr; }
    }
    auto host_alloc = [&](void ** p, size_t bytes) { if (r->host_plain) { *p = malloc(bytes); return *p != nullptr; } return hipHostMalloc(p, bytes, hipHostMallocDefault) == hipSuccess; };
    auto host_free = [&](void * p) { if (r->host_plain) free(p); else (void) hipHostFree(p); };
    const size_t need = ctx->is_last ? (size_t) n_tokens * ctx->model->hparams.n_vocab : 0;
    if (need > r->logits_cap) {
        if (r->logits_host) host_free(r->logits_host);
        if (r->topk_ids) host_free(r->topk_ids);
        r->logits_cap = std::max(need, (size_t) 16 * ctx->model->hparams.n_vocab);
        const size_t rows = r->logits_cap / ctx->model->hparams.n_vocab;
        if (!host_alloc((void **) &r->logits_host, r->logits_cap * 4)) return nullptr;
        if (!host_alloc((void **) &r->topk_ids, rows * (8 * 4 + 8 * 4 + 4))) return nullptr;
        r->topk_vals = (float *) (r->topk_ids + rows * 8);
        r->topk_sum  = r->topk_vals + rows * 8;
    }
    if (meta_bytes > r->meta_cap) {
        if (r->meta_host) host_free(r->meta_host);
        r->meta_cap = std::max(meta_bytes, (size_t) 4096);
        if (!host_alloc((void **) &r->meta_host, r->meta_cap)) return nullptr;
    }
    r->canceled = r->launched = r->finished = r->failed = false;
    return r;
}

// ---- the forward pass of this stage: layers [l0, l1) ----------------------------------------------------------------------
// Kernel sequence per Llama layer (10 launches + attention):
//   norm+quant -> wq, wk, wv -> rope+KV store -> attention -> quant -> wo(+residual) -> norm+quant -> up, gate -> silu*up+quant -> down(+residual)
// The fused launches compute exactly what the reference's separate graph nodes compute (same fp32 rounding points).
static bool trace_enabled() { static const bool on = getenv("PI_TRACE") != nullptr; return on; }     // debug dump of intermediates: read once
static void trace(llama_context * ctx, const char * what, int il, const float * x, int64_t n, int n_tok, int64_t stride = 0) {
    if (!trace_enabled()) return;
    if (!stride) stride = n;
    std::vector<float> h((size_t) stride * n_tok);
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipMemcpy(h.data(), x, h.size() * 4, hipMemcpyDeviceToHost);
    fprintf(stderr, "TRACE %d %-8s", il, what);
    for (int t = 0; t < n_tok; t++) { double s = 0; for (int64_t i = 0; i < n; i++) s += std::fabs((double) h[(size_t) t * stride + i]); fprintf(stderr, " %.6e", s); }
    fprintf(stderr, "\n");
}

static int vdt_of(const pi_weight * w) { return pi_vec_dot_type(pi_weight_type(w)); }

// norm(x) (* w + b) quantised for the given consumers into q8; falls back to fp32 + per-type quantisation when they disagree
static bool norm_for(llama_context * ctx, const float * x, const float * w, const float * b, int n, int n_tok, float eps, int rms, int vdt, void * q8) {
    OP_OK(pi_op_norm_quantize(x, w, b, nullptr, n, n_tok, eps, rms, vdt, q8, ctx->stream));
    return true;
}

// weights of the same type that read the same quantised activations go out as one launch (pi_op_mul_mat_multi)
static bool mul_mat_grouped(int n, const pi_weight * const * ws, const void * const * ins, float * const * outs, int n_tok, hipStream_t s) {
    bool done[3] = { false, false, false };
    for (int i = 0; i < n; i++) {
        if (done[i]) continue;
        const pi_weight * gw[3]; float * go[3]; int ng = 0;
        for (int j = i; j < n; j++)
            if (!done[j] && pi_weight_type(ws[j]) == pi_weight_type(ws[i]) && ins[j] == ins[i]) { gw[ng] = ws[j]; go[ng] = outs[j]; ng++; done[j] = true; }
        if (pi_op_mul_mat_multi(ng, gw, ins[i], n_tok, go, nullptr, s) != 0) return false;
    }
    return true;
}

// layers [la, lb) of this stage; `head`: the embedding lookup goes first, `tail`: final norm + lm_head + top-k go last
static bool forward(llama_context * ctx, int n_tok, int kv_head, int n_kv, int la, int lb, bool head, bool tail, bool multi = false) {
    llama_model * m = ctx->model;
    const pi_hparams & hp = m->hparams;
    const int n_embd = hp.n_embd, n_gqa = hp.n_embd_gqa(), n_ff = hp.n_ff, hd = hp.head_dim();
    hipStream_t s = ctx->stream;
    float * x = ctx->x, * x2 = ctx->x2;
    if (head && ctx->is_first) {
        OP_OK(pi_op_get_rows(m->tok_embd, ctx->tokens_dev, n_tok, x, s));
        if (m->pos_embd) {                                                                                          // build_starcoder: inpL + get_rows(pos_embd, inp_pos) (llama.cpp:4251-4255)
            OP_OK(pi_op_get_rows(m->pos_embd, ctx->pos_dev, n_tok, ctx->att, s));
            OP_OK(pi_op_add(x, ctx->att, x, (int64_t) n_tok * n_embd, s));
        }
        if (m->tok_norm) OP_OK(pi_op_norm(x, m->tok_norm, m->tok_norm_b, x, n_embd, n_tok, hp.f_norm_eps, 0, s));   // build_bloom: LayerNorm of the embeddings (llama.cpp:4650-4654)
    }
    const size_t kv_layer = (size_t) ctx->n_ctx * n_gqa;
    for (int il = la; il < lb; il++) {
        const pi_layer & L = m->layers[il];
        uint16_t * kc = ctx->k_cache + (size_t) (il - ctx->l0) * kv_layer, * vc = ctx->v_cache + (size_t) (il - ctx->l0) * kv_layer;
        if (hp.arch == PI_ARCH_LLAMA) {
            float * q = ctx->qkv, * k = q + (size_t) n_tok * n_embd, * v = k + (size_t) n_tok * n_gqa;
            const int vq = vdt_of(L.wq);
            if (!norm_for(ctx, x, L.attn_norm, nullptr, n_embd, n_tok, hp.f_norm_rms_eps, 1, vq, ctx->q8a)) return false;
            if (!trace_enabled() && pi_op_mul_mat_qkv_rope_ok(L.wq, L.wk, L.wv, 0, hd)) {
                // projections, RoPE, f16 rounding and the K/V cache writes in the mat-vec launch(es) themselves
                OP_OK(pi_op_mul_mat_qkv_rope(L.wq, L.wk, L.wv, ctx->q8a, n_tok, ctx->pos_dev, ctx->rope, hd, hp.n_head, hp.n_head_kv, kv_head, ctx->dyn_dev, ctx->q16, kc, vc, s));
            } else {
                if (vdt_of(L.wk) != vq && !norm_for(ctx, x, L.attn_norm, nullptr, n_embd, n_tok, hp.f_norm_rms_eps, 1, vdt_of(L.wk), ctx->q8c)) return false;
                if (vdt_of(L.wv) != vq && vdt_of(L.wv) != vdt_of(L.wk) && !norm_for(ctx, x, L.attn_norm, nullptr, n_embd, n_tok, hp.f_norm_rms_eps, 1, vdt_of(L.wv), ctx->q8c)) return false;
                {   // projections of the same type share one launch
                    const pi_weight * ws[3] = { L.wq, L.wk, L.wv };
                    const void * ins[3] = { ctx->q8a, vdt_of(L.wk) == vq ? ctx->q8a : ctx->q8c, vdt_of(L.wv) == vq ? ctx->q8a : ctx->q8c };
                    float * outs[3] = { q, k, v };
                    if (!mul_mat_grouped(3, ws, ins, outs, n_tok, s)) { ctx->last_error = std::string("mul_mat_grouped: ") + pi_last_error(); return false; }
                }
                trace(ctx, "q", il, q, n_embd, n_tok); trace(ctx, "k", il, k, n_gqa, n_tok); trace(ctx, "v", il, v, n_gqa, n_tok);
                OP_OK(pi_op_rope_kv_store(q, k, v, n_embd, n_gqa, n_gqa, ctx->pos_dev, ctx->rope, 0, hd, hp.n_head, hp.n_head_kv, n_tok, kv_head, ctx->q16, kc, vc, ctx->dyn_dev, s));
            }
            OP_OK(pi_op_attention_ex(ctx->q16, kc, vc, ctx->mask_dev, hd, hp.n_head, hp.n_head_kv, n_tok, n_kv, ctx->att, ctx->dyn_dev, multi ? 1 : 0, s));
            trace(ctx, "att", il, ctx->att, n_embd, n_tok);
            OP_OK(pi_op_quantize(vdt_of(L.wo), ctx->att, n_embd, n_tok, ctx->q8a, s));
            OP_OK(pi_op_mul_mat(L.wo, ctx->q8a, n_tok, x2, x, s));                                                   // ffn_inp = wo.att + inpSA
            const int vu = vdt_of(L.ffn_up);
            if (!norm_for(ctx, x2, L.ffn_norm, nullptr, n_embd, n_tok, hp.f_norm_rms_eps, 1, vu, ctx->q8a)) return false;
            if (vdt_of(L.ffn_gate) != vu && !norm_for(ctx, x2, L.ffn_norm, nullptr, n_embd, n_tok, hp.f_norm_rms_eps, 1, vdt_of(L.ffn_gate), ctx->q8c)) return false;
            {
                const pi_weight * ws[2] = { L.ffn_up, L.ffn_gate };
                const void * ins[2] = { ctx->q8a, vdt_of(L.ffn_gate) == vu ? ctx->q8a : ctx->q8c };
                float * outs[2] = { ctx->up, ctx->gate };
                if (!mul_mat_grouped(2, ws, ins, outs, n_tok, s)) { ctx->last_error = std::string("mul_mat_grouped: ") + pi_last_error(); return false; }
            }
            OP_OK(pi_op_act_quantize(ctx->gate, ctx->up, 0, n_ff, n_tok, vdt_of(L.ffn_down), ctx->q8b, s));        // silu(gate) * up
            OP_OK(pi_op_mul_mat(L.ffn_down, ctx->q8b, n_tok, x, x2, s));                                             // l_out = down(..) + ffn_inp
            trace(ctx, "lout", il, x, n_embd, n_tok);
        } else if (hp.generic()) {
            // the sequential block of build_baichuan / _starcoder / _refact / _bloom / _mpt / _stablelm: every operator in the reference's order, biases and the
            // residual as separate fp32 adds where the reference has separate ggml_add nodes
            const pi_block & B = hp.block;
            const float eps = B.rms ? hp.f_norm_rms_eps : hp.f_norm_eps;
            const int rms = B.rms ? 1 : 0;
            const int64_t qw = n_embd + 2 * n_gqa;
            const float * q, * k, * v; int64_t qs, ks, vs;
            if (B.fused_qkv) {
                if (!norm_for(ctx, x, L.attn_norm, L.attn_norm_b, n_embd, n_tok, eps, rms, vdt_of(L.wqkv), ctx->q8a)) return false;
                OP_OK(pi_op_mul_mat(L.wqkv, ctx->q8a, n_tok, ctx->qkv, nullptr, s));
                if (L.bqkv) OP_OK(pi_op_add_bias(ctx->qkv, L.bqkv, qw, n_tok, s));
                if (B.clamp_kqv > 0.0f) OP_OK(pi_op_clamp(ctx->qkv, -B.clamp_kqv, B.clamp_kqv, (int64_t) n_tok * qw, s));
                q = ctx->qkv; k = q + n_embd; v = k + n_gqa; qs = ks = vs = qw;
            } else {
                float * qo = ctx->qkv, * ko = qo + (size_t) n_tok * n_embd, * vo = ko + (size_t) n_tok * n_gqa;
                const int vq = vdt_of(L.wq);
                if (!norm_for(ctx, x, L.attn_norm, L.attn_norm_b, n_embd, n_tok, eps, rms, vq, ctx->q8a)) return false;
                if (vdt_of(L.wk) != vq && !norm_for(ctx, x, L.attn_norm, L.attn_norm_b, n_embd, n_tok, eps, rms, vdt_of(L.wk), ctx->q8c)) return false;
                if (vdt_of(L.wv) != vq && vdt_of(L.wv) != vdt_of(L.wk) && !norm_for(ctx, x, L.attn_norm, L.attn_norm_b, n_embd, n_tok, eps, rms, vdt_of(L.wv), ctx->q8c)) return false;
                const pi_weight * ws[3] = { L.wq, L.wk, L.wv };
                const void * ins[3] = { ctx->q8a, vdt_of(L.wk) == vq ? ctx->q8a : ctx->q8c, vdt_of(L.wv) == vq ? ctx->q8a : ctx->q8c };
                float * outs[3] = { qo, ko, vo };
                if (!mul_mat_grouped(3, ws, ins, outs, n_tok, s)) { ctx->last_error = std::string("mul_mat_grouped: ") + pi_last_error(); return false; }
                q = qo; k = ko; v = vo; qs = n_embd; ks = vs = n_gqa;
            }
            trace(ctx, "q", il, q, n_embd, n_tok, qs); trace(ctx, "k", il, k, n_gqa, n_tok, ks); trace(ctx, "v", il, v, n_gqa, n_tok, vs);
            if (B.qk_norm)                                                                                            // Persimmon: rows are [head][q | k | v]; per-head LayerNorm of Q and K, RoPE on half a head
                OP_OK(pi_op_persimmon_qkv_store(ctx->qkv, hp.n_head, hd, L.attn_q_norm, L.attn_q_norm_b, L.attn_k_norm, L.attn_k_norm_b, hp.f_norm_eps, ctx->pos_dev, ctx->rope,
                                                n_tok, kv_head, ctx->q16, kc, vc, ctx->dyn_dev, s));
            else
            // (without RoPE the context's table is the identity: cos 1, sin 0 — x0 * 1 - x1 * 0 is x0 bit for bit — so the f16 rounding and the cache writes stay one kernel)
            OP_OK(pi_op_rope_kv_store(q, k, v, qs, ks, vs, ctx->pos_dev, ctx->rope, B.rope == 2, hd, hp.n_head, hp.n_head_kv, n_tok, kv_head, ctx->q16, kc, vc, ctx->dyn_dev, s));
            OP_OK(pi_op_attention_alibi(ctx->q16, kc, vc, ctx->mask_dev, hd, hp.n_head, hp.n_head_kv, n_tok, n_kv, ctx->att, ctx->dyn_dev, multi ? 1 : 0, m->alibi_slopes, s));
            trace(ctx, "att", il, ctx->att, n_embd, n_tok);
            OP_OK(pi_op_quantize(vdt_of(L.wo), ctx->att, n_embd, n_tok, ctx->q8a, s));
            if (L.bo) {                                                                                               // wo . att, + bo, + inpL (llama.cpp:3766-3772, 4283)
                OP_OK(pi_op_mul_mat(L.wo, ctx->q8a, n_tok, x2, nullptr, s));
                OP_OK(pi_op_add_bias(x2, L.bo, n_embd, n_tok, s));
                OP_OK(pi_op_add(x2, x, x2, (int64_t) n_tok * n_embd, s));
            } else OP_OK(pi_op_mul_mat(L.wo, ctx->q8a, n_tok, x2, x, s));
            trace(ctx, "ffn_inp", il, x2, n_embd, n_tok);
            const int vu = vdt_of(L.ffn_up);
            if (!norm_for(ctx, x2, L.ffn_norm, L.ffn_norm_b, n_embd, n_tok, eps, rms, vu, ctx->q8a)) return false;
            if (B.gated) {                                                                                            // llm_build_ffn LLM_FFN_SILU / LLM_FFN_PAR (llama.cpp:3611-3695)
                if (vdt_of(L.ffn_gate) != vu && !norm_for(ctx, x2, L.ffn_norm, L.ffn_norm_b, n_embd, n_tok, eps, rms, vdt_of(L.ffn_gate), ctx->q8c)) return false;
                const pi_weight * ws[2] = { L.ffn_up, L.ffn_gate };
                const void * ins[2] = { ctx->q8a, vdt_of(L.ffn_gate) == vu ? ctx->q8a : ctx->q8c };
                float * outs[2] = { ctx->up, ctx->gate };
                if (!mul_mat_grouped(2, ws, ins, outs, n_tok, s)) { ctx->last_error = std::string("mul_mat_grouped: ") + pi_last_error(); return false; }
                OP_OK(pi_op_act_quantize(ctx->gate, ctx->up, 0, n_ff, n_tok, vdt_of(L.ffn_down), ctx->q8b, s));
            } else {                                                                                                  // LLM_FFN_GELU / LLM_FFN_SEQ: up (+ bias) -> gelu -> down (+ bias)
                OP_OK(pi_op_mul_mat(L.ffn_up, ctx->q8a, n_tok, ctx->up, nullptr, s));
                if (L.ffn_up_b) OP_OK(pi_op_add_bias(ctx->up, L.ffn_up_b, n_ff, n_tok, s));
                OP_OK(pi_op_act_quantize(ctx->up, nullptr, B.relu_sqr ? 2 : 1, n_ff, n_tok, vdt_of(L.ffn_down), ctx->q8b, s));
            }
            if (L.ffn_down_b) {
                OP_OK(pi_op_mul_mat(L.ffn_down, ctx->q8b, n_tok, x, nullptr, s));
                OP_OK(pi_op_add_bias(x, L.ffn_down_b, n_embd, n_tok, s));
                OP_OK(pi_op_add(x, x2, x, (int64_t) n_tok * n_embd, s));
            } else OP_OK(pi_op_mul_mat(L.ffn_down, ctx->q8b, n_tok, x, x2, s));
            trace(ctx, "lout", il, x, n_embd, n_tok);
        } else {
            const int64_t qw = n_embd + 2 * n_gqa;
            const int vqkv = vdt_of(L.wqkv), vup = vdt_of(L.ffn_up);
            // attn_norm feeds the MLP (llama.cpp:4194) and, without a second norm, the QKV projection too
            if (!norm_for(ctx, x, L.attn_norm, L.attn_norm_b, n_embd, n_tok, hp.f_norm_eps, 0, vup, ctx->q8c)) return false;
            const void * qkv_in = ctx->q8c;
            if (L.attn_norm_2) { if (!norm_for(ctx, x, L.attn_norm_2, L.attn_norm_2_b, n_embd, n_tok, hp.f_norm_eps, 0, vqkv, ctx->q8a)) return false; qkv_in = ctx->q8a; }
            else if (vqkv != vup) { if (!norm_for(ctx, x, L.attn_norm, L.attn_norm_b, n_embd, n_tok, hp.f_norm_eps, 0, vqkv, ctx->q8a)) return false; qkv_in = ctx->q8a; }
            OP_OK(pi_op_mul_mat(L.wqkv, qkv_in, n_tok, ctx->qkv, nullptr, s));
            OP_OK(pi_op_rope_kv_store(ctx->qkv, ctx->qkv + n_embd, ctx->qkv + n_embd + n_gqa, qw, qw, qw, ctx->pos_dev, ctx->rope, 1, hd, hp.n_head, hp.n_head_kv,
                                      n_tok, kv_head, ctx->q16, kc, vc, ctx->dyn_dev, s));
            trace(ctx, "q", il, ctx->qkv, n_embd, n_tok, qw); trace(ctx, "k", il, ctx->qkv + n_embd, n_gqa, n_tok, qw); trace(ctx, "v", il, ctx->qkv + n_embd + n_gqa, n_gqa, n_tok, qw);
            OP_OK(pi_op_attention_ex(ctx->q16, kc, vc, ctx->mask_dev, hd, hp.n_head, hp.n_head_kv, n_tok, n_kv, ctx->att, ctx->dyn_dev, multi ? 1 : 0, s));
            trace(ctx, "att", il, ctx->att, n_embd, n_tok);
            OP_OK(pi_op_quantize(vdt_of(L.wo), ctx->att, n_embd, n_tok, ctx->q8a, s));
            OP_OK(pi_op_mul_mat(L.wo, ctx->q8a, n_tok, x2, nullptr, s));                                             // ffn_inp = attention output
            trace(ctx, "wo", il, x2, n_embd, n_tok);
            OP_OK(pi_op_mul_mat(L.ffn_up, ctx->q8c, n_tok, ctx->up, nullptr, s));
            OP_OK(pi_op_act_quantize(ctx->up, nullptr, 1, n_ff, n_tok, vdt_of(L.ffn_down), ctx->q8b, s));           // gelu
            OP_OK(pi_op_mul_mat(L.ffn_down, ctx->q8b, n_tok, ctx->att, x2, s));                                      // ffn_out + ffn_inp
            OP_OK(pi_op_add(ctx->att, x, x, (int64_t) n_tok * n_embd, s));                                            // + inpL
            trace(ctx, "lout", il, x, n_embd, n_tok);
        }
    }
    if (tail && ctx->is_last) {
        const int vo = vdt_of(m->output);
        if (hp.arch == PI_ARCH_LLAMA || (hp.generic() && hp.block.rms)) { if (!norm_for(ctx, x, m->output_norm, nullptr, n_embd, n_tok, hp.f_norm_rms_eps, 1, vo, ctx->q8a)) return false; }
        else                                                              { if (!norm_for(ctx, x, m->output_norm, m->output_norm_b, n_embd, n_tok, hp.f_norm_eps, 0, vo, ctx->q8a)) return false; }
        OP_OK(pi_op_mul_mat(m->output, ctx->q8a, n_tok, ctx->logits_dev, nullptr, s));
        OP_OK(pi_op_topk(ctx->logits_dev, hp.n_vocab, n_tok, 8, ctx->topk_ids_dev, ctx->topk_vals_dev, ctx->topk_sum_dev, ctx->topk_scratch_dev, s));
    }
    return true;
}

// ---- launching a run: metadata block -> device, then the stage's forward pass as `n_segments` graph replays (or eager launches) ------------
static int n_segments(const llama_context * ctx) {
    const int nl = ctx->l1 - ctx->l0;
    return ctx->seg_layers > 0 && nl > 0 ? (nl + ctx->seg_layers - 1) / ctx->seg_layers : 1;
}
bool hop_store(void * dst, const void * src, size_t bytes, hipStream_t s);
// x_in (device, [n_tokens][n_embd]) replaces the embedding lookup on stages after the first
static bool launch_inputs(llama_context * ctx, int n_tokens, const void * meta, const float * x_in) {
    const int words = (ctx->n_ctx + 31) / 32;
    // block: tokens[n] | pos[n] | {kv_head, n_kv, n, has_table} | (cell[n], n_kv[n] when has_table: a pass that carries several runs) | mask[n][words]
    const int32_t * h_tok = (const int32_t *) meta, * h_pos = h_tok + n_tokens, * h_dyn = h_pos + n_tokens;
    const size_t n_table = h_dyn[3] ? (size_t) 2 * n_tokens : 0;
    const uint32_t * h_mask = (const uint32_t *) (h_dyn + 4 + n_table);
    hipStream_t s = ctx->stream;
    bool ok = hipMemcpyAsync(ctx->tokens_dev, h_tok, (size_t) 4 * n_tokens, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(ctx->pos_dev, h_pos, (size_t) 4 * n_tokens, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(ctx->dyn_dev, h_dyn, 16 + 4 * n_table, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(ctx->mask_dev, h_mask, (size_t) 4 * n_tokens * words, hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok && x_in) ok = hop_store(ctx->x, x_in, (size_t) 4 * n_tokens * ctx->model->hparams.n_embd, s);      // device (this stage's ring: a kernel) or host (staged transport: a copy)
    if (!ok) ctx->last_error = "H2D of batch metadata failed";
    return ok;
}
// PI_NO_ABORT_WORD=1 (A/B measurements, tests): kernels are launched without an abort word, cancels then act between segments only
static bool abort_checks_on() { static const bool on = getenv("PI_NO_ABORT_WORD") == nullptr; return on; }
// ahead of a run's kernels, in stream order: the abort word takes the run's identity (and a cancel that has arrived already)
static bool launch_abort_begin(llama_context * ctx, int32_t run_id) {
    if (!ctx->abort_dev) return true;
    if (pi_op_abort_begin(ctx->abort_dev, run_id, ctx->abort_host + (run_id & 255), ctx->stream) != 0) { ctx->last_error = std::string("pi_op_abort_begin: ") + pi_last_error(); return false; }
    return true;
}
// segment `seg` of the forward pass: every per-call quantity (tokens, positions, mask, kv_head, n_kv) lives in device memory that
// launch_inputs refreshed, so replaying the captured graph IS this call's segment
// acct_n_kv: the cells the pass really attends over (launch accounting of eager launches only; the kernels read n_kv from dyn_dev)
static bool launch_segment(llama_context * ctx, int n_tokens, int kv_head, int seg, pi_kt_sink ** sink_out, bool multi = false, bool capture_only = false, int acct_n_kv = 0) {
    const bool tracing = trace_enabled();
    pi_set_cu_count(ctx->cu_hi > ctx->cu_lo ? ctx->cu_hi - ctx->cu_lo : 0);     // (thread-local: the launchers of this thread size their grids for the CUs this context's stream may use)
    const int nseg = n_segments(ctx);
    const int la = ctx->seg_layers > 0 ? ctx->l0 + seg * ctx->seg_layers : ctx->l0;
    const int lb = ctx->seg_layers > 0 ? std::min(ctx->l1, la + ctx->seg_layers) : ctx->l1;
    const bool head = seg == 0, tail = seg == nseg - 1;
    hipStream_t s = ctx->stream;
    if (ctx->use_graphs && !tracing) {
        llama_context::graph_entry * gep = nullptr;
        { std::lock_guard<std::mutex> glk(ctx->graphs_mu); gep = &ctx->graphs[std::make_pair(n_tokens, 2 * seg + (multi ? 1 : 0))]; }     // (passes of several runs: their own graphs — per-token n_kv in the attention launches)
        llama_context::graph_entry & ge = *gep;
        bool ok = true;
        if (!ge.exec) {
            // One capture at a time in the process, and a few attempts: on this runtime a capture (thread-local mode) is now and then invalidated by what ANOTHER
            // thread does meanwhile — the caller's thread starting the draft's first runs (pinned allocations, its own first captures) while this context's feeder
            // captures — and every launch then fails with "previous error during capture".  Nothing has run at that point: end the capture and record it again.
            static std::mutex capture_mu;
            std::lock_guard<std::mutex> lk(capture_mu);
            for (int attempt = 0; attempt < 8 && !ge.exec; attempt++) {
                hipGraph_t graph = nullptr;
                if (attempt) { (void) hipGetLastError(); std::this_thread::sleep_for(std::chrono::microseconds(300 * attempt)); }
                ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (ok) {
                    pi_kt_set_sink(&ge.sink);                      // launch counts are part of the graph; they are only read while timing is on
                    if (attempt) ge.sink.launches.clear();
                    pi_set_abort_word(abort_checks_on() ? ctx->abort_dev : nullptr);      // the captured kernels carry the context's abort word
                    const bool fwd = forward(ctx, n_tokens, kv_head, ctx->n_ctx, la, lb, head, tail, multi);
                    pi_set_abort_word(nullptr);
                    pi_kt_set_sink(nullptr);
                    ok = hipStreamEndCapture(s, &graph) == hipSuccess && fwd && graph != nullptr;
                }
                if (ok) ok = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0) == hipSuccess;
                if (graph) (void) hipGraphDestroy(graph);
                if (!ok) { ge.exec = nullptr; fprintf(stderr, "llama_decode: graph capture of %d token(s), segment %d, attempt %d failed (%s): %s\n", n_tokens, seg, attempt, hipGetErrorString(hipGetLastError()), ctx->last_error.str().c_str()); }
            }
            if (!ok) ctx->last_error = "HIP graph capture of the forward pass failed: " + ctx->last_error;
        }
        if (ok && !capture_only) { ok = hipGraphLaunch(ge.exec, s) == hipSuccess; if (ok) ge.replays.fetch_add(1, std::memory_order_relaxed); }
        if (ok && sink_out && seg == 0) *sink_out = &ge.sink;
        return ok;
    }
    if (capture_only) return true;
    pi_set_abort_word(abort_checks_on() ? ctx->abort_dev : nullptr);
    pi_set_account_n_kv(acct_n_kv);
    const bool fwd = forward(ctx, n_tokens, kv_head, ctx->n_ctx, la, lb, head, tail, multi);
    pi_set_account_n_kv(0);
    pi_set_abort_word(nullptr);
    return fwd;
}
static bool launch_outputs(llama_context * ctx, pi_run * run) {
    if (!ctx->is_last) return true;
    hipStream_t s = ctx->stream;
    const size_t nv = ctx->model->hparams.n_vocab, r0 = (size_t) run->row0; const int n_tokens = run->n_tokens;
    // by a kernel that writes the run's page-locked buffers (pi_op_copy_out: a copy-engine transfer waits for the OTHER context's kernels); PIPEINFER_COPY_OUT=memcpy: the copies of rounds 1-3
    static const bool by_memcpy = getenv("PIPEINFER_COPY_OUT") && !strcmp(getenv("PIPEINFER_COPY_OUT"), "memcpy");
    if (!by_memcpy && !run->host_plain) {
        if (pi_op_copy_out(ctx->logits_dev + r0 * nv, run->logits_host, (int64_t) n_tokens * (int64_t) nv, ctx->topk_ids_dev + r0 * 8, run->topk_ids, ctx->topk_vals_dev + r0 * 8, run->topk_vals,
                           ctx->topk_sum_dev + r0, run->topk_sum, n_tokens, s) != 0) { ctx->last_error = std::string("pi_op_copy_out: ") + pi_last_error(); return false; }
        return true;
    }
    const bool ok = hipMemcpyAsync(run->logits_host, ctx->logits_dev + r0 * nv, (size_t) 4 * n_tokens * nv, hipMemcpyDeviceToHost, s) == hipSuccess &&
                    hipMemcpyAsync(run->topk_ids, ctx->topk_ids_dev + r0 * 8, (size_t) 4 * n_tokens * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
                    hipMemcpyAsync(run->topk_vals, ctx->topk_vals_dev + r0 * 8, (size_t) 4 * n_tokens * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
                    hipMemcpyAsync(run->topk_sum, ctx->topk_sum_dev + r0, (size_t) 4 * n_tokens, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (!ok) ctx->last_error = "D2H of logits failed";
    return ok;
}
static bool launch_forward(llama_context * ctx, int n_tokens, const void * meta, const float * x_in, pi_kt_sink ** sink_out) {
    if (!launch_inputs(ctx, n_tokens, meta, x_in)) return false;
    const int kv_head = ((const int32_t *) meta)[2 * n_tokens];
    const int n_kv = ((const int32_t *) meta)[2 * n_tokens + 1];
    for (int seg = 0, n = n_segments(ctx); seg < n; seg++) if (!launch_segment(ctx, n_tokens, kv_head, seg, sink_out, false, false, n_kv)) return false;
    return true;
}

// ---- device arbiter: contexts with a positive stream priority (the draft of a single-GPU run) go first -------------------------------------
// While such a context has a run pending, a feeder does not hand the GPU its next segment — except for ONE segment per completed urgent
// run (a credit).  The driver issues draft steps back to back and, with continuous speculation, never stops by itself while the target
// is not back: without the credit the target would make no progress at all while the draft is busy; with it the two alternate, one target
// segment per draft step, whatever the timing of the calls.
static std::atomic<int> g_urgent{0}, g_credit{0};
// (every context counts its own pending urgent runs: a priority context that is freed, or changes its priority, with runs it never finished must not
// leave g_urgent raised — every segmented feeder in the process would wait for a credit that no one is left to give)
static void urgent_begin(llama_context * ctx) { if (ctx->stream_priority > 0) { ctx->urgent_pending++; g_urgent.fetch_add(1); } }
static void urgent_end(llama_context * ctx) { if (ctx->urgent_pending > 0) { ctx->urgent_pending--; g_credit.store(1); g_urgent.fetch_sub(1); } }
void urgent_drop_all(llama_context * ctx) { while (ctx->urgent_pending > 0) urgent_end(ctx); }
static bool yield_to_urgent(const llama_context * ctx) {                      // true: it had to wait for an urgent run
    if (ctx->stream_priority > 0 || g_urgent.load() <= 0) return false;
    // bounded: the claim is released when the CALLER finishes the urgent run, and a caller that blocks on THIS context's run first would wait for ever.
    // 50 ms is several times the longest legitimate wait (a draft step queued behind one 70B target segment, ~7 ms).
    const int64_t t0 = time_us();
    for (int spin = 0; g_urgent.load() > 0; spin++) {
        int c = g_credit.load();
        if (c > 0 && g_credit.compare_exchange_strong(c, c - 1)) return true;
        if (spin < 4000) std::this_thread::yield();
        else { if (time_us() - t0 > 50000) return true; std::this_thread::sleep_for(std::chrono::microseconds(20)); }
    }
    return true;
}

// ---- feeder thread of a segmented context ---------------------------------------------------------------------------------------------------
static bool runs_cancelled(const std::vector<pi_run *> & runs) {
    for (const pi_run * r : runs) if (!r->canceled && !r->cancel_req.load()) return false;
    return true;
}
static bool pass_cancelled(const pi_pass & p) { return runs_cancelled(p.runs); }    // (callers hold fmu: finished members leave p.runs under it)
// May run q join a pass whose members (all started before q) are `members`?  Inside a pass every token's K/V of a layer are stored before that layer's attention, so a
// member's reads of an EARLIER member's cells see what they would see one run after the other.  What a pass cannot reproduce is a cell that changes hands inside it: q
// writing a cell that a member also writes, or that a member attends to expecting older content — e.g. the PipeInfer driver frees the cell of a speculative run's first
// token and hands it to the non-speculative run of the same token while the speculative run is still queued (speculative.cpp:1252-1255, find_slot llama.cpp:1593-1644).
// One case is exact all the same, and it is that one: the two tokens are the SAME computation — same token id, position, cell, n_kv and visibility row — so they store
// identical bits, and whoever reads the cell reads those bits whichever store lands last.
static bool pass_conflict_n(int n_ctx, const std::vector<pi_run *> & members, const pi_run * q) {
    const int words = (n_ctx + 31) / 32;
    auto tok = [](const pi_run * r) { return (const int32_t *) r->meta_host; };
    auto msk = [&](const pi_run * r, int t) { return (const uint32_t *) ((const int32_t *) r->meta_host + 2 * r->n_tokens + 4) + (size_t) t * words; };
    for (int j = 0; j < q->n_tokens; j++) {
        const int x = q->kv_head + j;
        bool touched = false, same = false;                                    // a member writes or reads x / a member's token at x is the same computation as q's token j
        for (const pi_run * m : members) {
            if (x >= m->kv_head && x < m->kv_head + m->n_tokens) {
                touched = true;
                const int i = x - m->kv_head;
                if (tok(m)[i] == tok(q)[j] && tok(m)[m->n_tokens + i] == tok(q)[q->n_tokens + j] && m->n_kv == q->n_kv && memcmp(msk(m, i), msk(q, j), (size_t) words * 4) == 0) same = true;
                else return true;                                              // two different tokens written to one cell in one pass
            } else {
                for (int i = 0; i < m->n_tokens && !touched; i++) touched = (msk(m, i)[x >> 5] >> (x & 31)) & 1u;
            }
        }
        if (touched && !same) return true;                                     // a member expects older content in x
    }
    return false;
}

static bool pass_conflict(const llama_context * ctx, const std::vector<pi_run *> & members, const pi_run * q) { return pass_conflict_n(ctx->n_ctx, members, q); }

// one input block for a pass of several runs: their tokens, positions and mask rows one after the other, the cell of every token, n_kv = the widest run's
// (cells past a run's own n_kv are masked for its tokens and contribute exact zeros, like every masked cell inside it)
static const void * compose_pass(llama_context * ctx, const std::vector<pi_run *> & members, int n_tokens) {
    const int n = n_tokens, words = (ctx->n_ctx + 31) / 32;
    const size_t need = (size_t) n * 8 + 16 + (size_t) n * 8 + (size_t) n * words * 4;
    if (need > ctx->pass_host_cap) {
        const size_t cap = std::max(need, (size_t) std::max(ctx->coalesce_max.load(), 1) * (16 + (size_t) words * 4) + 16);
        (void) hipStreamSynchronize(ctx->stream);                              // (an earlier pass may still be copying out of the old blocks)
        for (uint8_t *& b : ctx->pass_host) { if (b) (void) hipHostFree(b); b = nullptr; }
        for (uint8_t *& b : ctx->pass_host) if (hipHostMalloc((void **) &b, cap, hipHostMallocDefault) != hipSuccess) { b = nullptr; ctx->last_error = "out of pinned host memory (pass staging)"; return nullptr; }
        ctx->pass_host_cap = cap;
    }
    uint8_t * blk = ctx->pass_host[ctx->pass_slot++ & 3];
    int32_t * tok = (int32_t *) blk, * pos = tok + n, * dyn = pos + n, * cell = dyn + 4, * nkv = cell + n; uint32_t * mask = (uint32_t *) (nkv + n);
    int at = 0, n_kv = 0;
    for (const pi_run * r : members) {
        const int m = r->n_tokens;
        const int32_t * rt = (const int32_t *) r->meta_host, * rp = rt + m, * rd = rp + m; const uint32_t * rm = (const uint32_t *) (rd + 4);
        memcpy(tok + at, rt, (size_t) 4 * m); memcpy(pos + at, rp, (size_t) 4 * m);
        for (int i = 0; i < m; i++) { cell[at + i] = rd[0] + i; nkv[at + i] = rd[1]; }
        memcpy(mask + (size_t) at * words, rm, (size_t) 4 * m * words);
        n_kv = std::max(n_kv, (int) rd[1]);
        at += m;
    }
    dyn[0] = cell[0]; dyn[1] = n_kv; dyn[2] = n; dyn[3] = 1;
    return blk;
}
// Round 6 (tools_dev/timeline.py, profiles/r06_timeline_*): a cancelled run at the HEAD of the queue used to travel as a pass of its own, and the feeder then sat in the
// reaction window of that empty pass until the caller finished the cancelled run - which the PipeInfer driver does only between two speculative runs (three blocking draft
// steps, ~4-6 ms): every cancelled run cost the target one spec-run's drafting time of idle GPU.  Now they are completed on the spot (events only) and the feeder goes on
// to the first live run.  PIPEINFER_FEED_SKIP_CANCELLED=0: the old behaviour (A/B).
static bool feed_skips_cancelled() { static const bool on = !(getenv("PIPEINFER_FEED_SKIP_CANCELLED") && atoi(getenv("PIPEINFER_FEED_SKIP_CANCELLED")) == 0); return on; }
// enqueues one pass (one run, or several runs the feeder took together) on the context's stream; every member's events and state are final afterwards
static void feed_pass(llama_context * ctx, const std::shared_ptr<pi_pass> & pass) {
    // the members as they are NOW: once a member reaches state 2 the caller may finish it, which erases it from pass->runs (under fmu) and recycles the object —
    // the loops below walk this copy, and the last thing the feeder does with a member is that very store
    const int64_t t_feed0 = time_us();
    std::vector<pi_run *> runs;
    { std::lock_guard<std::mutex> lk(ctx->fmu); runs = pass->runs; }
    for (pi_run * r : runs) r->state.store(1);
    bool ok = true, full = false;
    for (pi_run * r : runs) ok = hipEventRecord(r->begin, ctx->stream) == hipSuccess && ok;
    hipEvent_t seg_done[2] = { ctx->seg_events[0], ctx->seg_events[1] };
    const int n = pass->n_tokens;
    bool gpu_work = false;
    if (!runs_cancelled(runs)) {
        gpu_work = true;
        const void * meta = runs.size() == 1 ? (const void *) runs[0]->meta_host : compose_pass(ctx, runs, pass->n_tokens);
        ok = ok && meta != nullptr;
        const int nseg = n_segments(ctx), kv_head = ok ? ((const int32_t *) meta)[2 * n] : 0;
        if (ok) ok = launch_inputs(ctx, n, meta, nullptr);
        int seg = 0;
        for (; ok && seg < nseg; seg++) {
            if (seg >= 2) (void) hipEventSynchronize(seg_done[seg & 1]);       // at most two segments ahead of the GPU
            (void) yield_to_urgent(ctx);
            if (runs_cancelled(runs)) break;                                    // llama_cancel_run reached every run of the pass while it was in flight
            ok = launch_abort_begin(ctx, pass->id);
            if (ok) ok = launch_segment(ctx, n, kv_head, seg, &runs[0]->sink, runs.size() > 1, false, pass->n_kv);
            if (ok) ok = hipEventRecord(seg_done[seg & 1], ctx->stream) == hipSuccess;
        }
        for (pi_run * r : runs) r->segments_run = seg;
        ctx->stat_segments += (uint64_t) seg; ctx->stat_segments_skipped += (uint64_t) (nseg - seg);
        full = ok && seg == nseg;
        if (full) for (pi_run * r : runs) ok = launch_outputs(ctx, r) && ok;
        if (full) { ctx->stat_passes++; ctx->stat_pass_runs += runs.size(); ctx->stat_pass_tokens += (uint64_t) n; if (runs.size() > 1) ctx->stat_coalesced_passes++; }
    }
    for (pi_run * r : runs) {
        r->launched = ok && full;
        if (hipEventRecord(r->done, ctx->stream) != hipSuccess) ok = false;
    }
    // (a pass that put no work on the GPU - every member was cancelled before its first segment - is nothing the caller reacts to: no reaction window behind it)
    if ((gpu_work || !feed_skips_cancelled()) && ctx->coalesce_max.load() > 1 && ctx->pass_event && hipEventRecord(ctx->pass_event, ctx->stream) == hipSuccess) { ctx->pass_pending = true; ctx->pass_first_id = runs[0]->run_id; ctx->pass_fed_us = t_feed0; }
    if (!ok) {                                                                 // NOT the same thing as a cancelled run: finish reports it (the driver ends the generation)
        for (pi_run * r : runs) r->failed = true;
        fprintf(stderr, "llama_decode (feeder): %s\n", ctx->last_error.str().c_str());
    }
    for (pi_run * r : runs) r->state.store(2);
}
// coalescing contexts: the feeder does not run ahead of the GPU across passes, and before it composes the next pass it gives the caller the chance to react to the one
// that has just finished — cancel what it invalidates, start what follows from it — so that those runs travel together.  Ends at once when the caller is blocked waiting
// for a run, when it has finished a run of that pass and called start / iprobe since, or after PIPEINFER_REACTION_US (10 ms).
// Measured (tools_dev/reaction_sweep.sh): a window about as long as ONE draft step cuts the caller off in the middle of its reaction — 3 ms gave 97 tok/s on 70B + TinyLlama (2.5 ms
// per draft step) against 102 from 6 ms up, and 65 against 76 on Falcon-40B + Falcon-7B (4.6 ms per step); beyond two steps the other conditions end the wait first, 7B is indifferent.
static void wait_for_reaction(llama_context * ctx) {
    if (!ctx->pass_pending) return;
    (void) hipEventSynchronize(ctx->pass_event);
    ctx->pass_pending = false;
    ctx->stat_reaction_windows++;
    tl("pass_done", ctx->pass_first_id);
    // bounded by PIPEINFER_REACTION_US (10 ms) and by three quarters of what the pass took from its hand-over to here (a context whose passes take a millisecond does not sit
    // out ten for a caller that is busy rather than blocked); the variable, when set, is the bound whatever the pass took
    static const bool tmo_fixed = getenv("PIPEINFER_REACTION_US") != nullptr;
    static const int64_t tmo_max = getenv("PIPEINFER_REACTION_US") ? atoll(getenv("PIPEINFER_REACTION_US")) : 10000;
    const int64_t t0 = time_us();
    const int64_t tmo = tmo_fixed ? tmo_max : std::min<int64_t>(tmo_max, std::max<int64_t>(500, (t0 - ctx->pass_fed_us) * 3 / 4));
    for (int spin = 0; ; spin++) {
        if (ctx->waiters.load() > 0) { tl("react_end", 1); return; }
        if (ctx->last_finished_id.load() >= ctx->pass_first_id && ctx->caller_epoch.load() > ctx->epoch_after_finish.load()) { tl("react_end", 2); return; }
        if (time_us() - t0 > tmo) { tl("react_end", 3); return; }
        { std::lock_guard<std::mutex> lk(ctx->fmu); if (ctx->fstop) return; }
        if (spin < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}
static void feeder_main(llama_context * ctx) {
    (void) hipSetDevice(ctx->model->device);
    while (true) {
        {
            std::unique_lock<std::mutex> lk(ctx->fmu);
            if (ctx->fqueue.empty()) tl("feeder_idle");
            ctx->fcv.wait(lk, [&] { return ctx->fstop || !ctx->fqueue.empty(); });
            if (ctx->fqueue.empty()) return;                                   // fstop and drained
        }
        const int cmax = ctx->coalesce_max.load();
        bool coalesce = cmax > 1;
        if (coalesce && !ctx->pass_event && hipEventCreateWithFlags(&ctx->pass_event, hipEventDisableTiming) != hipSuccess) { ctx->pass_event = nullptr; coalesce = false; }
        if (coalesce) wait_for_reaction(ctx); else ctx->pass_pending = false;
        std::vector<std::shared_ptr<pi_pass>> todo;                            // cancelled runs met on the way (no GPU work, events only), then the pass itself
        auto single = [&](pi_run * r) { auto p = std::make_shared<pi_pass>(); p->runs.push_back(r); p->id = r->run_id; p->n_tokens = r->n_tokens; p->n_kv = r->n_kv; r->pass = p; r->row0 = 0; r->pass_leader = true; r->pass_tokens = r->n_tokens; return p; };
        auto feed = [&](std::vector<std::shared_ptr<pi_pass>> & v) { for (auto & p : v) { tl("pass", p->id, (int) p->runs.size(), p->n_tokens); feed_pass(ctx, p); tl("fed", p->id); } v.clear(); };
        pi_run * first = nullptr;
        std::shared_ptr<pi_pass> pass;
        {
            std::lock_guard<std::mutex> lk(ctx->fmu);
            if (coalesce && feed_skips_cancelled())                             // cancelled runs at the head: completed on the spot, no reaction window of their own
                while (!ctx->fqueue.empty() && (ctx->fqueue.front()->canceled || ctx->fqueue.front()->cancel_req.load())) { todo.push_back(single(ctx->fqueue.front())); ctx->fqueue.pop_front(); }
            if (!ctx->fqueue.empty()) { first = ctx->fqueue.front(); ctx->fqueue.pop_front(); pass = single(first); }
        }
        feed(todo);
        if (!first) continue;
        {
            std::lock_guard<std::mutex> lk(ctx->fmu);
            if (coalesce && !first->canceled && !first->cancel_req.load()) {
                std::vector<pi_run *> dropped;
                while (!ctx->fqueue.empty()) {
                    pi_run * q = ctx->fqueue.front();
                    if (q->canceled || q->cancel_req.load()) { ctx->fqueue.pop_front(); dropped.push_back(q); continue; }
                    if (pass->n_tokens + q->n_tokens > cmax || pass_conflict(ctx, pass->runs, q)) break;
                    ctx->fqueue.pop_front();
                    q->pass = pass; q->row0 = pass->n_tokens; q->pass_leader = false;
                    pass->runs.push_back(q); pass->n_tokens += q->n_tokens; pass->n_kv = std::max(pass->n_kv, q->n_kv);
                }
                for (pi_run * r : pass->runs) r->pass_tokens = pass->n_tokens;
                for (pi_run * r : dropped) todo.push_back(single(r));
            }
            todo.push_back(pass);
        }
        feed(todo);
        if (g_timeline_on && coalesce && ctx->pass_pending) { (void) hipEventSynchronize(ctx->pass_event); tl("pass_gpu_done", ctx->pass_first_id); }   // (timeline only: the host clock of the pass leaving the GPU; the feeder would wait for it before its next pass anyway)
    }
}

pi_run * context_start(llama_context * ctx, const llama_batch & batch_in, int & rc) {
    rc = -1;
    std::string err;
    if (!context_prepare(ctx, err)) { ctx->last_error = err; fprintf(stderr, "llama_decode: %s\n", err.c_str()); return nullptr; }
    tl("start_in", ctx->stream_priority);
    const int n_tokens = batch_in.n_tokens;
    if (n_tokens <= 0) { ctx->last_error = "n_tokens == 0"; return nullptr; }
    if (n_tokens > ctx->n_batch) { ctx->last_error = "n_tokens > n_batch"; fprintf(stderr, "llama_decode: n_tokens (%d) > n_batch (%d)\n", n_tokens, ctx->n_batch); return nullptr; }
    if (!batch_in.token) { ctx->last_error = "embedding input batches are not supported on this path"; return nullptr; }
    // defaults for the legacy all_pos/all_seq_id form (llama.cpp:5519-5543)
    std::vector<llama_pos> pos_buf; std::vector<int32_t> nseq_buf; std::vector<llama_seq_id> seq_store; std::vector<llama_seq_id *> seq_ptrs;
    const llama_pos * pos = batch_in.pos; const int32_t * n_seq_id = batch_in.n_seq_id; llama_seq_id * const * seq_id = batch_in.seq_id;
    if (!pos) { pos_buf.resize(n_tokens); for (int i = 0; i < n_tokens; i++) pos_buf[i] = batch_in.all_pos_0 + i * batch_in.all_pos_1; pos = pos_buf.data(); }
    if (!seq_id) {
        nseq_buf.assign(n_tokens, 1); seq_store.assign(n_tokens, batch_in.all_seq_id); seq_ptrs.resize(n_tokens);
        for (int i = 0; i < n_tokens; i++) seq_ptrs[i] = &seq_store[i];
        n_seq_id = nseq_buf.data(); seq_id = seq_ptrs.data();
    }
    for (int i = 0; i < n_tokens; i++) if (pos[i] < 0 || pos[i] >= std::max(ctx->n_ctx, 32)) { ctx->last_error = "token position outside the RoPE table"; return nullptr; }

    ctx->cells->pre_decode(n_tokens);
    const int kv_head = ctx->cells->find_slot(n_tokens, pos, n_seq_id, seq_id);
    if (kv_head < 0) {                                                        // reference prints "Cannot find cache slot" and fails the decode (llama.cpp:5560-5563)
        ctx->last_error = "no KV cache slot for a batch of " + std::to_string(n_tokens) + " token(s): n_ctx " + std::to_string(ctx->n_ctx) + " is too small for the runs in flight";
        rc = 1; return nullptr;
    }
    const int n_kv = ctx->cells->n_attend();
    const int words = (ctx->n_ctx + 31) / 32;                                   // fixed row stride of the visibility bitmap
    const size_t meta_bytes = (size_t) n_tokens * 8 + 16 + (size_t) n_tokens * words * 4;
    pi_run * run = acquire_run(ctx, n_tokens, meta_bytes);
    // (a run that cannot be started gives the cells find_slot just allotted back: nothing was or will be written to them)
    auto give_back = [&]() { ctx->cells->release_cells(kv_head, n_tokens); };
    if (!run) { ctx->last_error = "out of pinned host memory"; give_back(); return nullptr; }
    run->batch_id = batch_in.batch_id; run->n_tokens = n_tokens; run->t_start_us = time_us(); run->sink = nullptr;
    run->run_id = (++ctx->run_counter) & 0x3fffffff; run->kv_head = kv_head; run->n_kv = n_kv;
    if (run->pass) {                                                          // the object is about to be recycled: it leaves its pass
        std::lock_guard<std::mutex> lk(ctx->fmu);
        auto & v = run->pass->runs;
        v.erase(std::remove(v.begin(), v.end(), run), v.end());
        run->pass.reset();
    } run->row0 = 0; run->pass_leader = true; run->pass_tokens = n_tokens;
    ctx->caller_epoch.fetch_add(1);
    // (a segmented context's stream belongs to its feeder thread: the begin event is recorded there, in stream order with the run's segments)
    if (!ctx->remote && ctx->seg_layers == 0 && hipEventRecord(run->begin, ctx->stream) != hipSuccess) { ctx->last_error = "hipEventRecord failed"; give_back(); ctx->pool.push_back(run); return nullptr; }
    // which rows phase 1 hands back (llama.cpp:5776-5789): flagged rows, or all with logits_all, or only the last one (stored at row 0)
    run->want_logits.assign(n_tokens, 1);
    run->last_only = false;
    if (batch_in.logits) for (int i = 0; i < n_tokens; i++) run->want_logits[i] = batch_in.logits[i] != 0;
    else if (!ctx->cparams.logits_all) run->last_only = true;
    int32_t * h_tok = (int32_t *) run->meta_host, * h_pos = h_tok + n_tokens, * h_dyn = h_pos + n_tokens; uint32_t * h_mask = (uint32_t *) (h_dyn + 4);
    memcpy(h_tok, batch_in.token, (size_t) 4 * n_tokens);
    memcpy(h_pos, pos, (size_t) 4 * n_tokens);
    h_dyn[0] = kv_head; h_dyn[1] = n_kv; h_dyn[2] = n_tokens; h_dyn[3] = 0;
    ctx->cells->build_mask(n_kv, n_tokens, pos, n_seq_id, seq_id, h_mask, words);
    ctx->cells->post_decode(n_tokens);

    run->meta_bytes = meta_bytes;
    if (ctx->run_tap) ctx->run_tap(ctx->run_tap_user, run->run_id, run->batch_id, n_tokens, run->meta_host, meta_bytes, nullptr);
    run->canceled = ctx->canceled_batches.count(run->batch_id) > 0;
    if (ctx->abort_host) ctx->abort_host[run->run_id & 255] = run->canceled ? 1 : 0;
    bool ok = true;
    if (ctx->remote) {                                                        // the layers live in other processes: ship the metadata block
        if (!run->canceled) {
            run->run_seq = ctx->remote_seq++;
            ok = ctx->remote_ops.start(ctx->remote_ops.user, run->run_seq, run->batch_id, n_tokens, run->meta_host, meta_bytes) == 0;
            if (!ok) ctx->last_error = "remote start failed";
            run->launched = ok;
            ctx->stat_calls++;
        }
        if (!ok) { fprintf(stderr, "llama_decode: %s\n", ctx->last_error.str().c_str()); give_back(); ctx->pool.push_back(run); return nullptr; }
        ctx->inflight.push_back(run);
        rc = 0;
        return run;
    }
    run->state.store(0); run->cancel_req.store(false); run->segments_run = 0;
    tl("start", ctx->stream_priority, run->run_id, n_tokens);
    urgent_begin(ctx);
    if (ctx->seg_layers > 0) {                                                // segmented: the feeder thread launches (and may cut short) the run
        if (!ctx->feeder.joinable()) {
            if (hipEventCreateWithFlags(&ctx->seg_events[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->seg_events[1], hipEventDisableTiming) != hipSuccess) {
                ctx->last_error = "hipEventCreate failed"; urgent_end(ctx); give_back(); ctx->pool.push_back(run); return nullptr;
            }
            ctx->feeder = std::thread(feeder_main, ctx);
        }
        ctx->inflight.push_back(run);
        { std::lock_guard<std::mutex> lk(ctx->fmu); ctx->fqueue.push_back(run); }
        ctx->fcv.notify_one();
        rc = 0;
        return run;
    }
    if (!run->canceled) {                                                     // a cancelled run still occupies its cells and completes "empty" (llama.cpp:5627-5716)
        ok = launch_abort_begin(ctx, run->run_id);
        if (ok) ok = launch_forward(ctx, n_tokens, run->meta_host, nullptr, &run->sink);
        if (ok) ok = launch_outputs(ctx, run);
        run->launched = ok;
    }
    if (ok) ok = hipEventRecord(run->done, ctx->stream) == hipSuccess;
    run->state.store(2);
    if (!ok) {
        // the cells find_slot handed to this run go back: nothing was (or will be) written to them
        fprintf(stderr, "llama_decode: %s\n", ctx->last_error.str().c_str());
        give_back();
        urgent_end(ctx);
        ctx->pool.push_back(run);
        return nullptr;
    }
    ctx->inflight.push_back(run);
    tl("start_out", ctx->stream_priority, run->run_id);
    rc = 0;
    return run;
}

// captures the graphs of every pass width up to max_tokens NOW, on the calling thread, while nothing else uses the context: a graph captured later, by the feeder in the
// middle of a generation, competes with the caller's own launches (draft runs, cancels) — on this runtime a capture is now and then invalidated by them
bool context_precapture(llama_context * ctx, int max_tokens) {
    if (!ctx->prepared || !ctx->use_graphs || trace_enabled() || ctx->remote) return true;
    { std::lock_guard<std::mutex> lk(ctx->fmu); if (!ctx->fqueue.empty()) return true; }
    if (!ctx->inflight.empty()) return true;
    const int nseg = n_segments(ctx);
    for (int n = 1; n <= std::min(max_tokens, ctx->n_batch); n++)
        for (int multi = 0; multi < 2; multi++) {
            if (multi && n < 2) continue;
            for (int seg = 0; seg < nseg; seg++)
                if (!launch_segment(ctx, n, 0, seg, nullptr, multi != 0, true)) return false;
        }
    return hipStreamSynchronize(ctx->stream) == hipSuccess;
}

void context_cancel_run(llama_context * ctx, pi_run * r) {
    if (r->cancel_req.load()) return;
    // Not handed to the GPU yet: the feeder drops it whole.  Handed over (partly): the kernels of its pass return on entry from now on — the flag is picked up by the
    // pass's next pi_op_abort_begin, the compare-and-swap hits the pass if it is the one executing (kernel granularity, like the reference's poll between graph nodes:
    // ggml.c:15748-15811); a segmented context also stops launching its remaining segments.  A pass that carries several runs stops when ALL of them are cancelled.
    r->cancel_req.store(true);
    tl("cancel", ctx->stream_priority, r->run_id, r->state.load());
    int32_t id = r->run_id; bool all = true, handed = r->state.load() != 0;
    {
        std::lock_guard<std::mutex> lk(ctx->fmu);
        if (r->pass) { id = r->pass->id; all = pass_cancelled(*r->pass); for (const pi_run * q : r->pass->runs) handed = handed || q->state.load() != 0; }
    }
    if (!all) return;
    if (ctx->abort_host) ctx->abort_host[id & 255] = 1;
    if (ctx->abort_dev && ctx->ctl_stream && handed && pi_op_abort_cancel(ctx->abort_dev, id, ctx->ctl_stream) != 0)
        fprintf(stderr, "llama_cancel_run: %s\n", pi_last_error());
}

// Stops run `rid` of this stage — and only that run.  Everything below is keyed by the id: the host flag is the slot of THAT id, the compare-and-swap on the device word
// hits only while THAT run is the one executing, so a call that arrives after the serve loop has moved on to the next run changes nothing for it.
void stage_abort_run(llama_context * ctx, int32_t rid) {
    if (rid == 0 || ctx->stage_run_id.load(std::memory_order_acquire) != rid) return;
    if (!ctx->abort_host || !ctx->abort_dev || !ctx->ctl_stream) return;
    volatile int32_t * flag = ctx->abort_host + (rid & 255);
    if (*flag) return;
    *flag = 1;                                                                 // picked up by the run's next pi_op_abort_begin ...
    (void) pi_op_abort_cancel(ctx->abort_dev, rid, ctx->ctl_stream);           // ... and hits the kernels already enqueued (only if that run is still the one executing)
}

bool context_poll_oldest(llama_context * ctx) {
    ctx->caller_epoch.fetch_add(1);
    if (ctx->inflight.empty()) return false;
    if (ctx->remote) return !ctx->inflight.front()->launched || ctx->remote_ops.poll(ctx->remote_ops.user) != 0;
    static const int64_t dbg_delay_us = getenv("PI_DEBUG_RUN_LATENCY_US") ? atoll(getenv("PI_DEBUG_RUN_LATENCY_US")) : 0;   // tests: pretend every run takes at least this long
    if (dbg_delay_us > 0 && time_us() - ctx->inflight.front()->t_start_us < dbg_delay_us) return false;
    if (ctx->inflight.front()->state.load() != 2) return false;                // the feeder has not recorded its completion event yet
    const bool back = hipEventQuery(ctx->inflight.front()->done) == hipSuccess;
    if (back) tl("probe_back", ctx->stream_priority, ctx->inflight.front()->run_id);
    return back;
}

bool context_wait(llama_context * ctx, pi_run * run) {
    if (ctx->remote) return true;                                             // context_finish blocks in the transport's wait()
    struct waiting { llama_context * c; waiting(llama_context * c_) : c(c_) { c->waiters.fetch_add(1); c->caller_epoch.fetch_add(1); } ~waiting() { c->waiters.fetch_sub(1); } } w(ctx);
    while (run->state.load() != 2) std::this_thread::sleep_for(std::chrono::microseconds(20));
    return hipEventSynchronize(run->done) == hipSuccess;
}

// One run on a pipeline stage (a process that owns layers [l0, l1)).  Mirrors the worker side of llama_decode_internal
// (llama.cpp:5588-5760): receive activations, compute the local graph, hand the result on; here the transport is the caller's.
// ---- the hop's store -------------------------------------------------------------------------------------------------------------------------
// A stage hands its activations to the next one by STORING them into the consumer's ring slot (a peer mapping of the next stage's HBM).  Until round 4 that was a
// hipMemcpyAsync; device memory is now written by a kernel of the producing stream - the copy-engine path of this runtime has been seen to wait for kernels of other queues
// (the draft's D2H beside the target), and a run's 32 KB per token are nothing a copy engine is needed for.  Host destinations (the staged transport) and
// PIPEINFER_HOP_STORE=memcpy keep the copy.
__global__ void __launch_bounds__(256) k_hop_store(const uint4 * __restrict__ src, uint4 * __restrict__ dst, size_t n16) {
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) dst[i] = src[i];
}
// How the store is done is decided ONCE, at the transport's set-up, from what the transport knows about its rings (hop_store_set_mode: 1 = kernel, both ends are device
// memory — this stage's HBM and the peer mapping of the next stage's ring; 2 = hipMemcpyAsync, host rings) and logged there per rank.  Mode 0 (no transport: the in-process
// stage chains of the tests) asks the runtime what the two pointers are; hipPointerGetAttributes on an IPC-mapped peer pointer is not something to rely on silently.
static std::atomic<int> g_hop_store_mode{0};
void hop_store_set_mode(int mode) { g_hop_store_mode.store(mode == 1 || mode == 2 ? mode : 0); }
int hop_store_mode() { return g_hop_store_mode.load(); }
bool hop_store(void * dst, const void * src, size_t bytes, hipStream_t s) {
    static const bool by_memcpy = getenv("PIPEINFER_HOP_STORE") && !strcmp(getenv("PIPEINFER_HOP_STORE"), "memcpy");
    bool dev = false;
    const int mode = g_hop_store_mode.load(std::memory_order_relaxed);
    const bool aligned = bytes % 16 == 0 && (((uintptr_t) dst | (uintptr_t) src) & 15) == 0;
    if (mode == 1) dev = !by_memcpy && aligned;
    else if (mode == 0 && !by_memcpy && aligned) {
        hipPointerAttribute_t a, b;                                             // both ends in device memory (the staged transport hands over host buffers)
        if (hipPointerGetAttributes(&a, dst) == hipSuccess && hipPointerGetAttributes(&b, src) == hipSuccess) dev = a.type == hipMemoryTypeDevice && b.type == hipMemoryTypeDevice;
        else (void) hipGetLastError();
    }
    if (!dev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s) == hipSuccess;
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(k_hop_store, dim3((unsigned) std::max<size_t>(1, std::min<size_t>(256, (n16 + 1023) / 1024))), dim3(256), 0, s, (const uint4 *) src, (uint4 *) dst, n16);
    return hipGetLastError() == hipSuccess;
}

int stage_run(llama_context * ctx, int n_tokens, const void * meta, size_t meta_bytes, const float * x_in, float * x_out,
              float * logits_host, int32_t * topk_ids, float * topk_vals, float * topk_sum, int (*should_abort)(void *), void * abort_user, const pi_stage_hop * hop) {
    std::string err;
    if (!context_prepare(ctx, err)) { ctx->last_error = err; return -1; }
    const int words = (ctx->n_ctx + 31) / 32;
    if (n_tokens <= 0 || n_tokens > ctx->n_batch || meta_bytes < (size_t) n_tokens * 8 + 16 + (size_t) n_tokens * words * 4) { ctx->last_error = "stage_run: bad metadata block"; return -1; }
    if (!ctx->is_first && !x_in) { ctx->last_error = "stage_run: activations missing"; return -1; }
    hipStream_t s = ctx->stream;
    const bool flag_in = hop && hop->in_flag && !ctx->is_first && ctx->abort_dev, flag_out = hop && hop->out_flag && !ctx->is_last;
    if ((flag_in || flag_out) && !ctx->hop_dev) {
        ctx->hop_dev = (uint8_t *) pi_malloc(64);
        if (!ctx->hop_dev || hipHostMalloc((void **) &ctx->hop_host, 64, hipHostMallocDefault) != hipSuccess) { ctx->last_error = "stage_run: no memory for the arrival-flag block"; return -1; }
        pi_memset(ctx->hop_dev, 0, 64, nullptr); memset(ctx->hop_host, 0, 64);
        if (hipDeviceSynchronize() != hipSuccess) { ctx->last_error = "stage_run: hipDeviceSynchronize failed"; return -1; }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr, ec = nullptr;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1); (void) hipEventCreate(&ec);
    // the run's identity on the device: its kernels carry the context's abort word, so a cancel that reaches this process while the stage is inside the run
    // (stage_abort_run, called by the transport's watcher) stops it at the next KERNEL, not at the next segment (llama.cpp:5665-5687 polls between graph nodes)
    const int32_t rid = (++ctx->run_counter) & 0x3fffffff;
    if (ctx->abort_host) ctx->abort_host[rid & 255] = 0;
    ctx->stage_run_id.store(rid);
    bool ok = true, cut = false;
    // Arrival flag: the upstream stage has only ANNOUNCED this run — the kernel below waits, on the device, until its activations have landed in x_in, and arms the abort
    // word (with the abort the upstream run ended with).  Everything behind it is enqueued now: when the flag arrives no host thread stands between the stages.
    // (e0 is recorded behind the wait: the GPU time of the run does not include the time it waited for its input.)
    if (flag_in) {
        if (pi_op_abort_begin_wait(ctx->abort_dev, rid, ctx->abort_host + (rid & 255), hop->in_flag, hop->in_ticket, hop->wait_timeout_us, (int32_t *) ctx->hop_dev, (uint64_t *) (ctx->hop_dev + 8), s) != 0) {
            ctx->last_error = std::string("pi_op_abort_begin_wait: ") + pi_last_error(); ok = false;
        }
    }
    (void) hipEventRecord(e0, s);
    // With an abort callback and a segmented context (pipeinfer_context_set_segments) the calling thread plays the feeder: segments go
    // out at most two ahead of the GPU and the callback is asked before each — the unlaunched rest of a cancelled run is never enqueued.
    const int nseg = n_segments(ctx);
    if (should_abort && nseg > 1) {
        if (!ctx->seg_events[0]) ok = ok && hipEventCreateWithFlags(&ctx->seg_events[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ctx->seg_events[1], hipEventDisableTiming) == hipSuccess;
        const int kv_head = ((const int32_t *) meta)[2 * n_tokens];
        if (ok) ok = launch_inputs(ctx, n_tokens, meta, ctx->is_first ? nullptr : x_in);
        int seg = 0;
        for (; ok && seg < nseg; seg++) {
            if (seg >= 2) (void) hipEventSynchronize(ctx->seg_events[seg & 1]);
            if (seg > 0 && should_abort(abort_user)) { cut = true; break; }
            if (!(flag_in && seg == 0)) ok = launch_abort_begin(ctx, rid);      // (the wait kernel armed the word for the first segment; later ones keep an abort it carries)
            if (ok) ok = launch_segment(ctx, n_tokens, kv_head, seg, nullptr, false, false, ((const int32_t *) meta)[2 * n_tokens + 1]);
            if (ok) ok = hipEventRecord(ctx->seg_events[seg & 1], s) == hipSuccess;
        }
        ctx->stat_segments += (uint64_t) seg; ctx->stat_segments_skipped += (uint64_t) (nseg - seg);
    } else {
        if (ok && !flag_in) ok = launch_abort_begin(ctx, rid);
        if (ok) ok = launch_forward(ctx, n_tokens, meta, ctx->is_first ? nullptr : x_in, nullptr);
    }
    const size_t nv = ctx->model->hparams.n_vocab;
    (void) hipEventRecord(ec, s);
    // the hop: the activations are stored by THIS stream straight into the consumer's ring slot (x_out: a peer mapping of the next stage's HBM, or host memory) ...
    if (ok && !cut && !ctx->is_last) ok = hop_store(x_out, ctx->x, (size_t) 4 * n_tokens * ctx->model->hparams.n_embd, s);
    // ... and the consumer's arrival flag right behind them (with the abort this run ended with: a run cut short here must not be computed downstream either)
    if (ok && flag_out && pi_op_flag_publish(hop->out_flag, hop->out_ticket, ctx->abort_dev, cut ? 1 : 0, s) != 0) { ctx->last_error = std::string("pi_op_flag_publish: ") + pi_last_error(); ok = false; }
    if (ok && !cut && ctx->is_last)
        ok = hipMemcpyAsync(logits_host, ctx->logits_dev, (size_t) 4 * n_tokens * nv, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipMemcpyAsync(topk_ids, ctx->topk_ids_dev, (size_t) 4 * n_tokens * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipMemcpyAsync(topk_vals, ctx->topk_vals_dev, (size_t) 4 * n_tokens * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipMemcpyAsync(topk_sum, ctx->topk_sum_dev, (size_t) 4 * n_tokens, hipMemcpyDeviceToHost, s) == hipSuccess;
    (void) hipEventRecord(e1, s);
    if (ok && flag_in)                                                          // what the wait saw (time-out, clock ticks) and the abort word the run ended with
        ok = hipMemcpyAsync(ctx->hop_host, ctx->hop_dev, 32, hipMemcpyDeviceToHost, s) == hipSuccess && hipMemcpyAsync(ctx->hop_host + 32, ctx->abort_dev, 4, hipMemcpyDeviceToHost, s) == hipSuccess;
    // a failed launch leaves the wait kernel (and whatever was enqueued) on the stream: raise the run's host flag — the wait kernel polls it — and always drain the stream
    // before the events are destroyed and the caller closes the transport (it used to return at once and the wait could spin for its whole time-out)
    if (!ok && ctx->abort_host) ctx->abort_host[rid & 255] = 1;
    if (hipStreamSynchronize(s) != hipSuccess) ok = false;
    ctx->stage_run_id.store(0);
    // cancelled while its (last) kernels were enqueued: they returned on entry, what the copies above moved is not a result
    if (ok && !cut && ((ctx->abort_host && ctx->abort_host[rid & 255]) || (should_abort && should_abort(abort_user)))) cut = true;
    if (ok && flag_in) {
        const int32_t w = *(const int32_t *) (ctx->hop_host + 32);
        if ((w >> 1) == rid && (w & 1)) cut = true;                            // aborted upstream (or while it waited): the kernels returned on entry
        if (*(const int32_t *) ctx->hop_host) { ctx->last_error = "stage_run: the run's activations never arrived (arrival flag timed out: the upstream stage is gone)"; ok = false; }
        const uint64_t * tk = (const uint64_t *) (ctx->hop_host + 8);
        ctx->stat_flag_seen_us = (double) tk[0] / 100.0; ctx->stat_flag_wait_ms = (double) tk[1] / 1e5; ctx->stat_flag_runs = tk[2];      // (running sums kept on the device)
    }
    float ms = 0.0f;
    if (ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) { ctx->stat_gpu_ms += ms; if (n_tokens == 1 && !cut) ctx->stat_gpu_ms_1 += ms; }
    if (ok && !cut && hipEventElapsedTime(&ms, ec, e1) == hipSuccess) ctx->stat_hop_ms += ms;
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); (void) hipEventDestroy(ec);
    if (!ok) { if (ctx->last_error.empty()) ctx->last_error = "stage_run failed"; return -1; }
    if (cut) return 1;
    const int32_t * h_dyn = (const int32_t *) meta + 2 * n_tokens;
    ctx->stat_calls++;
    ctx->stat_alg_bytes += decode_alg_bytes(ctx, n_tokens, h_dyn[1]);
    if (n_tokens == 1) { ctx->stat_calls_1++; ctx->stat_alg_bytes_1 += decode_alg_bytes(ctx, 1, h_dyn[1]); }
    return 0;
}

// ---- a pass of several runs on a stage that is the whole target (--gpus 2: draft + driver on GPU 0, the target alone on GPU 1) --------------------------------------
// What run coalescing (feeder_main / compose_pass) does for the in-process target, for the transport's serve loop: the runs whose headers wait in the stage's ring — members
// [0, n_members), in start order — are carried by ONE forward pass over the stage's weights; every member's rows go to its own result buffers.  Same rules: a member that would
// write or expect older content in a cell another member touches stays out (stage_pass_conflict == pass_conflict on the metadata blocks), a token attends over the n_kv of ITS run,
// every token keeps the bits it gets alone.  First-and-last stages only (no activations cross a boundary).  0 done, 1 cut short (every member cancelled), < 0 error.
static void stage_member(pi_run & r, const void * meta, int n_tokens) {
    const int32_t * m = (const int32_t *) meta;
    r.meta_host = (uint8_t *) const_cast<void *>(meta); r.n_tokens = n_tokens; r.kv_head = m[2 * n_tokens]; r.n_kv = m[2 * n_tokens + 1];
}
bool stage_pass_conflict(int n_ctx, int n_members, const void * const * metas, const int32_t * n_tokens, const void * q_meta, int q_tokens) {
    std::vector<pi_run> tmp((size_t) n_members + 1);                           // (pass_conflict reads n_tokens, kv_head, n_kv and the block itself)
    std::vector<pi_run *> members;
    for (int i = 0; i < n_members; i++) { stage_member(tmp[(size_t) i], metas[i], n_tokens[i]); members.push_back(&tmp[(size_t) i]); }
    stage_member(tmp[(size_t) n_members], q_meta, q_tokens);
    const bool c = pass_conflict_n(n_ctx, members, &tmp[(size_t) n_members]);
    for (auto & r : tmp) r.meta_host = nullptr;                                // (borrowed blocks: nothing of these stand-ins is freed)
    return c;
}
int stage_run_pass(llama_context * ctx, int n_members, const void * const * metas, const int32_t * n_tokens, float * const * logits_host, int32_t * const * topk_ids,
                   float * const * topk_vals, float * const * topk_sum, int (*should_abort)(void *), void * abort_user) {
    std::string err;
    if (!context_prepare(ctx, err)) { ctx->last_error = err; return -1; }
    if (!ctx->is_first || !ctx->is_last || n_members < 1) { ctx->last_error = "stage_run_pass: a pass of several runs needs a stage that is first and last"; return -1; }
    if (n_members == 1) return stage_run(ctx, n_tokens[0], metas[0], (size_t) n_tokens[0] * 8 + 16 + (size_t) n_tokens[0] * ((ctx->n_ctx + 31) / 32) * 4, nullptr, nullptr,
                                         logits_host[0], topk_ids[0], topk_vals[0], topk_sum[0], should_abort, abort_user, nullptr);
    std::vector<pi_run> tmp((size_t) n_members);
    std::vector<pi_run *> members;
    int n = 0, n_kv = 0;
    for (int i = 0; i < n_members; i++) { stage_member(tmp[(size_t) i], metas[i], n_tokens[i]); members.push_back(&tmp[(size_t) i]); n += n_tokens[i]; n_kv = std::max(n_kv, tmp[(size_t) i].n_kv); }
    struct unborrow { std::vector<pi_run> & v; ~unborrow() { for (auto & r : v) r.meta_host = nullptr; } } ub{ tmp };
    if (n > ctx->n_batch) { ctx->last_error = "stage_run_pass: the pass is wider than n_batch"; return -1; }
    const void * meta = compose_pass(ctx, members, n);
    if (!meta) return -1;
    hipStream_t s = ctx->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    const int32_t rid = (++ctx->run_counter) & 0x3fffffff;
    if (ctx->abort_host) ctx->abort_host[rid & 255] = 0;
    ctx->stage_run_id.store(rid);
    bool ok = true, cut = false;
    (void) hipEventRecord(e0, s);
    const int nseg = n_segments(ctx), kv_head = ((const int32_t *) meta)[2 * n];
    if (nseg > 1 && !ctx->seg_events[0]) ok = hipEventCreateWithFlags(&ctx->seg_events[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ctx->seg_events[1], hipEventDisableTiming) == hipSuccess;
    if (ok) ok = launch_inputs(ctx, n, meta, nullptr);
    int seg = 0;
    for (; ok && seg < nseg; seg++) {
        if (seg >= 2) (void) hipEventSynchronize(ctx->seg_events[seg & 1]);
        if (seg > 0 && should_abort && should_abort(abort_user)) { cut = true; break; }
        ok = launch_abort_begin(ctx, rid);
        if (ok) ok = launch_segment(ctx, n, kv_head, seg, nullptr, true, false, n_kv);
        if (ok && nseg > 1) ok = hipEventRecord(ctx->seg_events[seg & 1], s) == hipSuccess;
    }
    ctx->stat_segments += (uint64_t) seg; ctx->stat_segments_skipped += (uint64_t) (nseg - seg);
    const size_t nv = ctx->model->hparams.n_vocab;
    if (ok && !cut) {
        size_t r0 = 0;
        for (int i = 0; ok && i < n_members; i++) {                            // every member's rows into its own result slot
            const size_t m = (size_t) n_tokens[i];
            ok = hipMemcpyAsync(logits_host[i], ctx->logits_dev + r0 * nv, 4 * m * nv, hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipMemcpyAsync(topk_ids[i], ctx->topk_ids_dev + r0 * 8, 4 * m * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipMemcpyAsync(topk_vals[i], ctx->topk_vals_dev + r0 * 8, 4 * m * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipMemcpyAsync(topk_sum[i], ctx->topk_sum_dev + r0, 4 * m, hipMemcpyDeviceToHost, s) == hipSuccess;
            r0 += m;
        }
    }
    (void) hipEventRecord(e1, s);
    if (!ok && ctx->abort_host) ctx->abort_host[rid & 255] = 1;
    if (hipStreamSynchronize(s) != hipSuccess) ok = false;
    ctx->stage_run_id.store(0);
    if (ok && !cut && ((ctx->abort_host && ctx->abort_host[rid & 255]) || (should_abort && should_abort(abort_user)))) cut = true;
    float ms = 0.0f;
    if (ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ctx->stat_gpu_ms += ms;
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    if (!ok) { if (ctx->last_error.empty()) ctx->last_error = "stage_run_pass failed"; return -1; }
    if (cut) return 1;
    ctx->stat_calls++; ctx->stat_alg_bytes += decode_alg_bytes(ctx, n, n_kv);
    ctx->stat_passes++; ctx->stat_pass_runs += (uint64_t) n_members; ctx->stat_pass_tokens += (uint64_t) n; ctx->stat_coalesced_passes++;
    return 0;
}

int context_finish(llama_context * ctx, pi_run * run) {
    if (!run) return -1;
    if (ctx->remote) {
        if (run->launched) {
            const int rc = ctx->remote_ops.wait(ctx->remote_ops.user, run->run_seq, run->logits_host, run->topk_ids, run->topk_vals, run->topk_sum);
            if (rc < 0) { ctx->last_error = "remote wait failed"; return -1; }
            run->launched = rc == 1;                                          // 0: a stage skipped the run after a cancel
        }
    } else {
        {
            struct waiting { llama_context * c; waiting(llama_context * c_) : c(c_) { c->waiters.fetch_add(1); } ~waiting() { c->waiters.fetch_sub(1); } } w(ctx);
            tl("finish_wait", ctx->stream_priority, run->run_id, run->state.load());
            while (run->state.load() != 2) std::this_thread::sleep_for(std::chrono::microseconds(20));
            if (hipEventSynchronize(run->done) != hipSuccess) { ctx->last_error = "hipEventSynchronize failed"; urgent_end(ctx); return -1; }
        }
        tl("finish", ctx->stream_priority, run->run_id, (run->canceled || run->cancel_req.load()) ? 1 : 0);
        if (run->run_id > ctx->last_finished_id.load()) ctx->last_finished_id.store(run->run_id);
        ctx->epoch_after_finish.store(ctx->caller_epoch.fetch_add(1) + 1);
        // experiment knob (tools_dev/draft_sensitivity.sh): what does the schedule lose per microsecond of draft latency?  > 0: the priority context's run holds the
        // device that much longer; < 0: only its caller sees the result later
        static const int prio_delay_us = getenv("PIPEINFER_PRIORITY_DELAY_US") ? atoi(getenv("PIPEINFER_PRIORITY_DELAY_US")) : 0;
        if (prio_delay_us > 0 && ctx->stream_priority > 0) std::this_thread::sleep_for(std::chrono::microseconds(prio_delay_us));
        urgent_end(ctx);
        if (prio_delay_us < 0 && ctx->stream_priority > 0) std::this_thread::sleep_for(std::chrono::microseconds(-prio_delay_us));
        if (run->failed) {
            // a launch of this run failed (graph capture, a kernel launch, a copy): that is an error, not a cancelled run — the caller gets -1, the cells
            // find_slot allotted go back (nothing valid was written to them) and the run leaves the queue
            if (ctx->last_error.empty()) ctx->last_error = "a launch of the run failed";
            ctx->cells->release_cells(run->kv_head, run->n_tokens);
            ctx->last_n_tokens = 0;
            if (run->pass) { std::lock_guard<std::mutex> lk(ctx->fmu); auto & v = run->pass->runs; v.erase(std::remove(v.begin(), v.end(), run), v.end()); run->pass.reset(); }
            for (auto it = ctx->inflight.begin(); it != ctx->inflight.end(); ++it) if (*it == run) { ctx->inflight.erase(it); break; }
            run->finished = true;
            ctx->pool.push_back(run);
            return -1;
        }
        const bool cut = run->cancel_req.load();                               // cancelled after the start: its kernels returned on entry from some point on — nothing it wrote is a result
        if (run->canceled || cut) {
            float ms = 0.0f;
            ctx->stat_cancel_runs++;
            if (!run->canceled) {
                ctx->stat_cancel_runs_started++;                              // (a run that shared its pass with live runs cost nothing of its own: the pass was computed for them)
                if (run->pass_tokens == run->n_tokens && hipEventElapsedTime(&ms, run->begin, run->done) == hipSuccess) ctx->stat_cancel_gpu_ms += ms;
            }
        }
        if (cut) run->launched = false;
    }
    const size_t nv = ctx->model->hparams.n_vocab;
    const int n = run->n_tokens;
    if (ctx->run_tap) ctx->run_tap(ctx->run_tap_user, run->run_id, run->batch_id, n, run->meta_host, run->meta_bytes, run->launched && ctx->is_last && !run->canceled ? run->logits_host : nullptr);
    if (ctx->is_last) {
        ctx->logits.resize(run->last_only ? nv : (size_t) n * nv);             // llama.cpp:5734-5743
        if (run->launched) {
            if (run->last_only) memcpy(ctx->logits.data(), run->logits_host + (size_t) (n - 1) * nv, nv * 4);
            else for (int i = 0; i < n; i++) if (run->want_logits[i]) memcpy(ctx->logits.data() + (size_t) i * nv, run->logits_host + (size_t) i * nv, nv * 4);
            ctx->last_topk_ids.assign(run->topk_ids, run->topk_ids + (size_t) n * 8);
            ctx->last_topk_vals.assign(run->topk_vals, run->topk_vals + (size_t) n * 8);
            ctx->last_topk_sum.assign(run->topk_sum, run->topk_sum + n);
            ctx->last_n_tokens = n;
        } else {
            ctx->last_n_tokens = 0;
        }
    }
    if (run->launched && !ctx->remote && run->pass_leader) {
        // one pass = one decode call of the roofline accounting, whatever the number of runs it carried; the single-token figures count passes of ONE token only
        const int pn = run->pass_tokens;
        const int n_kv = run->pass ? std::max(run->n_kv, run->pass->n_kv) : run->n_kv;
        ctx->stat_calls++; ctx->stat_alg_bytes += decode_alg_bytes(ctx, pn, n_kv);
        if (pn == 1) { ctx->stat_calls_1++; ctx->stat_alg_bytes_1 += decode_alg_bytes(ctx, 1, n_kv); }
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, run->begin, run->done) == hipSuccess) { ctx->stat_gpu_ms += ms; if (pn == 1) ctx->stat_gpu_ms_1 += ms; tl("gpu_us", ctx->stream_priority, run->run_id, (int) (ms * 1000.0f)); }
    }
    if (run->pass) {                                                          // the object is about to be recycled: it leaves its pass
        std::lock_guard<std::mutex> lk(ctx->fmu);
        auto & v = run->pass->runs;
        v.erase(std::remove(v.begin(), v.end(), run), v.end());
        run->pass.reset();
    }
    const int64_t dt = time_us() - run->t_start_us;
    if (n == 1) { ctx->t_eval_us += dt; ctx->n_eval++; } else { ctx->t_p_eval_us += dt; ctx->n_p_eval += n; }     // llama.cpp:5807-5815
    for (auto it = ctx->inflight.begin(); it != ctx->inflight.end(); ++it) if (*it == run) { ctx->inflight.erase(it); break; }
    run->finished = true;
    ctx->pool.push_back(run);
    tl("finish_out", ctx->stream_priority, run->run_id);
    return 1;
}

}  // namespace pi
