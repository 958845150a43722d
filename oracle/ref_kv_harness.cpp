// TEST INFRASTRUCTURE.  The REAL reference KV-cache bookkeeping (and, at the end, its tokenizers): this translation unit includes the reference's llama.cpp from where it
// lies under $(REF) (never copied) and calls its file-static functions — llama_kv_cache_find_slot (llama.cpp:1593-1644), _cell_max (:1647),
// _clear (:1657), _seq_rm (:1666), _seq_cp (:1699), _seq_keep (:1717) — on a bare `struct llama_kv_cache` (cells only, no tensors).
// llama.cpp is compiled with -DGGML_USE_MPI exactly as PipeInfer builds it: ggml-mpi.h declares the MPI layer without including <mpi.h>, so
// the file compiles with plain g++; the ggml_mpi_* functions stay undefined in the shared object and are never reached from here (the
// KV-cell functions touch no communicator), the library is opened with lazy binding.  Nothing stands in for MPI.
// The decode-level rules around find_slot (head reset :5548-5550, n = min(n_ctx, max(32, cell_max)) :5571, head advance :5708-5713) sit
// inside llama_decode_internal and cannot be called on their own: the test applies them from oracle/kvcache.c on top of these functions.
#include "llama.cpp"

#define REFKV_API extern "C" __attribute__((visibility("default")))

REFKV_API void * refkv_new(int n_ctx) {
    llama_kv_cache * c = new llama_kv_cache();
    c->size = (uint32_t) n_ctx;
    c->cells.clear();
    c->cells.resize((size_t) n_ctx);                                       // llama_kv_cache_init, llama.cpp:1538-1539
    return c;
}
REFKV_API void refkv_free(void * c) { delete (llama_kv_cache *) c; }

REFKV_API int refkv_find_slot(void * vc, int n_tokens, const int32_t * pos, const int32_t * n_seq_id, const int32_t * seq_ids_flat, int stride) {
    llama_kv_cache & c = *(llama_kv_cache *) vc;
    std::vector<llama_seq_id *> rows((size_t) n_tokens);
    for (int i = 0; i < n_tokens; i++) rows[i] = const_cast<llama_seq_id *>(seq_ids_flat) + (size_t) i * stride;
    llama_batch b;
    memset(&b, 0, sizeof b);
    b.n_tokens = n_tokens; b.pos = const_cast<llama_pos *>(pos); b.n_seq_id = const_cast<int32_t *>(n_seq_id); b.seq_id = rows.data();
    return llama_kv_cache_find_slot(c, b) ? (int) c.head : -1;
}
REFKV_API int  refkv_cell_max(void * c) { return llama_kv_cache_cell_max(*(llama_kv_cache *) c); }
REFKV_API void refkv_clear(void * c) { llama_kv_cache_clear(*(llama_kv_cache *) c); }
REFKV_API void refkv_seq_rm(void * c, int seq, int p0, int p1) { llama_kv_cache_seq_rm(*(llama_kv_cache *) c, seq, p0, p1); }
REFKV_API void refkv_seq_cp(void * c, int src, int dst, int p0, int p1) { llama_kv_cache_seq_cp(*(llama_kv_cache *) c, src, dst, p0, p1); }
REFKV_API void refkv_seq_keep(void * c, int seq) { llama_kv_cache_seq_keep(*(llama_kv_cache *) c, seq); }
REFKV_API int  refkv_head(void * c) { return (int) ((llama_kv_cache *) c)->head; }
REFKV_API void refkv_set_head(void * c, int h) { ((llama_kv_cache *) c)->head = (uint32_t) h; }
REFKV_API int  refkv_used(void * c) { return (int) ((llama_kv_cache *) c)->used; }
REFKV_API int  refkv_cell_pos(void * c, int i) { return ((llama_kv_cache *) c)->cells[(size_t) i].pos; }
REFKV_API int  refkv_cell_has(void * c, int i, int seq) { return ((llama_kv_cache *) c)->cells[(size_t) i].has_seq_id(seq) ? 1 : 0; }
REFKV_API int  refkv_cell_nseq(void * c, int i) { return (int) ((llama_kv_cache *) c)->cells[(size_t) i].seq_id.size(); }

// ---- the reference's tokenizers (llm_tokenizer_spm / llm_tokenizer_bpe behind llama_tokenize, llama.cpp:5970-6640) on a vocabulary-only model ----------
// llama_load_model_from_file with vocab_only reads the GGUF header and the vocabulary and reaches no MPI call.
static void reftok_quiet(ggml_log_level, const char *, void *) {}
REFKV_API void * reftok_load(const char * path) {
    llama_log_set(reftok_quiet, NULL);
    llama_model_params p = llama_model_default_params();
    p.vocab_only = true;
    return llama_load_model_from_file(path, p);
}
REFKV_API void reftok_free(void * m) { llama_free_model((llama_model *) m); }
REFKV_API int  reftok_n_vocab(void * m) { return llama_n_vocab((llama_model *) m); }
REFKV_API int  reftok_tokenize(void * m, const char * text, int len, int32_t * out, int cap, int add_bos, int special) {
    try { return llama_tokenize((llama_model *) m, text, len, out, cap, add_bos != 0, special != 0); }
    catch (const std::exception &) { return -1000000; }                   // (the reference lets "byte not found in vocab" escape to its caller)
}
REFKV_API int  reftok_piece(void * m, int32_t token, char * buf, int cap) { return llama_token_to_piece((llama_model *) m, token, buf, cap); }

// ---- whole-model decode through the reference's OWN graph builders (llm_build_context::build_llama / build_falcon, llama.cpp:3846-4180) -------------------
// Real: llama_load_model_from_file (tensors mapped from the GGUF), llama_kv_cache_init, llama_kv_cache_find_slot, llama_build_graph (the graph, the RoPE /
// mask / KQ-scale inputs, the K/V cache writes), ggml-alloc, ggml_graph_compute.  Ours, because llama_new_context_with_model and llama_decode_internal also
// call the MPI layer: the few lines around them, taken in the reference's order — context parameters (llama.cpp:8739-8777), the measure / allocate
// sequence of the compute buffer (:8803-8833), and per call the head rule, find_slot, n, build, alloc, compute, head advance (:5545-5714).
struct refmodel { llama_model * model; llama_context * ctx; int n_threads; };

REFKV_API void * refmodel_load(const char * path, int n_ctx, int n_threads) {
    llama_log_set(reftok_quiet, NULL);
    ggml_time_init();
    { struct ggml_init_params ip = { 0, NULL, false }; ggml_free(ggml_init(ip)); }      // f16 tables (llama_backend_init, llama.cpp:8640-8650)
    llama_model * model = llama_load_model_from_file(path, llama_model_default_params());
    if (!model) return NULL;
    const llama_context_params dp = llama_context_default_params();
    llama_context * ctx = new llama_context(*model);
    const auto & hp = model->hparams;
    auto & cp = ctx->cparams;
    cp.n_batch = 512; cp.n_threads = cp.n_threads_batch = (uint32_t) n_threads;
    cp.yarn_ext_factor = 0.0f; cp.yarn_attn_factor = dp.yarn_attn_factor; cp.yarn_beta_fast = dp.yarn_beta_fast; cp.yarn_beta_slow = dp.yarn_beta_slow;
    cp.mul_mat_q = dp.mul_mat_q;
    cp.n_ctx = (uint32_t) n_ctx; cp.rope_freq_base = hp.rope_freq_base_train; cp.rope_freq_scale = hp.rope_freq_scale_train;
    cp.n_yarn_orig_ctx = hp.n_yarn_orig_ctx != 0 ? hp.n_yarn_orig_ctx : hp.n_ctx_train;
    if (hp.rope_scaling_type_train == LLAMA_ROPE_SCALING_NONE) cp.rope_freq_scale = 1.0f;
    ctx->logits_all = true;
    if (!llama_kv_cache_init(hp, ctx->kv_self, GGML_TYPE_F16, cp.n_ctx, 0)) { delete ctx; llama_free_model(model); return NULL; }
    static const size_t tensor_alignment = 32;
    ctx->buf_compute.resize(ggml_tensor_overhead() * LLAMA_MAX_NODES + ggml_graph_overhead());
    ctx->alloc = ggml_allocr_new_measure(tensor_alignment);
    const int nt = (int) std::min(cp.n_ctx, cp.n_batch), n_past = (int) cp.n_ctx - nt;
    llama_token token = llama_token_bos(model);
    ggml_cgraph * gf = llama_build_graph(*ctx, llama_batch_get_one(&token, nt, n_past, 0));
    const size_t alloc_size = ggml_allocr_alloc_graph(ctx->alloc, gf) + tensor_alignment;
    ggml_allocr_free(ctx->alloc);
    ctx->buf_alloc.resize(alloc_size);
    ctx->alloc = ggml_allocr_new(ctx->buf_alloc.data, ctx->buf_alloc.size, tensor_alignment);
    return new refmodel{ model, ctx, n_threads };
}
REFKV_API void refmodel_free(void * h) { refmodel * m = (refmodel *) h; delete m->ctx; llama_free_model(m->model); delete m; }
REFKV_API void * refmodel_kv(void * h) { return &((refmodel *) h)->ctx->kv_self; }       // for the refkv_* functions above
REFKV_API int refmodel_n_vocab(void * h) { return llama_n_vocab(((refmodel *) h)->model); }

// logits_out: [n_tokens][n_vocab].  Returns the slot head, -1 when no slot was found.
REFKV_API int refmodel_decode(void * h, int n_tokens, const int32_t * tokens, const int32_t * pos, const int32_t * n_seq_id, const int32_t * seq_ids_flat, int stride, float * logits_out) {
    refmodel * m = (refmodel *) h;
    llama_context & lctx = *m->ctx;
    auto & kv = lctx.kv_self;
    std::vector<llama_seq_id *> rows((size_t) n_tokens);
    for (int i = 0; i < n_tokens; i++) rows[i] = const_cast<llama_seq_id *>(seq_ids_flat) + (size_t) i * stride;
    llama_batch b;
    memset(&b, 0, sizeof b);
    b.n_tokens = n_tokens; b.token = const_cast<llama_token *>(tokens); b.pos = const_cast<llama_pos *>(pos); b.n_seq_id = const_cast<int32_t *>(n_seq_id); b.seq_id = rows.data();
    if (kv.head > kv.used + 2 * (uint32_t) n_tokens) kv.head = 0;
    if (!llama_kv_cache_find_slot(kv, b)) return -1;
    const int head = (int) kv.head;
    kv.n = std::min((int32_t) lctx.cparams.n_ctx, std::max(32, llama_kv_cache_cell_max(kv)));
    ggml_allocr_reset(lctx.alloc);
    ggml_cgraph * gf = llama_build_graph(lctx, b);
    ggml_allocr_alloc_graph(lctx.alloc, gf);
    struct ggml_tensor * res = gf->nodes[gf->n_nodes - 1];
    if (strcmp(res->name, "result_output") != 0) return -2;
    ggml_graph_compute_helper(lctx.work_buffer, gf, m->n_threads);
    kv.head += (uint32_t) n_tokens;
    if (kv.head >= kv.size) kv.head = 0;
    memcpy(logits_out, ggml_get_data(res), sizeof(float) * (size_t) llama_n_vocab(m->model) * (size_t) n_tokens);
    return head;
}
