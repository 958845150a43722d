// The rest of the llama.h surface that the reference's common/ sources (common.cpp, sampling.cpp, grammar-parser.cpp) and
// examples/speculative reference, so that they compile and link against this library unchanged:
//   * host-side samplers over llama_token_data_array (llama.cpp:7011-7545) — model independent arithmetic, restated here;
//   * ggml.h odds and ends (timers, CPU feature queries);
//   * entry points that are OUT OF SCOPE for the decode hot path (grammar, LoRA, quantiser, session files, beam search, embeddings):
//     they exist, say loudly that this engine does not provide them, and fail — nothing silently pretends to work.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <ctime>
#include <numeric>
#include <random>
#include <string>
#include <unordered_map>

namespace {
void out_of_scope(const char * what) {
    fprintf(stderr, "libpipeinfer_amd: %s is outside the decode hot path this engine implements (DESIGN.md section 9)\n", what);
}
struct t_sample {                                         // accumulates into ctx->t_sample_us like the reference's samplers
    llama_context * ctx; int64_t t0;
    explicit t_sample(llama_context * c) : ctx(c), t0(pi::time_us()) {}
    ~t_sample() { if (ctx) ctx->t_sample_us += pi::time_us() - t0; }
};
}

// ---- ggml.h -----------------------------------------------------------------------------------------------------------------------
extern "C" void    ggml_time_init(void) {}
extern "C" int64_t ggml_time_ms(void) { return pi::time_us() / 1000; }
extern "C" int64_t ggml_cycles(void) { return pi::time_us(); }                 // the drivers only take differences and divide by cycles_per_ms
extern "C" int64_t ggml_cycles_per_ms(void) { return 1000; }
#define PI_CPU_HAS(name, expr) extern "C" int ggml_cpu_has_##name(void) { return (expr); }
PI_CPU_HAS(avx, __builtin_cpu_supports("avx"))
PI_CPU_HAS(avx2, __builtin_cpu_supports("avx2"))
PI_CPU_HAS(avx512, __builtin_cpu_supports("avx512f"))
PI_CPU_HAS(avx512_vbmi, __builtin_cpu_supports("avx512vbmi"))
PI_CPU_HAS(avx512_vnni, __builtin_cpu_supports("avx512vnni"))
PI_CPU_HAS(fma, __builtin_cpu_supports("fma"))
PI_CPU_HAS(sse3, __builtin_cpu_supports("sse3"))
PI_CPU_HAS(ssse3, __builtin_cpu_supports("ssse3"))
PI_CPU_HAS(f16c, 1)
PI_CPU_HAS(neon, 0)
PI_CPU_HAS(arm_fma, 0)
PI_CPU_HAS(metal, 0)
PI_CPU_HAS(fp16_va, 0)
PI_CPU_HAS(wasm_simd, 0)
PI_CPU_HAS(blas, 0)
PI_CPU_HAS(cublas, 0)
PI_CPU_HAS(clblast, 0)
PI_CPU_HAS(gpublas, 0)
PI_CPU_HAS(vsx, 0)

// ---- small queries (llama.h:315-345) ------------------------------------------------------------------------------------------------
extern "C" int  llama_max_devices(void) { return LLAMA_MAX_DEVICES; }
extern "C" bool llama_mmap_supported(void) { return false; }                   // weights go to HBM; the host file is read, not mapped
extern "C" bool llama_mlock_supported(void) { return false; }
extern "C" float llama_rope_freq_scale_train(const struct llama_model * model) { return model->hparams.rope_freq_scale_train; }
extern "C" llama_token llama_token_prefix(const struct llama_model *) { return 32007; }       // llama.cpp:1372-1375 defaults (CodeLlama infill ids)
extern "C" llama_token llama_token_middle(const struct llama_model *) { return 32009; }
extern "C" llama_token llama_token_suffix(const struct llama_model *) { return 32008; }
extern "C" llama_token llama_token_eot(const struct llama_model *) { return 32010; }
extern "C" void llama_log_set(ggml_log_callback, void *) {}                    // the engine logs to stderr only
extern "C" void llama_dump_timing_info_yaml(FILE * stream, const struct llama_context * ctx) {   // llama.cpp:10208-10232
    const llama_timings t = llama_get_timings(const_cast<llama_context *>(ctx));
    fprintf(stream, "\n# Timings #\n###########\n\n");
    fprintf(stream, "mst_eval: %.2f  # ms / token during generation\n", t.n_eval ? t.t_eval_ms / t.n_eval : 0.0);
    fprintf(stream, "mst_p_eval: %.2f  # ms / token during prompt processing\n", t.n_p_eval ? t.t_p_eval_ms / t.n_p_eval : 0.0);
    fprintf(stream, "mst_sample: %.2f  # ms / token during sampling\n", t.n_sample ? t.t_sample_ms / t.n_sample : 0.0);
    fprintf(stream, "n_eval: %d  # number of tokens generated (excluding the first one)\n", t.n_eval);
    fprintf(stream, "n_p_eval: %d  # number of tokens processed in batches at the beginning\n", t.n_p_eval);
    fprintf(stream, "n_sample: %d  # number of sampled tokens\n", t.n_sample);
    fprintf(stream, "t_eval_us: %ld  # total microseconds spent generating tokens\n", (long) (1e3 * t.t_eval_ms));
    fprintf(stream, "t_load_us: %ld  # total microseconds spent loading the model\n", (long) (1e3 * t.t_load_ms));
    fprintf(stream, "t_p_eval_us: %ld  # total microseconds spent prompt processing\n", (long) (1e3 * t.t_p_eval_ms));
    fprintf(stream, "t_sample_us: %ld  # total microseconds spent sampling\n", (long) (1e3 * t.t_sample_ms));
    fprintf(stream, "ts_eval: %.2f  # tokens / second during generation\n", t.t_eval_ms > 0 ? 1e3 * t.n_eval / t.t_eval_ms : 0.0);
    fprintf(stream, "ts_p_eval: %.2f  # tokens / second during prompt processing\n", t.t_p_eval_ms > 0 ? 1e3 * t.n_p_eval / t.t_p_eval_ms : 0.0);
    fprintf(stream, "ts_sample: %.2f  # tokens / second during sampling\n", t.t_sample_ms > 0 ? 1e3 * t.n_sample / t.t_sample_ms : 0.0);
}

// ---- out of scope: present, loud, failing --------------------------------------------------------------------------------------------
extern "C" struct llama_model_quantize_params llama_model_quantize_default_params(void) {
    struct llama_model_quantize_params p; memset(&p, 0, sizeof p); p.ftype = LLAMA_FTYPE_MOSTLY_Q5_1; p.quantize_output_tensor = true; return p;   // llama.cpp:8614-8625
}
// llama_model_quantize: quantize.cpp
extern "C" int llama_apply_lora_from_file(struct llama_context *, const char *, float, const char *, int) { out_of_scope("llama_apply_lora_from_file (LoRA)"); return 1; }
extern "C" int llama_model_apply_lora_from_file(const struct llama_model *, const char *, float, const char *, int) { out_of_scope("llama_model_apply_lora_from_file (LoRA)"); return 1; }
extern "C" struct ggml_tensor * llama_get_model_tensor(struct llama_model *, const char *) { out_of_scope("llama_get_model_tensor (weights live in HBM in kernel layouts)"); return nullptr; }
extern "C" size_t llama_get_state_size(const struct llama_context *) { out_of_scope("llama_get_state_size (session state)"); return 0; }
extern "C" size_t llama_copy_state_data(struct llama_context *, uint8_t *) { out_of_scope("llama_copy_state_data (session state)"); return 0; }
extern "C" size_t llama_set_state_data(struct llama_context *, uint8_t *) { out_of_scope("llama_set_state_data (session state)"); return 0; }
extern "C" bool llama_load_session_file(struct llama_context *, const char *, llama_token *, size_t, size_t *) { out_of_scope("llama_load_session_file"); return false; }
extern "C" bool llama_save_session_file(struct llama_context *, const char *, const llama_token *, size_t) { out_of_scope("llama_save_session_file"); return false; }
extern "C" float * llama_get_embeddings(struct llama_context *) { out_of_scope("llama_get_embeddings"); return nullptr; }
extern "C" void llama_beam_search(struct llama_context *, llama_beam_search_callback_fn_t, void *, size_t, int, int) { out_of_scope("llama_beam_search"); }
extern "C" void llama_kv_cache_seq_shift(struct llama_context *, llama_seq_id, llama_pos, llama_pos, llama_pos) {
    // the reference's own MPI path cannot run a shifted cache either (GGML_ASSERT(inp0 == gf->nodes[0]), ggml-mpi.c:610)
    out_of_scope("llama_kv_cache_seq_shift (K-shift graphs)"); abort();
}
extern "C" void llama_kv_cache_seq_cp_back(struct llama_context * ctx, llama_seq_id src, llama_seq_id dst, llama_pos p0, llama_pos p1) { llama_kv_cache_seq_cp(ctx, src, dst, p0, p1); }        // llama.cpp:9298-9320: the same cell update, other message direction
extern "C" void llama_kv_cache_seq_cp_sync_bi(struct llama_context * ctx, llama_seq_id src, llama_seq_id dst, llama_pos p0, llama_pos p1) { llama_kv_cache_seq_cp(ctx, src, dst, p0, p1); }
extern "C" struct llama_grammar * llama_grammar_init(const llama_grammar_element **, size_t, size_t) { out_of_scope("llama_grammar_init (GBNF grammars)"); return nullptr; }
extern "C" void llama_grammar_free(struct llama_grammar *) {}
extern "C" struct llama_grammar * llama_grammar_copy(const struct llama_grammar *) { out_of_scope("llama_grammar_copy (GBNF grammars)"); return nullptr; }
extern "C" void llama_sample_grammar(struct llama_context *, llama_token_data_array *, const struct llama_grammar *) { out_of_scope("llama_sample_grammar (GBNF grammars)"); abort(); }
extern "C" void llama_grammar_accept_token(struct llama_context *, struct llama_grammar *, llama_token) { out_of_scope("llama_grammar_accept_token (GBNF grammars)"); abort(); }
// deprecated single-sequence entry points (llama.h:525-545): thin forms of llama_decode
extern "C" int llama_eval(struct llama_context * ctx, llama_token * tokens, int32_t n_tokens, int n_past) {
    llama_kv_cache_seq_rm(ctx, -1, n_past, -1);                                                   // llama.cpp:9804-9813
    return llama_decode(ctx, llama_batch_get_one(tokens, n_tokens, n_past, 0));
}
extern "C" int llama_eval_embd(struct llama_context *, float *, int32_t, int) { out_of_scope("llama_eval_embd (embedding inputs)"); return -1; }

// ---- samplers (llama.cpp:7011-7545) -----------------------------------------------------------------------------------------------------
extern "C" void llama_set_rng_seed(struct llama_context * ctx, uint32_t seed) {
    if (seed == LLAMA_DEFAULT_SEED) seed = (uint32_t) time(nullptr);
    ctx->rng.seed(seed);
}

extern "C" void llama_sample_top_k(struct llama_context * ctx, llama_token_data_array * c, int k, size_t min_keep) {            // :7047-7072
    t_sample ts(ctx);
    k = std::min(std::max(k, (int) min_keep), (int) c->size);
    if (!c->sorted) {
        auto by_logit = [](const llama_token_data & a, const llama_token_data & b) { return a.logit > b.logit; };
        if (k == (int) c->size) std::sort(c->data, c->data + c->size, by_logit);
        else std::partial_sort(c->data, c->data + k, c->data + c->size, by_logit);
        c->sorted = true;
    }
    c->size = (size_t) k;
}

extern "C" void llama_sample_top_p(struct llama_context * ctx, llama_token_data_array * c, float p, size_t min_keep) {          // :7074-7105
    if (p >= 1.0f) return;
    llama_sample_softmax(ctx, c);
    t_sample ts(ctx);
    float cum = 0.0f;
    size_t last = c->size;
    for (size_t i = 0; i < c->size; ++i) {
        cum += c->data[i].p;
        if (cum >= p && i + 1 >= min_keep) { last = i + 1; break; }
    }
    c->size = last;
}

extern "C" void llama_sample_min_p(struct llama_context * ctx, llama_token_data_array * c, float p, size_t min_keep) {          // :7107-7132
    if (p <= 0.0f || !c->size) return;
    llama_sample_softmax(ctx, c);
    t_sample ts(ctx);
    const float scale = c->data[0].p;
    size_t i = 1;
    for (; i < c->size; ++i) if (c->data[i].p < p * scale && i >= min_keep) break;
    c->size = i;
}

extern "C" void llama_sample_tail_free(struct llama_context * ctx, llama_token_data_array * c, float z, size_t min_keep) {      // :7134-7191
    if (z >= 1.0f || c->size <= 2) return;
    llama_sample_softmax(nullptr, c);
    t_sample ts(ctx);
    std::vector<float> d1(c->size - 1), d2(c->size - 2);
    for (size_t i = 0; i < d1.size(); ++i) d1[i] = c->data[i].p - c->data[i + 1].p;
    for (size_t i = 0; i < d2.size(); ++i) d2[i] = std::abs(d1[i] - d1[i + 1]);
    const float sum = std::accumulate(d2.begin(), d2.end(), 0.0f);
    if (sum > 1e-6f) { for (float & v : d2) v /= sum; } else { for (float & v : d2) v = 1.0f / d2.size(); }
    float cum = 0.0f;
    size_t last = c->size;
    for (size_t i = 0; i < d2.size(); ++i) {
        cum += d2[i];
        if (cum > z && i >= min_keep) { last = i; break; }
    }
    c->size = last;
}

extern "C" void llama_sample_typical(struct llama_context * ctx, llama_token_data_array * c, float p, size_t min_keep) {        // :7193-7250
    if (p >= 1.0f) return;
    llama_sample_softmax(nullptr, c);
    t_sample ts(ctx);
    float entropy = 0.0f;
    for (size_t i = 0; i < c->size; ++i) entropy += -c->data[i].p * logf(c->data[i].p);
    std::vector<float> shifted(c->size);
    for (size_t i = 0; i < c->size; ++i) shifted[i] = fabsf(-logf(c->data[i].p) - entropy);
    std::vector<size_t> idx(c->size);
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return shifted[a] < shifted[b]; });
    float cum = 0.0f;
    size_t last = idx.size();
    for (size_t i = 0; i < idx.size(); ++i) {
        cum += c->data[idx[i]].p;
        if (cum > p && i >= min_keep - 1) { last = i + 1; break; }
    }
    std::vector<llama_token_data> kept;
    for (size_t i = 0; i < last; ++i) kept.push_back(c->data[idx[i]]);
    std::copy(kept.begin(), kept.end(), c->data);
    c->size = kept.size();
}

extern "C" void llama_sample_temp(struct llama_context * ctx, llama_token_data_array * c, float temp) {                          // :7252-7262
    t_sample ts(ctx);
    for (size_t i = 0; i < c->size; ++i) c->data[i].logit /= temp;
}
extern "C" void llama_sample_temperature(struct llama_context * ctx, llama_token_data_array * c, float temp) { llama_sample_temp(ctx, c, temp); }

extern "C" void llama_sample_repetition_penalties(struct llama_context * ctx, llama_token_data_array * c, const llama_token * last_tokens, size_t penalty_last_n,
                                                  float penalty_repeat, float penalty_freq, float penalty_present) {            // :7268-7312
    if (penalty_last_n == 0 || (penalty_repeat == 1.0f && penalty_freq == 0.0f && penalty_present == 0.0f)) return;
    t_sample ts(ctx);
    std::unordered_map<llama_token, int> count;
    for (size_t i = 0; i < penalty_last_n; ++i) count[last_tokens[i]]++;
    for (size_t i = 0; i < c->size; ++i) {
        const auto it = count.find(c->data[i].id);
        if (it == count.end()) continue;
        const int n = it->second;
        if (c->data[i].logit <= 0) c->data[i].logit *= penalty_repeat; else c->data[i].logit /= penalty_repeat;
        c->data[i].logit -= float(n) * penalty_freq + float(n > 0) * penalty_present;
    }
    c->sorted = false;
}

static void log_softmax_inplace(float * a, size_t n) {                                                                           // :7354-7366
    const float mx = *std::max_element(a, a + n);
    float sum = 0.f;
    for (size_t i = 0; i < n; ++i) { const float p = expf(a[i] - mx); sum += p; a[i] = p; }
    for (size_t i = 0; i < n; ++i) a[i] = logf(a[i] / sum);
}
extern "C" void llama_sample_classifier_free_guidance(struct llama_context * ctx, llama_token_data_array * c, struct llama_context * guidance_ctx, float scale) {   // :7368-7401
    GGML_ASSERT(ctx);
    t_sample ts(ctx);
    const int n_vocab = llama_n_vocab(llama_get_model(ctx));
    GGML_ASSERT(n_vocab == (int) c->size);
    GGML_ASSERT(!c->sorted);
    std::vector<float> base(c->size);
    for (size_t i = 0; i < c->size; ++i) base[i] = c->data[i].logit;
    log_softmax_inplace(base.data(), c->size);
    float * guide = llama_get_logits(guidance_ctx);
    log_softmax_inplace(guide, (size_t) n_vocab);
    for (int i = 0; i < n_vocab; ++i) c->data[i].logit = scale * (base[i] - guide[i]) + guide[i];
}

extern "C" llama_token llama_sample_token(struct llama_context * ctx, llama_token_data_array * c) {                              // :7511-7533
    GGML_ASSERT(ctx);
    t_sample ts(ctx);
    llama_sample_softmax(nullptr, c);
    std::vector<float> probs(c->size);
    for (size_t i = 0; i < c->size; ++i) probs[i] = c->data[i].p;
    std::discrete_distribution<> dist(probs.begin(), probs.end());
    const int idx = dist(ctx->rng);
    ctx->n_sample++;
    return c->data[idx].id;
}

extern "C" llama_token llama_sample_token_mirostat(struct llama_context * ctx, llama_token_data_array * c, float tau, float eta, int m, float * mu) {   // :7403-7451
    GGML_ASSERT(ctx);
    const float N = float(llama_n_vocab(llama_get_model(ctx)));
    llama_sample_softmax(nullptr, c);
    float sum_ti_bi = 0.0f, sum_ti_sq = 0.0f;
    for (size_t i = 0; i < size_t(m - 1) && i < c->size - 1; ++i) {
        const float t_i = logf(float(i + 2) / float(i + 1)), b_i = logf(c->data[i].p / c->data[i + 1].p);
        sum_ti_bi += t_i * b_i; sum_ti_sq += t_i * t_i;
    }
    const float s_hat = sum_ti_bi / sum_ti_sq, eps_hat = s_hat - 1;
    const float k = powf((eps_hat * powf(2, *mu)) / (1 - powf(N, -eps_hat)), 1 / s_hat);
    llama_sample_top_k(nullptr, c, int(k), 1);
    const llama_token X = llama_sample_token(ctx, c);
    const size_t xi = std::distance(c->data, std::find_if(c->data, c->data + c->size, [&](const llama_token_data & d) { return d.id == X; }));
    *mu = *mu - eta * (-log2f(c->data[xi].p) - tau);
    return X;
}

extern "C" llama_token llama_sample_token_mirostat_v2(struct llama_context * ctx, llama_token_data_array * c, float tau, float eta, float * mu) {     // :7453-7493
    llama_sample_softmax(ctx, c);
    c->size = std::distance(c->data, std::find_if(c->data, c->data + c->size, [&](const llama_token_data & d) { return -log2f(d.p) > *mu; }));
    if (c->size == 0) c->size = 1;
    llama_sample_softmax(ctx, c);
    const llama_token X = llama_sample_token(ctx, c);
    const size_t xi = std::distance(c->data, std::find_if(c->data, c->data + c->size, [&](const llama_token_data & d) { return d.id == X; }));
    *mu = *mu - eta * (-log2f(c->data[xi].p) - tau);
    return X;
}

// llama.h:670-673 (C++): one line per cell row showing which sequences hold it (llama.cpp:9188-9236)
std::string dump_kv_cache_view_seqs(const llama_kv_cache_view & view, int row_size) {
    static const char slot_chars[] = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz+";
    std::string out;
    char buf[256];
    snprintf(buf, sizeof buf, "=== Dumping KV cache. total cells %d, max sequences per cell %d, populated cells %d, total tokens in cache %d, largest empty slot=%d @ %d\n",
             view.n_cells, view.n_max_seq, view.used_cells, view.token_count, view.max_contiguous, view.max_contiguous_idx);
    out += buf;
    std::unordered_map<llama_seq_id, size_t> seqs;
    const llama_seq_id * cs = view.cells_sequences;
    for (int i = 0; i < view.n_cells; i++, cs += view.n_max_seq)
        for (int j = 0; j < view.n_max_seq; j++) {
            if (cs[j] < 0 || seqs.count(cs[j])) continue;
            if (seqs.size() + 1 >= sizeof(slot_chars)) break;
            const size_t id = seqs.size();
            seqs[cs[j]] = id;
        }
    out += "=== Sequence legend: ";
    for (const auto & it : seqs) { snprintf(buf, sizeof buf, "%zu=%d, ", it.second, it.first); out += buf; }
    out += "'+'=other sequence ids";
    cs = view.cells_sequences;
    for (int i = 0; i < view.n_cells; i++, cs += view.n_max_seq) {
        if (i % row_size == 0) { snprintf(buf, sizeof buf, "\n%5d: ", i); out += buf; }
        for (int j = 0; j < view.n_max_seq; j++) {
            if (cs[j] >= 0) { const auto it = seqs.find(cs[j]); out += it != seqs.end() ? slot_chars[it->second] : '+'; }
            else out += '.';
        }
        out += ' ';
    }
    out += "\n=== Done dumping\n";
    return out;
}

// ---- GGUF metadata as strings (llama.cpp:8976-9013; values formatted like gguf_kv_to_str :611-663, arrays are not listed :2213-2221) ----
static std::string meta_value_str(const pi::gguf_value & v) {
    switch (v.type) {
        case 0: case 2: case 4: case 10: return std::to_string(v.u);
        case 1: case 3: case 5: case 11: return std::to_string(v.i);
        case 6: return std::to_string((float) v.f);
        case 12: return std::to_string(v.f);
        case 7: return v.b ? "true" : "false";
        case 8: return v.s;
    }
    return "unknown type " + std::to_string(v.type);
}
static const pi::gguf_value * meta_at(const struct llama_model * model, int i, const std::string ** key) {
    int n = 0;
    for (const auto & kv : model->file.kv) {                  // std::map: keys in order, like the reference's unordered listing by index is not pinned
        if (kv.second.type == 9) continue;
        if (n++ == i) { if (key) *key = &kv.first; return &kv.second; }
    }
    return nullptr;
}
extern "C" int llama_model_meta_count(const struct llama_model * model) {
    int n = 0;
    for (const auto & kv : model->file.kv) if (kv.second.type != 9) n++;
    return n;
}
extern "C" int llama_model_meta_val_str(const struct llama_model * model, const char * key, char * buf, size_t buf_size) {
    const pi::gguf_value * v = model->file.find(key);
    if (!v || v->type == 9) { if (buf_size > 0) buf[0] = '\0'; return -1; }
    return snprintf(buf, buf_size, "%s", meta_value_str(*v).c_str());
}
extern "C" int llama_model_meta_key_by_index(const struct llama_model * model, int i, char * buf, size_t buf_size) {
    const std::string * key = nullptr;
    if (!meta_at(model, i, &key)) { if (buf_size > 0) buf[0] = '\0'; return -1; }
    return snprintf(buf, buf_size, "%s", key->c_str());
}
extern "C" int llama_model_meta_val_str_by_index(const struct llama_model * model, int i, char * buf, size_t buf_size) {
    const pi::gguf_value * v = meta_at(model, i, nullptr);
    if (!v) { if (buf_size > 0) buf[0] = '\0'; return -1; }
    return snprintf(buf, buf_size, "%s", meta_value_str(*v).c_str());
}
