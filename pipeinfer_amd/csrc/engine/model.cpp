// GGUF -> llama_model: hyper-parameters (llama.cpp:2203-2381), vocabulary, and per-stage upload of the weights into HBM
// (tensor names/shapes: llama.cpp:2756-2824 Llama, :2891-2968 Falcon).
#include "engine.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pi {
int64_t time_us() {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
bool g_timeline_on = getenv("PI_TIMELINE") != nullptr;
namespace { struct tl_event { int64_t t; const char * tag; int a, b, c; }; std::mutex tl_mu; std::vector<tl_event> tl_events; }
void timeline_note(const char * tag, int a, int b, int c) {
    const int64_t t = time_us();
    std::lock_guard<std::mutex> lk(tl_mu);
    if (tl_events.size() >= (size_t) 4 << 20) return;                      // (a caller that polls in a tight loop must not grow this without bound: the first 4 M events are what is kept)
    tl_events.push_back(tl_event{ t, tag, a, b, c });
}
}

// writes "t_us,tag,a,b,c" lines (events since the last dump) and forgets them; returns the number of events, -1 when the file cannot be written
extern "C" LLAMA_API int pipeinfer_timeline_dump(const char * path) {
    std::vector<pi::tl_event> ev;
    { std::lock_guard<std::mutex> lk(pi::tl_mu); ev.swap(pi::tl_events); }
    if (!path) return (int) ev.size();
    FILE * f = fopen(path, "w");
    if (!f) return -1;
    for (const auto & e : ev) fprintf(f, "%lld,%s,%d,%d,%d\n", (long long) e.t, e.tag, e.a, e.b, e.c);
    fclose(f);
    return (int) ev.size();
}

llama_model::~llama_model() {
    for (auto & l : layers) {
        pi_weight_free(l.wq); pi_weight_free(l.wk); pi_weight_free(l.wv); pi_weight_free(l.wo); pi_weight_free(l.wqkv);
        pi_weight_free(l.ffn_gate); pi_weight_free(l.ffn_down); pi_weight_free(l.ffn_up);
    }
    pi_weight_free(tok_embd); pi_weight_free(output); pi_weight_free(pos_embd);
    for (void * p : f32_allocs) pi_free(p);
}

static pi_weight * upload_matrix(llama_model & m, const std::string & name, int64_t ne0, int64_t ne1, std::string & err, int layer = -1) {
    const pi::gguf_tensor * t = m.file.tensor(name);
    if (!t) { err = "missing tensor " + name; return nullptr; }
    if (t->ne[0] != ne0 || t->ne[1] != ne1) {
        err = "tensor " + name + " has shape [" + std::to_string(t->ne[0]) + "," + std::to_string(t->ne[1]) + "], expected [" + std::to_string(ne0) + "," + std::to_string(ne1) + "]";
        return nullptr;
    }
    pi_weight * w = pi_weight_upload(t->type, ne0, ne1, t->data, nullptr);
    if (!w) { err = std::string("upload of ") + name + " failed: " + pi_last_error(); return nullptr; }
    m.hbm_bytes += pi_weight_nbytes(w);
    m.hbm_copy_bytes += pi_weight_copy_nbytes(w);
    m.matrices.push_back({ w, t->data, layer });
    return w;
}

static float * upload_f32(llama_model & m, const std::string & name, int64_t n, bool required, std::string & err) {
    const pi::gguf_tensor * t = m.file.tensor(name);
    if (!t) { if (required) err = "missing tensor " + name; return nullptr; }
    if (t->type != PI_F32 || t->ne[0] != n) { err = "tensor " + name + " must be f32[" + std::to_string(n) + "]"; return nullptr; }
    float * d = (float *) pi_malloc((size_t) n * 4);
    if (!d || pi_memcpy_h2d(d, t->data, (size_t) n * 4, nullptr) || pi_stream_sync(nullptr)) { err = std::string("upload of ") + name + " failed: " + pi_last_error(); return nullptr; }
    m.f32_allocs.push_back(d);
    m.hbm_bytes += (size_t) n * 4;
    return d;
}

bool llama_model::ensure_loaded(int l0, int l1, bool first, bool last, bool want_mq, std::string & err) {
    const pi_hparams & hp = hparams;
    const int n_embd = hp.n_embd, n_gqa = hp.n_embd_gqa(), n_ff = hp.n_ff;
    struct mq_scope { ~mq_scope() { pi_weight_set_mq_copies(1); } } restore;      // (the setting is the calling thread's)
    pi_weight_set_mq_copies(0);                                                  // the embedding table is only ever gathered from (get_rows): no matrix-core copy
    if (first && !tok_embd) {
        if (!(tok_embd = upload_matrix(*this, "token_embd.weight", n_embd, hp.n_vocab, err))) return false;
        if (hp.arch == PI_ARCH_STARCODER && !(pos_embd = upload_matrix(*this, "position_embd.weight", n_embd, hp.n_ctx_train, err))) return false;
        if (hp.arch == PI_ARCH_BLOOM) {
            if (!(tok_norm = upload_f32(*this, "token_embd_norm.weight", n_embd, true, err))) return false;
            if (!(tok_norm_b = upload_f32(*this, "token_embd_norm.bias", n_embd, true, err))) return false;
        }
    }
    if (hp.block.alibi > 0.0f && !alibi_slopes) {
        std::vector<float> m((size_t) hp.n_head);
        pi_alibi_slopes(hp.n_head, hp.block.alibi, m.data());
        alibi_slopes = (float *) pi_malloc(m.size() * 4);
        if (!alibi_slopes || pi_memcpy_h2d(alibi_slopes, m.data(), m.size() * 4, nullptr) || pi_stream_sync(nullptr)) { err = std::string("upload of the ALiBi slopes failed: ") + pi_last_error(); return false; }
        f32_allocs.push_back(alibi_slopes);
    }
    // tile-major copies (+78 % HBM on a Q4_K_M model) only where a pass can be wide enough to use them
    pi_weight_set_mq_copies(want_mq ? 1 : 0);
    for (int il = l0; il < l1; il++) {
        pi_layer & L = layers[il];
        if (L.loaded) {
            // loaded by a context whose passes never reach the matrix-core kernel, and now a context with wide passes (prompts, verify batches, coalesced runs) prepares:
            // make the copies now — without them every wide pass of this context would silently stay on the VALU kernel (same bits, 2-4x the time)
            if (want_mq && !L.has_mq) {
                for (const matrix_ref & r : matrices)
                    if (r.layer == il) {
                        const size_t before = pi_weight_copy_nbytes(r.w);
                        if (pi_weight_ensure_mq_copy(r.w, r.host, nullptr) != 0) { err = std::string("tile-major copy of layer ") + std::to_string(il) + " failed: " + pi_last_error(); return false; }
                        hbm_copy_bytes += pi_weight_copy_nbytes(r.w) - before;
                    }
                L.has_mq = true;
            }
            continue;
        }
        const std::string p = "blk." + std::to_string(il) + ".";
        if (hp.arch == PI_ARCH_LLAMA) {
            if (!(L.attn_norm = upload_f32(*this, p + "attn_norm.weight", n_embd, true, err))) return false;
            if (!(L.wq = upload_matrix(*this, p + "attn_q.weight", n_embd, n_embd, err, il))) return false;
            if (!(L.wk = upload_matrix(*this, p + "attn_k.weight", n_embd, n_gqa, err, il))) return false;
            if (!(L.wv = upload_matrix(*this, p + "attn_v.weight", n_embd, n_gqa, err, il))) return false;
            if (!(L.wo = upload_matrix(*this, p + "attn_output.weight", n_embd, n_embd, err, il))) return false;
            if (!(L.ffn_norm = upload_f32(*this, p + "ffn_norm.weight", n_embd, true, err))) return false;
            if (!(L.ffn_gate = upload_matrix(*this, p + "ffn_gate.weight", n_embd, n_ff, err, il))) return false;
            if (!(L.ffn_down = upload_matrix(*this, p + "ffn_down.weight", n_ff, n_embd, err, il))) return false;
            if (!(L.ffn_up = upload_matrix(*this, p + "ffn_up.weight", n_embd, n_ff, err, il))) return false;
        } else if (hp.generic()) {
            // tensor sets of llama.cpp:2825-2890 (baichuan; refact shares llama's, :2756-2824), :2969-3047 (starcoder), :3116-3197 (bloom), :3198-3264 (mpt), :3265-3345 (stablelm)
            const bool norm_b = hp.arch == PI_ARCH_STARCODER || hp.arch == PI_ARCH_BLOOM || hp.arch == PI_ARCH_STABLELM || hp.arch == PI_ARCH_PERSIMMON;
            const bool proj_b = hp.arch == PI_ARCH_STARCODER || hp.arch == PI_ARCH_BLOOM || hp.arch == PI_ARCH_PERSIMMON;
            if (!(L.attn_norm = upload_f32(*this, p + "attn_norm.weight", n_embd, true, err))) return false;
            if (norm_b && !(L.attn_norm_b = upload_f32(*this, p + "attn_norm.bias", n_embd, true, err))) return false;
            if (hp.block.fused_qkv) {
                if (!(L.wqkv = upload_matrix(*this, p + "attn_qkv.weight", n_embd, n_embd + 2 * n_gqa, err, il))) return false;
                if (proj_b && !(L.bqkv = upload_f32(*this, p + "attn_qkv.bias", n_embd + 2 * n_gqa, true, err))) return false;
            } else {
                if (!(L.wq = upload_matrix(*this, p + "attn_q.weight", n_embd, n_embd, err, il))) return false;
                if (!(L.wk = upload_matrix(*this, p + "attn_k.weight", n_embd, n_gqa, err, il))) return false;
                if (!(L.wv = upload_matrix(*this, p + "attn_v.weight", n_embd, n_gqa, err, il))) return false;
            }
            if (hp.block.qk_norm) {                                                       // llama.cpp:3105-3108: four vectors of 64
                if (!(L.attn_q_norm = upload_f32(*this, p + "attn_q_norm.weight", 64, true, err)) || !(L.attn_q_norm_b = upload_f32(*this, p + "attn_q_norm.bias", 64, true, err))) return false;
                if (!(L.attn_k_norm = upload_f32(*this, p + "attn_k_norm.weight", 64, true, err)) || !(L.attn_k_norm_b = upload_f32(*this, p + "attn_k_norm.bias", 64, true, err))) return false;
            }
            if (!(L.wo = upload_matrix(*this, p + "attn_output.weight", n_embd, n_embd, err, il))) return false;
            if (proj_b && !(L.bo = upload_f32(*this, p + "attn_output.bias", n_embd, true, err))) return false;
            if (!(L.ffn_norm = upload_f32(*this, p + "ffn_norm.weight", n_embd, true, err))) return false;
            if (norm_b && !(L.ffn_norm_b = upload_f32(*this, p + "ffn_norm.bias", n_embd, true, err))) return false;
            if (hp.block.gated && !(L.ffn_gate = upload_matrix(*this, p + "ffn_gate.weight", n_embd, n_ff, err, il))) return false;
            if (!(L.ffn_down = upload_matrix(*this, p + "ffn_down.weight", n_ff, n_embd, err, il))) return false;
            if (proj_b && !(L.ffn_down_b = upload_f32(*this, p + "ffn_down.bias", n_embd, true, err))) return false;
            if (!(L.ffn_up = upload_matrix(*this, p + "ffn_up.weight", n_embd, n_ff, err, il))) return false;
            if (proj_b && !(L.ffn_up_b = upload_f32(*this, p + "ffn_up.bias", n_ff, true, err))) return false;
        } else {
            if (!(L.attn_norm = upload_f32(*this, p + "attn_norm.weight", n_embd, true, err))) return false;
            if (!(L.attn_norm_b = upload_f32(*this, p + "attn_norm.bias", n_embd, true, err))) return false;
            if (file.tensor(p + "attn_norm_2.weight")) {                       // Falcon-40B style second norm (llama.cpp:2921-2926)
                if (!(L.attn_norm_2 = upload_f32(*this, p + "attn_norm_2.weight", n_embd, true, err))) return false;
                if (!(L.attn_norm_2_b = upload_f32(*this, p + "attn_norm_2.bias", n_embd, true, err))) return false;
            }
            if (!(L.wqkv = upload_matrix(*this, p + "attn_qkv.weight", n_embd, n_embd + 2 * n_gqa, err, il))) return false;
            if (!(L.wo = upload_matrix(*this, p + "attn_output.weight", n_embd, n_embd, err, il))) return false;
            if (!(L.ffn_down = upload_matrix(*this, p + "ffn_down.weight", n_ff, n_embd, err, il))) return false;
            if (!(L.ffn_up = upload_matrix(*this, p + "ffn_up.weight", n_embd, n_ff, err, il))) return false;
        }
        L.loaded = true;
        L.has_mq = want_mq;
    }
    if (last && !output) {
        if (!(output_norm = upload_f32(*this, "output_norm.weight", n_embd, true, err))) return false;
        const bool out_b = hp.arch == PI_ARCH_FALCON || hp.arch == PI_ARCH_STARCODER || hp.arch == PI_ARCH_BLOOM || hp.arch == PI_ARCH_STABLELM || hp.arch == PI_ARCH_PERSIMMON;
        if (out_b && !(output_norm_b = upload_f32(*this, "output_norm.bias", n_embd, true, err))) return false;
        if (!(output = upload_matrix(*this, "output.weight", n_embd, hp.n_vocab, err))) return false;
        output_has_mq = want_mq;
    } else if (last && want_mq && !output_has_mq) {
        for (const matrix_ref & r : matrices)
            if (r.w == output) {
                const size_t before = pi_weight_copy_nbytes(r.w);
                if (pi_weight_ensure_mq_copy(r.w, r.host, nullptr) != 0) { err = std::string("tile-major copy of the output head failed: ") + pi_last_error(); return false; }
                hbm_copy_bytes += pi_weight_copy_nbytes(r.w) - before;
            }
        output_has_mq = true;
    }
    return true;
}

// sum of ggml_nbytes of the tensors one decode call on this stage streams: the embedding contributes n_tok rows
size_t llama_model::stage_weight_bytes(int l0, int l1, bool first, bool last, int n_tok) const {
    size_t b = 0;
    auto add = [&](const std::string & name, int64_t rows_used) {
        const pi::gguf_tensor * t = file.tensor(name);
        if (t) b += rows_used < 0 ? t->nbytes : pi::type_row_size(t->type, t->ne[0]) * (size_t) rows_used;
    };
    if (first) add("token_embd.weight", n_tok);
    if (first) { add("position_embd.weight", n_tok); add("token_embd_norm.weight", -1); add("token_embd_norm.bias", -1); }
    static const char * names[] = { "attn_norm.weight", "attn_norm.bias", "attn_norm_2.weight", "attn_norm_2.bias", "attn_q.weight", "attn_k.weight", "attn_v.weight",
                                    "attn_qkv.weight", "attn_qkv.bias", "attn_output.weight", "attn_output.bias", "ffn_norm.weight", "ffn_norm.bias", "ffn_gate.weight",
                                    "ffn_down.weight", "ffn_down.bias", "ffn_up.weight", "ffn_up.bias", "attn_q_norm.weight", "attn_q_norm.bias", "attn_k_norm.weight", "attn_k_norm.bias" };
    for (int il = l0; il < l1; il++) for (const char * n : names) add("blk." + std::to_string(il) + "." + n, -1);
    if (last) { add("output_norm.weight", -1); add("output_norm.bias", -1); add("output.weight", -1); }
    return b;
}

extern "C" struct llama_model * llama_load_model_from_file(const char * path, struct llama_model_params params) {
    std::unique_ptr<llama_model> m(new llama_model());
    std::string err;
    m->path = path;
    m->device = params.main_gpu;
    if (!m->file.open(path, err)) { fprintf(stderr, "llama_load_model_from_file: %s\n", err.c_str()); return nullptr; }
    const pi::gguf_file & f = m->file;
    const std::string arch = f.get_str("general.architecture", "");
    pi_hparams & hp = m->hparams;
    if (arch == "llama") hp.arch = PI_ARCH_LLAMA;
    else if (arch == "falcon") hp.arch = PI_ARCH_FALCON;
    else if (arch == "baichuan") hp.arch = PI_ARCH_BAICHUAN;
    else if (arch == "starcoder") hp.arch = PI_ARCH_STARCODER;
    else if (arch == "refact") hp.arch = PI_ARCH_REFACT;
    else if (arch == "bloom") hp.arch = PI_ARCH_BLOOM;
    else if (arch == "mpt") hp.arch = PI_ARCH_MPT;
    else if (arch == "stablelm") hp.arch = PI_ARCH_STABLELM;
    else if (arch == "persimmon") hp.arch = PI_ARCH_PERSIMMON;
    else { fprintf(stderr, "llama_load_model_from_file: architecture '%s' is outside this engine's scope (llama, falcon, baichuan, starcoder, persimmon, refact, bloom, mpt, stablelm)\n", arch.c_str()); return nullptr; }
    m->name = f.get_str("general.name", "");
    hp.n_ctx_train = (int) f.get_u32(arch + ".context_length", 0);
    hp.n_embd      = (int) f.get_u32(arch + ".embedding_length", 0);
    hp.n_layer     = (int) f.get_u32(arch + ".block_count", 0);
    hp.n_ff        = (int) f.get_u32(arch + ".feed_forward_length", 0);
    hp.n_head      = (int) f.get_u32(arch + ".attention.head_count", 0);
    hp.n_head_kv   = (int) f.get_u32(arch + ".attention.head_count_kv", (uint32_t) hp.n_head);
    if (!hp.n_embd || !hp.n_head || !hp.n_layer) { fprintf(stderr, "llama_load_model_from_file: missing hyper-parameters in %s\n", path); return nullptr; }
    hp.n_rot       = (int) f.get_u32(arch + ".rope.dimension_count", (uint32_t) (hp.n_embd / hp.n_head));
    hp.f_norm_rms_eps = f.get_f32(arch + ".attention.layer_norm_rms_epsilon", 0.0f);
    hp.f_norm_eps     = f.get_f32(arch + ".attention.layer_norm_epsilon", 0.0f);
    hp.rope_freq_base_train = f.get_f32(arch + ".rope.freq_base", 10000.0f);
    float ropescale = f.get_f32(arch + ".rope.scaling.factor", 0.0f);
    if (ropescale == 0.0f) ropescale = f.get_f32(arch + ".rope.scale_linear", 0.0f);
    hp.rope_freq_scale_train = ropescale == 0.0f ? 1.0f : 1.0f / ropescale;
    if ((hp.arch == PI_ARCH_LLAMA || hp.arch == PI_ARCH_FALCON) && hp.n_rot != hp.n_embd / hp.n_head) { fprintf(stderr, "llama_load_model_from_file: invalid n_rot %d (expected %d)\n", hp.n_rot, hp.n_embd / hp.n_head); return nullptr; }   // llama.cpp:2269-2273
    hp.f_max_alibi_bias = f.get_f32(arch + ".attention.max_alibi_bias", 0.0f);
    hp.f_clamp_kqv      = f.get_f32(arch + ".attention.clamp_kqv", 0.0f);
    {   // the block of each generic architecture, as its build_* function strings it together
        pi_block & b = hp.block;
        switch (hp.arch) {
            case PI_ARCH_BAICHUAN:                                                       // llama.cpp:4030-4063: the 7B rotates, the 13B carries ALiBi instead; other depths assert
                b.rms = true; b.gated = true;
                if (hp.n_layer == 32) b.rope = 0;
                else if (hp.n_layer == 40) b.alibi = 8.0f;
                else { fprintf(stderr, "llama_load_model_from_file: baichuan with %d layers is neither the 7B (32) nor the 13B (40) the reference builds a graph for (llama.cpp:4030-4050)\n", hp.n_layer); return nullptr; }
                break;
            case PI_ARCH_REFACT:    b.rms = true; b.gated = true; b.alibi = 8.0f; break;      // llama.cpp:4591
            case PI_ARCH_STARCODER: b.fused_qkv = true; break;
            case PI_ARCH_BLOOM:     b.fused_qkv = true; b.alibi = 8.0f; break;               // llama.cpp:4683
            case PI_ARCH_MPT:
                b.fused_qkv = true; b.alibi = hp.f_max_alibi_bias; b.clamp_kqv = hp.f_clamp_kqv;
                if (!f.find(arch + ".attention.max_alibi_bias")) { fprintf(stderr, "llama_load_model_from_file: mpt needs %s.attention.max_alibi_bias (llama.cpp:2359)\n", arch.c_str()); return nullptr; }
                break;
            case PI_ARCH_STABLELM:  b.rope = 2; b.gated = true; break;
            case PI_ARCH_PERSIMMON:                                                      // llama.cpp:4329-4538: MHA asserted (:4368), vectors of 64 (:3105-3108)
                b.fused_qkv = true; b.rope = 2; b.relu_sqr = true; b.qk_norm = true;
                if (hp.n_head_kv != hp.n_head || hp.head_dim() != 64) { fprintf(stderr, "llama_load_model_from_file: persimmon needs n_head_kv == n_head and a head width of 64 (llama.cpp:4368, 3105)\n"); return nullptr; }
                break;
            default: break;
        }
        if (b.rope == 2 && !b.qk_norm && (hp.n_rot <= 0 || hp.n_rot % 2 || hp.head_dim() % hp.n_rot)) { fprintf(stderr, "llama_load_model_from_file: n_rot %d does not divide head_dim %d\n", hp.n_rot, hp.head_dim()); return nullptr; }
    }
    if (hp.head_dim() <= 0 || hp.head_dim() > 256 || hp.head_dim() % 2 || hp.n_embd != hp.head_dim() * hp.n_head) { fprintf(stderr, "llama_load_model_from_file: head width %d (n_embd %d / n_head %d) is not supported: the attention kernels take even head widths up to 256 (64 and 128 tuned, the others through the plain kernel; the reference takes any width, llama.cpp:3698-3778)\n", hp.head_dim(), hp.n_embd, hp.n_head); return nullptr; }
    if (!m->vocab.load(f, err)) { fprintf(stderr, "llama_load_model_from_file: %s\n", err.c_str()); return nullptr; }
    if (hp.n_ff == 0) { const pi::gguf_tensor * t = f.tensor("blk.0.ffn_up.weight"); if (t) hp.n_ff = (int) t->ne[1]; }
    if (!params.vocab_only) {
        const pi::gguf_tensor * te = f.tensor("token_embd.weight");
        if (!te) { fprintf(stderr, "llama_load_model_from_file: no token_embd.weight in %s\n", path); return nullptr; }
        hp.n_vocab = (int) te->ne[1];
        m->layers.resize(hp.n_layer);
    } else {
        hp.n_vocab = (int) m->vocab.id_to_token.size();
    }
    for (const auto & t : f.tensors) { m->n_elements += (uint64_t) (t.ne[0] * t.ne[1] * t.ne[2] * t.ne[3]); m->n_bytes += t.nbytes; }
    return m.release();
}

extern "C" void llama_free_model(struct llama_model * model) { delete model; }

extern "C" int llama_n_vocab(const struct llama_model * m) { return m->hparams.n_vocab; }
extern "C" int llama_n_ctx_train(const struct llama_model * m) { return m->hparams.n_ctx_train; }
extern "C" int llama_n_embd(const struct llama_model * m) { return m->hparams.n_embd; }
extern "C" enum llama_vocab_type llama_vocab_type(const struct llama_model * m) { return (enum llama_vocab_type) m->vocab.type; }
extern "C" uint64_t llama_model_size(const struct llama_model * m) { return m->n_bytes; }
extern "C" uint64_t llama_model_n_params(const struct llama_model * m) { return m->n_elements; }
extern "C" int llama_model_desc(const struct llama_model * m, char * buf, size_t n) {
    static const char * names[] = { "llama", "falcon", "baichuan", "starcoder", "refact", "bloom", "mpt", "stablelm", "persimmon" };
    return snprintf(buf, n, "%s %s", names[m->hparams.arch], m->name.c_str());
}

extern "C" const char * llama_token_get_text(const struct llama_model * m, llama_token t) { return m->vocab.id_to_token[t].text.c_str(); }
extern "C" float llama_token_get_score(const struct llama_model * m, llama_token t) { return m->vocab.id_to_token[t].score; }
extern "C" enum llama_token_type llama_token_get_type(const struct llama_model * m, llama_token t) { return (enum llama_token_type) m->vocab.id_to_token[t].type; }
extern "C" llama_token llama_token_bos(const struct llama_model * m) { return m->vocab.bos_id; }
extern "C" llama_token llama_token_eos(const struct llama_model * m) { return m->vocab.eos_id; }
extern "C" llama_token llama_token_nl(const struct llama_model * m) { return m->vocab.linefeed_id; }
extern "C" int llama_add_bos_token(const struct llama_model * m) { return m->vocab.add_bos; }
extern "C" int llama_add_eos_token(const struct llama_model * m) { return m->vocab.add_eos; }

extern "C" int llama_tokenize(const struct llama_model * m, const char * text, int text_len, llama_token * tokens, int n_max, bool add_bos, bool special) {
    std::vector<int32_t> res;
    try { res = m->vocab.tokenize(std::string(text, (size_t) text_len), add_bos, special); }
    catch (const std::exception & e) { fprintf(stderr, "llama_tokenize: %s\n", e.what()); return 0; }
    if (n_max < (int) res.size()) return -(int) res.size();                  // llama.cpp:10092-10110
    for (size_t i = 0; i < res.size(); i++) tokens[i] = res[i];
    return (int) res.size();
}

extern "C" int llama_token_to_piece(const struct llama_model * m, llama_token token, char * buf, int length) {
    return m->vocab.token_to_piece(token, buf, length);
}
