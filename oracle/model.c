// TEST INFRASTRUCTURE — see oracle.h.  Whole-model forward of the reference (Llama and Falcon graphs) over the
// restated operators, plus a minimal GGUF reader and the greedy samplers.
#include "oracle.h"

#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void o_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void) n;
#endif
}

// ---- GGUF (layout: ggml.c:18083-18330) -------------------------------------------------------------------
typedef struct { char name[96]; int type; int n_dims; int64_t ne[4]; const void * data; } o_tensor;

struct o_model {
    void * map; size_t map_size;
    int n_tensors; o_tensor * tensors;
    int is_falcon, n_layer, n_embd, n_head, n_head_kv, n_ff, n_vocab, n_ctx;
    float eps, freq_base;
    // the other builders (build_baichuan / _starcoder / _refact / _bloom / _mpt / _stablelm, llama.cpp:3985-4939) as one sequential block
    int generic, rms, fused_qkv, rope_mode /* -1 none, 0, 2 */, gated, n_rot, persimmon;
    float alibi, clamp_kqv;
    o_kv_cache * kv;
    uint16_t * k_cache, * v_cache;     // per layer: K [n_ctx][n_embd_gqa], V [n_embd_gqa][n_ctx]  (llama.cpp:1529-1560, 3565-3576)
};

typedef struct { const uint8_t * p; } rd;
static uint32_t rd_u32(rd * r) { uint32_t v; memcpy(&v, r->p, 4); r->p += 4; return v; }
static uint64_t rd_u64(rd * r) { uint64_t v; memcpy(&v, r->p, 8); r->p += 8; return v; }
static void rd_str(rd * r, char * out, size_t cap) {
    const uint64_t n = rd_u64(r);
    const size_t c = n < cap - 1 ? n : cap - 1;
    memcpy(out, r->p, c); out[c] = 0; r->p += n;
}
static const size_t gguf_scalar_size[] = { 1, 1, 2, 2, 4, 4, 4, 1, 0, 0, 8, 8, 8 };   // u8 i8 u16 i16 u32 i32 f32 bool str arr u64 i64 f64

static void skip_value(rd * r, uint32_t type) {
    if (type == 8) { const uint64_t n = rd_u64(r); r->p += n; }
    else if (type == 9) {
        const uint32_t et = rd_u32(r); const uint64_t n = rd_u64(r);
        if (et == 8) for (uint64_t i = 0; i < n; i++) { const uint64_t l = rd_u64(r); r->p += l; }
        else r->p += n * gguf_scalar_size[et];
    } else r->p += gguf_scalar_size[type];
}

static const o_tensor * find_tensor(const o_model * m, const char * name) {
    for (int i = 0; i < m->n_tensors; i++) if (strcmp(m->tensors[i].name, name) == 0) return &m->tensors[i];
    return NULL;
}
static const o_tensor * layer_tensor(const o_model * m, int il, const char * name, int required) {
    char full[128]; snprintf(full, sizeof full, "blk.%d.%s", il, name);
    const o_tensor * t = find_tensor(m, full);
    if (!t && required) { fprintf(stderr, "oracle: missing tensor %s\n", full); abort(); }
    return t;
}

o_model * o_model_load(const char * path, int n_ctx, int max_seq) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return NULL;
    struct stat st; fstat(fd, &st);
    void * map = mmap(NULL, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return NULL;
    o_model * m = calloc(1, sizeof *m);
    m->map = map; m->map_size = st.st_size;
    rd r = { map };
    if (rd_u32(&r) != 0x46554747u) { o_model_free(m); return NULL; }    // "GGUF"
    rd_u32(&r);                                                          // version 2 or 3
    const uint64_t n_tensors = rd_u64(&r), n_kv = rd_u64(&r);
    char arch[64] = "llama";
    uint32_t alignment = 32;
    float max_alibi = 0.0f, clamp_kqv = 0.0f;
    m->freq_base = 10000.0f; m->n_head_kv = -1;
    for (uint64_t i = 0; i < n_kv; i++) {
        char key[256]; rd_str(&r, key, sizeof key);
        const uint32_t type = rd_u32(&r);
        const char * dot = strchr(key, '.');
        const char * sub = dot ? dot + 1 : key;
        if (strcmp(key, "general.architecture") == 0 && type == 8) { rd_str(&r, arch, sizeof arch); continue; }
        if (strcmp(key, "general.alignment") == 0 && type == 4) { alignment = rd_u32(&r); continue; }
        if (type == 4 && dot && strncmp(key, arch, dot - key) == 0) {
            rd peek = r; const uint32_t v = rd_u32(&peek);
            if      (!strcmp(sub, "block_count"))              m->n_layer = v;
            else if (!strcmp(sub, "embedding_length"))         m->n_embd = v;
            else if (!strcmp(sub, "feed_forward_length"))      m->n_ff = v;
            else if (!strcmp(sub, "attention.head_count"))     m->n_head = v;
            else if (!strcmp(sub, "attention.head_count_kv"))  m->n_head_kv = v;
            else if (!strcmp(sub, "rope.dimension_count"))     m->n_rot = v;
        }
        if (type == 6 && dot) {
            rd peek = r; float v; memcpy(&v, peek.p, 4);
            if (!strcmp(sub, "attention.layer_norm_rms_epsilon") || !strcmp(sub, "attention.layer_norm_epsilon")) m->eps = v;
            else if (!strcmp(sub, "rope.freq_base")) m->freq_base = v;
            else if (!strcmp(sub, "attention.max_alibi_bias")) max_alibi = v;
            else if (!strcmp(sub, "attention.clamp_kqv")) clamp_kqv = v;
        }
        skip_value(&r, type);
    }
    m->is_falcon = strcmp(arch, "falcon") == 0;
    if (m->n_head_kv < 0) m->n_head_kv = m->n_head;
    if (!m->n_rot) m->n_rot = m->n_embd / m->n_head;                          // llama.cpp:2265-2267
    m->rope_mode = -1;
    if (!strcmp(arch, "baichuan")) {                                          // 7B (32 layers) rotates, 13B (40) has ALiBi (llama.cpp:4030-4063)
        m->generic = m->rms = m->gated = 1;
        if (m->n_layer == 32) m->rope_mode = 0; else m->alibi = 8.0f;
    } else if (!strcmp(arch, "refact"))    { m->generic = m->rms = m->gated = 1; m->alibi = 8.0f; }
    else if (!strcmp(arch, "starcoder"))   { m->generic = m->fused_qkv = 1; }
    else if (!strcmp(arch, "bloom"))       { m->generic = m->fused_qkv = 1; m->alibi = 8.0f; }
    else if (!strcmp(arch, "mpt"))         { m->generic = m->fused_qkv = 1; m->alibi = max_alibi; m->clamp_kqv = clamp_kqv; }
    else if (!strcmp(arch, "stablelm"))    { m->generic = m->gated = 1; m->rope_mode = 2; }
    else if (!strcmp(arch, "persimmon"))   { m->generic = m->fused_qkv = m->persimmon = 1; m->rope_mode = 2; }      // build_persimmon, llama.cpp:4329-4538
    m->n_tensors = (int) n_tensors;
    m->tensors = calloc(n_tensors, sizeof(o_tensor));
    uint64_t * offs = calloc(n_tensors, 8);
    for (uint64_t i = 0; i < n_tensors; i++) {
        o_tensor * t = &m->tensors[i];
        rd_str(&r, t->name, sizeof t->name);
        t->n_dims = rd_u32(&r);
        for (int d = 0; d < 4; d++) t->ne[d] = 1;
        for (int d = 0; d < t->n_dims; d++) t->ne[d] = rd_u64(&r);
        t->type = rd_u32(&r);
        offs[i] = rd_u64(&r);
    }
    size_t data_off = (size_t) (r.p - (const uint8_t *) map);
    data_off = (data_off + alignment - 1) / alignment * alignment;
    for (uint64_t i = 0; i < n_tensors; i++) m->tensors[i].data = (const uint8_t *) map + data_off + offs[i];
    free(offs);
    const o_tensor * te = find_tensor(m, "token_embd.weight");
    m->n_vocab = (int) te->ne[1];
    m->n_ctx = n_ctx;
    m->kv = o_kv_new(n_ctx, max_seq);
    const size_t kvn = (size_t) m->n_layer * n_ctx * (m->n_embd / m->n_head) * m->n_head_kv;
    m->k_cache = calloc(kvn, 2); m->v_cache = calloc(kvn, 2);
    o_tables_init();
    return m;
}

void o_model_free(o_model * m) {
    if (m->kv) o_kv_free(m->kv);
    free(m->k_cache); free(m->v_cache); free(m->tensors);
    munmap(m->map, m->map_size);
    free(m);
}

int o_model_info(const o_model * m, int what) {
    switch (what) {
        case 0: return m->n_vocab; case 1: return m->n_embd; case 2: return m->n_layer; case 3: return m->n_head;
        case 4: return m->n_head_kv; case 5: return m->n_ff; case 6: return m->is_falcon; case 7: return m->n_ctx;
    }
    return -1;
}
o_kv_cache * o_model_kv(o_model * m) { return m->kv; }

static void trace(const char * what, int il, const float * x, int n, int n_tok) {
    if (!getenv("PI_TRACE")) return;
    fprintf(stderr, "TRACE %d %-8s", il, what);
    for (int t = 0; t < n_tok; t++) { double s = 0; for (int i = 0; i < n; i++) s += fabs((double) x[(size_t) t * n + i]); fprintf(stderr, " %.6e", s); }
    fprintf(stderr, "\n");
}

static void mm(const o_tensor * w, const float * act, int n_tok, float * out) {
    o_mul_mat(w->type, w->data, (int) w->ne[0], (int) w->ne[1], act, n_tok, out);
}

// llm_build_norm (llama.cpp:3579-3609): norm, then ggml_mul by weight, then ggml_add bias — separate fp32 ops
static void norm_wb(const float * x, float * y, int n, int rows, int rms, float eps, const o_tensor * w, const o_tensor * b) {
    if (rms) o_rms_norm(x, y, n, rows, eps); else o_norm(x, y, n, rows, eps);
    const float * wd = w ? w->data : NULL, * bd = b ? b->data : NULL;
    for (int r = 0; r < rows; r++) for (int i = 0; i < n; i++) {
        float v = y[(size_t) r * n + i];
        if (wd) v = v * wd[i];
        if (bd) v = v + bd[i];
        y[(size_t) r * n + i] = v;
    }
}

// ggml_add of a row vector to every token's row (the projections' biases, llama.cpp:3628-3630, 3770-3772); no-op without the tensor
static void add_bias(float * x, const o_tensor * b, int n, int rows) {
    if (!b) return;
    const float * bd = b->data;
    for (int r = 0; r < rows; r++) for (int i = 0; i < n; i++) x[(size_t) r * n + i] = x[(size_t) r * n + i] + bd[i];
}

// build_llama (llama.cpp:3872-3983) / build_falcon (llama.cpp:4106-4227) / the sequential block of the six other builders (llama.cpp:3985-4939)
int o_model_forward(o_model * m, const int32_t * tokens, const int32_t * pos, int n_tok, int kv_head, int n_kv, const float * mask, float * logits) {
    const int n_embd = m->n_embd, hd = n_embd / m->n_head, n_embd_gqa = hd * m->n_head_kv, n_ff = m->n_ff;
    const size_t T = n_tok;
    float * inpL = malloc(4 * T * n_embd), * cur = malloc(4 * T * n_embd), * attn_norm = malloc(4 * T * n_embd);
    float * qkv = malloc(4 * T * (n_embd + 2 * n_embd_gqa));
    float * Q = malloc(4 * T * n_embd), * K = malloc(4 * T * n_embd_gqa), * V = malloc(4 * T * n_embd_gqa);
    float * Qr = malloc(4 * T * n_embd), * Kr = malloc(4 * T * n_embd_gqa);
    float * att = malloc(4 * T * n_embd), * ffn_inp = malloc(4 * T * n_embd);
    float * up = malloc(4 * T * n_ff), * gate = malloc(4 * T * n_ff), * tmp = malloc(4 * T * n_embd);
    const o_tensor * te = find_tensor(m, "token_embd.weight");
    o_get_rows(te->type, te->data, n_embd, tokens, n_tok, inpL);
    const size_t kv_layer = (size_t) m->n_ctx * n_embd_gqa;

    if (find_tensor(m, "position_embd.weight")) {                            // build_starcoder (llama.cpp:4251-4255)
        const o_tensor * pe = find_tensor(m, "position_embd.weight");
        o_get_rows(pe->type, pe->data, n_embd, pos, n_tok, cur);
        for (size_t i = 0; i < T * n_embd; i++) inpL[i] = inpL[i] + cur[i];
    }
    if (find_tensor(m, "token_embd_norm.weight")) {                          // build_bloom (llama.cpp:4650-4654)
        norm_wb(inpL, cur, n_embd, n_tok, 0, m->eps, find_tensor(m, "token_embd_norm.weight"), find_tensor(m, "token_embd_norm.bias"));
        memcpy(inpL, cur, 4 * T * n_embd);
    }

    for (int il = 0; il < m->n_layer; il++) {
        if (m->generic) {
            const int nq = n_embd + 2 * n_embd_gqa;
            norm_wb(inpL, cur, n_embd, n_tok, m->rms, m->eps, layer_tensor(m, il, "attn_norm.weight", 1), layer_tensor(m, il, "attn_norm.bias", 0));
            if (m->fused_qkv) {
                mm(layer_tensor(m, il, "attn_qkv.weight", 1), cur, n_tok, qkv);
                add_bias(qkv, layer_tensor(m, il, "attn_qkv.bias", 0), nq, n_tok);
                if (m->clamp_kqv > 0.0f) for (size_t i = 0; i < T * nq; i++) { const float lo = qkv[i] < m->clamp_kqv ? qkv[i] : m->clamp_kqv; qkv[i] = lo > -m->clamp_kqv ? lo : -m->clamp_kqv; }   // MAX(MIN(x, max), min), ggml.c:10928
                if (m->persimmon) {
                    // rows are [head][q | k | v][hd] (reshape_4d(hd, 3, n_head, n_tokens), llama.cpp:4372-4393); Q and K get a LayerNorm over the hd values of their head
                    for (size_t t = 0; t < T; t++) for (int h = 0; h < m->n_head; h++) {
                        const float * src = qkv + t * nq + (size_t) h * 3 * hd;
                        memcpy(Q + t * n_embd + (size_t) h * hd, src,          4 * (size_t) hd);
                        memcpy(K + t * n_embd + (size_t) h * hd, src + hd,     4 * (size_t) hd);
                        memcpy(V + t * n_embd + (size_t) h * hd, src + 2 * hd, 4 * (size_t) hd);
                    }
                    norm_wb(Q, Qr, hd, n_tok * m->n_head, 0, m->eps, layer_tensor(m, il, "attn_q_norm.weight", 1), layer_tensor(m, il, "attn_q_norm.bias", 1));
                    norm_wb(K, Kr, hd, n_tok * m->n_head, 0, m->eps, layer_tensor(m, il, "attn_k_norm.weight", 1), layer_tensor(m, il, "attn_k_norm.bias", 1));
                    memcpy(Q, Qr, 4 * T * n_embd); memcpy(K, Kr, 4 * T * n_embd);
                } else
                for (size_t t = 0; t < T; t++) {
                    memcpy(Q + t * n_embd,     qkv + t * nq,                       4 * (size_t) n_embd);
                    memcpy(K + t * n_embd_gqa, qkv + t * nq + n_embd,              4 * (size_t) n_embd_gqa);
                    memcpy(V + t * n_embd_gqa, qkv + t * nq + n_embd + n_embd_gqa, 4 * (size_t) n_embd_gqa);
                }
            } else {
                mm(layer_tensor(m, il, "attn_q.weight", 1), cur, n_tok, Q);
                mm(layer_tensor(m, il, "attn_k.weight", 1), cur, n_tok, K);
                mm(layer_tensor(m, il, "attn_v.weight", 1), cur, n_tok, V);
            }
            trace("q", il, Q, n_embd, n_tok); trace("k", il, K, n_embd_gqa, n_tok); trace("v", il, V, n_embd_gqa, n_tok);
            if (m->persimmon) {
                // NeoX RoPE of a VIEW of the first hd / 2 values of every head with n_dims = hd / 2, the other half passes (llama.cpp:4331, 4395-4455)
                const int nr = hd / 2;
                float * a = malloc(4 * (size_t) nr), * b = malloc(4 * (size_t) nr);
                for (int which = 0; which < 2; which++) {
                    const float * src = which ? K : Q; float * dst = which ? Kr : Qr;
                    memcpy(dst, src, 4 * T * n_embd);
                    for (size_t t = 0; t < T; t++) for (int h = 0; h < m->n_head; h++) {
                        memcpy(a, src + t * n_embd + (size_t) h * hd, 4 * (size_t) nr);
                        o_rope_ex(a, b, nr, nr, 1, 1, pos + t, 2, m->freq_base, 1.0f);
                        memcpy(dst + t * n_embd + (size_t) h * hd, b, 4 * (size_t) nr);
                    }
                }
                free(a); free(b);
            } else
            if (m->rope_mode >= 0) {
                o_rope_ex(Q, Qr, hd, m->rope_mode == 2 ? m->n_rot : hd, m->n_head,    n_tok, pos, m->rope_mode, m->freq_base, 1.0f);
                o_rope_ex(K, Kr, hd, m->rope_mode == 2 ? m->n_rot : hd, m->n_head_kv, n_tok, pos, m->rope_mode, m->freq_base, 1.0f);
            } else { memcpy(Qr, Q, 4 * T * n_embd); memcpy(Kr, K, 4 * T * n_embd_gqa); }
            o_attention_alibi(m->k_cache + il * kv_layer, m->v_cache + il * kv_layer, m->n_ctx, Qr, Kr, V, mask, hd, m->n_head, m->n_head_kv, n_tok, kv_head, n_kv, m->alibi, att);
            trace("att", il, att, n_embd, n_tok);
            mm(layer_tensor(m, il, "attn_output.weight", 1), att, n_tok, cur);
            add_bias(cur, layer_tensor(m, il, "attn_output.bias", 0), n_embd, n_tok);
            for (size_t i = 0; i < T * n_embd; i++) ffn_inp[i] = cur[i] + inpL[i];
            trace("ffn_inp", il, ffn_inp, n_embd, n_tok);
            norm_wb(ffn_inp, cur, n_embd, n_tok, m->rms, m->eps, layer_tensor(m, il, "ffn_norm.weight", 1), layer_tensor(m, il, "ffn_norm.bias", 0));
            mm(layer_tensor(m, il, "ffn_up.weight", 1), cur, n_tok, up);          // llm_build_ffn (llama.cpp:3611-3695)
            add_bias(up, layer_tensor(m, il, "ffn_up.bias", 0), n_ff, n_tok);
            if (m->gated) {
                mm(layer_tensor(m, il, "ffn_gate.weight", 1), cur, n_tok, gate);
                o_silu(gate, gate, (int) (T * n_ff));
                for (size_t i = 0; i < T * n_ff; i++) gate[i] = gate[i] * up[i];
            } else if (m->persimmon) {                                               // LLM_FFN_RELU_SQR: ggml_relu, ggml_sqr (ggml.c:1415, 1407)
                for (size_t i = 0; i < T * n_ff; i++) { const float a = up[i] > 0.f ? up[i] : 0.f; gate[i] = a * a; }
            } else {
                o_gelu(up, gate, (int) (T * n_ff));
            }
            mm(layer_tensor(m, il, "ffn_down.weight", 1), gate, n_tok, tmp);
            add_bias(tmp, layer_tensor(m, il, "ffn_down.bias", 0), n_embd, n_tok);
            for (size_t i = 0; i < T * n_embd; i++) inpL[i] = tmp[i] + ffn_inp[i];
            trace("lout", il, inpL, n_embd, n_tok);
            continue;
        }
        if (!m->is_falcon) {
            norm_wb(inpL, cur, n_embd, n_tok, 1, m->eps, layer_tensor(m, il, "attn_norm.weight", 1), NULL);
            mm(layer_tensor(m, il, "attn_q.weight", 1), cur, n_tok, Q);
            mm(layer_tensor(m, il, "attn_k.weight", 1), cur, n_tok, K);
            mm(layer_tensor(m, il, "attn_v.weight", 1), cur, n_tok, V);
        } else {
            norm_wb(inpL, attn_norm, n_embd, n_tok, 0, m->eps, layer_tensor(m, il, "attn_norm.weight", 1), layer_tensor(m, il, "attn_norm.bias", 1));
            const o_tensor * n2 = layer_tensor(m, il, "attn_norm_2.weight", 0);
            const float * src = attn_norm;
            if (n2) { norm_wb(inpL, cur, n_embd, n_tok, 0, m->eps, n2, layer_tensor(m, il, "attn_norm_2.bias", 1)); src = cur; }
            mm(layer_tensor(m, il, "attn_qkv.weight", 1), src, n_tok, qkv);
            const int nq = n_embd + 2 * n_embd_gqa;
            for (size_t t = 0; t < T; t++) {
                memcpy(Q + t * n_embd,     qkv + t * nq,                       4 * (size_t) n_embd);
                memcpy(K + t * n_embd_gqa, qkv + t * nq + n_embd,              4 * (size_t) n_embd_gqa);
                memcpy(V + t * n_embd_gqa, qkv + t * nq + n_embd + n_embd_gqa, 4 * (size_t) n_embd_gqa);
            }
        }
        const int mode = m->is_falcon ? 2 : 0;
        o_rope(Q, Qr, hd, m->n_head,    n_tok, pos, mode, m->freq_base, 1.0f);
        o_rope(K, Kr, hd, m->n_head_kv, n_tok, pos, mode, m->freq_base, 1.0f);
        trace("q", il, Q, n_embd, n_tok); trace("k", il, K, n_embd_gqa, n_tok); trace("v", il, V, n_embd_gqa, n_tok);
        o_attention(m->k_cache + il * kv_layer, m->v_cache + il * kv_layer, m->n_ctx, Qr, Kr, V, mask,
                    hd, m->n_head, m->n_head_kv, n_tok, kv_head, n_kv, att);
        trace("att", il, att, n_embd, n_tok);
        mm(layer_tensor(m, il, "attn_output.weight", 1), att, n_tok, cur);
        trace("wo", il, cur, n_embd, n_tok);

        if (!m->is_falcon) {
            for (size_t i = 0; i < T * n_embd; i++) ffn_inp[i] = cur[i] + inpL[i];
            norm_wb(ffn_inp, cur, n_embd, n_tok, 1, m->eps, layer_tensor(m, il, "ffn_norm.weight", 1), NULL);
            mm(layer_tensor(m, il, "ffn_up.weight", 1),   cur, n_tok, up);      // llm_build_ffn, llama.cpp:3611-3695
            mm(layer_tensor(m, il, "ffn_gate.weight", 1), cur, n_tok, gate);
            o_silu(gate, gate, (int) (T * n_ff));
            for (size_t i = 0; i < T * n_ff; i++) gate[i] = gate[i] * up[i];
            mm(layer_tensor(m, il, "ffn_down.weight", 1), gate, n_tok, tmp);
            for (size_t i = 0; i < T * n_embd; i++) inpL[i] = tmp[i] + ffn_inp[i];
            trace("lout", il, inpL, n_embd, n_tok);
        } else {
            mm(layer_tensor(m, il, "ffn_up.weight", 1), attn_norm, n_tok, up);  // parallel block: MLP consumes attn_norm (llama.cpp:4194)
            o_gelu(up, up, (int) (T * n_ff));
            mm(layer_tensor(m, il, "ffn_down.weight", 1), up, n_tok, tmp);
            trace("up", il, up, n_ff, n_tok);
            for (size_t i = 0; i < T * n_embd; i++) { const float a = tmp[i] + cur[i]; inpL[i] = a + inpL[i]; }
            trace("lout", il, inpL, n_embd, n_tok);
        }
    }
    if (m->generic)        norm_wb(inpL, cur, n_embd, n_tok, m->rms, m->eps, find_tensor(m, "output_norm.weight"), find_tensor(m, "output_norm.bias"));
    else if (m->is_falcon) norm_wb(inpL, cur, n_embd, n_tok, 0, m->eps, find_tensor(m, "output_norm.weight"), find_tensor(m, "output_norm.bias"));
    else                   norm_wb(inpL, cur, n_embd, n_tok, 1, m->eps, find_tensor(m, "output_norm.weight"), NULL);
    mm(find_tensor(m, "output.weight"), cur, n_tok, logits);
    free(inpL); free(cur); free(attn_norm); free(qkv); free(Q); free(K); free(V); free(Qr); free(Kr);
    free(att); free(ffn_inp); free(up); free(gate); free(tmp);
    return 0;
}

// llama_decode_internal_phased phase 0 + 1 on a single rank (llama.cpp:5461-5827)
int o_model_decode(o_model * m, int n_tok, const int32_t * tokens, const int32_t * pos, const int32_t * n_seq_id,
                   const int32_t * seq_ids, int stride, float * logits) {
    o_kv_pre_decode(m->kv, n_tok);
    const int head = o_kv_find_slot(m->kv, n_tok, pos, n_seq_id, seq_ids, stride);
    if (head < 0) return 1;
    const int n_kv = o_kv_n(m->kv);
    float * mask = malloc((size_t) 4 * n_kv * n_tok);
    o_kv_build_mask(m->kv, n_kv, n_tok, pos, n_seq_id, seq_ids, stride, mask);
    o_model_forward(m, tokens, pos, n_tok, head, n_kv, mask, logits);
    o_kv_post_decode(m->kv, n_tok);
    free(mask);
    return 0;
}

// llama_sample_token_greedy (llama.cpp:7496-7510): std::max_element keeps the FIRST maximum
int o_argmax(const float * x, int n) {
    int best = 0;
    for (int i = 1; i < n; i++) if (x[i] > x[best]) best = i;
    return best;
}

// llama_sample_softmax (llama.cpp:7019-7046): sort descending by logit, p = expf(l - max)/sum in fp32 over the FULL vocab.
// Only the first k entries are returned (the driver reads the top 8, speculative.cpp:981,1009). Ties: lower id first.
void o_softmax_topk(const float * logits, int n, int k, int32_t * ids, float * logit, float * p) {
    // selection of the k best by (logit desc, id asc)
    for (int j = 0; j < k; j++) { ids[j] = -1; logit[j] = -INFINITY; }
    for (int i = 0; i < n; i++) {
        int j = k;
        while (j > 0 && (ids[j - 1] < 0 || logits[i] > logit[j - 1])) j--;
        if (j < k) {
            for (int q = k - 1; q > j; q--) { ids[q] = ids[q - 1]; logit[q] = logit[q - 1]; }
            ids[j] = i; logit[j] = logits[i];
        }
    }
    // the reference accumulates cum_sum over the SORTED array; fp32 addition order = descending logit.
    // Reproducing that order needs the full sort; do it for exactness.
    int32_t * order = malloc((size_t) 4 * n);
    for (int i = 0; i < n; i++) order[i] = i;
    // simple merge sort by (logit desc, id asc)
    int32_t * tmp = malloc((size_t) 4 * n);
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, a = lo, b = mid, o = lo;
            while (a < mid && b < hi) tmp[o++] = (logits[order[b]] > logits[order[a]]) ? order[b++] : order[a++];
            while (a < mid) tmp[o++] = order[a++];
            while (b < hi) tmp[o++] = order[b++];
        }
        memcpy(order, tmp, (size_t) 4 * n);
    }
    const float max_l = logits[order[0]];
    float cum = 0.0f;
    for (int i = 0; i < n; i++) cum += expf(logits[order[i]] - max_l);
    for (int j = 0; j < k; j++) p[j] = expf(logit[j] - max_l) / cum;
    free(order); free(tmp);
}
