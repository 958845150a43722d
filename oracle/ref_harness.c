// TEST INFRASTRUCTURE — not product code.
//
// Thin C harness around the *real* reference tensor library (ggml.c / ggml-quants.c /
// ggml-alloc.c / ggml-backend.c compiled from /root/reference by oracle/Makefile into
// oracle/_ref/libggml_ref.so).  Nothing from the reference is copied here: this file only
// *calls* the reference's public API (ggml.h) so that
//   (1) oracle/*.c (our CPU restatement) can be pinned against the reference's own kernels, and
//   (2) bench.py can time the reference CPU kernels on the host cores ("cpu_baseline.kind=reference").
//
// The model-level entry points drive the reference's ops in the order llama.cpp's graph builders use
// (llm_build_norm / llm_build_kv_store / llm_build_kqv / llm_build_ffn, llama.cpp:3546-3778;
// build_llama llama.cpp:3872-3983; build_falcon llama.cpp:4106-4227).  llama.cpp itself is NOT
// buildable in this image (it needs <mpi.h>, llama.cpp:5457-5479), so KV-cell placement and the
// attention mask are supplied by the caller (oracle/kvcache.c restates llama.cpp:1593-1765,5200-5240).
//
// Built only where /root/reference exists; the resulting .so travels to the GPU box, sources do not.

#include "ggml.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REF_API __attribute__((visibility("default")))

static void ref_init_once(void) {
    static int done = 0;
    if (!done) {
        // ggml_init fills the fp16 lookup tables (ggml.c:2190-2201) on first call
        struct ggml_init_params ip = { 1024, NULL, false };
        struct ggml_context * c = ggml_init(ip);
        ggml_free(c);
        done = 1;
    }
}

// ----------------------------------------------------------------------------------------------
// type traits level (ggml.h:2141-2157)
// ----------------------------------------------------------------------------------------------

REF_API int ref_blck_size(int type) { return ggml_blck_size((enum ggml_type) type); }
REF_API int ref_type_size(int type) { return (int) ggml_type_size((enum ggml_type) type); }
REF_API int ref_vec_dot_type(int type) { ref_init_once(); return (int) ggml_internal_get_type_traits((enum ggml_type) type).vec_dot_type; }

// runtime activation/weight quantiser (SIMD variant where the reference has one, e.g. q8_0 AVX path)
REF_API void ref_quantize(int type, const float * x, void * y, int k) {
    ref_init_once();
    ggml_internal_get_type_traits((enum ggml_type) type).from_float(x, y, k);
}

// deterministic "reference" quantiser used when model files are produced (llama_model_quantize)
REF_API void ref_quantize_reference(int type, const float * x, void * y, int k) {
    ref_init_once();
    ggml_internal_get_type_traits((enum ggml_type) type).from_float_reference(x, y, k);
}

REF_API void ref_dequantize(int type, const void * x, float * y, int k) {
    ref_init_once();
    ggml_internal_get_type_traits((enum ggml_type) type).to_float(x, y, k);
}

// x: n elements of `type`, y: n elements already in type's vec_dot_type
REF_API float ref_vec_dot(int type, int n, const void * x, const void * y) {
    ref_init_once();
    float s = 0.0f;
    ggml_internal_get_type_traits((enum ggml_type) type).vec_dot(n, &s, x, y);
    return s;
}

REF_API void ref_fp32_to_fp16_row(const float * x, uint16_t * y, int n) { ref_init_once(); ggml_fp32_to_fp16_row(x, (ggml_fp16_t *) y, n); }
REF_API void ref_fp16_to_fp32_row(const uint16_t * x, float * y, int n) { ref_init_once(); ggml_fp16_to_fp32_row((const ggml_fp16_t *) x, y, n); }

// ----------------------------------------------------------------------------------------------
// single-op graphs
// ----------------------------------------------------------------------------------------------

typedef struct {
    struct ggml_context * ctx;
    void * buf;
} ref_scratch;

static ref_scratch scratch_new(size_t bytes) {
    ref_init_once();
    ref_scratch s;
    s.buf = malloc(bytes);
    struct ggml_init_params ip = { bytes, s.buf, false };
    s.ctx = ggml_init(ip);
    return s;
}

static void scratch_free(ref_scratch s) {
    ggml_free(s.ctx);
    free(s.buf);
}

static void compute(struct ggml_context * ctx, struct ggml_tensor * out, int n_threads) {
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, out);
    ggml_graph_compute_with_ctx(ctx, gf, n_threads);
}

// dst[ne01, n_tok] = W[ne01 rows x ne00] * act[ne00, n_tok]  — the whole ggml_mul_mat op (ggml.c:9388-9597)
REF_API void ref_mul_mat(int type, const void * w, int ne00, int ne01, const float * act, int n_tok, float * out, int n_threads) {
    const size_t wbytes = (size_t) ne01 * ggml_type_size((enum ggml_type) type) * (ne00 / ggml_blck_size((enum ggml_type) type));
    ref_scratch s = scratch_new(wbytes + (size_t) 4 * ne00 * n_tok * 3 + (size_t) 4 * ne01 * n_tok + (64u << 20));
    struct ggml_tensor * W = ggml_new_tensor_2d(s.ctx, (enum ggml_type) type, ne00, ne01);
    memcpy(W->data, w, wbytes);
    struct ggml_tensor * A = ggml_new_tensor_2d(s.ctx, GGML_TYPE_F32, ne00, n_tok);
    memcpy(A->data, act, (size_t) 4 * ne00 * n_tok);
    struct ggml_tensor * O = ggml_mul_mat(s.ctx, W, A);
    compute(s.ctx, O, n_threads);
    memcpy(out, O->data, (size_t) 4 * ne01 * n_tok);
    scratch_free(s);
}

REF_API void ref_rms_norm(const float * x, float * y, int n, int rows, float eps) {
    ref_scratch s = scratch_new((size_t) 8 * n * rows + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_2d(s.ctx, GGML_TYPE_F32, n, rows);
    memcpy(X->data, x, (size_t) 4 * n * rows);
    struct ggml_tensor * O = ggml_rms_norm(s.ctx, X, eps);
    compute(s.ctx, O, 1);
    memcpy(y, O->data, (size_t) 4 * n * rows);
    scratch_free(s);
}

REF_API void ref_norm(const float * x, float * y, int n, int rows, float eps) {
    ref_scratch s = scratch_new((size_t) 8 * n * rows + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_2d(s.ctx, GGML_TYPE_F32, n, rows);
    memcpy(X->data, x, (size_t) 4 * n * rows);
    struct ggml_tensor * O = ggml_norm(s.ctx, X, eps);
    compute(s.ctx, O, 1);
    memcpy(y, O->data, (size_t) 4 * n * rows);
    scratch_free(s);
}

// x,y: [head_dim, n_head, n_tok] f32; mode 0 = llama pairs, 2 = NeoX (falcon)
REF_API void ref_rope(const float * x, float * y, int head_dim, int n_head, int n_tok, const int32_t * pos,
                      int mode, int n_orig_ctx, float freq_base, float freq_scale) {
    const size_t n = (size_t) head_dim * n_head * n_tok;
    ref_scratch s = scratch_new(8 * n + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_3d(s.ctx, GGML_TYPE_F32, head_dim, n_head, n_tok);
    memcpy(X->data, x, 4 * n);
    struct ggml_tensor * P = ggml_new_tensor_1d(s.ctx, GGML_TYPE_I32, n_tok);
    memcpy(P->data, pos, (size_t) 4 * n_tok);
    struct ggml_tensor * O = ggml_rope_custom(s.ctx, X, P, head_dim, mode, 0, n_orig_ctx, freq_base, freq_scale, 0.0f, 1.0f, 32.0f, 1.0f);
    compute(s.ctx, O, 1);
    memcpy(y, O->data, 4 * n);
    scratch_free(s);
}

REF_API void ref_soft_max(const float * x, float * y, int nc, int nr) {
    ref_scratch s = scratch_new((size_t) 8 * nc * nr + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_2d(s.ctx, GGML_TYPE_F32, nc, nr);
    memcpy(X->data, x, (size_t) 4 * nc * nr);
    struct ggml_tensor * O = ggml_soft_max(s.ctx, X);
    compute(s.ctx, O, 1);
    memcpy(y, O->data, (size_t) 4 * nc * nr);
    scratch_free(s);
}

REF_API void ref_unary(int which, const float * x, float * y, int n) {
    ref_scratch s = scratch_new((size_t) 8 * n + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_1d(s.ctx, GGML_TYPE_F32, n);
    memcpy(X->data, x, (size_t) 4 * n);
    struct ggml_tensor * O = which == 0 ? ggml_silu(s.ctx, X) : ggml_gelu(s.ctx, X);
    compute(s.ctx, O, 1);
    memcpy(y, O->data, (size_t) 4 * n);
    scratch_free(s);
}

REF_API void ref_get_rows(int type, const void * w, int ne00, int ne01, const int32_t * ids, int n, float * out) {
    const size_t wbytes = (size_t) ne01 * ggml_type_size((enum ggml_type) type) * (ne00 / ggml_blck_size((enum ggml_type) type));
    ref_scratch s = scratch_new(wbytes + (size_t) 4 * ne00 * n + (16u << 20));
    struct ggml_tensor * W = ggml_new_tensor_2d(s.ctx, (enum ggml_type) type, ne00, ne01);
    memcpy(W->data, w, wbytes);
    struct ggml_tensor * I = ggml_new_tensor_1d(s.ctx, GGML_TYPE_I32, n);
    memcpy(I->data, ids, (size_t) 4 * n);
    struct ggml_tensor * O = ggml_get_rows(s.ctx, W, I);
    compute(s.ctx, O, 1);
    memcpy(out, O->data, (size_t) 4 * ne00 * n);
    scratch_free(s);
}

// One attention block exactly as llm_build_kv_store + llm_build_kqv wire it (llama.cpp:3546-3577, 3698-3764),
// minus the wo projection.  k_cache/v_cache are the reference layouts: K [n_ctx][n_embd_gqa] f16 rows,
// V transposed [n_embd_gqa][n_ctx] f16 (both for ONE layer, updated in place).
REF_API void ref_attention(uint16_t * k_cache, uint16_t * v_cache, int n_ctx,
                           const float * q_cur /*[hd,n_head,n_tok] roped*/, const float * k_cur /*[hd,n_head_kv,n_tok] roped*/,
                           const float * v_cur /*[n_embd_gqa,n_tok]*/, const float * mask /*[n_kv,n_tok]*/,
                           int head_dim, int n_head, int n_head_kv, int n_tok, int kv_head, int n_kv,
                           float * out /*[n_embd,n_tok]*/, int n_threads) {
    const int n_embd = head_dim * n_head, n_embd_gqa = head_dim * n_head_kv;
    const size_t kvb = (size_t) 2 * n_ctx * n_embd_gqa;
    ref_scratch s = scratch_new(2 * kvb + (size_t) 16 * n_embd * n_tok + (size_t) 16 * n_kv * n_tok * n_head + (64u << 20));
    struct ggml_context * c = s.ctx;
    struct ggml_tensor * K = ggml_new_tensor_1d(c, GGML_TYPE_F16, (int64_t) n_ctx * n_embd_gqa);
    struct ggml_tensor * V = ggml_new_tensor_1d(c, GGML_TYPE_F16, (int64_t) n_ctx * n_embd_gqa);
    memcpy(K->data, k_cache, kvb);
    memcpy(V->data, v_cache, kvb);
    struct ggml_tensor * Q  = ggml_new_tensor_3d(c, GGML_TYPE_F32, head_dim, n_head, n_tok);     memcpy(Q->data,  q_cur, (size_t) 4 * n_embd * n_tok);
    struct ggml_tensor * Kc = ggml_new_tensor_3d(c, GGML_TYPE_F32, head_dim, n_head_kv, n_tok);  memcpy(Kc->data, k_cur, (size_t) 4 * n_embd_gqa * n_tok);
    struct ggml_tensor * Vc = ggml_new_tensor_2d(c, GGML_TYPE_F32, n_embd_gqa, n_tok);           memcpy(Vc->data, v_cur, (size_t) 4 * n_embd_gqa * n_tok);
    struct ggml_tensor * M  = ggml_new_tensor_3d(c, GGML_TYPE_F32, n_kv, n_tok, 1);              memcpy(M->data,  mask,  (size_t) 4 * n_kv * n_tok);
    struct ggml_tensor * S  = ggml_new_tensor_1d(c, GGML_TYPE_F32, 1);
    ggml_set_f32(S, 1.0f / sqrtf((float) head_dim));

    struct ggml_cgraph * gf = ggml_new_graph(c);
    const size_t es = ggml_element_size(K);
    struct ggml_tensor * v_cur_t = ggml_transpose(c, ggml_reshape_2d(c, Vc, n_embd_gqa, n_tok));
    struct ggml_tensor * k_view  = ggml_view_1d(c, K, (int64_t) n_tok * n_embd_gqa, es * n_embd_gqa * kv_head);
    struct ggml_tensor * v_view  = ggml_view_2d(c, V, n_tok, n_embd_gqa, (size_t) n_ctx * es, kv_head * es);
    ggml_build_forward_expand(gf, ggml_cpy(c, Kc, k_view));
    ggml_build_forward_expand(gf, ggml_cpy(c, v_cur_t, v_view));

    struct ggml_tensor * q  = ggml_permute(c, Q, 0, 2, 1, 3);
    struct ggml_tensor * k  = ggml_view_3d(c, K, head_dim, n_kv, n_head_kv, es * n_embd_gqa, es * head_dim, 0);
    struct ggml_tensor * kq = ggml_mul_mat(c, k, q);
    kq = ggml_scale(c, kq, S);
    kq = ggml_add(c, kq, M);
    kq = ggml_soft_max(c, kq);
    struct ggml_tensor * v   = ggml_view_3d(c, V, n_kv, head_dim, n_head_kv, es * n_ctx, es * n_ctx * head_dim, 0);
    struct ggml_tensor * kqv = ggml_mul_mat(c, v, kq);
    struct ggml_tensor * mrg = ggml_permute(c, kqv, 0, 2, 1, 3);
    struct ggml_tensor * cur = ggml_cont_2d(c, mrg, n_embd, n_tok);
    ggml_build_forward_expand(gf, cur);
    ggml_graph_compute_with_ctx(c, gf, n_threads);

    memcpy(out, cur->data, (size_t) 4 * n_embd * n_tok);
    memcpy(k_cache, K->data, kvb);
    memcpy(v_cache, V->data, kvb);
    scratch_free(s);
}

// ----------------------------------------------------------------------------------------------
// model level: GGUF through the reference's own reader (ggml.c:18083-18330), graphs as llama.cpp builds them
// ----------------------------------------------------------------------------------------------

typedef struct {
    struct gguf_context * gguf;
    struct ggml_context * wctx;   // weights
    struct ggml_context * kctx;   // kv cache
    struct ggml_tensor  * k, * v;
    int is_falcon;
    int n_layer, n_embd, n_head, n_head_kv, n_ff, n_vocab, n_ctx_train, n_ctx;
    float eps, freq_base, freq_scale;
    void * cbuf; size_t cbuf_size;
} ref_model;

static uint32_t kv_u32(struct gguf_context * g, const char * arch, const char * key, uint32_t dflt) {
    char full[256];
    snprintf(full, sizeof full, "%s.%s", arch, key);
    int id = gguf_find_key(g, full);
    return id < 0 ? dflt : gguf_get_val_u32(g, id);
}

static float kv_f32(struct gguf_context * g, const char * arch, const char * key, float dflt) {
    char full[256];
    snprintf(full, sizeof full, "%s.%s", arch, key);
    int id = gguf_find_key(g, full);
    return id < 0 ? dflt : gguf_get_val_f32(g, id);
}

REF_API void * ref_model_load(const char * path, int n_ctx) {
    ref_init_once();
    ref_model * m = calloc(1, sizeof *m);
    struct gguf_init_params gp = { false, &m->wctx };
    m->gguf = gguf_init_from_file(path, gp);
    if (!m->gguf) { free(m); return NULL; }
    const char * arch = gguf_get_val_str(m->gguf, gguf_find_key(m->gguf, "general.architecture"));
    m->is_falcon   = strcmp(arch, "falcon") == 0;
    m->n_layer     = kv_u32(m->gguf, arch, "block_count", 0);
    m->n_embd      = kv_u32(m->gguf, arch, "embedding_length", 0);
    m->n_head      = kv_u32(m->gguf, arch, "attention.head_count", 0);
    m->n_head_kv   = kv_u32(m->gguf, arch, "attention.head_count_kv", m->n_head);
    m->n_ff        = kv_u32(m->gguf, arch, "feed_forward_length", 0);
    m->n_ctx_train = kv_u32(m->gguf, arch, "context_length", 0);
    m->eps         = kv_f32(m->gguf, arch, m->is_falcon ? "attention.layer_norm_epsilon" : "attention.layer_norm_rms_epsilon", 0.0f);
    m->freq_base   = kv_f32(m->gguf, arch, "rope.freq_base", 10000.0f);
    m->freq_scale  = 1.0f;
    m->n_ctx       = n_ctx;
    struct ggml_tensor * te = ggml_get_tensor(m->wctx, "token_embd.weight");
    m->n_vocab = (int) te->ne[1];

    const int64_t n_el = (int64_t) m->n_layer * n_ctx * (m->n_embd / m->n_head) * m->n_head_kv;
    struct ggml_init_params ip = { (size_t) 4 * n_el + (1u << 20), NULL, false };
    m->kctx = ggml_init(ip);
    m->k = ggml_new_tensor_1d(m->kctx, GGML_TYPE_F16, n_el);   // llama.cpp:1529-1560
    m->v = ggml_new_tensor_1d(m->kctx, GGML_TYPE_F16, n_el);
    memset(m->k->data, 0, (size_t) 2 * n_el);
    memset(m->v->data, 0, (size_t) 2 * n_el);
    m->cbuf_size = 0; m->cbuf = NULL;
    return m;
}

REF_API void ref_model_free(void * h) {
    ref_model * m = h;
    ggml_free(m->kctx); ggml_free(m->wctx); gguf_free(m->gguf); free(m->cbuf); free(m);
}

REF_API int ref_model_info(void * h, int what) {
    ref_model * m = h;
    switch (what) {
        case 0: return m->n_vocab; case 1: return m->n_embd; case 2: return m->n_layer; case 3: return m->n_head;
        case 4: return m->n_head_kv; case 5: return m->n_ff; case 6: return m->is_falcon; case 7: return m->n_ctx;
    }
    return -1;
}

static struct ggml_tensor * W(ref_model * m, int il, const char * name, int required) {
    char full[128];
    if (il >= 0) snprintf(full, sizeof full, "blk.%d.%s", il, name); else snprintf(full, sizeof full, "%s", name);
    struct ggml_tensor * t = ggml_get_tensor(m->wctx, full);
    if (!t && required) { fprintf(stderr, "ref_harness: missing tensor %s\n", full); abort(); }
    return t;
}

static struct ggml_tensor * norm(struct ggml_context * c, struct ggml_tensor * x, struct ggml_tensor * w, struct ggml_tensor * b, int rms, float eps) {
    struct ggml_tensor * cur = rms ? ggml_rms_norm(c, x, eps) : ggml_norm(c, x, eps);
    if (w) cur = ggml_mul(c, cur, w);
    if (b) cur = ggml_add(c, cur, b);
    return cur;
}

// logits_out: [n_vocab, n_tok] for every token.  mask: [n_kv, n_tok] additive 0/-inf.  kv_head: first cell written.
REF_API int ref_model_decode(void * h, const int32_t * tokens, const int32_t * pos, int n_tok, int kv_head, int n_kv,
                             const float * mask, float * logits_out, int n_threads) {
    ref_model * m = h;
    const int hd = m->n_embd / m->n_head, n_embd = m->n_embd, n_head = m->n_head, n_head_kv = m->n_head_kv;
    const int n_embd_gqa = hd * n_head_kv, n_ctx = m->n_ctx;
    const size_t need = (size_t) 4 * n_tok * ((size_t) (6 * n_embd + 3 * m->n_ff + 2 * (size_t) n_kv * n_head) * (m->n_layer + 1) + 2 * (size_t) m->n_vocab)
                      + (size_t) 4 * n_kv * n_tok + (256u << 20);
    if (need > m->cbuf_size) { free(m->cbuf); m->cbuf = malloc(need); m->cbuf_size = need; }
    struct ggml_init_params ip = { m->cbuf_size, m->cbuf, false };
    struct ggml_context * c = ggml_init(ip);
    struct ggml_cgraph * gf = ggml_new_graph_custom(c, 8192, false);

    struct ggml_tensor * inp_tokens = ggml_new_tensor_1d(c, GGML_TYPE_I32, n_tok); memcpy(inp_tokens->data, tokens, (size_t) 4 * n_tok);
    struct ggml_tensor * inp_pos    = ggml_new_tensor_1d(c, GGML_TYPE_I32, n_tok); memcpy(inp_pos->data, pos, (size_t) 4 * n_tok);
    struct ggml_tensor * KQ_scale   = ggml_new_tensor_1d(c, GGML_TYPE_F32, 1);     ggml_set_f32(KQ_scale, 1.0f / sqrtf((float) hd));
    struct ggml_tensor * KQ_mask    = ggml_new_tensor_3d(c, GGML_TYPE_F32, n_kv, n_tok, 1); memcpy(KQ_mask->data, mask, (size_t) 4 * n_kv * n_tok);

    struct ggml_tensor * inpL = ggml_get_rows(c, W(m, -1, "token_embd.weight", 1), inp_tokens);
    const size_t es = ggml_element_size(m->k);
    const int rope_mode = m->is_falcon ? 2 : 0;

    for (int il = 0; il < m->n_layer; ++il) {
        struct ggml_tensor * cur, * Qcur, * Kcur, * Vcur, * attn_norm = NULL;
        if (!m->is_falcon) {
            cur  = norm(c, inpL, W(m, il, "attn_norm.weight", 1), NULL, 1, m->eps);
            Qcur = ggml_mul_mat(c, W(m, il, "attn_q.weight", 1), cur);
            Kcur = ggml_mul_mat(c, W(m, il, "attn_k.weight", 1), cur);
            Vcur = ggml_mul_mat(c, W(m, il, "attn_v.weight", 1), cur);
            Qcur = ggml_reshape_3d(c, Qcur, hd, n_head, n_tok);
            Kcur = ggml_reshape_3d(c, Kcur, hd, n_head_kv, n_tok);
        } else {
            attn_norm = norm(c, inpL, W(m, il, "attn_norm.weight", 1), W(m, il, "attn_norm.bias", 1), 0, m->eps);
            struct ggml_tensor * n2w = W(m, il, "attn_norm_2.weight", 0);
            cur = n2w ? norm(c, inpL, n2w, W(m, il, "attn_norm_2.bias", 1), 0, m->eps) : attn_norm;
            cur = ggml_mul_mat(c, W(m, il, "attn_qkv.weight", 1), cur);
            Qcur = ggml_cont(c, ggml_view_2d(c, cur, n_embd,     n_tok, cur->nb[1], 0));
            Kcur = ggml_cont(c, ggml_view_2d(c, cur, n_embd_gqa, n_tok, cur->nb[1], sizeof(float) * n_embd));
            Vcur = ggml_cont(c, ggml_view_2d(c, cur, n_embd_gqa, n_tok, cur->nb[1], sizeof(float) * (n_embd + n_embd_gqa)));
            Qcur = ggml_reshape_3d(c, Qcur, hd, n_head, n_tok);
            Kcur = ggml_reshape_3d(c, Kcur, hd, n_head_kv, n_tok);
        }
        Qcur = ggml_rope_custom(c, Qcur, inp_pos, hd, rope_mode, 0, m->n_ctx_train, m->freq_base, m->freq_scale, 0.0f, 1.0f, 32.0f, 1.0f);
        Kcur = ggml_rope_custom(c, Kcur, inp_pos, hd, rope_mode, 0, m->n_ctx_train, m->freq_base, m->freq_scale, 0.0f, 1.0f, 32.0f, 1.0f);

        // kv store
        struct ggml_tensor * v_cur_t = ggml_transpose(c, ggml_reshape_2d(c, Vcur, n_embd_gqa, n_tok));
        struct ggml_tensor * k_view  = ggml_view_1d(c, m->k, (int64_t) n_tok * n_embd_gqa, (es * n_embd_gqa) * ((size_t) il * n_ctx + kv_head));
        struct ggml_tensor * v_view  = ggml_view_2d(c, m->v, n_tok, n_embd_gqa, (size_t) n_ctx * es, ((size_t) il * n_ctx) * es * n_embd_gqa + kv_head * es);
        ggml_build_forward_expand(gf, ggml_cpy(c, Kcur, k_view));
        ggml_build_forward_expand(gf, ggml_cpy(c, v_cur_t, v_view));

        // kqv
        struct ggml_tensor * q  = ggml_permute(c, Qcur, 0, 2, 1, 3);
        struct ggml_tensor * k  = ggml_view_3d(c, m->k, hd, n_kv, n_head_kv, es * n_embd_gqa, es * hd, es * n_embd_gqa * n_ctx * il);
        struct ggml_tensor * kq = ggml_mul_mat(c, k, q);
        kq = ggml_scale(c, kq, KQ_scale);
        kq = ggml_add(c, kq, KQ_mask);
        kq = ggml_soft_max(c, kq);
        struct ggml_tensor * v   = ggml_view_3d(c, m->v, n_kv, hd, n_head_kv, es * n_ctx, es * n_ctx * hd, es * n_ctx * n_embd_gqa * il);
        struct ggml_tensor * kqv = ggml_mul_mat(c, v, kq);
        cur = ggml_cont_2d(c, ggml_permute(c, kqv, 0, 2, 1, 3), n_embd, n_tok);
        cur = ggml_mul_mat(c, W(m, il, "attn_output.weight", 1), cur);

        if (!m->is_falcon) {
            struct ggml_tensor * ffn_inp = ggml_add(c, cur, inpL);
            cur = norm(c, ffn_inp, W(m, il, "ffn_norm.weight", 1), NULL, 1, m->eps);
            struct ggml_tensor * up   = ggml_mul_mat(c, W(m, il, "ffn_up.weight", 1), cur);
            struct ggml_tensor * gate = ggml_mul_mat(c, W(m, il, "ffn_gate.weight", 1), cur);
            gate = ggml_silu(c, gate);
            cur  = ggml_mul(c, gate, up);
            cur  = ggml_mul_mat(c, W(m, il, "ffn_down.weight", 1), cur);
            cur  = ggml_add(c, cur, ffn_inp);
        } else {
            struct ggml_tensor * ffn_inp = cur;
            cur = ggml_mul_mat(c, W(m, il, "ffn_up.weight", 1), attn_norm);
            cur = ggml_gelu(c, cur);
            cur = ggml_mul_mat(c, W(m, il, "ffn_down.weight", 1), cur);
            cur = ggml_add(c, cur, ffn_inp);
            cur = ggml_add(c, cur, inpL);
        }
        inpL = cur;
    }

    struct ggml_tensor * cur = m->is_falcon
        ? norm(c, inpL, W(m, -1, "output_norm.weight", 1), W(m, -1, "output_norm.bias", 1), 0, m->eps)
        : norm(c, inpL, W(m, -1, "output_norm.weight", 1), NULL, 1, m->eps);
    cur = ggml_mul_mat(c, W(m, -1, "output.weight", 1), cur);
    ggml_build_forward_expand(gf, cur);
    ggml_graph_compute_with_ctx(c, gf, n_threads);
    memcpy(logits_out, cur->data, (size_t) 4 * m->n_vocab * n_tok);
    ggml_free(c);
    return 0;
}

// raw access to the reference-layout caches (for cell-level checks)
REF_API void * ref_model_kcache(void * h) { return ((ref_model *) h)->k->data; }
REF_API void * ref_model_vcache(void * h) { return ((ref_model *) h)->v->data; }

REF_API int64_t ref_time_us(void) { ggml_time_init(); return ggml_time_us(); }
