// A llama.h caller in the style of the reference's examples/speculative, reduced to what a parity check needs: token ids in,
// token ids out.  It includes nothing but include/llama.h (+ pipeinfer.h for the packaged driver) and links libpipeinfer_amd.so:
//
//   g++ -std=c++11 -O2 -Iinclude examples/speculative_min.cpp -Lpipeinfer_amd -lpipeinfer_amd -Wl,-rpath,$PWD/pipeinfer_amd -o speculative_min
//   ./speculative_min target.gguf draft.gguf N_PREDICT tok tok tok ...
//
// Part 1 drives the target alone through the plain llama.cpp calls the reference's drivers use (llama_batch_init, llama_decode,
// llama_get_logits_ith, llama_sample_token_greedy: speculative.cpp:223-246, sampling.cpp); part 2 runs PipeInfer's asynchronous
// speculative loop (pipeinfer_speculative = main() of speculative.cpp) with a forced speculation iteration per token.  Both
// lines must be the same tokens (greedy verification is lossless) — tests/test_gpu_engine.py compares them with the golden
// continuation of the reference.
#include "llama.h"
#include "pipeinfer.h"

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <vector>

// the boundary is a struct layout as much as a set of names: llama_batch is passed BY VALUE (llama.h:142-162; 80 bytes on x86-64 = int32 + pad + 6 pointers + 5 int32,
// SURVEY 8b), llama_token_data is 12 bytes (llama.h:116-120) — a caller compiled against the reference's header must find the same offsets here
static_assert(sizeof(llama_batch) == 80, "llama_batch layout differs from the reference (llama.h:142-162)");
static_assert(offsetof(llama_batch, n_tokens) == 0 && offsetof(llama_batch, token) == 8 && offsetof(llama_batch, embd) == 16 && offsetof(llama_batch, pos) == 24 &&
              offsetof(llama_batch, n_seq_id) == 32 && offsetof(llama_batch, seq_id) == 40 && offsetof(llama_batch, logits) == 48 && offsetof(llama_batch, all_pos_0) == 56 &&
              offsetof(llama_batch, all_pos_1) == 60 && offsetof(llama_batch, all_seq_id) == 64 && offsetof(llama_batch, batch_id) == 68 && offsetof(llama_batch, max_n_seq) == 72,
              "llama_batch field offsets differ from the reference (llama.h:142-162)");
static_assert(sizeof(llama_token_data) == 12 && offsetof(llama_token_data, logit) == 4 && offsetof(llama_token_data, p) == 8, "llama_token_data layout (llama.h:116-120)");

static void batch_add(llama_batch & b, llama_token id, llama_pos pos, llama_seq_id seq, bool logits) {
    b.token[b.n_tokens] = id; b.pos[b.n_tokens] = pos; b.n_seq_id[b.n_tokens] = 1; b.seq_id[b.n_tokens][0] = seq; b.logits[b.n_tokens] = logits;
    b.n_tokens++;
}

static llama_token greedy(llama_context * ctx, int i_batch) {
    const int n_vocab = llama_n_vocab(llama_get_model(ctx));
    const float * logits = llama_get_logits_ith(ctx, i_batch);
    std::vector<llama_token_data> cur(n_vocab);
    for (llama_token t = 0; t < n_vocab; t++) cur[t] = llama_token_data{ t, logits[t], 0.0f };
    llama_token_data_array arr = { cur.data(), cur.size(), false };
    return llama_sample_token_greedy(ctx, &arr);
}

int main(int argc, char ** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s target.gguf draft.gguf n_predict token...\n", argv[0]); return 2; }
    const int n_predict = atoi(argv[3]);
    std::vector<llama_token> prompt;
    for (int i = 4; i < argc; i++) prompt.push_back(atoi(argv[i]));

    llama_backend_init(false);
    llama_model * mt = llama_load_model_from_file(argv[1], llama_model_default_params());
    llama_model * md = llama_load_model_from_file(argv[2], llama_model_default_params());
    if (!mt || !md) { fprintf(stderr, "model load failed\n"); return 1; }
    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = 256; cp.n_batch = 64; cp.logits_all = true;                  // speculative.cpp:115
    llama_context * ct = llama_new_context_with_model(mt, cp);
    llama_context * cd = llama_new_context_with_model(md, cp);
    if (!ct || !cd) { fprintf(stderr, "context creation failed\n"); return 1; }

    // ---- part 1: target only, plain llama.cpp calls ----
    llama_batch batch = llama_batch_init(64, 0, 1);
    for (size_t i = 0; i < prompt.size(); i++) batch_add(batch, prompt[i], (llama_pos) i, 0, i + 1 == prompt.size());
    if (llama_decode(ct, batch) != 0) { fprintf(stderr, "llama_decode failed\n"); return 1; }
    llama_token id = greedy(ct, batch.n_tokens - 1);
    printf("plain:");
    for (int i = 0; i < n_predict; i++) {
        printf(" %d", id);
        batch.n_tokens = 0;
        batch_add(batch, id, (llama_pos) (prompt.size() + i), 0, true);
        if (llama_decode(ct, batch) != 0) { fprintf(stderr, "llama_decode failed\n"); return 1; }
        id = greedy(ct, 0);
    }
    printf("\n");
    llama_batch_free(&batch);

    // ---- part 2: the asynchronous speculative driver ----
    llama_kv_cache_clear(ct);
    pipeinfer_spec_params sp = pipeinfer_spec_default_params();
    sp.n_predict = n_predict - 1; sp.n_draft = 3; sp.n_parallel = 3; sp.p_accept = 0.05f; sp.p_split = 0.9f; sp.ignore_eos = 1; sp.min_spec_iters = 1;
    std::vector<llama_token> out(n_predict + 16);
    pipeinfer_spec_result res;
    const int n = pipeinfer_speculative(ct, cd, &sp, prompt.data(), (int32_t) prompt.size(), out.data(), (int32_t) out.size(), &res);
    if (n < 0) { fprintf(stderr, "pipeinfer_speculative failed: %d\n", n); return 1; }
    printf("spec:");
    for (int i = 0; i < n && i < n_predict; i++) printf(" %d", out[i]);
    printf("\nspec_runs %d accepted %d drafted %d\n", res.n_runs_spec, res.n_accept_clean, res.n_drafted_tokens);

    llama_free(ct); llama_free(cd); llama_free_model(mt); llama_free_model(md);
    llama_backend_free();
    return 0;
}
