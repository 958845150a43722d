// Command-line driver with the flags and the closing report of the reference's examples/speculative (speculative.cpp:30-120 for
// the options that matter on this path, :688-718 for the report and results.csv), on top of include/llama.h + include/pipeinfer.h:
//
//   speculative_amd -m target.gguf -md draft.gguf [-p "text" | --prompt-tokens 1,2,3] [-n 128] [--draft 3] [-np 3] [-pa 0.1]
//                   [-ps 0.9] [-pr 0.0] [-pd 0.0] [-c 1024] [-b 512] [--ignore-eos] [--target-only] [--results results.csv]
//
// It is a caller of the library, not part of it; multi-GPU stages are launched through bench.py / pipeline.py.
#include "llama.h"
#include "pipeinfer.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void usage(const char * a0) {
    fprintf(stderr, "usage: %s -m target.gguf -md draft.gguf (-p text | --prompt-tokens 1,2,3) [-n N] [--draft N] [-np N] [-pa F] [-ps F] [-pr F] [-pd F] "
                    "[-c N] [-b N] [--ignore-eos] [--target-only] [--results file.csv]\n", a0);
}

int main(int argc, char ** argv) {
    std::string path_tgt, path_dft, prompt_text, results_path;
    std::vector<llama_token> prompt;
    pipeinfer_spec_params sp = pipeinfer_spec_default_params();          // -n 128 --draft 3 -np 3 -pa 0.1 -ps 0.9 (dolphin_tinyllama_8_nodes.sh)
    int n_ctx = 1024, n_batch = 512;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { usage(argv[0]); exit(2); } return argv[++i]; };
        if      (a == "-m" || a == "--model")            path_tgt = next();
        else if (a == "-md" || a == "--model-draft")     path_dft = next();
        else if (a == "-p" || a == "--prompt")           prompt_text = next();
        else if (a == "--prompt-tokens")                 { const char * s = next(); while (*s) { prompt.push_back((llama_token) strtol(s, (char **) &s, 10)); if (*s == ',') s++; } }
        else if (a == "-n" || a == "--n-predict")        sp.n_predict = atoi(next());
        else if (a == "--draft")                         sp.n_draft = atoi(next());
        else if (a == "-np" || a == "--parallel")        sp.n_parallel = atoi(next());
        else if (a == "-pa" || a == "--p-accept")        sp.p_accept = (float) atof(next());
        else if (a == "-ps" || a == "--p-split")         sp.p_split = (float) atof(next());
        else if (a == "-pr" || a == "--p-recovery")      sp.p_recovery = (float) atof(next());
        else if (a == "-pd" || a == "--p-decay")         sp.p_decay = (float) atof(next());
        else if (a == "-c" || a == "--ctx-size")         n_ctx = atoi(next());
        else if (a == "-b" || a == "--batch-size")       n_batch = atoi(next());
        else if (a == "--ignore-eos")                    sp.ignore_eos = 1;
        else if (a == "--target-only")                   sp.mode = 2;
        else if (a == "--orig" || a == "--synchronous")  sp.mode = 1;       // the schedule of examples/speculative_orig: draft a tree, verify it in one target batch
        else if (a == "--results")                       results_path = next();
        else { fprintf(stderr, "unknown argument: %s\n", a.c_str()); usage(argv[0]); return 2; }
    }
    if (path_tgt.empty() || path_dft.empty() || (prompt_text.empty() && prompt.empty())) { usage(argv[0]); return 2; }

    llama_backend_init(false);
    llama_model * mt = llama_load_model_from_file(path_tgt.c_str(), llama_model_default_params());
    llama_model * md = llama_load_model_from_file(path_dft.c_str(), llama_model_default_params());
    if (!mt || !md) { fprintf(stderr, "%s: unable to load model\n", argv[0]); return 1; }
    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = n_ctx; cp.n_batch = n_batch; cp.logits_all = true;          // speculative.cpp:115
    llama_context * ct = llama_new_context_with_model(mt, cp);
    llama_context * cd = llama_new_context_with_model(md, cp);
    if (!ct || !cd) { fprintf(stderr, "%s: failed to create the contexts\n", argv[0]); return 1; }

    if (prompt.empty()) {                                                  // llama_tokenize with add_bos, like ::llama_tokenize(ctx, prompt, true)
        prompt.resize(prompt_text.size() + 8);
        const int n = llama_tokenize(mt, prompt_text.c_str(), (int) prompt_text.size(), prompt.data(), (int) prompt.size(), true, true);
        if (n < 0) { fprintf(stderr, "%s: tokenisation failed\n", argv[0]); return 1; }
        prompt.resize(n);
    }
    if ((int) prompt.size() < 2 || (int) prompt.size() > n_ctx - 4) { fprintf(stderr, "%s: the prompt needs between 2 and n_ctx - 4 tokens\n", argv[0]); return 1; }

    std::vector<llama_token> out(sp.n_predict + 16);
    pipeinfer_spec_result r;
    const int n = pipeinfer_speculative(ct, cd, &sp, prompt.data(), (int32_t) prompt.size(), out.data(), (int32_t) out.size(), &r);
    if (n < 0) { fprintf(stderr, "%s: generation failed (%d)\n", argv[0], n); return 1; }

    char piece[64];
    for (int i = 0; i < n; i++) {
        const int len = llama_token_to_piece(mt, out[i], piece, (int) sizeof piece);
        if (len > 0) fwrite(piece, 1, (size_t) len, stdout);
    }
    printf("\n\ntokens:");
    for (int i = 0; i < n; i++) printf(" %d", out[i]);
    printf("\n\n");

    if (!results_path.empty()) {                                           // results.csv columns of the reference: encode t/s, decode t/s, inter-token latency, ttft
        FILE * f = fopen(results_path.c_str(), "a");
        if (f) { fprintf(f, "%g,%g,%g,%g\n", r.encode_tps, r.decode_tps, r.itl_s, r.ttft_s); fclose(f); }
    }
    printf("OVERALL SYSTEM MEASUREMENTS:\n");
    printf("encoded %4d tokens in %8.3f seconds, speed: %8.3f t/s\n", r.n_input, r.t_enc_s, r.encode_tps);
    printf("decoded %4d tokens in %8.3f seconds, speed: %8.3f t/s\n", r.n_predict, r.t_dec_s, r.decode_tps);
    printf("Average inter-token latency: %f seconds\n", r.itl_s);
    printf("Time-to-first-token: %f seconds\n", r.ttft_s);
    printf("\nTARGET PIPELINE MEASUREMENTS\n");
    printf("n_draft   = %d\n", sp.n_draft);
    printf("n_predict = %d\n", r.n_predict);
    printf("n_drafted = %d\n", r.n_drafted);
    printf("n_accept  = %d\n", r.n_accept);
    printf("accept    = %.3f%%\n", r.n_drafted > 0 ? 100.0f * r.n_accept / r.n_drafted : 0.0f);
    printf("\nspeculative runs %d (cancelled %d), drafted tokens verified %d, of which accepted %d\n", r.n_runs_spec, r.n_runs_canceled, r.n_drafted_tokens, r.n_accept_clean);

    llama_free(ct); llama_free(cd); llama_free_model(mt); llama_free_model(md);
    llama_backend_free();
    return 0;
}
