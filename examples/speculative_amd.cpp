// Command-line driver with the flags and the closing report of the reference's examples/speculative (speculative.cpp:30-120 for
// the options that matter on this path, :688-718 for the report and results.csv), on top of include/llama.h + include/pipeinfer.h:
//
//   speculative_amd -m target.gguf -md draft.gguf [-p "text" | --prompt-tokens 1,2,3] [-n 128] [--draft 3] [-np 3] [-pa 0.1]
//                   [-ps 0.9] [-pr 0.0] [-pd 0.0] [-c 1024] [-b 512] [--ignore-eos] [--target-only] [--results results.csv]
//                   [--mpi-layer-split w1,w2,...[/1.0]] [--rank R --world N --job NAME --device D]
//
// It is a caller of the library, not part of it.  With --mpi-layer-split the same command line is started once per GPU, the way the
// reference is started once per MPI rank (speculative.cpp:116-139, common.cpp:574-596): rank 0 runs the driver and the draft model,
// ranks 1..N-1 serve the target's layer ranges (weights of the FIRST list, ggml_mpi_split_range rule) through pipeinfer_transport_*.
// The reference's rank 0 also holds target layers and its last rank the draft; here the draft sits with the driver, so the first list
// names the N-1 stage ranks and the second list, if given, must be a single entry.  Rank / world come from --rank/--world or the
// environment of the launcher (PIPEINFER_RANK/WORLD, OMPI_COMM_WORLD_RANK/SIZE, PMI_RANK/SIZE, RANK/WORLD_SIZE).
#include "llama.h"
#include "pipeinfer.h"
#include "pipeinfer_ops.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void usage(const char * a0) {
    fprintf(stderr, "usage: %s -m target.gguf -md draft.gguf (-p text | --prompt-tokens 1,2,3) [-n N] [--draft N] [-np N] [-pa F] [-ps F] [-pr F] [-pd F] "
                    "[-c N] [-b N] [--ignore-eos] [--target-only] [--orig] [--results file.csv] [--seg-layers N] [--coalesce N] [--mpi-layer-split w1,w2,...[/1.0] [--rank R] [--world N] [--job NAME] [--device D]]\n", a0);
}

static int env_int(const char * const * names, int fallback) {
    for (; *names; names++) { const char * v = getenv(*names); if (v && *v) return atoi(v); }
    return fallback;
}

int main(int argc, char ** argv) {
    std::string path_tgt, path_dft, prompt_text, results_path;
    std::vector<llama_token> prompt;
    pipeinfer_spec_params sp = pipeinfer_spec_default_params();          // -n 128 --draft 3 -np 3 -pa 0.1 -ps 0.9 (dolphin_tinyllama_8_nodes.sh)
    int n_ctx = 1024, n_batch = 512;
    std::vector<float> split;                                             // --mpi-layer-split: weights of the target stages (ranks 1..)
    static const char * const rank_env[]  = { "PIPEINFER_RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "RANK", NULL };
    static const char * const world_env[] = { "PIPEINFER_WORLD", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "WORLD_SIZE", NULL };
    static const char * const local_env[] = { "PIPEINFER_DEVICE", "OMPI_COMM_WORLD_LOCAL_RANK", "LOCAL_RANK", NULL };
    int rank = env_int(rank_env, 0), world = env_int(world_env, 1), device = env_int(local_env, -1);
    std::string job = getenv("PIPEINFER_JOB") ? getenv("PIPEINFER_JOB") : "";
    int seg_layers = -1, coalesce = 16;                                    // -1: a quarter of the target's layers
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
        else if (a == "--mpi-layer-split") {                                // common.cpp:574-596: "w,w,w/w": lists separated by '/', weights by ','
            const char * v = next();
            int list = 0, n_second = 0;
            while (*v) {
                char * end = NULL;
                const float w = strtof(v, &end);
                if (end == v) { fprintf(stderr, "%s: bad --mpi-layer-split\n", argv[0]); return 2; }
                if (list == 0) split.push_back(w); else n_second++;
                v = end;
                if (*v == '/') { list++; v++; } else if (*v == ',') v++;
            }
            if (list > 1 || (list == 1 && n_second != 1)) { fprintf(stderr, "%s: --mpi-layer-split: the draft model is not split here (second list: one entry)\n", argv[0]); return 2; }
        }
        else if (a == "--seg-layers")                    seg_layers = atoi(next());   // one GPU: layers per segment of the target's graphs (0: one graph, no feeder)
        else if (a == "--coalesce")                      coalesce = atoi(next());     // one GPU: tokens the target's feeder may carry in one pass (0: one pass per run)
        else if (a == "--rank")                          rank = atoi(next());
        else if (a == "--world")                         world = atoi(next());
        else if (a == "--job")                           job = next();
        else if (a == "--device")                        device = atoi(next());
        else { fprintf(stderr, "unknown argument: %s\n", a.c_str()); usage(argv[0]); return 2; }
    }
    if (path_tgt.empty() || path_dft.empty() || (prompt_text.empty() && prompt.empty())) { usage(argv[0]); return 2; }

    const bool pipelined = !split.empty();
    if (pipelined) {
        if (world != (int) split.size() + 1) { fprintf(stderr, "%s: --mpi-layer-split names %d target stage(s): start %d processes (rank 0 = driver + draft), not %d\n", argv[0], (int) split.size(), (int) split.size() + 1, world); return 2; }
        if (rank < 0 || rank >= world) { fprintf(stderr, "%s: rank %d outside 0..%d\n", argv[0], rank, world - 1); return 2; }
        if (job.empty()) { fprintf(stderr, "%s: --mpi-layer-split needs --job NAME (or PIPEINFER_JOB), the same on every rank\n", argv[0]); return 2; }
    } else if (world > 1 && rank > 0) {
        return 0;                                                          // started under a multi-rank launcher without a split: one rank does the work
    }
    const int n_dev = pi_device_count();
    if (n_dev <= 0) { fprintf(stderr, "%s: no GPU\n", argv[0]); return 1; }
    if (device < 0) device = pipelined ? rank : 0;
    device %= n_dev;

    llama_backend_init(false);
    llama_model_params mp = llama_model_default_params();
    mp.main_gpu = device;
    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = n_ctx; cp.n_batch = n_batch; cp.logits_all = true;          // speculative.cpp:115
    llama_model * mt = llama_load_model_from_file(path_tgt.c_str(), mp);
    if (!mt) { fprintf(stderr, "%s: unable to load model\n", argv[0]); return 1; }
    llama_context * ct = llama_new_context_with_model(mt, cp);
    if (!ct) { fprintf(stderr, "%s: failed to create the contexts\n", argv[0]); return 1; }

    pipeinfer_transport * tp = NULL;
    if (pipelined) {
        pipeinfer_transport_params tpp;
        memset(&tpp, 0, sizeof tpp);
        tpp.name = job.c_str(); tpp.rank = rank; tpp.world = world; tpp.n_embd = llama_n_embd(mt); tpp.n_vocab = llama_n_vocab(mt);
        tpp.n_batch = n_batch; tpp.n_ctx = n_ctx; tpp.device = device; tpp.timeout_ms = 0;
        if (rank > 0) {                                                    // a target stage: llama_split_layers_weighted (speculative.cpp:137) + the worker loop of llama_decode (llama.cpp:5588-5760)
            int32_t l0 = 0, l1 = 0;
            if (pipeinfer_context_set_stage_weighted(ct, split.data(), world - 1, rank - 1, &l0, &l1) != 0 ||
                pipeinfer_context_set_segments(ct, (l1 - l0 + 3) / 4 > 0 ? (l1 - l0 + 3) / 4 : 1) != 0 || pipeinfer_context_prepare(ct) != 0) {
                fprintf(stderr, "%s: rank %d: %s\n", argv[0], rank, pipeinfer_context_error(ct)); return 1;
            }
            fprintf(stderr, "%s: rank %d serves layers [%d, %d) on GPU %d\n", argv[0], rank, l0, l1, device);
            tp = pipeinfer_transport_open(&tpp);
            if (!tp) { fprintf(stderr, "%s: rank %d: pipeinfer_transport_open failed\n", argv[0], rank); return 1; }
            pipeinfer_transport_stats st;
            const int rc = pipeinfer_transport_serve(tp, ct, NULL, &st);
            if (rc != 0) fprintf(stderr, "%s: rank %d: %s\n", argv[0], rank, pipeinfer_transport_error(tp));
            else fprintf(stderr, "%s: rank %d: %llu runs, %llu skipped, %llu cut short\n", argv[0], rank, (unsigned long long) st.runs, (unsigned long long) st.skipped, (unsigned long long) st.cut_short);
            pipeinfer_transport_close(tp);
            llama_free(ct); llama_free_model(mt); llama_backend_free();
            return rc != 0;
        }
        tp = pipeinfer_transport_open(&tpp);
        if (!tp) { fprintf(stderr, "%s: pipeinfer_transport_open failed\n", argv[0]); return 1; }
        if (pipeinfer_transport_attach_head(tp, ct) != 0) { fprintf(stderr, "%s: %s\n", argv[0], pipeinfer_transport_error(tp)); return 1; }
        fprintf(stderr, "%s: %d target stage(s), activations through %s\n", argv[0], world - 1, pipeinfer_transport_device_mode(tp) ? "peer stores into the next stage's HBM ring" : "host rings");
    }
    llama_model * md = llama_load_model_from_file(path_dft.c_str(), mp);
    if (!md) { fprintf(stderr, "%s: unable to load model\n", argv[0]); return 1; }
    llama_context * cd = llama_new_context_with_model(md, cp);
    if (!cd) { fprintf(stderr, "%s: failed to create the contexts\n", argv[0]); return 1; }
    if (!pipelined && sp.mode == 0) {
        // both models on ONE GPU (BASELINE configs[1]): the target's stage is launched segment by segment by a feeder thread that lets the (priority) draft go first
        // between segments and carries the runs waiting in its queue in one forward pass; none of this changes a token
        if (seg_layers < 0) seg_layers = pipeinfer_model_n_layer(mt) / 4 > 0 ? pipeinfer_model_n_layer(mt) / 4 : 1;
        if (seg_layers > 0) {
            if (pipeinfer_context_set_segments(ct, seg_layers) != 0 || pipeinfer_context_set_priority(cd, 1) != 0 || pipeinfer_context_prepare(ct) != 0 || pipeinfer_context_prepare(cd) != 0 ||
                (coalesce > 1 && pipeinfer_context_set_coalescing(ct, coalesce) != 0)) {
                fprintf(stderr, "%s: %s / %s\n", argv[0], pipeinfer_context_error(ct), pipeinfer_context_error(cd)); return 1;
            }
        }
    }

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
    if (tp) { pipeinfer_transport_shutdown(tp); }
    if (n < 0) { fprintf(stderr, "%s: generation failed (%d)\n", argv[0], n); if (tp) pipeinfer_transport_close(tp); return 1; }

    std::vector<char> piece(64);
    for (int i = 0; i < n; i++) {
        int len = llama_token_to_piece(mt, out[i], piece.data(), (int) piece.size());
        if (len < 0) { piece.resize((size_t) -len); len = llama_token_to_piece(mt, out[i], piece.data(), (int) piece.size()); }   // -len = the size the piece needs (common.cpp:1112-1127)
        if (len > 0) fwrite(piece.data(), 1, (size_t) len, stdout);
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

    if (tp) pipeinfer_transport_close(tp);
    llama_free(ct); llama_free(cd); llama_free_model(mt); llama_free_model(md);
    llama_backend_free();
    return 0;
}
