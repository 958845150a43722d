// The PipeInfer driver as a library call: asynchronous, continuous, tree-structured speculative decoding with early
// cancellation.  This restates the state machine of the reference's examples/speculative/speculative.cpp (main loop
// :316-679, run_speculation_loop :765-844, start_async_spec_run :881-1180, begin_async_run :1182-1275,
// begin_non_spec_run :850-879, check_for_cancel :1277-1359, calc_p_adjust :846-848) on top of the llama.h calls above:
// same bookkeeping of sequence partitions, same accept / cancel rules, same counters.  Differences, all host-side:
//   * the per-sequence-id KV loops (up to 3001 ids x 2 contexts per accepted run) use the closed-form range operations
//     of kv_cells (identical final state);
//   * MPI root synchronisation (llama_sync_token*, llama_swap_comm) disappears: one process is every root;
//   * greedy sampling reads the device-side top-8 + log-sum-exp that the last stage produces beside the logits instead
//     of sorting 32 000 candidates on the host (llama_sample_softmax semantics, ties -> lower id);
//   * repetition penalties / grammar are not applied (the reference benchmark runs --repeat-last-n 0, greedy).
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>

namespace {

struct host_batch {                                       // owns the arrays behind a llama_batch
    std::vector<llama_token> token; std::vector<llama_pos> pos; std::vector<int32_t> n_seq; std::vector<std::vector<llama_seq_id>> seq; std::vector<int8_t> logits;
    std::vector<llama_seq_id *> seq_ptr;
    int32_t batch_id = 0;
    int n_tokens() const { return (int) token.size(); }
    void clear() { token.clear(); pos.clear(); n_seq.clear(); seq.clear(); logits.clear(); }
    void add(llama_token id, llama_pos p, const std::vector<llama_seq_id> & s) { token.push_back(id); pos.push_back(p); seq.push_back(s); logits.push_back(1); }
    llama_batch view() {
        n_seq.resize(seq.size()); seq_ptr.resize(seq.size());
        for (size_t i = 0; i < seq.size(); i++) { n_seq[i] = (int32_t) seq[i].size(); seq_ptr[i] = seq[i].data(); }
        llama_batch b;
        memset(&b, 0, sizeof b);
        b.n_tokens = n_tokens(); b.token = token.data(); b.pos = pos.data(); b.n_seq_id = n_seq.data(); b.seq_id = seq_ptr.data(); b.logits = logits.data();
        b.batch_id = batch_id; b.max_n_seq = 1;
        return b;
    }
};

struct seq_draft {                                        // speculative.cpp:16-28
    bool active = false, drafting = false, skip = false;
    int i_batch_dft = 0;
    std::vector<int> i_batch_tgt;
    std::vector<llama_token> tokens, prefix_tokens;
};

struct async_run {                                        // speculative.cpp:30-43
    ggml_cgraph * ticket = nullptr;
    host_batch batch;
    std::vector<seq_draft> drafts;
    int n_past_tgt = 0, prefix_n_past_tgt = 0, n_past_dft = 0, s_keep = 0, seq_offset = 0, n_past_max = 0;
    bool speculative = false, canceled = false;
};

struct cand { llama_token id; float logit, p; };

struct driver {
    llama_context * tgt, * dft;
    pipeinfer_spec_params P;
    pipeinfer_spec_result R;
    int n_seq_dft, max_seq;
    llama_token eos;
    std::deque<int> free_offsets;
    std::deque<std::shared_ptr<async_run>> tgt_runs, dft_runs;      // front = newest, back = oldest (speculative.cpp:1262, :331)
    std::vector<llama_token> generated;
    std::vector<seq_draft> drafts;
    async_run current;
    int32_t batch_id = 0;
    int n_past_tgt = 0, n_past_dft = 0, spec_past_tgt = 0, spec_past_dft = 0, n_rejected = 0;
    bool first_run = true;
    llama_token id = 0;
    host_batch batch_tgt, batch_dft;
    bool failed = false;                                      // a decode could not be started: drain what is in flight and return an error

    // ---- sampling -----------------------------------------------------------------------------------------------------------------
    // top-8 candidates of row idx with probabilities over the full vocabulary (llama_sampling_sample with temp < 0:
    // sampling.cpp:179-183 -> llama_sample_softmax).  --ignore-eos = logit_bias[eos] = -inf (common.cpp:1061-1063).
    void top_candidates(llama_context * ctx, int idx, cand * out, int & n) {
        llama_token ids[8]; float lg[8]; float sum;
        n = 0;
        if (pipeinfer_get_topk(ctx, idx, ids, lg, &sum) != 0) return;
        float mx = lg[0];
        if (P.ignore_eos) {
            const float l_eos = llama_get_logits_ith(ctx, idx)[eos];
            if (l_eos != -INFINITY) {
                sum -= expf(l_eos - mx);
                llama_get_logits_ith(ctx, idx)[eos] = -INFINITY;
            }
            int k = 0;
            for (int i = 0; i < 8; i++) if (ids[i] != eos) { ids[k] = ids[i]; lg[k] = lg[i]; k++; }
            const int n_keep = k;
            if (n_keep > 0 && lg[0] != mx) { sum *= expf(mx - lg[0]); mx = lg[0]; }
            for (int i = 0; i < n_keep; i++) out[i] = cand{ ids[i], lg[i], expf(lg[i] - mx) / sum };
            n = n_keep;
            return;
        }
        for (int i = 0; i < 8; i++) out[i] = cand{ ids[i], lg[i], expf(lg[i] - mx) / sum };
        n = 8;
    }
    llama_token sample_greedy(llama_context * ctx, int idx) {
        cand c[8]; int n;
        top_candidates(ctx, idx, c, n);
        ctx->n_sample++;
        if (n == 0) {                                                     // no results behind this row (a run that failed or was cut short must never be sampled): an error, not a token
            fprintf(stderr, "pipeinfer_speculative: no logits for row %d of the last %s run: %s\n", idx, ctx == tgt ? "target" : "draft", pipeinfer_context_error(ctx));
            failed = true;
            return 0;
        }
        return c[0].id;
    }

    // ---- KV helpers (both contexts hold the same sequence layout) -------------------------------------------------------------------
    static void rm(llama_context * c, int s, int p0, int p1) { llama_kv_cache_seq_rm(c, s, p0, p1); }
    static void cp(llama_context * c, int a, int b, int p0, int p1) { llama_kv_cache_seq_cp(c, a, b, p0, p1); }

    float p_adjust(int iter) const { return iter * P.p_recovery - std::max(n_rejected * P.p_decay, 0.0f); }      // speculative.cpp:846-848

    // speculative.cpp:1182-1275
    void begin_async_run(llama_context * ctx, std::deque<std::shared_ptr<async_run>> & runs, const std::vector<seq_draft> & dr, int & n_past, int n_past_dft_,
                         bool is_spec, host_batch & batch, int n_past_max, int prefix_n_past, int seq_offset) {
        batch_id++;
        auto run = std::make_shared<async_run>();
        run->seq_offset = seq_offset;
        run->batch = batch; run->batch.batch_id = batch_id;
        run->drafts = dr;
        run->n_past_tgt = n_past; run->prefix_n_past_tgt = prefix_n_past; run->n_past_max = n_past_max; run->n_past_dft = n_past_dft_;
        run->speculative = is_spec;
        if (!is_spec) pipeinfer_kv_cache_seq_rm_range(ctx, 0, max_seq, n_past, n_past + 1);
        else          pipeinfer_kv_cache_seq_rm_range(ctx, seq_offset, seq_offset + n_seq_dft - 1, n_past, n_past + 1);
        run->ticket = llama_start_async_decode(ctx, run->batch.view());
        if (!run->ticket) {                                               // no KV slot / n_tokens > n_batch / HIP error: the run never entered the context's queue
            fprintf(stderr, "pipeinfer_speculative: llama_start_async_decode failed (%s run, %d tokens): %s\n", is_spec ? "speculative" : "non-speculative", run->batch.n_tokens(), pipeinfer_context_error(ctx));
            failed = true;
            return;
        }
        if (P.verbose) {
            fprintf(stderr, "[spec] launch %s batch=%d ctx=%s off=%d n_past=%d n_past_max=%d prefix=%d tokens:", is_spec ? "SPEC" : "nonspec", batch_id, ctx == tgt ? "tgt" : "dft", seq_offset, n_past, n_past_max, prefix_n_past);
            for (int i = 0; i < run->batch.n_tokens(); i++) { fprintf(stderr, " %d@%d{", run->batch.token[i], run->batch.pos[i]); for (auto q : run->batch.seq[i]) fprintf(stderr, "%d,", q); fprintf(stderr, "}"); }
            fprintf(stderr, "\n");
        }
        runs.push_front(run);
        if (!is_spec) pipeinfer_kv_cache_seq_cp_range(ctx, 0, 1, max_seq, n_past, n_past + 1);
        if (is_spec) R.n_runs_spec++; else if (ctx == tgt) R.n_runs_nonspec++;
    }

    // speculative.cpp:850-879
    void begin_non_spec_run(llama_context * ctx, std::deque<std::shared_ptr<async_run>> & runs, int & n_past) {
        std::vector<seq_draft> ns(n_seq_dft);
        for (int s = 0; s < n_seq_dft; ++s) {
            ns[s].i_batch_tgt = std::vector<int>(1, 0);
            ns[s].i_batch_dft = drafts[s].i_batch_dft;
            ns[s].tokens = std::vector<llama_token>(1, id);
            ns[s].active = drafts[s].active; ns[s].drafting = drafts[s].drafting; ns[s].skip = drafts[s].skip;
        }
        host_batch b;
        b.add(id, n_past, { 0 });
        begin_async_run(ctx, runs, ns, n_past, n_past_dft, false, b, n_past + 1, n_past, 0);
        n_past++;
    }

    // speculative.cpp:1277-1359
    void check_for_cancel() {
        std::vector<int32_t> canceled;
        for (auto & rp : tgt_runs) {
            async_run & run = *rp;
            if (run.canceled) continue;
            bool correct_prefix = true;
            if (run.speculative && n_past_tgt >= run.prefix_n_past_tgt) {
                for (int d = n_seq_dft - 1; d >= 0; d--) {
                    if (run.drafts[d].tokens.empty()) continue;
                    correct_prefix = true;
                    std::vector<llama_token> concat = run.drafts[d].prefix_tokens;
                    concat.insert(concat.end(), run.drafts[d].tokens.begin(), run.drafts[d].tokens.end());
                    size_t di = 0, index = (size_t) run.prefix_n_past_tgt;
                    while (index < generated.size() && di < concat.size() && generated.size() > (size_t) run.prefix_n_past_tgt) {
                        if (generated[index] != concat[di]) { correct_prefix = false; break; }
                        di++; index = (size_t) run.prefix_n_past_tgt + di;
                    }
                    if (correct_prefix) run.s_keep = d;
                }
            }
            if (run.n_past_max < n_past_tgt || !correct_prefix) {
                if (run.speculative) canceled.push_back(run.batch.batch_id);
                run.canceled = true;
                if (P.verbose) fprintf(stderr, "[spec] cancel batch=%d spec=%d n_past_max=%d n_past_tgt=%d correct_prefix=%d\n", run.batch.batch_id, run.speculative, run.n_past_max, n_past_tgt, correct_prefix);
                R.n_runs_canceled++;
            }
        }
        if (!canceled.empty()) llama_cancel_run(tgt, canceled.data(), (int) canceled.size());
    }

    // speculative.cpp:881-1180 — returns true when nothing could be drafted
    bool start_async_spec_run(int prefix_n_past, int orig_offset, float padj) {
        const int off = current.seq_offset;
        for (int i = 0; i < n_seq_dft; i++) {
            rm(tgt, i + off, first_run ? prefix_n_past : prefix_n_past - 1, -1);
            rm(dft, i + off, first_run ? n_past_dft : n_past_dft - 1, -1);
            cp(tgt, first_run ? 0 : orig_offset, i + off, first_run ? prefix_n_past : prefix_n_past - 1, spec_past_tgt + 1);
            cp(dft, first_run ? 0 : orig_offset, i + off, first_run ? n_past_dft : n_past_dft - 1, spec_past_dft + 1);
        }
        batch_tgt.clear();
        for (int s = 0; s < n_seq_dft; ++s) {
            drafts[s].active = false;
            if (!first_run) { if (!drafts[s].tokens.empty()) drafts[s].prefix_tokens.insert(drafts[s].prefix_tokens.end(), drafts[s].tokens.begin(), drafts[s].tokens.end()); }
            else drafts[s].prefix_tokens.clear();
            drafts[s].tokens.clear();
            drafts[s].i_batch_tgt.clear();
        }
        drafts[0].tokens.push_back(id);                              // erased again after the drafting phase
        batch_dft.clear();
        int n_seq_cur = 0, max_ran_seq = 0, n_past_cur = spec_past_dft;
        for (int s = 0; s < n_seq_dft; ++s) { drafts[s].skip = true; drafts[s].active = false; drafts[s].drafting = false; }
        drafts[0].active = true; drafts[0].drafting = true; drafts[0].skip = false; drafts[0].i_batch_dft = 0;

        for (int i = 0; i < P.n_draft; ++i) {
            batch_dft.clear();
            for (int s = 0; s <= max_ran_seq; ++s) {
                if (!drafts[s].drafting || drafts[s].skip) continue;
                cand cur[8]; int nc;
                top_candidates(dft, drafts[s].i_batch_dft, cur, nc);
                dft->n_sample++;
                if (nc == 0 || cur[0].p < P.p_accept + padj) { drafts[s].drafting = false; continue; }
                std::vector<int> sa(1, s);
                for (int f = 1; f < 8 && f < nc; ++f) {                  // try to split the branch
                    if (n_seq_cur < n_seq_dft - 1 && cur[f].p > P.p_split + padj) {
                        n_seq_cur++;
                        rm(dft, n_seq_cur + off, n_past_dft, n_past_cur);
                        cp(dft, s + off, n_seq_cur + off, n_past_dft, n_past_cur);
                        for (auto & sq : batch_tgt.seq)                  // earlier tokens of this branch also belong to the new one
                            for (size_t p = 0; p < sq.size(); ++p) if (sq[p] == s + off) { sq.push_back(n_seq_cur + off); break; }
                        drafts[n_seq_cur].active = true; drafts[n_seq_cur].drafting = true; drafts[n_seq_cur].skip = false;
                        drafts[n_seq_cur].tokens = drafts[s].tokens;
                        drafts[n_seq_cur].i_batch_tgt = drafts[s].i_batch_tgt;
                        sa.push_back(n_seq_cur);
                    } else break;
                }
                for (size_t is = 0; is < sa.size(); ++is) {
                    const llama_token tid = cur[is].id;
                    const int sq = sa[is];
                    drafts[sq].tokens.push_back(tid);
                    drafts[sq].i_batch_tgt.push_back(batch_tgt.n_tokens());
                    batch_tgt.add(tid, spec_past_tgt + i, { sq + off });
                    drafts[sq].i_batch_dft = batch_dft.n_tokens();
                    batch_dft.add(tid, n_past_cur, { sq + off });
                    if (batch_tgt.n_tokens() > P.n_draft) drafts[sq].drafting = false;
                }
            }
            if (batch_dft.n_tokens() == 0) break;
            batch_dft.batch_id = batch_id;
            if (llama_decode(dft, batch_dft.view()) != 0) { failed = true; break; }      // (stale top-k rows must not be sampled)
            ++n_past_cur;
            ++R.n_drafted;
            max_ran_seq = n_seq_cur;
            batch_dft.clear();
            if (batch_tgt.n_tokens() > P.n_draft) break;
        }
        if (batch_tgt.n_tokens() == 0 || failed) {
            free_offsets.push_back(off);
            n_rejected++;
            return true;
        }
        for (int s = 0; s < n_seq_dft; ++s) {
            if (!drafts[s].active) continue;
            drafts[s].tokens.erase(drafts[s].tokens.begin());
        }
        if (first_run) ++R.n_drafted;
        const int n0 = (int) drafts[0].tokens.size();
        R.n_drafted_tokens += batch_tgt.n_tokens();
        begin_async_run(tgt, tgt_runs, drafts, spec_past_tgt, n_past_dft, true, batch_tgt, spec_past_tgt + n0, prefix_n_past, off);
        spec_past_tgt += n0; spec_past_dft += n0;
        id = drafts[0].tokens.back();
        first_run = false;
        return false;
    }

    // ---- adaptive draft pause (our addition, pipeinfer.h: adaptive_window / adaptive_pause; off by default) ---------------------------
    // Called once per accepted token.  While the draft is paused its KV cache lags: the accepted tokens are remembered and decoded
    // in one batch (same cells, same sequence layout as the per-token begin_non_spec_run) right before drafting resumes.
    int pause_left = 0, pause_len = 0, win_tokens = 0, win_spec0 = 0;
    bool loop_found_target_back = false;                                  // the last speculation loop could not start because the target run was already back
    std::vector<std::pair<llama_token, int>> lag_tokens;                   // (token, position): drafted tokens accepted meanwhile keep their own cells
    bool draft_paused() const { return pause_left > 0; }
    void adaptive_step() {
        if (P.mode == 2 || P.adaptive_window == 0 || P.adaptive_pause <= 0) return;
        if (pause_left > 0) { if (--pause_left == 0) { win_tokens = 0; win_spec0 = R.n_runs_spec; } return; }
        if (P.adaptive_window < 0) {                                       // test hook: pause after every |window| tokens no matter what
            if (++win_tokens >= -P.adaptive_window) pause_left = P.adaptive_pause;
            return;
        }
        if (R.n_runs_spec != win_spec0) { win_tokens = 0; win_spec0 = R.n_runs_spec; pause_len = 0; return; }     // speculation is happening: keep drafting
        if (!loop_found_target_back) { win_tokens = 0; return; }          // no run for another reason (an unsure draft, no free partition): that is not what the pause is for
        if (++win_tokens >= P.adaptive_window) {
            pause_len = pause_len == 0 ? P.adaptive_pause : std::min(pause_len * 2, P.adaptive_pause * 16);
            pause_left = pause_len;
        }
    }
    bool draft_catch_up() {
        if (lag_tokens.empty()) return true;
        // every lagged token is decoded at ITS position: a speculative run launched before the pause can be accepted while the draft sits
        // out, which advances n_past_dft past cells that already hold the draft's K/V of those accepted tokens
        const int n = (int) lag_tokens.size();
        for (const auto & tp : lag_tokens) pipeinfer_kv_cache_seq_rm_range(dft, 0, max_seq, tp.second, tp.second + 1);
        const int n_batch = std::max(1, dft->n_batch);
        for (int i0 = 0; i0 < n; i0 += n_batch) {
            host_batch b;
            for (int i = i0; i < std::min(n, i0 + n_batch); i++) b.add(lag_tokens[i].first, lag_tokens[i].second, { 0 });
            if (llama_decode(dft, b.view()) != 0) return false;
        }
        for (const auto & tp : lag_tokens) pipeinfer_kv_cache_seq_cp_range(dft, 0, 1, max_seq, tp.second, tp.second + 1);
        lag_tokens.clear();
        R.n_draft_catchups++;
        return true;
    }

    // speculative.cpp:765-844.  min_spec_iters > 0 (our addition, default 0 = reference behaviour) runs that many speculation
    // iterations even if the oldest target run has already finished — on one fast GPU the target is often back before the
    // first draft token, and tests need the accept / cancel paths exercised.
    void run_speculation_loop() {
        bool is_waiting = llama_mpi_iprobe(tgt);
        pi::tl("specloop", is_waiting ? 1 : 0, (int) tgt_runs.size());
        loop_found_target_back = is_waiting;                              // (adaptive draft pause: the only reason it counts)
        const int forced = P.min_spec_iters;
        bool returned_offset = false;
        if (is_waiting && forced == 0) { free_offsets.push_back(current.seq_offset); returned_offset = true; }
        int iter = 0, launched = 0;
        float padj;
        while ((!is_waiting || iter < forced) && P.p_accept + (padj = p_adjust(iter)) < 1.0f) {
            if (P.max_inflight > 0 && (int) tgt_runs.size() >= P.max_inflight) break;
            const int orig_offset = current.seq_offset;
            bool should_run_spec = true;
            do {
                should_run_spec = true;
                for (const auto & r : tgt_runs) {
                    if (r->seq_offset == current.seq_offset && r->speculative) {
                        should_run_spec = false;
                        if (!free_offsets.empty()) { current.seq_offset = free_offsets.front(); free_offsets.pop_front(); }
                        break;
                    }
                }
            } while (!should_run_spec && !free_offsets.empty());
            if (!should_run_spec) break;
            auto it = std::find(free_offsets.begin(), free_offsets.end(), current.seq_offset);
            if (it != free_offsets.end()) free_offsets.erase(it);
            if (start_async_spec_run(n_past_tgt, orig_offset, padj)) { launched = -1; break; }      // (that path already returned the offset)
            launched++;
            is_waiting = llama_mpi_iprobe(tgt);
            first_run = false;
            iter++;
        }
        if (is_waiting && forced > 0 && launched == 0 && !returned_offset &&
            std::find(free_offsets.begin(), free_offsets.end(), current.seq_offset) == free_offsets.end()) {
            bool in_use = false;
            for (const auto & r : tgt_runs) if (r->speculative && r->seq_offset == current.seq_offset) in_use = true;
            if (!in_use) free_offsets.push_back(current.seq_offset);
        }
    }

    int run(const llama_token * prompt, int n_prompt, llama_token * out, int out_cap) {
        memset(&R, 0, sizeof R);
        n_seq_dft = std::max(1, P.n_parallel);
        const int n_simul_seqs = 1000;                                    // speculative.cpp:140-145
        max_seq = n_simul_seqs * n_seq_dft + 1;
        for (int i = 0; i < n_simul_seqs; i++) free_offsets.push_back(i * n_seq_dft + 1);
        eos = llama_token_eos(llama_get_model(tgt));
        const int n_input = n_prompt;
        if (n_input < 2 || n_input > llama_n_ctx(tgt) - 4) return -1;
        // the KV cache must hold the prompt, the generation and the cells of the speculative runs that can be in flight at once
        // (the reference aborts in find_slot when it cannot; say so before anything is launched)
        // Hard bound only (like run_sync): the reference fails when find_slot cannot place a batch (llama.cpp:5562-5565), not up front; a run that cannot be
        // placed ends the generation through the failed / drain path.  The cells speculative runs in flight may occupy are a warning.
        if (n_input + P.n_predict + 2 > llama_n_ctx(tgt)) {
            fprintf(stderr, "pipeinfer_speculative: n_ctx %d cannot hold %d prompt + %d generated tokens\n", llama_n_ctx(tgt), n_input, P.n_predict);
            return -1;
        }
        if (P.mode != 2 && P.verbose) {
            const int need = n_input + P.n_predict + 2 + (P.n_draft + 1) * n_seq_dft * (P.max_inflight > 0 ? P.max_inflight : 4);
            if (need > llama_n_ctx(tgt)) fprintf(stderr, "pipeinfer_speculative: warning: n_ctx %d may run short of cells with several speculative runs in flight (%d wanted)\n", llama_n_ctx(tgt), need);
        }
        R.n_input = n_input;

        // ---- prompt: every token carries every sequence id (speculative.cpp:223-246) ----
        const int64_t t_enc_start = pi::time_us();
        std::vector<llama_seq_id> all_seqs(max_seq + 1);
        for (int i = 0; i <= max_seq; i++) all_seqs[i] = i;
        {
            host_batch bt, bd;
            for (int i = 0; i < n_input - 1; i++) { bt.add(prompt[i], i, all_seqs); bd.add(prompt[i], i, all_seqs); }
            if (llama_decode(tgt, bt.view()) != 0) return -2;
            bt.clear();
            bt.add(prompt[n_input - 1], n_input - 1, all_seqs); bd.add(prompt[n_input - 1], n_input - 1, all_seqs);
            if (llama_decode(tgt, bt.view()) != 0) return -2;
            if (P.mode != 2 && llama_decode(dft, bd.view()) != 0) return -2;
        }
        const int64_t t_enc_end = pi::time_us();

        n_past_tgt = n_past_dft = n_input;
        bool has_eos = false;
        drafts.assign(n_seq_dft, seq_draft());
        const int64_t t_dec_start = pi::time_us();
        drafts[0].i_batch_tgt = { 0 };
        current = async_run();
        current.n_past_tgt = n_past_tgt - 1; current.n_past_max = n_past_tgt; current.n_past_dft = n_past_dft - 1;
        current.seq_offset = free_offsets.front();
        current.speculative = false;
        generated.assign(prompt, prompt + n_input);
        spec_past_tgt = n_past_tgt; spec_past_dft = n_past_dft;
        int64_t ttft = pi::time_us(), itt_start = 0;
        std::vector<int64_t> itl;
        bool first_token = false, has_run_first_token = false;
        first_run = true;
        n_rejected = 0;
        int n_out = 0;

        while (true) {
            if (failed) {                                                 // drain and report (the output so far is valid, the generation is not complete)
                // (the draft first: on one GPU its pending runs hold the device arbiter, and the target's feeder launches its remaining segments only on their credit)
                while (!dft_runs.empty()) { auto r = dft_runs.back(); llama_finish_async_decode(dft, r->batch.view(), r->ticket); dft_runs.pop_back(); }
                while (!tgt_runs.empty()) { auto r = tgt_runs.back(); llama_finish_async_decode(tgt, r->batch.view(), r->ticket); tgt_runs.pop_back(); }
                return -3;
            }
            int i_dft = 0, s_keep = 0;
            const bool is_waiting = llama_mpi_iprobe(tgt);
            if (!tgt_runs.empty() && is_waiting) {
                check_for_cancel();
                std::shared_ptr<async_run> run = tgt_runs.back();
                if (P.verbose) fprintf(stderr, "[spec] finish batch=%d spec=%d canceled=%d s_keep=%d n_past_tgt=%d\n", run->batch.batch_id, run->speculative, run->canceled, run->s_keep, n_past_tgt);
                if (!run->canceled) {
                    drafts = run->drafts;
                    current.speculative = run->speculative; current.n_past_max = run->n_past_max; current.n_past_tgt = run->n_past_tgt;
                    current.n_past_dft = run->n_past_dft; current.seq_offset = run->seq_offset;
                    s_keep = run->s_keep;
                    if (run->speculative) R.n_accept_clean++;             // its first drafted token was confirmed by the run before it (prefix check)
                    for (int s = 0; s < n_seq_dft; ++s) { if (!drafts[s].active) continue; drafts[s].tokens.erase(drafts[s].tokens.begin()); }
                } else {
                    llama_finish_async_decode(tgt, run->batch.view(), run->ticket);
                    tgt_runs.pop_back();
                    if (run->speculative) free_offsets.push_back(run->seq_offset);
                    continue;
                }
                const int frc = llama_finish_async_decode(tgt, run->batch.view(), run->ticket);
                tgt_runs.pop_back();
                if (frc < 0) { fprintf(stderr, "pipeinfer_speculative: target run %d failed: %s\n", run->batch.batch_id, pipeinfer_context_error(tgt)); failed = true; continue; }
                spec_past_tgt = n_past_tgt; spec_past_dft = n_past_dft;
                first_run = true;
            } else if (!tgt_runs.empty()) {
                if (P.mode == 2 || draft_paused()) { pi::context_wait(tgt, (pi_run *) tgt_runs.back()->ticket); if (tgt->remote) { while (!llama_mpi_iprobe(tgt)) {} } continue; }      // target only: just wait
                run_speculation_loop();
                continue;
            }

            bool any_match = false;
            std::deque<int> keeps;
            keeps.push_back(s_keep);
            while (!keeps.empty()) {
                id = sample_greedy(tgt, drafts[keeps[0]].i_batch_tgt[i_dft]);
                if (failed) break;
                if (current.n_past_tgt + i_dft == n_past_tgt - 1) {
                    any_match = true;
                    ++R.n_predict;
                    if (current.speculative) R.n_accept++;
                    if (has_run_first_token) {
                        if (first_token) { ttft = pi::time_us() - ttft; first_token = false; }
                        else itl.push_back(pi::time_us() - itt_start);
                        itt_start = pi::time_us();
                    }
                    generated.push_back(id);
                    if (P.verbose) fprintf(stderr, "[spec] ACCEPT %d at gen index %zu (i_dft=%d idx=%d cur.n_past_tgt=%d spec=%d)\n", id, generated.size() - 1, i_dft, drafts[keeps[0]].i_batch_tgt[i_dft], current.n_past_tgt, current.speculative);
                    if (n_out < out_cap) out[n_out++] = id;
                }
                if (id == eos) has_eos = true;
                if (!current.speculative) break;
                bool matches = false;
                keeps.clear();
                for (int s = 0; s < n_seq_dft; ++s) {
                    if (!drafts[s].active) continue;
                    if (i_dft < (int) drafts[s].tokens.size() && id == drafts[s].tokens[i_dft]) { matches = true; keeps.push_back(s); s_keep = keeps[0]; }
                    else drafts[s].active = false;
                }
                if (matches) {
                    if (current.n_past_tgt + i_dft == n_past_tgt - 1) { ++R.n_accept; ++R.n_accept_clean; ++n_past_tgt; ++n_past_dft; }
                    ++i_dft;
                    if (current.speculative && current.n_past_tgt + i_dft < n_past_tgt) continue;
                }
            }

            pi::tl("sampled", n_past_tgt, any_match ? 1 : 0, current.speculative ? 1 : 0);
            if (failed) continue;
            if (!any_match) {
                if (current.speculative) free_offsets.push_back(current.seq_offset);
                continue;
            }
            n_rejected = 0;
            check_for_cancel();

            if (current.speculative) {                                    // accept/reject tree prune (speculative.cpp:588-612)
                const int off = current.seq_offset;
                pipeinfer_kv_cache_seq_rm_range(tgt, off, off + n_seq_dft - 1, n_past_tgt, -1);
                pipeinfer_kv_cache_seq_rm_range(dft, off, off + n_seq_dft - 1, n_past_dft, -1);
                rm(tgt, 0, current.n_past_tgt + 1, n_past_tgt);
                rm(dft, 0, current.n_past_dft + 1, n_past_dft);
                cp(tgt, s_keep + off, 0, current.n_past_tgt + 1, n_past_tgt);
                cp(dft, s_keep + off, 0, current.n_past_dft + 1, n_past_dft);
                pipeinfer_kv_cache_seq_rm_cp_range(tgt, 0, 1, max_seq, current.n_past_tgt + 1, n_past_tgt);
                pipeinfer_kv_cache_seq_rm_cp_range(dft, 0, 1, max_seq, current.n_past_dft + 1, n_past_dft);
            }

            begin_non_spec_run(tgt, tgt_runs, n_past_tgt);
            if (!has_run_first_token) { has_run_first_token = true; first_token = true; }
            adaptive_step();
            if (P.mode != 2 && !draft_paused()) {
                if (!draft_catch_up()) return -2;
                begin_non_spec_run(dft, dft_runs, n_past_dft);
                std::shared_ptr<async_run> d = dft_runs.back();
                dft_runs.pop_back();
                if (llama_finish_async_decode(dft, d->batch.view(), d->ticket) < 0) { failed = true; continue; }
            } else {
                if (P.mode != 2) { lag_tokens.emplace_back(id, n_past_dft); R.n_draft_paused++; }      // the draft sits this token out (caught up later in one batch)
                n_past_dft++;
            }
            spec_past_tgt = n_past_tgt; spec_past_dft = n_past_dft;

            if (R.n_predict > P.n_predict || has_eos) break;              // (checked here so the last token does not launch more speculation)

            if (!current.speculative) {
                if (free_offsets.empty()) continue;
                current.seq_offset = free_offsets.front();
                free_offsets.pop_front();
            }
            if (P.mode != 2 && !draft_paused()) run_speculation_loop();
        }
        const int64_t t_dec_end = pi::time_us();

        // drain what is still in flight (speculative.cpp:735-747) — the draft first: a pending draft run holds the single-GPU arbiter (context.cpp), and waiting for a
        // target run whose feeder is waiting for that draft run's credit would never end
        while (!dft_runs.empty()) { auto r = dft_runs.back(); llama_finish_async_decode(dft, r->batch.view(), r->ticket); dft_runs.pop_back(); }
        while (!tgt_runs.empty()) { auto r = tgt_runs.back(); llama_finish_async_decode(tgt, r->batch.view(), r->ticket); tgt_runs.pop_back(); }

        R.t_enc_s = (t_enc_end - t_enc_start) / 1e6; R.t_dec_s = (t_dec_end - t_dec_start) / 1e6;
        R.encode_tps = n_input / R.t_enc_s; R.decode_tps = R.n_predict / R.t_dec_s;
        R.ttft_s = ttft / 1e6;
        double s = 0; for (int64_t v : itl) s += (double) v;
        R.itl_s = itl.empty() ? 0.0 : s / itl.size() / 1e6;
        return n_out;
    }

    // ---- mode 1: synchronous tree speculation — examples/speculative_orig/speculative.cpp:150-520 restated ----------------------------
    // One cycle = sample the target over the last verify batch (accepting drafted tokens while they match), tidy both KV caches down
    // to the kept branch, draft up to n_draft steps (tree branches split at p_split), verify the whole tree in ONE target batch.
    // Nothing runs concurrently: on a single GPU this is the schedule in which the draft never queues behind a target run.
    // n_drafted / n_accept are deterministic functions of the two models.
    int run_sync(const llama_token * prompt, int n_prompt, llama_token * out, int out_cap) {
        memset(&R, 0, sizeof R);
        n_seq_dft = std::max(1, P.n_parallel);
        eos = llama_token_eos(llama_get_model(tgt));
        const int n_input = n_prompt;
        if (n_input < 2 || n_input > llama_n_ctx(tgt) - 4) return -1;
        if (n_input + P.n_predict + 2 > llama_n_ctx(tgt)) {
            fprintf(stderr, "pipeinfer_speculative: n_ctx %d cannot hold %d prompt + %d generated tokens\n", llama_n_ctx(tgt), n_input, P.n_predict);
            return -1;
        }
        R.n_input = n_input;
        std::vector<llama_seq_id> seq_ids(n_seq_dft + 1);
        for (int i = 0; i <= n_seq_dft; i++) seq_ids[i] = i;

        const int64_t t_enc_start = pi::time_us();
        {   // speculative_orig:172-186
            host_batch bt, bd;
            for (int i = 0; i < n_input - 1; i++) { bt.add(prompt[i], i, seq_ids); bd.add(prompt[i], i, seq_ids); }
            if (llama_decode(tgt, bt.view()) != 0) return -2;
            bt.clear();
            bt.add(prompt[n_input - 1], n_input - 1, seq_ids); bd.add(prompt[n_input - 1], n_input - 1, seq_ids);
            if (llama_decode(tgt, bt.view()) != 0) return -2;
            if (llama_decode(dft, bd.view()) != 0) return -2;
        }
        const int64_t t_enc_end = pi::time_us();

        n_past_tgt = n_past_dft = n_input;
        bool has_eos = false;
        drafts.assign(n_seq_dft, seq_draft());
        const int64_t t_dec_start = pi::time_us();
        drafts[0].i_batch_tgt = { 0 };
        int64_t ttft = pi::time_us(), itt_start = 0;
        std::vector<int64_t> itl;
        bool first_token = false, has_run_first_token = false;
        int n_out = 0;

        while (true) {
            int i_dft = 0, s_keep = 0;
            while (true) {
                id = sample_greedy(tgt, drafts[s_keep].i_batch_tgt[i_dft]);
                if (failed) return -3;
                if (has_run_first_token) {
                    if (first_token) { ttft = pi::time_us() - ttft; first_token = false; }
                    else itl.push_back(pi::time_us() - itt_start);
                    itt_start = pi::time_us();
                }
                if (n_out < out_cap) out[n_out++] = id;
                if (id == eos) has_eos = true;
                ++R.n_predict;
                bool matches = false;
                for (int s = 0; s < n_seq_dft; ++s) {
                    if (!drafts[s].active) continue;
                    if (i_dft < (int) drafts[s].tokens.size() && id == drafts[s].tokens[i_dft]) { s_keep = s; matches = true; }
                    else drafts[s].active = false;
                }
                if (matches) { ++R.n_accept; ++R.n_accept_clean; ++n_past_tgt; ++n_past_dft; ++i_dft; continue; }

                llama_kv_cache_seq_keep(dft, s_keep); cp(dft, s_keep, 0, -1, -1); llama_kv_cache_seq_keep(dft, 0);
                rm(tgt, s_keep, n_past_tgt, -1);
                llama_kv_cache_seq_keep(tgt, s_keep); cp(tgt, s_keep, 0, -1, -1); llama_kv_cache_seq_keep(tgt, 0);
                for (int s = 0; s < n_seq_dft; ++s) { drafts[s].active = false; drafts[s].tokens.clear(); drafts[s].i_batch_tgt.clear(); }
                drafts[0].tokens.push_back(id);                           // erased again after the speculation phase
                drafts[0].i_batch_tgt.push_back(0);
                batch_dft.clear();
                batch_dft.add(id, n_past_dft, { 0 });
                rm(dft, -1, n_past_dft, -1);
                if (llama_decode(dft, batch_dft.view()) != 0) return -2;
                ++n_past_dft;
                break;
            }
            if (R.n_predict > P.n_predict || has_eos) break;

            int n_seq_cur = 1, n_past_cur = n_past_dft;
            for (int s = 0; s < n_seq_dft; ++s) { drafts[s].active = false; drafts[s].drafting = false; }
            drafts[0].active = true; drafts[0].drafting = true; drafts[0].i_batch_dft = 0;
            batch_tgt.clear();
            batch_tgt.add(drafts[0].tokens[0], n_past_tgt, { 0 });

            for (int i = 0; i < P.n_draft; ++i) {                          // tree drafting (speculative_orig:375-475)
                batch_dft.clear();
                for (int s = 0; s < n_seq_dft; ++s) drafts[s].skip = false;
                for (int s = 0; s < n_seq_dft; ++s) {
                    if (!drafts[s].drafting || drafts[s].skip) continue;
                    cand cur[8]; int nc;
                    top_candidates(dft, drafts[s].i_batch_dft, cur, nc);
                    dft->n_sample++;
                    if (nc == 0 || cur[0].p < P.p_accept) { drafts[s].drafting = false; continue; }
                    std::vector<int> sa(1, s);
                    for (int f = 1; f < 8 && f < nc; ++f) {
                        if (n_seq_cur < n_seq_dft && cur[f].p > P.p_split) {
                            rm(dft, n_seq_cur, -1, -1);
                            cp(dft, s, n_seq_cur, -1, -1);
                            for (auto & sq : batch_tgt.seq)              // earlier tokens of this branch also belong to the new one
                                for (size_t q = 0; q < sq.size(); ++q) if (sq[q] == s) { sq.push_back(n_seq_cur); break; }
                            drafts[n_seq_cur].active = true; drafts[n_seq_cur].drafting = true; drafts[n_seq_cur].skip = true;
                            drafts[n_seq_cur].tokens = drafts[s].tokens;
                            drafts[n_seq_cur].i_batch_dft = drafts[s].i_batch_dft;
                            drafts[n_seq_cur].i_batch_tgt = drafts[s].i_batch_tgt;
                            sa.push_back(n_seq_cur);
                            n_seq_cur++;
                        } else break;
                    }
                    for (size_t is = 0; is < sa.size(); ++is) {
                        const llama_token tid = cur[is].id;
                        const int sq = sa[is];
                        drafts[sq].tokens.push_back(tid);
                        drafts[sq].i_batch_tgt.push_back(batch_tgt.n_tokens());
                        batch_tgt.add(tid, n_past_tgt + i + 1, { sq });
                        drafts[sq].i_batch_dft = batch_dft.n_tokens();
                        batch_dft.add(tid, n_past_cur, { sq });
                        if (batch_tgt.n_tokens() > P.n_draft) drafts[sq].drafting = false;
                    }
                }
                if (batch_dft.n_tokens() == 0) break;
                if (llama_decode(dft, batch_dft.view()) != 0) return -2;
                ++n_past_cur;
                ++R.n_drafted;
                if (batch_tgt.n_tokens() > P.n_draft) break;
            }

            llama_kv_cache_seq_keep(tgt, 0);                               // verify the tree in one target batch (speculative_orig:478-495)
            for (int s = 1; s < n_seq_dft; ++s) cp(tgt, 0, s, -1, -1);
            if (!has_run_first_token) { has_run_first_token = true; first_token = true; }
            R.n_drafted_tokens += batch_tgt.n_tokens() - 1;
            if (batch_tgt.n_tokens() > 1) R.n_runs_spec++; else R.n_runs_nonspec++;
            if (llama_decode(tgt, batch_tgt.view()) != 0) return -2;
            ++n_past_tgt;
            for (int s = 0; s < n_seq_dft; ++s) { if (!drafts[s].active) continue; drafts[s].tokens.erase(drafts[s].tokens.begin()); }
        }
        const int64_t t_dec_end = pi::time_us();
        R.t_enc_s = (t_enc_end - t_enc_start) / 1e6; R.t_dec_s = (t_dec_end - t_dec_start) / 1e6;
        R.encode_tps = n_input / R.t_enc_s; R.decode_tps = R.n_predict / R.t_dec_s;
        R.ttft_s = ttft / 1e6;
        double sum = 0; for (int64_t v : itl) sum += (double) v;
        R.itl_s = itl.empty() ? 0.0 : sum / itl.size() / 1e6;
        return n_out;
    }
};

}  // namespace

extern "C" struct pipeinfer_spec_params pipeinfer_spec_default_params(void) {
    struct pipeinfer_spec_params p;
    memset(&p, 0, sizeof p);
    p.n_predict = 128; p.n_draft = 3; p.n_parallel = 3; p.p_accept = 0.1f; p.p_split = 0.9f; p.p_recovery = 0.0f; p.p_decay = 0.0f;   // dolphin_tinyllama_8_nodes.sh
    p.temp = -1.0f; p.ignore_eos = 1; p.mode = 0; p.max_inflight = 0; p.verbose = 0; p.adaptive_window = 0; p.adaptive_pause = 0;
    return p;
}

extern "C" int pipeinfer_speculative(struct llama_context * ctx_tgt, struct llama_context * ctx_dft, const struct pipeinfer_spec_params * params,
                                     const llama_token * prompt, int32_t n_prompt, llama_token * out_tokens, int32_t out_cap, struct pipeinfer_spec_result * result) {
    driver d;
    d.tgt = ctx_tgt; d.dft = ctx_dft; d.P = *params;
    if (params->mode < 0 || params->mode > 2) { fprintf(stderr, "pipeinfer_speculative: unknown mode %d\n", params->mode); return -1; }
    if (!ctx_tgt->cparams.logits_all || (params->mode != 2 && !ctx_dft->cparams.logits_all)) { fprintf(stderr, "pipeinfer_speculative: both contexts need logits_all (speculative.cpp:115)\n"); return -1; }
    ctx_tgt->canceled_batches.clear();                       // batch ids restart at 1 for every call
    if (ctx_dft) ctx_dft->canceled_batches.clear();
    for (llama_context * c : { ctx_tgt, ctx_dft })           // remote stages key their cancel notes by generation: -2 opens a new one
        if (c && c->remote && c->remote_ops.cancel) c->remote_ops.cancel(c->remote_ops.user, -2);
    const int n = params->mode == 1 ? d.run_sync(prompt, n_prompt, out_tokens, out_cap) : d.run(prompt, n_prompt, out_tokens, out_cap);
    if (result) *result = d.R;
    return n;
}
