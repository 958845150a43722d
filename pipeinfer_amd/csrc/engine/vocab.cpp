#include "vocab.h"

#include <cstring>
#include <queue>
#include <stdexcept>

namespace pi {

static size_t utf8_len(char c) {
    static const size_t lookup[16] = { 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4 };
    return lookup[(uint8_t) c >> 4];
}

static void replace_all(std::string & s, const std::string & from, const std::string & to) {
    std::string out;
    size_t pos = 0, hit;
    while ((hit = s.find(from, pos)) != std::string::npos) { out.append(s, pos, hit - pos); out += to; pos = hit + from.size(); }
    out.append(s, pos, std::string::npos);
    s.swap(out);
}

int32_t vocab::byte_to_token(uint8_t ch) const {
    static const char * hex = "0123456789ABCDEF";
    const char buf[7] = { '<', '0', 'x', hex[ch >> 4], hex[ch & 15], '>', 0 };
    auto it = token_to_id.find(buf);
    if (it == token_to_id.end()) throw std::runtime_error(std::string("vocab has no byte token ") + buf);
    return it->second;
}

bool vocab::load(const gguf_file & f, std::string & err) {
    const gguf_value * toks = f.find("tokenizer.ggml.tokens");
    if (!toks || toks->type != 9 || toks->arr_type != 8) { err = "cannot find tokenizer vocab in model file"; return false; }
    const gguf_value * scores = f.find("tokenizer.ggml.scores");
    const gguf_value * types  = f.find("tokenizer.ggml.token_type");
    const std::string model = f.get_str("tokenizer.ggml.model", "llama");
    if (model == "gpt2") {
        type = 1; bos_id = 11; eos_id = 11; unk_id = -1;
        if (const gguf_value * merges = f.find("tokenizer.ggml.merges")) {
            for (size_t i = 0; i < merges->arr_s.size(); i++) {
                const std::string & w = merges->arr_s[i];
                const size_t pos = w.find(' ', 1);
                if (pos != std::string::npos) bpe_ranks.emplace(std::make_pair(w.substr(0, pos), w.substr(pos + 1)), (int) i);
            }
        }
    } else {
        type = 0; bos_id = 1; eos_id = 2; unk_id = 0;
    }
    const size_t n = toks->arr_s.size();
    id_to_token.resize(n);
    for (size_t i = 0; i < n; i++) {
        id_to_token[i].text  = toks->arr_s[i];
        id_to_token[i].score = scores && scores->arr_data && i < scores->arr_n ? ((const float *) scores->arr_data)[i] : 0.0f;
        id_to_token[i].type  = types && types->arr_data && i < types->arr_n ? ((const int32_t *) types->arr_data)[i] : 1;
        token_to_id[toks->arr_s[i]] = (int32_t) i;
    }
    auto special = [&](const char * key, int32_t & id) {
        const gguf_value * v = f.find(key);
        if (v && v->type == 4 && (size_t) (v->u + 1) <= n) id = (int32_t) v->u;
    };
    special("tokenizer.ggml.bos_token_id", bos_id);
    special("tokenizer.ggml.eos_token_id", eos_id);
    special("tokenizer.ggml.unknown_token_id", unk_id);
    special("tokenizer.ggml.seperator_token_id", sep_id);
    special("tokenizer.ggml.padding_token_id", pad_id);
    if (const gguf_value * v = f.find("tokenizer.ggml.add_bos_token")) if (v->type == 7) add_bos = v->b;
    if (const gguf_value * v = f.find("tokenizer.ggml.add_eos_token")) if (v->type == 7) add_eos = v->b;
    if (type == 0) {
        try { linefeed_id = byte_to_token('\n'); } catch (const std::exception & e) { err = e.what(); return false; }
    } else {
        auto it = token_to_id.find("\xC4\x8A");     // U+010A, the byte-level BPE spelling of '\n'
        linefeed_id = it != token_to_id.end() ? it->second : -1;
    }
    // special-token cache: multi-character tokens that cannot be produced by joining two other tokens (llama.cpp:2540-2615)
    for (const auto & kv : token_to_id) {
        const std::string & tok = kv.first;
        if (tok.length() <= 1) continue;
        bool tokenizable = false;
        for (size_t i = 1; i < tok.length();) {
            const size_t utf = utf8_len(tok[i - 1]);
            if (utf == 1) {
                if (token_to_id.count(tok.substr(0, i)) && token_to_id.count(tok.substr(i))) { tokenizable = true; break; }
                i++;
            } else {
                i += utf - 1;
            }
        }
        if (tokenizable) continue;
        size_t chars = 0;
        for (size_t i = 0; i < tok.length(); i += utf8_len(tok[i])) chars++;
        if (chars > 1) special_tokens_cache[tok] = kv.second;
    }
    return true;
}

namespace {

struct symbol { int prev, next; const char * text; size_t n; };

struct bigram {
    int left, right; float score; size_t size;
};
struct bigram_less {   // max-heap on score, ties -> smaller left index first (llm_bigram_spm::comparator)
    bool operator()(const bigram & l, const bigram & r) const { return l.score < r.score || (l.score == r.score && l.left > r.left); }
};

class spm_session {
public:
    spm_session(const vocab & v, const std::string & text, std::vector<int32_t> & out) : v_(v), out_(out) {
        size_t offs = 0; int index = 0;
        while (offs < text.size()) {
            symbol s;
            const size_t len = utf8_len(text[offs]);
            s.text = text.c_str() + offs;
            s.n = std::min(len, text.size() - offs);
            offs += s.n;
            s.prev = index - 1;
            s.next = offs == text.size() ? -1 : index + 1;
            index++;
            syms_.push_back(s);
        }
        for (size_t i = 1; i < syms_.size(); ++i) try_add((int) i - 1, (int) i);
        while (!queue_.empty()) {
            const bigram b = queue_.top(); queue_.pop();
            symbol & l = syms_[b.left]; symbol & r = syms_[b.right];
            if (l.n == 0 || r.n == 0 || l.n + r.n != b.size) continue;        // one side already merged away
            l.n += r.n; r.n = 0;
            l.next = r.next;
            if (r.next >= 0) syms_[r.next].prev = b.left;
            try_add(l.prev, b.left);
            try_add(b.left, l.next);
        }
        if (!syms_.empty()) for (int i = 0; i != -1; i = syms_[i].next) emit(syms_[i]);
    }

private:
    void try_add(int left, int right) {
        if (left == -1 || right == -1) return;
        const std::string text(syms_[left].text, syms_[left].n + syms_[right].n);
        auto it = v_.token_to_id.find(text);
        if (it == v_.token_to_id.end() || (size_t) it->second >= v_.id_to_token.size()) return;
        queue_.push(bigram{ left, right, v_.id_to_token[it->second].score, text.size() });
        rev_merge_[text] = std::make_pair(left, right);
    }
    void emit(const symbol & s) {
        const std::string text(s.text, s.n);
        auto it = v_.token_to_id.find(text);
        if (it != v_.token_to_id.end()) { out_.push_back(it->second); return; }
        auto p = rev_merge_.find(text);
        if (p == rev_merge_.end()) {
            for (size_t j = 0; j < s.n; ++j) out_.push_back(v_.byte_to_token((uint8_t) s.text[j]));   // byte fallback
            return;
        }
        emit(syms_[p->second.first]);
        emit(syms_[p->second.second]);
    }
    const vocab & v_;
    std::vector<int32_t> & out_;
    std::vector<symbol> syms_;
    std::priority_queue<bigram, std::vector<bigram>, bigram_less> queue_;
    std::map<std::string, std::pair<int, int>> rev_merge_;
};

struct fragment { bool is_token; int32_t token; size_t off, len; };

}  // namespace

std::vector<int32_t> vocab::tokenize(const std::string & raw, bool bos, bool special) const {
    std::vector<int32_t> out;
    if (bos && bos_id != -1) out.push_back(bos_id);
    if (raw.empty()) return out;
    std::vector<fragment> frags{ fragment{ false, -1, 0, raw.size() } };
    if (special) {
        for (const auto & st : special_tokens_cache) {                      // same (lexicographic) order as the reference
            std::vector<fragment> next;
            for (const fragment & fr : frags) {
                if (fr.is_token) { next.push_back(fr); continue; }
                size_t base = fr.off, len = fr.len;
                while (true) {
                    const size_t m = raw.find(st.first, base);
                    if (m == std::string::npos || m + st.first.size() > base + len) { next.push_back(fragment{ false, -1, base, len }); break; }
                    if (m > base) next.push_back(fragment{ false, -1, base, m - base });
                    next.push_back(fragment{ true, st.second, 0, 0 });
                    const size_t r = m + st.first.size();
                    if (r < base + len) { len = base + len - r; base = r; } else break;
                }
            }
            frags.swap(next);
        }
    }
    if (type != 0) throw std::runtime_error("BPE (gpt2) tokenisation of raw text is not implemented on this path; pass token ids");
    for (size_t i = 0; i < frags.size(); i++) {
        const fragment & fr = frags[i];
        if (fr.is_token) { out.push_back(fr.token); continue; }
        std::string text = raw.substr(fr.off, fr.len);
        if (i == 0) text = " " + text;                                      // leading space only when the first fragment is text
        replace_all(text, " ", "\xe2\x96\x81");
        spm_session(*this, text, out);
    }
    return out;
}

int vocab::token_to_piece(int32_t token, char * buf, int length) const {
    if (token < 0 || (size_t) token >= id_to_token.size()) return 0;
    const vocab_entry & e = id_to_token[token];
    if (type == 0) {
        if (e.type == 1) {
            std::string r = e.text;
            replace_all(r, "\xe2\x96\x81", " ");
            if (length < (int) r.size()) return -(int) r.size();
            memcpy(buf, r.data(), r.size());
            return (int) r.size();
        }
        if (e.type == 2) { if (length < 3) return -3; memcpy(buf, "\xe2\x96\x85", 3); return 3; }
        if (e.type == 6) { if (length < 1) return -1; buf[0] = (char) strtol(e.text.substr(3, 2).c_str(), nullptr, 16); return 1; }
        return 0;
    }
    if (e.type == 1) {                                                        // byte-level BPE: undo the GPT-2 byte<->unicode map
        std::string r;
        const std::string & t = e.text;
        for (size_t i = 0; i < t.size();) {
            const size_t l = utf8_len(t[i]);
            uint32_t cp = 0;
            if (l == 1) cp = (uint8_t) t[i];
            else if (l == 2) cp = ((uint8_t) t[i] & 0x1f) << 6 | ((uint8_t) t[i + 1] & 0x3f);
            else if (l == 3) cp = ((uint8_t) t[i] & 0x0f) << 12 | ((uint8_t) t[i + 1] & 0x3f) << 6 | ((uint8_t) t[i + 2] & 0x3f);
            i += l;
            // GPT-2 map: printable latin bytes map to themselves, the rest to 256 + k in order
            if ((cp >= 33 && cp <= 126) || (cp >= 161 && cp <= 172) || (cp >= 174 && cp <= 255)) r.push_back((char) cp);
            else if (cp >= 256) {
                int k = (int) cp - 256, b = -1;
                for (int c = 0; c < 256; c++) {
                    const bool keep = (c >= 33 && c <= 126) || (c >= 161 && c <= 172) || (c >= 174 && c <= 255);
                    if (!keep) { if (k == 0) { b = c; break; } k--; }
                }
                if (b >= 0) r.push_back((char) b);
            }
        }
        if (length < (int) r.size()) return -(int) r.size();
        memcpy(buf, r.data(), r.size());
        return (int) r.size();
    }
    return 0;
}

}  // namespace pi
