#include "vocab.h"
#include "unicode_tables.h"

#include <algorithm>

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

// ---- byte-level BPE (gpt2 vocabularies: Falcon) ---------------------------------------------------------------------------------------
// llm_tokenizer_bpe (llama.cpp:6117-6247) with its GPT-2 style pre-tokenizer bpe_gpt2_preprocess (:6249-6395).  The pre-tokenizer is a hand
// written approximation of the regex  's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+  — its exact splits (including
// the places where it differs from the regex) decide the token ids, so the state machine below follows it decision by decision.
enum cp_class { CP_OTHER = 0, CP_DIGIT, CP_LETTER, CP_WHITESPACE };
template <size_t N> static bool in_ranges(const uint32_t (&r)[N][2], uint32_t cp) {
    size_t lo = 0, hi = N;
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (cp > r[mid][1]) lo = mid + 1; else hi = mid; }
    return lo < N && cp >= r[lo][0];
}
static cp_class classify(uint32_t cp) {
    if (in_ranges(pi_unicode_letter, cp)) return CP_LETTER;
    if (in_ranges(pi_unicode_digit, cp)) return CP_DIGIT;
    if (in_ranges(pi_unicode_whitespace, cp)) return CP_WHITESPACE;
    return CP_OTHER;
}
static uint32_t decode_utf8(const std::string & s, size_t & off) {               // unicode.h:259-291 on well-formed input; malformed bytes pass through one by one
    const uint8_t c = (uint8_t) s[off];
    const size_t left = s.size() - off;
    if (c < 0x80 || left < 2 || (c & 0xE0) == 0x80) { off += 1; return c; }
    if ((c & 0xE0) == 0xC0) { const uint32_t r = (c & 0x1Fu) << 6 | ((uint8_t) s[off + 1] & 0x3Fu); off += 2; return r; }
    if ((c & 0xF0) == 0xE0 && left >= 3) { const uint32_t r = (c & 0x0Fu) << 12 | ((uint8_t) s[off + 1] & 0x3Fu) << 6 | ((uint8_t) s[off + 2] & 0x3Fu); off += 3; return r; }
    if ((c & 0xF8) == 0xF0 && left >= 4) {
        const uint32_t r = (c & 0x07u) << 18 | ((uint8_t) s[off + 1] & 0x3Fu) << 12 | ((uint8_t) s[off + 2] & 0x3Fu) << 6 | ((uint8_t) s[off + 3] & 0x3Fu);
        off += 4; return r;
    }
    off += 1; return c;
}
static std::string encode_utf8(uint32_t cp) {
    std::string r;
    if (cp < 0x80) r.push_back((char) cp);
    else if (cp < 0x800) { r.push_back((char) (0xC0 | cp >> 6)); r.push_back((char) (0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { r.push_back((char) (0xE0 | cp >> 12)); r.push_back((char) (0x80 | ((cp >> 6) & 0x3F))); r.push_back((char) (0x80 | (cp & 0x3F))); }
    else { r.push_back((char) (0xF0 | cp >> 18)); r.push_back((char) (0x80 | ((cp >> 12) & 0x3F))); r.push_back((char) (0x80 | ((cp >> 6) & 0x3F))); r.push_back((char) (0x80 | (cp & 0x3F))); }
    return r;
}
// GPT-2's byte -> printable unicode map (unicode.h:405-427): printable latin-1 bytes stand for themselves, the others become U+0100 + k
static const std::string & byte_to_unicode(uint8_t b) {
    static const std::vector<std::string> table = [] {
        std::vector<std::string> t(256);
        int k = 0;
        for (int c = 0; c < 256; c++) {
            const bool keep = (c >= 33 && c <= 126) || (c >= 161 && c <= 172) || (c >= 174 && c <= 255);
            t[c] = encode_utf8(keep ? (uint32_t) c : (uint32_t) (256 + k++));
        }
        return t;
    }();
    return table[b];
}

static std::vector<std::string> bpe_pre_tokenize(const std::string & text) {
    std::vector<std::string> ch; std::vector<cp_class> cls;
    for (size_t off = 0; off < text.size();) { const uint32_t cp = decode_utf8(text, off); ch.push_back(encode_utf8(cp)); cls.push_back(classify(cp)); }
    const int n = (int) ch.size();
    std::vector<std::string> words;
    std::string token;
    enum { NONE, LETTERS, DIGITS, SPECIALS, SPACES_AHEAD } collecting = NONE;
    static const std::string empty;
    for (int i = 0; i < n; i++) {
        const std::string & c = ch[i], & c1 = i + 1 < n ? ch[i + 1] : empty, & c2 = i + 2 < n ? ch[i + 2] : empty;
        const cp_class t = cls[i], t1 = i + 1 < n ? cls[i + 1] : CP_OTHER;
        if (c == "'" && n - i >= 2 && (c1 == "s" || c1 == "t" || c1 == "m" || c1 == "d")) {                 // 's 't 'm 'd  (the collecting state is NOT reset: :6282-6293)
            if (!token.empty()) words.push_back(token);
            words.push_back(c + c1); token.clear(); i += 1; continue;
        }
        if (c == "'" && n - i >= 3 && ((c1 == "r" && c2 == "e") || (c1 == "v" && c2 == "e") || (c1 == "l" && c2 == "l"))) {    // 're 've 'll
            if (!token.empty()) words.push_back(token);
            words.push_back(c + c1 + c2); token.clear(); i += 2; continue;
        }
        bool split = false;
        if (collecting == NONE) {
            const bool lead_space = token.empty() && c == " ";                                               // " ?" of the regex alternatives
            if (t == CP_LETTER || (lead_space && t1 == CP_LETTER)) collecting = LETTERS;
            else if (t == CP_DIGIT || (lead_space && t1 == CP_DIGIT)) collecting = DIGITS;
            else if (t == CP_OTHER || (lead_space && t1 == CP_OTHER)) collecting = SPECIALS;                // (no next character counts as "other", like the reference's empty lookup)
            else if (t == CP_WHITESPACE && t1 == CP_WHITESPACE && i + 1 < n) collecting = SPACES_AHEAD;
            else if (t == CP_WHITESPACE) split = true;
        } else if ((collecting == LETTERS && t != CP_LETTER) || (collecting == DIGITS && t != CP_DIGIT) ||
                   (collecting == SPECIALS && t != CP_OTHER) || (collecting == SPACES_AHEAD && i + 1 < n && (t1 == CP_LETTER || t1 == CP_DIGIT))) {
            split = true;
        }
        if (i + 1 == n) { split = true; token += c; }                                                       // final character: it joins the token that is pushed now
        if (split) {
            if (!token.empty()) words.push_back(token);
            token = c; collecting = NONE;
        } else token += c;
    }
    for (std::string & w : words) { std::string e; for (char b : w) e += byte_to_unicode((uint8_t) b); w.swap(e); }
    return words;
}

struct bpe_bigram { int left, right; std::string text; int rank; };
struct bpe_bigram_after { bool operator()(const bpe_bigram & l, const bpe_bigram & r) const { return l.rank > r.rank || (l.rank == r.rank && l.left > r.left); } };

static void bpe_tokenize(const vocab & v, const std::string & text, std::vector<int32_t> & out) {
    for (const std::string & word : bpe_pre_tokenize(text)) {
        struct sym { size_t off, n; int prev, next; };
        std::vector<sym> syms;
        for (size_t off = 0; off < word.size();) {
            const size_t len = std::min(word.size() - off, utf8_len(word[off]));
            syms.push_back(sym{ off, len, (int) syms.size() - 1, -1 });
            off += len;
        }
        for (size_t i = 0; i + 1 < syms.size(); i++) syms[i].next = (int) i + 1;
        std::priority_queue<bpe_bigram, std::vector<bpe_bigram>, bpe_bigram_after> queue;
        auto str = [&](int i) { return word.substr(syms[i].off, syms[i].n); };
        auto try_add = [&](int l, int r) {
            if (l < 0 || r < 0) return;
            const std::string a = str(l), b = str(r);
            const auto it = v.bpe_ranks.find(std::make_pair(a, b));
            if (it != v.bpe_ranks.end()) queue.push(bpe_bigram{ l, r, a + b, it->second });
        };
        for (int i = 1; i < (int) syms.size(); i++) try_add(i - 1, i);
        while (!queue.empty()) {
            const bpe_bigram bg = queue.top(); queue.pop();
            sym & L = syms[bg.left], & R = syms[bg.right];
            if (L.n == 0 || R.n == 0 || str(bg.left) + str(bg.right) != bg.text) continue;                  // outdated
            L.n += R.n; R.n = 0;                                                                             // symbols of a word are contiguous: merging extends the left one
            L.next = R.next;
            if (R.next >= 0) syms[R.next].prev = bg.left;
            try_add(L.prev, bg.left);
            try_add(bg.left, L.next);
        }
        for (const sym & sy : syms) {
            if (sy.n == 0) continue;
            const std::string piece = word.substr(sy.off, sy.n);
            const auto it = v.token_to_id.find(piece);
            if (it != v.token_to_id.end()) { out.push_back(it->second); continue; }
            for (char b : piece) {                                                                           // unknown piece: its bytes, one token each (:6206-6214)
                const auto ib = v.token_to_id.find(std::string(1, b));
                if (ib == v.token_to_id.end()) throw std::runtime_error("ERROR: byte not found in vocab");
                out.push_back(ib->second);
            }
        }
    }
}

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
    if (type != 0) {                                                        // llama_tokenize_internal, LLAMA_VOCAB_TYPE_BPE arm (llama.cpp:6575-6594)
        for (const fragment & fr : frags) {
            if (fr.is_token) out.push_back(fr.token); else bpe_tokenize(*this, raw.substr(fr.off, fr.len), out);
        }
        return out;
    }
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
