#include "kv_cells.h"

#include <algorithm>
#include <climits>
#include <cstring>

namespace pi {

kv_cells::kv_cells(int n_ctx) : size_(n_ctx), pos_(n_ctx, -1), nseq_(n_ctx, 0), bits_((size_t) n_ctx, 0) {}

void kv_cells::ensure_seq(int seq) {
    if (seq < cap_) return;
    int cap = cap_;
    while (cap <= seq) cap *= 2;
    const int words = cap / 64;
    std::vector<uint64_t> nb((size_t) size_ * words, 0);
    for (int i = 0; i < size_; i++) memcpy(&nb[(size_t) i * words], &bits_[(size_t) i * words_], (size_t) words_ * 8);
    bits_.swap(nb);
    cap_ = cap; words_ = words;
}

void kv_cells::clear_cell(int i) {
    memset(&bits_[(size_t) i * words_], 0, (size_t) words_ * 8);
    nseq_[i] = 0;
}

int kv_cells::popcount(const uint64_t * w) const {
    int n = 0;
    for (int k = 0; k < words_; k++) n += __builtin_popcountll(w[k]);
    return n;
}

void kv_cells::set_range(uint64_t * w, int first, int last) const {
    for (int k = first >> 6; k <= last >> 6; k++) {
        const int lo = std::max(first, k * 64) & 63, hi = std::min(last, k * 64 + 63) & 63;
        const uint64_t m = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1)) & ~((1ull << lo) - 1);
        w[k] |= m;
    }
}

void kv_cells::clr_range(uint64_t * w, int first, int last) const {
    for (int k = first >> 6; k <= last >> 6; k++) {
        const int lo = std::max(first, k * 64) & 63, hi = std::min(last, k * 64 + 63) & 63;
        const uint64_t m = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1)) & ~((1ull << lo) - 1);
        w[k] &= ~m;
    }
}

int kv_cells::first_free() const {
    for (int i = 0; i < size_; i++) if (pos_[i] < 0) return i;
    return size_;
}

void kv_cells::pre_decode(int n_tokens) {
    if ((uint32_t) head_ > (uint32_t) (used_ + 2 * n_tokens)) head_ = 0;
}

int kv_cells::find_slot(int n_tokens, const int32_t * pos, const int32_t * n_seq_id, int32_t * const * seq_id) {
    const uint32_t n_ctx = (uint32_t) size_;
    if ((uint32_t) n_tokens > n_ctx) return -1;
    uint32_t n_tested = 0, head = (uint32_t) head_;
    while (true) {
        if (head + n_tokens > n_ctx) { n_tested += n_ctx - head; head = 0; continue; }
        bool found = true;
        for (int i = 0; i < n_tokens; i++) {
            if (pos_[head + i] >= 0) { found = false; head += i + 1; n_tested += i + 1; break; }
        }
        if (found) break;
        if (n_tested >= n_ctx) { head_ = (int) head; return -1; }
    }
    head_ = (int) head;
    for (int i = 0; i < n_tokens; i++) {
        const int c = head_ + i;
        pos_[c] = pos[i];
        for (int j = 0; j < n_seq_id[i]; j++) {
            const int s = seq_id[i][j];
            ensure_seq(s);
            uint64_t & w = bits_[(size_t) c * words_ + (s >> 6)];
            if (!((w >> (s & 63)) & 1)) { w |= 1ull << (s & 63); nseq_[c]++; }
        }
    }
    used_ += n_tokens;
    return head_;
}

void kv_cells::post_decode(int n_tokens) {
    head_ += n_tokens;
    if (head_ >= size_) head_ = 0;
}

int kv_cells::cell_max() const {          // never looks at cell 0, like the reference loop `for (i = size-1; i > 0; --i)`
    for (int i = size_ - 1; i > 0; --i) if (pos_[i] >= 0 && nseq_[i] > 0) return i + 1;
    return 0;
}

int kv_cells::n_attend() const { return std::min(size_, std::max(32, cell_max())); }

void kv_cells::clear() {
    for (int i = 0; i < size_; i++) { pos_[i] = -1; clear_cell(i); }
    head_ = 0; used_ = 0;
}

void kv_cells::seq_rm(int seq_id, int p0, int p1) {
    uint32_t new_head = (uint32_t) size_;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT_MAX;
    for (int i = 0; i < size_; ++i) {
        if ((pos_[i] >= p0 || pos_[i] < 0) && pos_[i] < p1) {
            if (seq_id < 0) clear_cell(i);
            else if (has(i, seq_id)) { bits_[(size_t) i * words_ + (seq_id >> 6)] &= ~(1ull << (seq_id & 63)); nseq_[i]--; }
            if (nseq_[i] == 0) {
                if (pos_[i] >= 0) used_--;
                pos_[i] = -1;
                if (new_head == (uint32_t) size_) new_head = (uint32_t) i;
            }
        }
    }
    if (new_head != (uint32_t) size_ && new_head < (uint32_t) head_) head_ = (int) new_head;
}

void kv_cells::release_cells(int first, int n) {
    for (int i = first; i < first + n && i < size_; i++) {
        if (i < 0) continue;
        clear_cell(i);
        if (pos_[i] >= 0) used_--;
        pos_[i] = -1;
    }
    if (first >= 0 && first < head_) head_ = first;
}

void kv_cells::seq_cp(int src, int dst, int p0, int p1) {
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT_MAX;
    head_ = 0;
    ensure_seq(dst);
    for (int i = 0; i < size_; ++i) {
        if (has(i, src) && pos_[i] >= p0 && pos_[i] < p1 && !has(i, dst)) {
            bits_[(size_t) i * words_ + (dst >> 6)] |= 1ull << (dst & 63);
            nseq_[i]++;
        }
    }
}

void kv_cells::seq_keep(int seq_id) {
    uint32_t new_head = (uint32_t) size_;
    ensure_seq(seq_id);
    for (int i = 0; i < size_; ++i) {
        if (!has(i, seq_id)) {
            if (pos_[i] >= 0) used_--;
            pos_[i] = -1; clear_cell(i);
            if (new_head == (uint32_t) size_) new_head = (uint32_t) i;
        } else {
            clear_cell(i);
            bits_[(size_t) i * words_ + (seq_id >> 6)] |= 1ull << (seq_id & 63);
            nseq_[i] = 1;
        }
    }
    if (new_head != (uint32_t) size_ && new_head < (uint32_t) head_) head_ = (int) new_head;
}

// Closed form of `for s in [first,last]: seq_rm(s, p0, p1)`.
// Every seq_rm visits the cells in range and ALL free cells; its new_head is the first visited cell that is empty
// afterwards.  Cells only ever become emptier, so the minimum over the loop is the first free cell at the end.
void kv_cells::seq_rm_range(int first, int last, int p0, int p1) {
    if (first > last) return;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT_MAX;
    ensure_seq(last);
    for (int i = 0; i < size_; ++i) {
        if (pos_[i] >= p0 && pos_[i] < p1) {
            uint64_t * w = &bits_[(size_t) i * words_];
            clr_range(w, first, last);
            nseq_[i] = popcount(w);
            if (nseq_[i] == 0) { used_--; pos_[i] = -1; }
        }
    }
    const int ff = first_free();
    if (ff != size_ && ff < head_) head_ = ff;
}

void kv_cells::seq_cp_range(int src, int first, int last, int p0, int p1) {
    if (first > last) return;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT_MAX;
    head_ = 0;
    ensure_seq(last);
    for (int i = 0; i < size_; ++i) {
        if (has(i, src) && pos_[i] >= p0 && pos_[i] < p1) {
            uint64_t * w = &bits_[(size_t) i * words_];
            set_range(w, first, last);
            nseq_[i] = popcount(w);
        }
    }
}

// Closed form of `for s in [first,last]: { seq_rm(s, p0, p1); seq_cp(src, s, p0, p1); }` with src outside the range:
// a cell in range that carries src keeps it throughout, so each s is erased and re-inserted (ends with all of
// [first,last]); a cell in range without src just loses [first,last] and is freed when that empties it (a freed cell has
// pos = -1 and can never match a later seq_cp).  The last operation is a seq_cp, hence head = 0.
void kv_cells::seq_rm_cp_range(int src, int first, int last, int p0, int p1) {
    if (first > last) return;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT_MAX;
    ensure_seq(last);
    for (int i = 0; i < size_; ++i) {
        if (pos_[i] >= p0 && pos_[i] < p1) {
            uint64_t * w = &bits_[(size_t) i * words_];
            if (has(i, src)) set_range(w, first, last); else clr_range(w, first, last);
            nseq_[i] = popcount(w);
            if (nseq_[i] == 0) { used_--; pos_[i] = -1; }
        }
    }
    head_ = 0;
}

void kv_cells::build_mask(int n_kv, int n_tokens, const int32_t * pos, const int32_t * n_seq_id, int32_t * const * seq_id, uint32_t * out, int words32) const {
    if (words32 <= 0) words32 = (n_kv + 31) / 32;          // row stride of the bitmap (the engine uses a fixed stride of n_ctx bits)
    memset(out, 0, (size_t) n_tokens * words32 * 4);
    std::vector<uint64_t> tmask((size_t) words_);
    for (int j = 0; j < n_tokens; j++) {
        uint32_t * row = out + (size_t) j * words32;
        const int ns = n_seq_id[j];
        if (ns <= 8) {
            for (int c = 0; c < n_kv; c++) {
                if (pos_[c] > pos[j] || nseq_[c] == 0) continue;
                bool vis = false;
                for (int s = 0; s < ns && !vis; s++) vis = has(c, seq_id[j][s]);
                if (vis) row[c >> 5] |= 1u << (c & 31);
            }
        } else {
            std::fill(tmask.begin(), tmask.end(), 0);
            for (int s = 0; s < ns; s++) { const int id = seq_id[j][s]; if (id < cap_) tmask[id >> 6] |= 1ull << (id & 63); }
            for (int c = 0; c < n_kv; c++) {
                if (pos_[c] > pos[j] || nseq_[c] == 0) continue;
                const uint64_t * w = &bits_[(size_t) c * words_];
                bool vis = false;
                for (int k = 0; k < words_ && !vis; k++) vis = (w[k] & tmask[k]) != 0;
                if (vis) row[c >> 5] |= 1u << (c & 31);
            }
        }
    }
}

int kv_cells::token_count() const {
    int n = 0;
    for (int i = 0; i < size_; i++) n += nseq_[i];
    return n;
}

}  // namespace pi
