// TEST INFRASTRUCTURE — see oracle.h.  KV-cell bookkeeping of the reference (llama.cpp:1292-1301, 1593-1765) and the
// PipeInfer attention-mask rule (llama.cpp:5200-5240), restated one operation at a time.  PINNED: tests/test_kv_cells.py runs random operation
// sequences through this file, the engine's table and the reference's own functions (llama.cpp compiled by ref_kv_harness.cpp into _ref/libref_kv.so);
// restated only: o_kv_pre_decode / o_kv_post_decode / o_kv_n / o_kv_build_mask, whose originals sit inside llama_decode_internal.
#include "oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t pos;        // -1 = free
    int32_t delta;
    int32_t nseq;       // population count of bits
    uint64_t * bits;    // std::set<llama_seq_id> as a bitset over [0, max_seq)
} o_cell;

struct o_kv_cache {
    int size, max_seq, words;
    int head, used, n;
    o_cell * cells;
    uint64_t * store;
};

static inline int  cell_has(const o_cell * c, int s)   { return (c->bits[s >> 6] >> (s & 63)) & 1; }
static inline void cell_insert(o_cell * c, int s)      { if (!cell_has(c, s)) { c->bits[s >> 6] |= 1ull << (s & 63); c->nseq++; } }
static inline void cell_erase(o_cell * c, int s)       { if (cell_has(c, s))  { c->bits[s >> 6] &= ~(1ull << (s & 63)); c->nseq--; } }
static inline void cell_clear(o_cell * c, int words)   { memset(c->bits, 0, (size_t) 8 * words); c->nseq = 0; }

o_kv_cache * o_kv_new(int n_ctx, int max_seq) {
    o_kv_cache * c = calloc(1, sizeof *c);
    c->size = n_ctx; c->max_seq = max_seq; c->words = (max_seq + 63) / 64;
    c->cells = calloc(n_ctx, sizeof(o_cell));
    c->store = calloc((size_t) n_ctx * c->words, 8);
    for (int i = 0; i < n_ctx; i++) { c->cells[i].pos = -1; c->cells[i].bits = c->store + (size_t) i * c->words; }
    return c;
}
void o_kv_free(o_kv_cache * c) { free(c->cells); free(c->store); free(c); }

int o_kv_head(const o_kv_cache * c) { return c->head; }
int o_kv_used(const o_kv_cache * c) { return c->used; }
int o_kv_cell_pos(const o_kv_cache * c, int i) { return c->cells[i].pos; }
int o_kv_cell_has(const o_kv_cache * c, int i, int s) { return cell_has(&c->cells[i], s); }
int o_kv_cell_nseq(const o_kv_cache * c, int i) { return c->cells[i].nseq; }

// llama_decode_internal_phased, llama.cpp:5548-5550
void o_kv_pre_decode(o_kv_cache * c, int n_tokens) {
    if ((uint32_t) c->head > (uint32_t) (c->used + 2 * n_tokens)) c->head = 0;
}

// llama_kv_cache_find_slot, llama.cpp:1593-1644 (first fit from head, wraps; writes pos + seq ids BEFORE compute)
int o_kv_find_slot(o_kv_cache * c, int n_tokens, const int32_t * pos, const int32_t * n_seq_id, const int32_t * seq_ids, int stride) {
    const uint32_t n_ctx = c->size;
    if ((uint32_t) n_tokens > n_ctx) return -1;
    uint32_t n_tested = 0, head = c->head;
    while (1) {
        if (head + n_tokens > n_ctx) { n_tested += n_ctx - head; head = 0; continue; }
        int found = 1;
        for (int i = 0; i < n_tokens; i++) {
            if (c->cells[head + i].pos >= 0) { found = 0; head += i + 1; n_tested += i + 1; break; }
        }
        if (found) break;
        if (n_tested >= n_ctx) { c->head = head; return -1; }
    }
    c->head = head;
    for (int i = 0; i < n_tokens; i++) {
        c->cells[head + i].pos = pos[i];
        for (int j = 0; j < n_seq_id[i]; j++) cell_insert(&c->cells[head + i], seq_ids[(size_t) i * stride + j]);
    }
    c->used += n_tokens;
    return head;
}

// llama.cpp:5708-5713
void o_kv_post_decode(o_kv_cache * c, int n_tokens) {
    c->head += n_tokens;
    if (c->head >= c->size) c->head = 0;
}

// llama_kv_cache_cell_max, llama.cpp:1647-1655 (note: never inspects cell 0)
int o_kv_cell_max(const o_kv_cache * c) {
    for (int i = c->size - 1; i > 0; --i) if (c->cells[i].pos >= 0 && c->cells[i].nseq > 0) return i + 1;
    return 0;
}

// llama.cpp:5571
int o_kv_n(const o_kv_cache * c) {
    int m = o_kv_cell_max(c); if (m < 32) m = 32;
    return m < c->size ? m : c->size;
}

// llama.cpp:1657-1664
void o_kv_clear(o_kv_cache * c) {
    for (int i = 0; i < c->size; i++) { c->cells[i].pos = -1; cell_clear(&c->cells[i], c->words); }
    c->head = 0; c->used = 0;
}

// llama_kv_cache_seq_rm, llama.cpp:1666-1697 — PipeInfer's predicate also visits free cells (pos < 0), :1677
void o_kv_seq_rm(o_kv_cache * c, int seq_id, int p0, int p1) {
    uint32_t new_head = c->size;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT_MAX;
    for (int i = 0; i < c->size; ++i) {
        o_cell * cell = &c->cells[i];
        if ((cell->pos >= p0 || cell->pos < 0) && cell->pos < p1) {
            if (seq_id < 0) cell_clear(cell, c->words);
            else cell_erase(cell, seq_id);
            if (cell->nseq == 0) {
                if (cell->pos >= 0) c->used--;
                cell->pos = -1;
                if (new_head == (uint32_t) c->size) new_head = i;
            }
        }
    }
    if (new_head != (uint32_t) c->size && new_head < (uint32_t) c->head) c->head = new_head;
}

// llama_kv_cache_seq_cp, llama.cpp:1699-1715 (resets head to 0)
void o_kv_seq_cp(o_kv_cache * c, int src, int dst, int p0, int p1) {
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT_MAX;
    c->head = 0;
    for (int i = 0; i < c->size; ++i) {
        o_cell * cell = &c->cells[i];
        if (cell_has(cell, src) && cell->pos >= p0 && cell->pos < p1) cell_insert(cell, dst);
    }
}

// llama_kv_cache_seq_keep, llama.cpp:1717-1734
void o_kv_seq_keep(o_kv_cache * c, int seq_id) {
    uint32_t new_head = c->size;
    for (int i = 0; i < c->size; ++i) {
        o_cell * cell = &c->cells[i];
        if (!cell_has(cell, seq_id)) {
            if (cell->pos >= 0) c->used--;
            cell->pos = -1; cell_clear(cell, c->words);
            if (new_head == (uint32_t) c->size) new_head = i;
        } else {
            cell_clear(cell, c->words); cell_insert(cell, seq_id);
        }
    }
    if (new_head != (uint32_t) c->size && new_head < (uint32_t) c->head) c->head = new_head;
}

// KQ_mask fill, llama.cpp:5200-5240: cell i is visible to token j iff ANY of the token's seq ids is on the cell
// AND cell.pos <= token pos; otherwise -INFINITY.
void o_kv_build_mask(const o_kv_cache * c, int n_kv, int n_tokens, const int32_t * pos, const int32_t * n_seq_id,
                     const int32_t * seq_ids, int stride, float * mask) {
    for (int j = 0; j < n_tokens; ++j) {
        for (int i = 0; i < n_kv; ++i) {
            int has = 0;
            for (int s = 0; s < n_seq_id[j] && !has; s++) has = cell_has(&c->cells[i], seq_ids[(size_t) j * stride + s]);
            mask[(size_t) j * n_kv + i] = (!has || c->cells[i].pos > pos[j]) ? -__builtin_inff() : 0.0f;
        }
    }
}
