// Host-side KV-cell table: which cache cell holds which position for which sequences.
//
// Semantics follow the reference one operation at a time (llama_kv_cell / llama_kv_cache, llama.cpp:1292-1330;
// find_slot :1593-1644; cell_max :1647-1655; clear :1657-1664; seq_rm :1666-1697 with PipeInfer's predicate that also
// visits free cells; seq_cp :1699-1715 which resets head; seq_keep :1717-1734) because cell placement decides what
// attention reads.  std::set<llama_seq_id> becomes a bitset (PipeInfer tags every canonical cell with ~3000 seq ids).
//
// The *_range operations are exact closed forms of the loops the PipeInfer driver runs over all sequence ids
// (speculative.cpp:603-612, 1252-1268): same final cells, `used` and `head` as issuing the single operations in order
// (proved in DESIGN.md, checked against the single-op loop by tests/test_kv_cells.py).
#pragma once

#include <cstdint>
#include <vector>

namespace pi {

class kv_cells {
public:
    explicit kv_cells(int n_ctx);

    int  size() const { return size_; }
    int  head() const { return head_; }
    int  used() const { return used_; }
    int  pos(int i) const { return pos_[i]; }
    bool has(int i, int seq) const { return seq < cap_ && ((bits_[(size_t) i * words_ + (seq >> 6)] >> (seq & 63)) & 1); }
    int  nseq(int i) const { return nseq_[i]; }

    void pre_decode(int n_tokens);                      // llama.cpp:5548-5550
    // returns the first cell of the slot (== head) or -1 when no contiguous free run exists
    int  find_slot(int n_tokens, const int32_t * pos, const int32_t * n_seq_id, int32_t * const * seq_id);
    void post_decode(int n_tokens);                     // llama.cpp:5708-5713
    int  cell_max() const;
    int  n_attend() const;                              // kv_self.n, llama.cpp:5571
    void clear();
    void seq_rm(int seq_id, int p0, int p1);
    void seq_cp(int src, int dst, int p0, int p1);
    void seq_keep(int seq_id);
    // gives the cells [first, first + n) back exactly (a run that was allotted them could not be started): no other cell is touched
    void release_cells(int first, int n);

    // == for (s = first; s <= last; s++) seq_rm(s, p0, p1)
    void seq_rm_range(int first, int last, int p0, int p1);
    // == for (s = first; s <= last; s++) seq_cp(src, s, p0, p1)          (src outside [first,last] or not: same result)
    void seq_cp_range(int src, int first, int last, int p0, int p1);
    // == for (s = first; s <= last; s++) { seq_rm(s, p0, p1); seq_cp(src, s, p0, p1); }      requires src outside [first,last]
    void seq_rm_cp_range(int src, int first, int last, int p0, int p1);

    // attention visibility (KQ_mask rule, llama.cpp:5200-5240): bit c of row j <=> cell c carries ANY seq id of token j
    // and cell.pos <= pos[j].  out: [n_tokens][words32] with words32 = ceil(n_kv / 32)
    void build_mask(int n_kv, int n_tokens, const int32_t * pos, const int32_t * n_seq_id, int32_t * const * seq_id, uint32_t * out, int words32 = 0) const;

    int  token_count() const;                           // llama_get_kv_cache_token_count, llama.cpp:9238-9247

private:
    void ensure_seq(int seq);
    void clear_cell(int i);
    void set_range(uint64_t * w, int first, int last) const;
    void clr_range(uint64_t * w, int first, int last) const;
    int  popcount(const uint64_t * w) const;
    int  first_free() const;

    int size_, head_ = 0, used_ = 0;
    int cap_ = 64, words_ = 1;                          // sequence-id capacity (bits) and 64-bit words per cell
    std::vector<int32_t>  pos_, nseq_;
    std::vector<uint64_t> bits_;
};

}  // namespace pi
