// Vocabulary + tokenizer (plumbing: prompt text <-> token ids on the host).
// Behaviour follows the reference: vocab loading llama.cpp:2387-2633, SPM tokenizer :5970-6093, special-token
// partitioning :6431-6524, llama_tokenize_internal :6526-6600, llama_token_to_piece :10112-10168.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "gguf.h"

namespace pi {

struct vocab_entry { std::string text; float score = 0.0f; int type = 1; };

struct vocab {
    int type = 0;                               // 0 = SPM, 1 = BPE
    std::vector<vocab_entry> id_to_token;
    std::unordered_map<std::string, int32_t> token_to_id;
    std::map<std::string, int32_t> special_tokens_cache;   // ordered like the reference's std::map
    std::map<std::pair<std::string, std::string>, int> bpe_ranks;
    int32_t bos_id = 1, eos_id = 2, unk_id = 0, sep_id = -1, pad_id = -1, linefeed_id = 13;
    int add_bos = -1, add_eos = -1;

    bool load(const gguf_file & f, std::string & err);
    std::vector<int32_t> tokenize(const std::string & text, bool bos, bool special) const;
    int  token_to_piece(int32_t token, char * buf, int length) const;
    int32_t byte_to_token(uint8_t ch) const;
};

}  // namespace pi
