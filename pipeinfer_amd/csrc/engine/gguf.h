// GGUF v2/v3 reader (layout: reference ggml.c:18083-18330; keys: llama.cpp:268-316).  The file stays mmapped; tensor data is
// handed to the device uploader straight from the mapping.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace pi {

struct gguf_value {
    uint32_t type = 0;                       // gguf value type id
    uint64_t u = 0; int64_t i = 0; double f = 0; bool b = false;
    std::string s;
    uint32_t arr_type = 0;
    std::vector<std::string> arr_s;
    const uint8_t * arr_data = nullptr;      // raw pointer into the mapping for numeric arrays
    uint64_t arr_n = 0;
};

struct gguf_tensor {
    std::string name;
    int      type = 0;
    int      n_dims = 0;
    int64_t  ne[4] = { 1, 1, 1, 1 };
    uint64_t offset = 0;                     // relative to the data section
    const uint8_t * data = nullptr;
    size_t   nbytes = 0;
};

class gguf_file {
public:
    ~gguf_file();
    bool open(const std::string & path, std::string & err);
    const gguf_value * find(const std::string & key) const;
    uint32_t    get_u32(const std::string & key, uint32_t dflt) const;
    float       get_f32(const std::string & key, float dflt) const;
    bool        get_bool(const std::string & key, bool dflt) const;
    std::string get_str(const std::string & key, const std::string & dflt) const;
    const gguf_tensor * tensor(const std::string & name) const;

    std::map<std::string, gguf_value> kv;
    struct raw_kv { std::string key; const uint8_t * begin; const uint8_t * end; };     // [type id + value] bytes in the mapping, in file order (the quantiser copies them through)
    std::vector<raw_kv> kv_raw;
    std::vector<gguf_tensor> tensors;
    uint32_t version = 0;
    size_t   file_size = 0;

private:
    void * map_ = nullptr;
    std::map<std::string, size_t> index_;
};

size_t type_row_size(int type, int64_t ne0);

}  // namespace pi
