#include "gguf.h"

#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace pi {

static const size_t k_scalar_size[13] = { 1, 1, 2, 2, 4, 4, 4, 1, 0, 0, 8, 8, 8 };   // u8 i8 u16 i16 u32 i32 f32 bool str arr u64 i64 f64

size_t type_row_size(int type, int64_t ne0) {
    switch (type) {
        case 0:  return (size_t) ne0 * 4;
        case 1:  return (size_t) ne0 * 2;
        case 2:  return (size_t) ne0 / 32 * 18;
        case 3:  return (size_t) ne0 / 32 * 20;      // Q4_1
        case 6:  return (size_t) ne0 / 32 * 22;      // Q5_0
        case 7:  return (size_t) ne0 / 32 * 24;      // Q5_1
        case 8:  return (size_t) ne0 / 32 * 34;
        case 10: return (size_t) ne0 / 256 * 84;
        case 11: return (size_t) ne0 / 256 * 110;
        case 12: return (size_t) ne0 / 256 * 144;
        case 13: return (size_t) ne0 / 256 * 176;
        case 14: return (size_t) ne0 / 256 * 210;
    }
    return 0;
}

gguf_file::~gguf_file() {
    if (map_) munmap(map_, file_size);
}

namespace {
struct reader {
    const uint8_t * p; const uint8_t * end; bool ok = true;
    template <typename T> T get() {
        T v{};
        if (p + sizeof(T) > end) { ok = false; return v; }
        memcpy(&v, p, sizeof(T)); p += sizeof(T);
        return v;
    }
    std::string str() {
        const uint64_t n = get<uint64_t>();
        if (!ok || p + n > end) { ok = false; return std::string(); }
        std::string s((const char *) p, (size_t) n); p += n;
        return s;
    }
};
}  // namespace

bool gguf_file::open(const std::string & path, std::string & err) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "cannot open " + path; return false; }
    struct stat st;
    fstat(fd, &st);
    file_size = (size_t) st.st_size;
    map_ = mmap(nullptr, file_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map_ == MAP_FAILED) { map_ = nullptr; err = "mmap failed for " + path; return false; }
    reader r{ (const uint8_t *) map_, (const uint8_t *) map_ + file_size };
    if (r.get<uint32_t>() != 0x46554747u) { err = "not a GGUF file: " + path; return false; }
    version = r.get<uint32_t>();
    if (version != 2 && version != 3) { err = "unsupported GGUF version " + std::to_string(version); return false; }   // v1 rejected like ggml.c:18124
    const uint64_t n_tensors = r.get<uint64_t>(), n_kv = r.get<uint64_t>();
    for (uint64_t i = 0; i < n_kv && r.ok; i++) {
        std::string key = r.str();
        gguf_value v;
        const uint8_t * const raw0 = r.p;
        v.type = r.get<uint32_t>();
        switch (v.type) {
            case 0: v.u = r.get<uint8_t>(); break;   case 1: v.i = r.get<int8_t>(); break;
            case 2: v.u = r.get<uint16_t>(); break;  case 3: v.i = r.get<int16_t>(); break;
            case 4: v.u = r.get<uint32_t>(); break;  case 5: v.i = r.get<int32_t>(); break;
            case 6: v.f = r.get<float>(); break;     case 7: v.b = r.get<uint8_t>() != 0; break;
            case 8: v.s = r.str(); break;
            case 9: {
                v.arr_type = r.get<uint32_t>(); v.arr_n = r.get<uint64_t>();
                if (v.arr_type == 8) { v.arr_s.reserve(v.arr_n); for (uint64_t k = 0; k < v.arr_n && r.ok; k++) v.arr_s.push_back(r.str()); }
                else if (v.arr_type < 13 && k_scalar_size[v.arr_type]) { v.arr_data = r.p; r.p += v.arr_n * k_scalar_size[v.arr_type]; if (r.p > r.end) r.ok = false; }
                else r.ok = false;
            } break;
            case 10: v.u = r.get<uint64_t>(); break; case 11: v.i = r.get<int64_t>(); break;
            case 12: v.f = r.get<double>(); break;
            default: r.ok = false;
        }
        if (v.type == 1 || v.type == 3 || v.type == 5 || v.type == 11) v.u = (uint64_t) v.i;
        if (v.type == 0 || v.type == 2 || v.type == 4 || v.type == 10) v.i = (int64_t) v.u;
        kv_raw.push_back({ key, raw0, r.p });
        kv.emplace(std::move(key), std::move(v));
    }
    tensors.resize(n_tensors);
    for (uint64_t i = 0; i < n_tensors && r.ok; i++) {
        gguf_tensor & t = tensors[i];
        t.name = r.str();
        t.n_dims = (int) r.get<uint32_t>();
        if (t.n_dims > 4) { r.ok = false; break; }
        for (int d = 0; d < t.n_dims; d++) t.ne[d] = (int64_t) r.get<uint64_t>();
        t.type = (int) r.get<uint32_t>();
        t.offset = r.get<uint64_t>();
    }
    if (!r.ok) { err = "corrupt GGUF header in " + path; return false; }
    const uint32_t align = get_u32("general.alignment", 32);
    size_t data_off = (size_t) (r.p - (const uint8_t *) map_);
    data_off = (data_off + align - 1) / align * align;
    for (size_t i = 0; i < tensors.size(); i++) {
        gguf_tensor & t = tensors[i];
        const size_t row = type_row_size(t.type, t.ne[0]);
        t.nbytes = row * (size_t) (t.ne[1] * t.ne[2] * t.ne[3]);
        t.data = (const uint8_t *) map_ + data_off + t.offset;
        if (row == 0) { err = "tensor " + t.name + " has unsupported type " + std::to_string(t.type); return false; }
        if (data_off + t.offset + t.nbytes > file_size) { err = "tensor " + t.name + " exceeds the file"; return false; }
        index_[t.name] = i;
    }
    return true;
}

const gguf_value * gguf_file::find(const std::string & key) const {
    auto it = kv.find(key);
    return it == kv.end() ? nullptr : &it->second;
}
uint32_t gguf_file::get_u32(const std::string & key, uint32_t dflt) const {
    const gguf_value * v = find(key);
    return v && (v->type <= 5 || v->type == 10 || v->type == 11) ? (uint32_t) v->u : dflt;
}
float gguf_file::get_f32(const std::string & key, float dflt) const {
    const gguf_value * v = find(key);
    return v && (v->type == 6 || v->type == 12) ? (float) v->f : dflt;
}
bool gguf_file::get_bool(const std::string & key, bool dflt) const {
    const gguf_value * v = find(key);
    return v && v->type == 7 ? v->b : dflt;
}
std::string gguf_file::get_str(const std::string & key, const std::string & dflt) const {
    const gguf_value * v = find(key);
    return v && v->type == 8 ? v->s : dflt;
}
const gguf_tensor * gguf_file::tensor(const std::string & name) const {
    auto it = index_.find(name);
    return it == index_.end() ? nullptr : &tensors[it->second];
}

}  // namespace pi
