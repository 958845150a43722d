// Command line of the reference's examples/quantize (quantize.cpp:70-200: `quantize [--allow-requantize] [--leave-output-tensor] [--pure]
// model-f32.gguf [model-quant.gguf] type [nthreads]`, type by name or by llama_ftype id) on top of include/llama.h's llama_model_quantize.
#include "llama.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static const struct { const char * name; enum llama_ftype ftype; } k_types[] = {
    { "Q4_0", LLAMA_FTYPE_MOSTLY_Q4_0 }, { "Q4_1", LLAMA_FTYPE_MOSTLY_Q4_1 }, { "Q5_0", LLAMA_FTYPE_MOSTLY_Q5_0 }, { "Q5_1", LLAMA_FTYPE_MOSTLY_Q5_1 }, { "Q8_0", LLAMA_FTYPE_MOSTLY_Q8_0 },
    { "Q2_K", LLAMA_FTYPE_MOSTLY_Q2_K }, { "Q3_K_S", LLAMA_FTYPE_MOSTLY_Q3_K_S }, { "Q3_K_M", LLAMA_FTYPE_MOSTLY_Q3_K_M }, { "Q3_K", LLAMA_FTYPE_MOSTLY_Q3_K_M }, { "Q3_K_L", LLAMA_FTYPE_MOSTLY_Q3_K_L },
    { "Q4_K_S", LLAMA_FTYPE_MOSTLY_Q4_K_S }, { "Q4_K_M", LLAMA_FTYPE_MOSTLY_Q4_K_M }, { "Q4_K", LLAMA_FTYPE_MOSTLY_Q4_K_M }, { "Q5_K_S", LLAMA_FTYPE_MOSTLY_Q5_K_S }, { "Q5_K_M", LLAMA_FTYPE_MOSTLY_Q5_K_M },
    { "Q5_K", LLAMA_FTYPE_MOSTLY_Q5_K_M }, { "Q6_K", LLAMA_FTYPE_MOSTLY_Q6_K }, { "F16", LLAMA_FTYPE_MOSTLY_F16 }, { "F32", LLAMA_FTYPE_ALL_F32 }, { "COPY", LLAMA_FTYPE_ALL_F32 },
};

static bool parse_type(const std::string & s, enum llama_ftype & out, std::string & name) {
    for (const auto & t : k_types) if (s == t.name) { out = t.ftype; name = t.name; return true; }
    char * end = NULL;
    const long id = strtol(s.c_str(), &end, 10);
    if (end == s.c_str() || *end) return false;
    for (const auto & t : k_types) if ((long) t.ftype == id) { out = t.ftype; name = t.name; return true; }
    return false;
}

int main(int argc, char ** argv) {
    llama_model_quantize_params p = llama_model_quantize_default_params();
    int a = 1;
    for (; a < argc && strncmp(argv[a], "--", 2) == 0; a++) {
        if (!strcmp(argv[a], "--leave-output-tensor")) p.quantize_output_tensor = false;
        else if (!strcmp(argv[a], "--allow-requantize")) p.allow_requantize = true;
        else if (!strcmp(argv[a], "--pure")) p.pure = true;
        else { fprintf(stderr, "unknown option %s\n", argv[a]); return 1; }
    }
    if (argc - a < 2) { fprintf(stderr, "usage: %s [--allow-requantize] [--leave-output-tensor] [--pure] model-f32.gguf [model-quant.gguf] type [nthreads]\n", argv[0]); return 1; }
    const std::string in = argv[a++];
    std::string out, type_name;
    if (!parse_type(argv[a], p.ftype, type_name)) {                        // the output name is optional: default ggml-model-<TYPE>.gguf beside the input
        out = argv[a++];
        if (a >= argc || !parse_type(argv[a], p.ftype, type_name)) { fprintf(stderr, "%s: invalid ftype\n", argv[0]); return 1; }
    } else {
        const size_t slash = in.find_last_of('/');
        out = (slash == std::string::npos ? std::string() : in.substr(0, slash + 1)) + "ggml-model-" + type_name + ".gguf";
    }
    if (type_name == "COPY") p.only_copy = true;
    a++;
    if (a < argc) p.nthread = atoi(argv[a]);
    fprintf(stderr, "%s: quantizing '%s' to '%s' as %s\n", argv[0], in.c_str(), out.c_str(), type_name.c_str());
    llama_backend_init(false);
    const int rc = llama_model_quantize(in.c_str(), out.c_str(), &p);
    llama_backend_free();
    if (rc) { fprintf(stderr, "%s: failed to quantize model from '%s'\n", argv[0], in.c_str()); return 1; }
    return 0;
}
