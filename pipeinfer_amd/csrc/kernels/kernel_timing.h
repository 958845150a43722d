// Sampled kernel timing shared between the mat-vec launcher and the engine's graph cache (see mmvq.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

struct pi_kt_sample { hipEvent_t a, b; int type, nt; double bytes; };
struct pi_kt_count  { int type, nt; uint64_t count; };
struct pi_kt_sink {                       // the samples recorded while one graph was captured
    std::vector<pi_kt_sample> samples;
    std::vector<pi_kt_count>  launches;   // launches of each instantiation inside the graph
    int every = 16;
};
void pi_kt_set_sink(pi_kt_sink * sink);   // thread-local: launches on this thread go to `sink` (nullptr = eager mode)
bool pi_kt_enabled();
void pi_kt_accumulate(pi_kt_sink * sink, uint64_t replays);   // after a replay of the graph has completed
