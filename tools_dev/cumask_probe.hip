// Do CU masks (hipExtStreamCreateWithCUMask) let a chain of small kernels run BESIDE kernels that fill the rest of the chip?  (the draft step on one GPU.)
// Stream A: back-to-back "target" kernels, each 2048 workgroups x 256 threads that spin ~20 us per workgroup (8 waves per CU-round => ~40-80 us per kernel, like a 70B mat-vec).
// Stream B: a dependent chain of 200 one-workgroup kernels (~3 us each alone), like a TinyLlama decode step.
// Every workgroup stamps s_memrealtime (100 MHz) at entry / exit and the hardware id of where it ran, so overlap is read from TIMESTAMPS, per kernel:
//   (a) no masks            (b) disjoint masks on both streams, eager launches            (c) as (b), chain B replayed as a captured graph
// build: hipcc --offload-arch=gfx950 -O3 tools_dev/cumask_probe.hip -o tools_dev/cumask_probe ; run: tools_dev/cumask_probe [cus_for_B=32]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <set>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct stamp { unsigned long long t0, t1; unsigned hw, xcc; };
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }
__global__ void __launch_bounds__(256) k_big(stamp * out, int kidx, int spin_ticks) {
    const unsigned long long t0 = now();
    while (now() - t0 < (unsigned long long) spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        stamp s; s.t0 = t0; s.t1 = now();
        s.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4); s.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[(size_t) kidx * gridDim.x + blockIdx.x] = s;
    }
}
__global__ void __launch_bounds__(64) k_small(stamp * out, int kidx, int spin_ticks, const int * dep, int * done) {
    const unsigned long long t0 = now();
    int v = dep ? *dep : 0;                                          // the chain: read what the previous kernel wrote
    while (now() - t0 < (unsigned long long) spin_ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) {
        stamp s; s.t0 = t0; s.t1 = now();
        s.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4); s.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[kidx] = s; *done = v + 1;
    }
}

static void summarize(const char * name, const std::vector<stamp> & big, int n_big, int wg_big, const std::vector<stamp> & small, int n_small) {
    // chain B: total, and how much of it lay inside some A kernel's span
    unsigned long long b0 = small[0].t0, b1 = small[n_small - 1].t1;
    double gap_sum = 0; for (int i = 1; i < n_small; i++) gap_sum += (double) (small[i].t0 - small[i - 1].t1);
    unsigned long long a0 = ~0ull, a1 = 0;
    std::vector<std::pair<unsigned long long, unsigned long long>> spans(n_big);
    for (int k = 0; k < n_big; k++) {
        unsigned long long s0 = ~0ull, s1 = 0;
        for (int w = 0; w < wg_big; w++) { const stamp & s = big[(size_t) k * wg_big + w]; s0 = std::min(s0, s.t0); s1 = std::max(s1, s.t1); }
        spans[k] = { s0, s1 }; a0 = std::min(a0, s0); a1 = std::max(a1, s1);
    }
    int inside = 0;                                                  // small kernels that started while an A kernel had workgroups running AND ended before that A kernel ended
    for (int i = 0; i < n_small; i++) for (auto & sp : spans) if (small[i].t0 > sp.first && small[i].t1 < sp.second) { inside++; break; }
    std::set<unsigned> cus_a, cus_b;
    for (auto & s : big) cus_a.insert((s.xcc & 0xf) << 16 | (s.hw & 0xff00)); for (int i = 0; i < n_small; i++) cus_b.insert((small[i].xcc & 0xf) << 16 | (small[i].hw & 0xff00));
    int shared = 0; for (unsigned c : cus_b) shared += cus_a.count(c);
    printf("%-44s chain B: %8.1f us for %d kernels (%.2f us per kernel, mean gap %.2f us) | A: %d kernels in %8.1f us (%.1f us each) | B kernels strictly inside an A kernel: %d / %d | "
           "distinct (xcc, se, cu) used: A %zu, B %zu, shared %d\n", name, (b1 - b0) / 100.0, n_small, (b1 - b0) / 100.0 / n_small, gap_sum / (n_small - 1) / 100.0, n_big, (a1 - a0) / 100.0,
           (a1 - a0) / 100.0 / n_big, inside, n_small, cus_a.size(), cus_b.size(), shared);
}

int main(int argc, char ** argv) {
    const int cus_b = argc > 1 ? atoi(argv[1]) : 32;
    const int n_big = 40, wg_big = 2048, n_small = 200;
    stamp * d_big, * d_small; int * d_dep;
    CK(hipMalloc(&d_big, sizeof(stamp) * n_big * wg_big)); CK(hipMalloc(&d_small, sizeof(stamp) * n_small)); CK(hipMalloc(&d_dep, 4 * (n_small + 1)));
    CK(hipMemset(d_dep, 0, 4 * (n_small + 1)));
    std::vector<stamp> big((size_t) n_big * wg_big), small(n_small);
    auto run = [&](const char * name, hipStream_t sa, hipStream_t sb, bool graph_b, bool with_a) {
        hipGraphExec_t ge = nullptr;
        if (graph_b) {
            hipGraph_t gr; CK(hipStreamBeginCapture(sb, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < n_small; i++) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, sb, d_small, i, 250, d_dep + i, d_dep + i + 1);
            CK(hipStreamEndCapture(sb, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0)); CK(hipGraphDestroy(gr));
        }
        CK(hipDeviceSynchronize());
        if (with_a) for (int k = 0; k < n_big; k++) hipLaunchKernelGGL(k_big, dim3(wg_big), dim3(256), 0, sa, d_big, k, 2000);
        if (graph_b) CK(hipGraphLaunch(ge, sb));
        else for (int i = 0; i < n_small; i++) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, sb, d_small, i, 250, d_dep + i, d_dep + i + 1);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(big.data(), d_big, sizeof(stamp) * big.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(small.data(), d_small, sizeof(stamp) * n_small, hipMemcpyDeviceToHost));
        if (with_a) summarize(name, big, n_big, wg_big, small, n_small);
        else printf("%-44s chain B alone: %8.1f us (%.2f us per kernel)\n", name, (small[n_small - 1].t1 - small[0].t0) / 100.0, (small[n_small - 1].t1 - small[0].t0) / 100.0 / n_small);
        if (ge) CK(hipGraphExecDestroy(ge));
    };
    hipStream_t a0, b0; CK(hipStreamCreateWithFlags(&a0, hipStreamNonBlocking));
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&b0, hipStreamNonBlocking, hi));
    run("B alone (eager)", a0, b0, false, false);
    run("B alone (graph)", a0, b0, true, false);
    run("(a) no masks, B high priority, eager", a0, b0, false, true);
    run("(a') no masks, B high priority, B as graph", a0, b0, true, true);
    // disjoint masks: B gets the first `cus_b` bits, A the rest of 256
    std::vector<uint32_t> ma(8, 0), mb(8, 0);
    for (int i = 0; i < 256; i++) (i < cus_b ? mb : ma)[i >> 5] |= 1u << (i & 31);
    hipStream_t a1, b1; CK(hipExtStreamCreateWithCUMask(&a1, 8, ma.data())); CK(hipExtStreamCreateWithCUMask(&b1, 8, mb.data()));
    run("(b) disjoint CU masks, eager", a1, b1, false, true);
    run("(c) disjoint CU masks, B as graph", a1, b1, true, true);
    // interleaved masks: every 8th CU bit to B (in case the bit order is not chip-linear)
    std::vector<uint32_t> mc(8, 0), md(8, 0);
    for (int i = 0; i < 256; i++) ((i % 8) == 0 ? md : mc)[i >> 5] |= 1u << (i & 31);
    hipStream_t a2, b2; CK(hipExtStreamCreateWithCUMask(&a2, 8, mc.data())); CK(hipExtStreamCreateWithCUMask(&b2, 8, md.data()));
    run("(d) every 8th CU bit to B, eager", a2, b2, false, true);
    run("(e) every 8th CU bit to B, B as graph", a2, b2, true, true);
    return 0;
}
