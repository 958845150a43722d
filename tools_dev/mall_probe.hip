// Does it pay to warm the 256 MB Infinity Cache with the NEXT mat-vec's weights while the latency-bound small launches of a decode layer run?
//   cold   : mat-vec over rotating copies (> 600 MB in all): every launch streams from HBM (what the decode path does today)
//   hot    : the same copy every time (<= 256 MB stays in the memory-side cache)
//   pref   : [prefetch copy i][mat-vec copy i] on one stream, minus the prefetch-only loop: the mat-vec alone when its bytes were just read
//   layer  : [norm_quant][mat-vec] per iteration, against the same with the prefetch of the mat-vec's copy on a SIDE stream next to norm_quant
// build: hipcc --offload-arch=gfx950 -O2 tools_dev/mall_probe.hip -Iinclude -Lpipeinfer_amd -lpipeinfer_amd -Wl,-rpath,$PWD/pipeinfer_amd -o tools_dev/mall_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pipeinfer_ops.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char ** argv) {
    if (pi_device_init(0) != 0) { printf("init: %s\n", pi_last_error()); return 1; }
    hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    const int shapes[][2] = { { 8192, 8192 }, { 28672, 8192 }, { 8192, 28672 } };
    const int pf_wg = argc > 1 ? atoi(argv[1]) : 0;
    for (auto & sh : shapes) {
        const int M = sh[0], K = sh[1];
        const size_t nbytes = (size_t) M * (K / 256) * 144;
        std::vector<uint8_t> host(nbytes);
        for (size_t i = 0; i < nbytes; i++) host[i] = (uint8_t) (i * 2654435761u >> 13);
        for (size_t b = 0; b < nbytes; b += 144) { host[b] = 0x00; host[b + 1] = 0x2c; host[b + 2] = 0x00; host[b + 3] = 0x28; }      // finite d, dmin
        const int ncopy = (int) (700e6 / nbytes) + 1;
        std::vector<pi_weight *> ws(ncopy);
        setenv("PI_MMQ_COPIES", "0", 1);
        for (auto & w : ws) { w = pi_weight_upload(PI_Q4_K, K, M, host.data(), nullptr); if (!w) { printf("upload: %s\n", pi_last_error()); return 1; } }
        float * x = (float *) pi_malloc((size_t) K * 4), * nw = (float *) pi_malloc((size_t) K * 4), * out = (float *) pi_malloc((size_t) M * 4);
        std::vector<float> hx(K); for (int i = 0; i < K; i++) hx[i] = (float) ((i * 37) % 101 - 50) / 50.0f;
        pi_memcpy_h2d(x, hx.data(), (size_t) K * 4, nullptr); pi_memcpy_h2d(nw, hx.data(), (size_t) K * 4, nullptr);
        void * q8 = pi_malloc(pi_q8_nbytes(PI_Q8_K, K, 1));
        pi_op_norm_quantize(x, nw, nullptr, nullptr, K, 1, 1e-5f, 1, PI_Q8_K, q8, A);
        CK(hipStreamSynchronize(A));
        const int n = 300;
        auto timeit = [&](auto body) {
            for (int i = 0; i < 10; i++) body(i);
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int i = 0; i < n; i++) body(i);
            CK(hipDeviceSynchronize());
            return (now() - t0) / n * 1e6;
        };
        const double cold = timeit([&](int i) { pi_op_mul_mat(ws[i % ncopy], q8, 1, out, nullptr, A); });
        const double hot = timeit([&](int i) { pi_op_mul_mat(ws[0], q8, 1, out, nullptr, A); });
        const double pf = timeit([&](int i) { pi_op_prefetch(pi_weight_data(ws[i % ncopy]), nbytes, pf_wg, A); });
        const double pfmv = timeit([&](int i) { pi_op_prefetch(pi_weight_data(ws[i % ncopy]), nbytes, pf_wg, A); pi_op_mul_mat(ws[i % ncopy], q8, 1, out, nullptr, A); });
        const double nq = timeit([&](int i) { pi_op_norm_quantize(x, nw, nullptr, nullptr, K, 1, 1e-5f, 1, PI_Q8_K, q8, A); });
        const double layer = timeit([&](int i) { pi_op_norm_quantize(x, nw, nullptr, nullptr, K, 1, 1e-5f, 1, PI_Q8_K, q8, A); pi_op_mul_mat(ws[i % ncopy], q8, 1, out, nullptr, A); });
        // side-stream prefetch of a part of the matrix next to norm_quant: B waits for the previous mat-vec (event on A), A never waits for B
        hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
        printf("Q4_K M=%5d K=%5d (%6.1f MB): mat-vec cold %6.2f us (%4.0f GB/s) | same copy %6.2f | prefetch alone %6.2f (%4.0f GB/s) | prefetch+mat-vec %6.2f -> mat-vec after prefetch %6.2f | norm_quant %5.2f | norm_quant+mat-vec %6.2f\n",
               M, K, nbytes / 1e6, cold, nbytes / cold / 1e3, hot, pf, nbytes / pf / 1e3, pfmv, pfmv - pf, nq, layer);
        for (double frac : { 0.125, 0.25, 0.5, 1.0 }) for (int wg : { 64, 256, 0 }) {
            const size_t part = (size_t) (nbytes * frac) / 4096 * 4096;
            const double t = timeit([&](int i) {
                CK(hipEventRecord(ev[i & 1], A));                                   // the previous mat-vec has finished
                CK(hipStreamWaitEvent(B, ev[i & 1], 0));
                pi_op_norm_quantize(x, nw, nullptr, nullptr, K, 1, 1e-5f, 1, PI_Q8_K, q8, A);
                pi_op_prefetch(pi_weight_data(ws[i % ncopy]), part, wg, B);
                pi_op_mul_mat(ws[i % ncopy], q8, 1, out, nullptr, A);
            });
            printf("    side-stream prefetch of %4.0f %% (%6.1f MB, %4d workgroups) next to norm_quant: %6.2f us per [norm_quant + mat-vec]  (%+.2f)\n", frac * 100, part / 1e6, wg, t, t - layer);
        }
        for (auto w : ws) pi_weight_free(w);
        pi_free(x); pi_free(nw); pi_free(out); pi_free(q8);
    }
    return 0;
}
