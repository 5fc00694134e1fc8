// PCIe duplex pattern probe (profiles/r03_experiments.md 12): 8 x (3 H2D copies) on an upload stream with an event behind each piece,
// a small kernel per piece on a second stream, a D2H copy per piece on a third; arguments: prior gated prio pageable ord (see main).
// hipcc --offload-arch=gfx950 -O2 -o /tmp/duplex tools/probes/pcie_duplex_pattern.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define Q(x) (void)(x)
__global__ void k_touch(const float* src, float* dst, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + 1.0f;
}
int main(int argc, char** argv) {
    const int prior = argc > 1 ? atoi(argv[1]) : 0;   // 1: s0 carries a whole upload + download before s1/s2 exist
    const int gated = argc > 2 ? atoi(argv[2]) : 0;   // 1: D2H issued by the host after hipEventSynchronize
    const int prio = argc > 3 ? atoi(argv[3]) : 0;
    const int pageable = argc > 4 ? atoi(argv[4]) : 0; const int ordm = argc > 5 ? atoi(argv[5]) : 0; // 0 stream wait, 1 host sync, 2 none // 1: a small pageable H2D on s0 per frame before the kernels, small D2H + sync after
    const size_t rows = 1110000, K = 8;
    const size_t in_bytes = rows * 40, out_bytes = rows * 48;
    char *h_in, *h_out, *d_in, *d_out, *d_small;
    Q(hipHostMalloc((void**)&h_in, in_bytes, hipHostMallocMapped)); Q(hipHostMalloc((void**)&h_out, out_bytes, hipHostMallocMapped));
    Q(hipMalloc((void**)&d_in, in_bytes)); Q(hipMalloc((void**)&d_out, out_bytes)); Q(hipMalloc((void**)&d_small, 65536));
    char small[4096] = {0};
    hipStream_t s0, s1, s2;
    Q(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    if (prior) {
        for (int i = 0; i < 3; ++i) {
            Q(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s0));
            Q(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s0));
            Q(hipStreamSynchronize(s0));
        }
    }
    if (prio) { int lo, hi; Q(hipDeviceGetStreamPriorityRange(&lo, &hi)); Q(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); Q(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo)); }
    else { Q(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); Q(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); }
    hipEvent_t up[K], fr[K], ord;
    for (size_t k = 0; k < K; ++k) { Q(hipEventCreateWithFlags(&up[k], hipEventDisableTiming)); Q(hipEventCreateWithFlags(&fr[k], hipEventDisableTiming)); }
    Q(hipEventCreateWithFlags(&ord, hipEventDisableTiming));
    auto run = [&](int mode) {
        Q(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < 5; ++it) {
            if (mode == 0) {
                Q(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s0));
                hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, s0, (const float*)d_in, (float*)d_out, (size_t)1 << 20);
                Q(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s0));
            } else {
                if (ordm == 0) { Q(hipEventRecord(ord, s0)); Q(hipStreamWaitEvent(s1, ord, 0)); } else if (ordm == 1) Q(hipStreamSynchronize(s0));
                for (size_t k = 0; k < K; ++k) {
                    const size_t lo = in_bytes * k / K, hi = in_bytes * (k + 1) / K;
                    for (int p = 0; p < 3; ++p) {
                        const size_t a = lo + (hi - lo) * p / 3, b = lo + (hi - lo) * (p + 1) / 3;
                        Q(hipMemcpyAsync(d_in + a, h_in + a, b - a, hipMemcpyHostToDevice, s1));
                    }
                    Q(hipEventRecord(up[k], s1));
                }
                if (pageable) Q(hipMemcpyAsync(d_small, small, 2048, hipMemcpyHostToDevice, s0));
                for (size_t k = 0; k < K; ++k) {
                    Q(hipStreamWaitEvent(s0, up[k], 0));
                    hipLaunchKernelGGL(k_touch, dim3(512), dim3(256), 0, s0, (const float*)d_in, (float*)d_out, (size_t)1 << 17);
                    Q(hipEventRecord(fr[k], s0));
                }
                for (size_t k = 0; k < K; ++k) {
                    const size_t lo = out_bytes * k / K, hi = out_bytes * (k + 1) / K;
                    if (gated) Q(hipEventSynchronize(fr[k])); else Q(hipStreamWaitEvent(s2, fr[k], 0));
                    Q(hipMemcpyAsync(h_out + lo, d_out + lo, hi - lo, hipMemcpyDeviceToHost, s2));
                }
                if (pageable) { Q(hipMemcpyAsync(small, d_small, 64, hipMemcpyDeviceToHost, s0)); Q(hipStreamSynchronize(s0)); Q(hipMemcpyAsync(h_out, d_out, 4400000, hipMemcpyDeviceToHost, s0)); Q(hipStreamSynchronize(s0)); Q(hipStreamSynchronize(s2)); }
            }
            Q(hipDeviceSynchronize());
        }
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 5;
    };
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 2; ++mode) if (rep) std::printf("ord %d prior %d gated %d prio %d pageable %d  mode %d: %8.1f us per frame\n", ordm, prior, gated, prio, pageable, mode, run(mode)); else run(mode);
    return 0;
}
