// multi_gpu_single_process.cpp -- one process, one thread, every GPU of the node: what a Bevy App is (one World, one schedule).
//
// configs[3] of BASELINE.json shards an entity range over the GPUs of a node and all-gathers the packed ViewVisibility masks.
// bench.py does that with one process per GPU under torch.distributed (what the scaling driver launches); a Bevy plugin cannot:
// it lives in ONE process.  This program drives the same sharded frame from one thread through the C ABI alone:
//   a context per device (mi_ctx_create), communicators from ncclCommInitAll, MI_EXCHANGE_GROUPED, and per frame
//       for every context: mi_propagate_and_cull (its row range, its block of the gathered buffer written in place)
//       mi_exchange_group_flush(contexts, ncclGroupStart, ncclGroupEnd)     <- the N all-gathers go out together
//   and checks that every rank's gathered buffer holds, block by block, the masks one unsharded context computes for the same rows.
// RCCL is loaded with dlopen (no link-time dependency of the library on it).  Runs with N = 1 on a one-GPU box (a 1-rank
// communicator: every code path but the wire); `multi_gpu_single_process [n_rows] [frames]`.  Exit code 0 = all equal.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/bevy_mi355x.h"

#define CK(call)                                                                                    \
    do {                                                                                            \
        const int32_t rc_ = (call);                                                                 \
        if (rc_ != MI_OK) { std::printf("FAILED %s -> %d (%s)\n", #call, rc_, mi_last_error_string(nullptr)); return 1; } \
    } while (0)
#define HCK(call)                                                                       \
    do {                                                                                \
        const hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) { std::printf("FAILED %s -> %s\n", #call, hipGetErrorString(e_)); return 1; } \
    } while (0)

typedef int (*fn_comm_init_all)(void** comms, int ndev, const int* devlist);
typedef int (*fn_comm_destroy)(void* comm);
typedef int (*fn_void)(void);

static uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const uint32_t n_rows = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 1000000u;
    const int frames = argc > 2 ? std::atoi(argv[2]) : 6;
    const uint32_t n_views = 4;
    int n_dev = 0;
    HCK(hipGetDeviceCount(&n_dev));
    if (n_dev < 1) { std::printf("no HIP device\n"); return 1; }
    void* rccl = nullptr;
    for (const char* path : {std::getenv("MI_RCCL_LIB") ? std::getenv("MI_RCCL_LIB") : "librccl.so", "/opt/rocm/lib/librccl.so",
                             "/usr/local/lib/python3.10/dist-packages/torch/lib/librccl.so"})
        if (!rccl) rccl = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!rccl) { std::printf("cannot load librccl.so: %s\n", dlerror()); return 1; }
    auto comm_init_all = (fn_comm_init_all)dlsym(rccl, "ncclCommInitAll");
    auto comm_destroy = (fn_comm_destroy)dlsym(rccl, "ncclCommDestroy");
    void* all_gather = dlsym(rccl, "ncclAllGather");
    void* group_start = dlsym(rccl, "ncclGroupStart");
    void* group_end = dlsym(rccl, "ncclGroupEnd");
    if (!comm_init_all || !comm_destroy || !all_gather || !group_start || !group_end) { std::printf("RCCL symbols missing\n"); return 1; }

    // ---- the scene: many_cubes-shaped rows (positions on a sphere, random rotations), 4 cameras at yaw 0 / 90 / 180 / 270
    std::vector<float> t(3 * (size_t)n_rows), r(4 * (size_t)n_rows), s(3 * (size_t)n_rows, 1.0f), c(3 * (size_t)n_rows, 0.0f), h(3 * (size_t)n_rows, 0.5f);
    uint64_t seed = 42;
    for (uint32_t i = 0; i < n_rows; ++i) {
        const double phi = std::acos(1.0 - 2.0 * (i + 0.36) / (n_rows - 1.0 + 0.72)), theta = 3.883222077450933 * i, R = 500.0;
        t[3 * (size_t)i] = (float)(R * std::cos(theta) * std::sin(phi));
        t[3 * (size_t)i + 1] = (float)(R * std::sin(theta) * std::sin(phi));
        t[3 * (size_t)i + 2] = (float)(R * std::cos(phi));
        double q[4], len = 0;
        for (double& v : q) { v = (double)(splitmix64(seed) >> 11) / 9007199254740992.0 * 2.0 - 1.0; len += v * v; }
        len = std::sqrt(len);
        for (int k = 0; k < 4; ++k) r[4 * (size_t)i + k] = (float)(q[k] / len);
    }
    float clip[16];
    mi_perspective_clip_from_view(3.14159265f / 4.0f, 16.0f / 9.0f, 0.1f, clip);
    auto frusta_of = [&](int frame, float* out /* 24 * n_views */) {
        for (uint32_t v = 0; v < n_views; ++v) {
            const float a = 0.5f * (1.5707963f * (float)v + 0.0025f * (float)frame), sy = std::sin(a), cy = std::cos(a);
            const float cam[12] = {1 - 2 * sy * sy, 0, -2 * sy * cy, 0, 1, 0, 2 * sy * cy, 0, 1 - 2 * sy * sy, 0, 0, 0};  // rotation about y
            mi_compute_frustum(clip, cam, 1000.0f, out + 24 * v);
        }
    };

    // ---- the sharded side: a context, a row range (256-aligned) and a set of gathered buffers per device
    const uint32_t N = (uint32_t)n_dev;
    const uint32_t rows_per = ((n_rows + N - 1) / N + 255u) / 256u * 256u;
    const uint64_t words_per_view = rows_per / 64u;         // one 64-bit word per wave
    const uint64_t block_bytes = n_views * words_per_view * 8;  // [rank][view][word]
    const uint32_t n_bufs = 3;
    std::vector<mi_ctx*> ctxs(N, nullptr);
    std::vector<void*> comms(N, nullptr);
    std::vector<int> devs(N);
    for (uint32_t d = 0; d < N; ++d) devs[d] = (int)d;
    if (comm_init_all(comms.data(), (int)N, devs.data()) != 0) { std::printf("ncclCommInitAll failed\n"); return 1; }
    std::vector<std::vector<void*>> bufs(N, std::vector<void*>(n_bufs, nullptr));
    std::vector<uint32_t> lo(N), cnt(N);
    for (uint32_t d = 0; d < N; ++d) {
        lo[d] = std::min(n_rows, d * rows_per);
        cnt[d] = std::min(n_rows - lo[d], rows_per);
        HCK(hipSetDevice((int)d));
        CK(mi_ctx_create((int32_t)d, nullptr, &ctxs[d]));
        CK(mi_columns_resize(ctxs[d], cnt[d]));
        if (cnt[d]) {
            CK(mi_upload_transforms(ctxs[d], 0, cnt[d], &t[3 * (size_t)lo[d]], &r[4 * (size_t)lo[d]], &s[3 * (size_t)lo[d]]));
            CK(mi_upload_bounds(ctxs[d], 0, cnt[d], &c[3 * (size_t)lo[d]], &h[3 * (size_t)lo[d]], nullptr, nullptr));
        }
        for (uint32_t b = 0; b < n_bufs; ++b) {
            HCK(hipMalloc(&bufs[d][b], N * block_bytes));
            HCK(hipMemset(bufs[d][b], 0, N * block_bytes));
        }
        CK(mi_exchange_set_mode(ctxs[d], MI_EXCHANGE_GROUPED));
        CK(mi_exchange_configure(ctxs[d], comms[d], all_gather, bufs[d].data(), n_bufs, words_per_view, (uint64_t)d * n_views * words_per_view, block_bytes, d));
    }
    // ---- the reference: the whole scene in one context on device 0
    HCK(hipSetDevice(0));
    mi_ctx* whole = nullptr;
    CK(mi_ctx_create(0, nullptr, &whole));
    CK(mi_columns_resize(whole, n_rows));
    CK(mi_upload_transforms(whole, 0, n_rows, t.data(), r.data(), s.data()));
    CK(mi_upload_bounds(whole, 0, n_rows, c.data(), h.data(), nullptr, nullptr));

    int bad = 0;
    double sharded_s = 0;
    std::vector<float> frusta(24 * n_views);
    std::vector<uint32_t> want((n_rows + 31) / 32);
    std::vector<uint64_t> gathered(N * n_views * words_per_view);
    for (int f = 0; f < frames; ++f) {
        frusta_of(f, frusta.data());
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t d = 0; d < N; ++d)  // one thread, every device: the frame calls only enqueue
            if (cnt[d]) CK(mi_propagate_and_cull(ctxs[d], frusta.data(), nullptr, nullptr, n_views, MI_CULL_END_FRAME));
        CK(mi_exchange_group_flush(ctxs.data(), N, group_start, group_end));
        for (uint32_t d = 0; d < N; ++d) {
            void* last = nullptr;
            CK(mi_exchange_last(ctxs[d], &last, 1));
        }
        sharded_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        CK(mi_propagate_and_cull(whole, frusta.data(), nullptr, nullptr, n_views, MI_CULL_END_FRAME));
        // every rank holds every rank's masks
        for (uint32_t d = 0; d < N; ++d) {
            void* last = nullptr;
            CK(mi_exchange_last(ctxs[d], &last, 1));
            HCK(hipSetDevice((int)d));
            HCK(hipMemcpy(gathered.data(), last, gathered.size() * 8, hipMemcpyDeviceToHost));
            for (uint32_t v = 0; v < n_views; ++v) {
                HCK(hipSetDevice(0));
                CK(mi_download_visibility(whole, v, want.data()));
                for (uint32_t row = 0; row < n_rows; ++row) {
                    const uint32_t rk = row / rows_per, local = row - rk * rows_per;
                    const uint64_t w = gathered[((size_t)rk * n_views + v) * words_per_view + local / 64u];
                    const bool got = (w >> (local & 63u)) & 1ull, exp = (want[row >> 5] >> (row & 31u)) & 1u;
                    if (got != exp && ++bad < 5) std::printf("  frame %d, rank %u's buffer, view %u, row %u: %d instead of %d\n", f, d, v, row, (int)got, (int)exp);
                }
            }
        }
    }
    std::printf("{\"devices\": %u, \"rows\": %u, \"views\": %u, \"frames\": %d, \"mismatches\": %d, \"sharded_ms_per_frame_incl_wait\": %.4f}\n", N, n_rows, n_views,
                frames, bad, 1e3 * sharded_s / frames);
    for (uint32_t d = 0; d < N; ++d) {
        CK(mi_exchange_configure(ctxs[d], nullptr, nullptr, nullptr, 0, 0, 0, 0, 0));
        CK(mi_ctx_destroy(ctxs[d]));
        comm_destroy(comms[d]);
        for (void* b : bufs[d]) hipFree(b);
    }
    CK(mi_ctx_destroy(whole));
    return bad ? 1 : 0;
}
