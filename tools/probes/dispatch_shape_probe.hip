// dispatch_shape_probe.hip -- GPU box probe behind VERDICT r05 item 1: is the metric frame's launch bound by the rate at which
// 256-thread workgroups are handed out?  A stand-in for k_frame<1,true,*>'s row path -- per row: Transform in (40 B), From(Transform),
// a five-plane sphere test, one ballot word per wave, GlobalTransform out (48 B, nontemporal, through the wave-private LDS transpose),
// the ViewVisibility byte read -- over 1.11 M rows (MALL-resident like the metric frame) and 10 M rows, launched as:
//   threads per workgroup 256 / 512 / 1024;  rows per lane 1 / 2 / 4 with every row's loads issued up front;
//   LDS per 256 threads 12 KB (8 waves per SIMD) or 32 KB (5 waves per SIMD: what the walk-carrying variant is held to);
//   a private segment reserved but never touched (what 22 spilled SGPRs leave k_frame<1,true,1> with);
//   a persistent grid (one workgroup per slot, tiles in a loop, the next tile's loads issued before this tile's arithmetic).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I bevy_amd/csrc tools/probes/dispatch_shape_probe.hip -o tools/probes/_scratch_dispatch_probe
// Output: one line per shape -- us per launch back to back (what bench.py's ms_per_step sees) and the dispatch's own duration.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#include "glam_math.h"

using namespace mi;
typedef float v4f __attribute__((ext_vector_type(4)));

struct Cols {
    const float *t, *q, *s;
    float* g;
    uint8_t* vv;
    uint64_t* mask;
    float planes[20];
    uint32_t n, magic;
};
struct F3 { float x, y, z; };

#define WAVE_LDS_SYNC()                                                 \
    do {                                                               \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
        __builtin_amdgcn_wave_barrier();                               \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
    } while (0)

struct RowIn {
    F3 t, s;
    float4 q;
    uint32_t vv;
};
__device__ __forceinline__ RowIn load_row(const Cols& c, uint32_t row) {
    RowIn r = {};
    if (row < c.n) {
        r.t = reinterpret_cast<const F3*>(c.t)[row];
        r.q = reinterpret_cast<const float4*>(c.q)[row];
        r.s = reinterpret_cast<const F3*>(c.s)[row];
        r.vv = c.vv[row];
    }
    return r;
}
__device__ __forceinline__ void process_row(const Cols& c, const RowIn& in, uint32_t row, float4* lds_wave, uint32_t lane) {
    const bool live = row < c.n;
    const uint32_t wave_row0 = row & ~63u;
    Affine a = {};
    if (live) a = affine_from_srt(V3{in.s.x, in.s.y, in.s.z}, V4{in.q.x, in.q.y, in.q.z, in.q.w}, V3{in.t.x, in.t.y, in.t.z});
    lds_wave[lane * 3u + 0u] = make_float4(a.m.x_axis.x, a.m.x_axis.y, a.m.x_axis.z, a.m.y_axis.x);
    lds_wave[lane * 3u + 1u] = make_float4(a.m.y_axis.y, a.m.y_axis.z, a.m.z_axis.x, a.m.z_axis.y);
    lds_wave[lane * 3u + 2u] = make_float4(a.m.z_axis.z, a.t.x, a.t.y, a.t.z);
    WAVE_LDS_SYNC();
    float4* dst = reinterpret_cast<float4*>(c.g) + 3ull * wave_row0;
    const uint32_t lim = wave_row0 < c.n ? (c.n - wave_row0 < 64u ? c.n - wave_row0 : 64u) * 3u : 0u;
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {
        const uint32_t i = k * 64u + lane;
        if (i < lim) {
            const float4 v = lds_wave[i];
            __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(dst) + i);
        }
    }
    // the sphere half of the rule: centre = a * (0,0,0) + t, radius = |M3 * (0.5,0.5,0.5)|
    const V3 h = mul(a.m, V3{0.5f, 0.5f, 0.5f});
    const float sr = length3(h);
    bool vis = live;
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const float d = dot4(V4{c.planes[4 * p], c.planes[4 * p + 1], c.planes[4 * p + 2], c.planes[4 * p + 3]}, V4{a.t.x, a.t.y, a.t.z, 1.0f});
        vis = vis && !(d + sr <= 0.0f);
    }
    const unsigned long long m = __ballot(vis);
    if (lane == 0 && wave_row0 < c.n) c.mask[row >> 6] = m;
    const uint32_t cur = vis ? 1u : 0u;
    if (live && cur != in.vv) c.vv[row] = (uint8_t)cur;
    WAVE_LDS_SYNC();  // (the next row of this lane reuses the transpose buffer)
}

extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];

template <int THREADS, int RPL, bool SCRATCH, bool PERSIST>
__global__ void __launch_bounds__(THREADS) k_probe(Cols c, uint32_t n_tiles) {
    float4* lds_wave = reinterpret_cast<float4*>(lds_dyn) + (threadIdx.x >> 6) * 192u;
    const uint32_t lane = threadIdx.x & 63u;
    if constexpr (SCRATCH) {
        if (c.magic == 0xdeadbeefu) {  // never true: the private segment is reserved, no scratch instruction runs
            volatile float buf[13];
            for (int i = 0; i < 13; ++i) buf[i] = (float)i;
            c.g[0] = buf[c.n % 13u];
        }
    }
    if constexpr (!PERSIST) {
        const uint32_t tile = blockIdx.x;
        RowIn in[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) in[r] = load_row(c, (tile * RPL + r) * THREADS + threadIdx.x);
#pragma unroll
        for (int r = 0; r < RPL; ++r) process_row(c, in[r], (tile * RPL + r) * THREADS + threadIdx.x, lds_wave, lane);
    } else {
        // RPL = how many tiles ahead the loads run
        RowIn in[RPL];
        uint32_t tile = blockIdx.x;
#pragma unroll
        for (int r = 0; r < RPL; ++r) in[r] = load_row(c, tile + r * gridDim.x < n_tiles ? (tile + r * gridDim.x) * THREADS + threadIdx.x : 0xFFFFFFFFu);
        for (; tile < n_tiles; tile += gridDim.x * RPL) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const uint32_t cur = tile + r * gridDim.x;
                if (cur >= n_tiles) break;
                const RowIn mine = in[r];
                const uint32_t nxt = cur + gridDim.x * RPL;
                in[r] = load_row(c, nxt < n_tiles ? nxt * THREADS + threadIdx.x : 0xFFFFFFFFu);
                process_row(c, mine, cur * THREADS + threadIdx.x, lds_wave, lane);
            }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static Cols g_c;
static hipStream_t g_stream;

template <int THREADS, int RPL, bool SCRATCH, bool PERSIST>
static void run(const char* name, uint32_t lds_per_256, uint32_t persist_wgs_per_cu = 0) {
    const uint32_t lds = lds_per_256 * (THREADS / 256);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<THREADS, RPL, SCRATCH, PERSIST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t n_tiles = (g_c.n + THREADS - 1) / THREADS;
    uint32_t grid = PERSIST ? 256u * persist_wgs_per_cu : (n_tiles + RPL - 1) / RPL;
    if (PERSIST && grid > n_tiles) grid = n_tiles;
    hipEvent_t e0, e1, k0, k1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_probe<THREADS, RPL, SCRATCH, PERSIST>), dim3(grid), dim3(THREADS), lds, g_stream, g_c, n_tiles);
    CK(hipStreamSynchronize(g_stream));
    std::vector<float> per;
    for (int rep = 0; rep < 7; ++rep) {
        CK(hipEventRecord(e0, g_stream));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k_probe<THREADS, RPL, SCRATCH, PERSIST>), dim3(grid), dim3(THREADS), lds, g_stream, g_c, n_tiles);
        CK(hipEventRecord(e1, g_stream));
        CK(hipStreamSynchronize(g_stream));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        per.push_back(ms * 10.0f);
    }
    std::sort(per.begin(), per.end());
    std::vector<float> kd;
    for (int i = 0; i < 30; ++i) {
        hipExtLaunchKernelGGL((k_probe<THREADS, RPL, SCRATCH, PERSIST>), dim3(grid), dim3(THREADS), lds, g_stream, k0, k1, 0, g_c, n_tiles);
        CK(hipStreamSynchronize(g_stream));
        float ms; CK(hipEventElapsedTime(&ms, k0, k1));
        kd.push_back(ms * 1000.0f);
    }
    std::sort(kd.begin(), kd.end());
    const double bytes = (double)g_c.n * (40.0 + 1.0 + 48.0 + 0.125);
    printf("%-44s grid %6u lds %6u  back-to-back %7.2f us (min %7.2f)  dispatch %7.2f us (min %7.2f)  %5.2f TB/s\n", name, grid, lds, per[per.size() / 2], per[0],
           kd[kd.size() / 2], kd[0], bytes / (per[per.size() / 2] * 1e-6) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&g_stream));
    for (uint32_t n : {1110000u, 10000000u}) {
        float *t, *q, *s, *g; uint8_t* vv; uint64_t* mask;
        CK(hipMalloc(&t, 12ull * n)); CK(hipMalloc(&q, 16ull * n)); CK(hipMalloc(&s, 12ull * n)); CK(hipMalloc(&g, 48ull * n));
        CK(hipMalloc(&vv, n)); CK(hipMalloc(&mask, 8ull * ((n + 63) / 64)));
        std::vector<float> ht(3ull * n), hq(4ull * n), hs(3ull * n);
        for (uint32_t i = 0; i < n; ++i) {
            ht[3 * i] = (float)(i % 1000) - 500.0f; ht[3 * i + 1] = (float)((i / 1000) % 1000) - 500.0f; ht[3 * i + 2] = -(float)(i % 97);
            hq[4 * i] = 0.0f; hq[4 * i + 1] = 0.38268343f; hq[4 * i + 2] = 0.0f; hq[4 * i + 3] = 0.92387953f;
            hs[3 * i] = hs[3 * i + 1] = hs[3 * i + 2] = 1.0f;
        }
        CK(hipMemcpy(t, ht.data(), 12ull * n, hipMemcpyHostToDevice)); CK(hipMemcpy(q, hq.data(), 16ull * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(s, hs.data(), 12ull * n, hipMemcpyHostToDevice)); CK(hipMemset(vv, 0, n));
        g_c = Cols{t, q, s, g, vv, mask, {1, 0, 0, 100, -1, 0, 0, 100, 0, 1, 0, 100, 0, -1, 0, 100, 0, 0, -1, -0.1f}, n, 0u};
        printf("---- %u rows (%.1f MB per launch)\n", n, n * 89.125 / 1e6);
        run<256, 1, false, false>("256 thr x 1 row, 12 KB LDS (8 waves/SIMD)", 12288);
        run<256, 2, false, false>("256 thr x 2 rows", 12288);
        run<256, 4, false, false>("256 thr x 4 rows", 12288);
        run<512, 1, false, false>("512 thr x 1 row", 12288);
        run<1024, 1, false, false>("1024 thr x 1 row", 12288);
        run<512, 2, false, false>("512 thr x 2 rows", 12288);
        run<256, 1, false, false>("256 thr x 1 row, 32 KB LDS (5 waves/SIMD)", 32768);
        run<256, 1, true, false>("256 thr x 1 row, 32 KB, private segment", 32768);
        run<256, 2, false, false>("256 thr x 2 rows, 32 KB", 32768);
        run<256, 2, true, false>("256 thr x 2 rows, 32 KB, private segment", 32768);
        run<256, 4, false, false>("256 thr x 4 rows, 32 KB", 32768);
        run<512, 1, false, false>("512 thr x 1 row, 32 KB / 256 thr", 32768);
        run<1024, 1, false, false>("1024 thr x 1 row, 32 KB / 256 thr", 32768);
        run<512, 2, false, false>("512 thr x 2 rows, 32 KB / 256 thr", 32768);
        run<256, 1, false, true>("persistent 256 thr, 8 WG/CU, 1 tile ahead", 12288, 8);
        run<256, 2, false, true>("persistent 256 thr, 8 WG/CU, 2 tiles ahead", 12288, 8);
        run<256, 2, false, true>("persistent 256 thr, 5 WG/CU, 2 tiles ahead", 32768, 5);
        run<256, 4, false, true>("persistent 256 thr, 5 WG/CU, 4 tiles ahead", 32768, 5);
        run<256, 2, false, true>("persistent 256 thr, 4 WG/CU, 2 tiles ahead", 32768, 4);
        CK(hipFree(t)); CK(hipFree(q)); CK(hipFree(s)); CK(hipFree(g)); CK(hipFree(vv)); CK(hipFree(mask));
    }
    return 0;
}
