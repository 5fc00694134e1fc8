// hipcc --offload-arch=gfx950 -O3 tools/probes/issue_rate_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
// What one wave's instruction stream costs on MI355X (ns per instruction, by s_memrealtime at 100 MHz around long unrolled runs):
// dependent / independent v_fma_f32, s_add, an LDS pointer chase (the round trip of a dependent ds_read), a workgroup barrier among k waves,
// a dependent global load (L2 hit).  The strip kernel's (kernels_tree.hip) round costs are priced with these.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
__global__ void k_dep_fma(float* out, unsigned long long* t, float a, float b) {
    float v = out[threadIdx.x];
    const unsigned long long t0 = now();
    for (int i = 0; i < 16; ++i) { REP256(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));) }
    const unsigned long long t1 = now();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
__global__ void k_indep_fma(float* out, unsigned long long* t, float a, float b) {
    float v0 = out[threadIdx.x], v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    const unsigned long long t0 = now();
    for (int i = 0; i < 16; ++i) {
        REP16(REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));))
    }
    const unsigned long long t1 = now();
    out[threadIdx.x] = v0 + v1 + v2 + v3;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
__global__ void k_salu(unsigned* out, unsigned long long* t, unsigned a) {
    unsigned v = a;
    const unsigned long long t0 = now();
    for (int i = 0; i < 16; ++i) { REP256(asm volatile("s_add_u32 %0, %0, %1" : "+s"(v) : "s"(a) : "scc");) }
    const unsigned long long t1 = now();
    if (threadIdx.x == 0) out[blockIdx.x] = v, t[blockIdx.x] = t1 - t0;
}
__global__ void k_lds_chase(unsigned* out, unsigned long long* t) {
    __shared__ unsigned ring[1024];
    for (unsigned i = threadIdx.x; i < 1024; i += blockDim.x) ring[i] = (i * 17u + 5u) & 1023u;
    __syncthreads();
    unsigned p = threadIdx.x;
    const unsigned long long t0 = now();
    for (int i = 0; i < 1024; ++i) p = ring[p];
    const unsigned long long t1 = now();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
__global__ void k_barrier(unsigned long long* t) {
    const unsigned long long t0 = now();
    for (int i = 0; i < 1024; ++i) __builtin_amdgcn_s_barrier();
    const unsigned long long t1 = now();
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
__global__ void k_lds_handoff(unsigned* out, unsigned long long* t) {  // wave 1 writes, barrier, wave 0 reads, barrier: the strip kernel's hand-over
    __shared__ unsigned box[64];
    unsigned acc = 0;
    const unsigned long long t0 = now();
    for (int i = 0; i < 1024; ++i) {
        if (threadIdx.x >= 64) box[threadIdx.x - 64] = i + acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        if (threadIdx.x < 64) acc += box[threadIdx.x];
    }
    const unsigned long long t1 = now();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
__global__ void k_gchase(const unsigned* ring, unsigned* out, unsigned long long* t) {
    unsigned p = threadIdx.x;
    const unsigned long long t0 = now();
    for (int i = 0; i < 256; ++i) p = ring[p];
    const unsigned long long t1 = now();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
int main() {
    float* df; unsigned* du; unsigned long long* dt; unsigned* ring;
    (void)hipMalloc(&df, 4096 * 4); (void)hipMalloc(&du, 4096 * 4); (void)hipMalloc(&dt, 4096 * 8); (void)hipMalloc(&ring, 1 << 20);
    hipMemset(df, 0, 4096 * 4);
    std::vector<unsigned> hr(1 << 18);
    for (unsigned i = 0; i < hr.size(); ++i) hr[i] = (i * 4099u + 64u) & (unsigned)(hr.size() - 1);
    hipMemcpy(ring, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    auto report = [&](const char* name, int blocks, double per) {
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), dt, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        printf("%-58s %8.2f ns each\n", name, 10.0 * s / blocks / per);
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_dep_fma, dim3(1), dim3(64), 0, 0, df, dt, 1.0001f, 0.5f); report("dependent v_fma_f32, one wave on the chip", 1, 4096);
        hipLaunchKernelGGL(k_dep_fma, dim3(1024), dim3(64), 0, 0, df, dt, 1.0001f, 0.5f); report("dependent v_fma_f32, one wave per SIMD (1024 WGs of 64)", 1024, 4096);
        hipLaunchKernelGGL(k_dep_fma, dim3(1024), dim3(128), 0, 0, df, dt, 1.0001f, 0.5f); report("dependent v_fma_f32, 1024 WGs of 128", 1024, 4096);
        hipLaunchKernelGGL(k_dep_fma, dim3(1024), dim3(256), 0, 0, df, dt, 1.0001f, 0.5f); report("dependent v_fma_f32, 1024 WGs of 256 (a wave per SIMD)", 1024, 4096);
        hipLaunchKernelGGL(k_dep_fma, dim3(1024), dim3(512), 0, 0, df, dt, 1.0001f, 0.5f); report("dependent v_fma_f32, 1024 WGs of 512 (two waves per SIMD)", 1024, 4096);
        hipLaunchKernelGGL(k_indep_fma, dim3(1), dim3(64), 0, 0, df, dt, 1.0001f, 0.5f); report("independent v_fma_f32 (4 chains), one wave", 1, 16384);
        hipLaunchKernelGGL(k_salu, dim3(1), dim3(64), 0, 0, du, dt, 3u); report("dependent s_add_u32, one wave", 1, 4096);
        hipLaunchKernelGGL(k_lds_chase, dim3(1), dim3(64), 0, 0, du, dt); report("dependent ds_read_b32 (LDS round trip), one wave", 1, 1024);
        hipLaunchKernelGGL(k_lds_chase, dim3(1280), dim3(320), 0, 0, du, dt); report("dependent ds_read_b32, 5 WGs of 5 waves per CU", 1280, 1024);
        hipLaunchKernelGGL(k_barrier, dim3(1), dim3(128), 0, 0, dt); report("s_barrier, 2 waves", 1, 1024);
        hipLaunchKernelGGL(k_barrier, dim3(1), dim3(320), 0, 0, dt); report("s_barrier, 5 waves", 1, 1024);
        hipLaunchKernelGGL(k_lds_handoff, dim3(1), dim3(128), 0, 0, du, dt); report("LDS write + fence + barrier + read (hand-over), 2 waves", 1, 1024);
        hipLaunchKernelGGL(k_lds_handoff, dim3(1280), dim3(128), 0, 0, du, dt); report("hand-over, 5 WGs per CU", 1280, 1024);
        hipLaunchKernelGGL(k_gchase, dim3(1), dim3(64), 0, 0, ring, du, dt); report("dependent global_load_dword (1 MB ring), one wave", 1, 256);
        hipLaunchKernelGGL(k_gchase, dim3(1024), dim3(64), 0, 0, ring, du, dt); report("dependent global_load_dword, 1024 waves", 1024, 256);
        printf("--\n");
    }
    return 0;
}
