// kernels_cells.hip -- the STATIC CULL ORDER: build of the cell-ordered copy of a static scene (the frame kernel that uses it,
// k_frame_cells, lives with the other frame kernels in kernels_flat.hip).
//
// check_visibility_cpu_culling (crates/bevy_camera/src/visibility/mod.rs:748-876) tests every entity against every view; the
// world-sphere path (k_frame_sph) already reduces a static row to 16 bytes, but it still touches every row of every frame, and --
// table order carries no spatial coherence (many_cubes' Fibonacci order puts 64 consecutive rows on a ring around the whole sphere)
// -- any survivor in a wave pulls the wave's GlobalTransforms: 588 MB per frame at 10 M rows x 4 views for a 237 MB model.
// When the scene has been static for a few frames the library therefore keeps, besides the columns, a copy IN CELL ORDER:
//   perm[slot]         the row in that slot: rows sorted by the 30-bit Morton code of their world-sphere centre
//   sph_s / g_s / vv_s the rows' world sphere (16 B), GlobalTransform (48 B) and ViewVisibility byte, in slot order
//   per 64 slots       a bounding sphere of the 64 world spheres, the flags / RenderLayers / half extents where the slots agree,
//                      and whether every one of them may be rejected by a frustum test at all (cell summary, below)
//   state              per 64 slots: every ViewVisibility byte is zero
// A wave of k_frame_cells owns 64 slots: it tests the bounding sphere against each view's five planes first and REJECTS the view
// (or the whole wave: 36 bytes read, nothing written) when the sphere lies behind a plane by more than an explicit f32 margin --
// never the other way round: a wave that is not rejected runs the reference's per-row rule unchanged.  Results are written BY ROW
// (atomicOr into the row-ordered masks, which the launch before zeroed), so nothing downstream knows about the order.
//
// The build is rare (a scene that has gone quiet pays it once): bounds of the centres, Morton keys, one radix sort of (key, row)
// pairs -- rocPRIM's, a library sort for a library-shaped job -- and a gather pass that writes the copies and the summaries.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "glam_math.h"
#include "kernels.h"

namespace mi {

// ---- bounds of the sphere centres: min / max per axis as order-preserving integers ----
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ordered_to_float(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}
__device__ __forceinline__ bool finite3(float x, float y, float z) {
    return (__float_as_uint(x) & 0x7F800000u) != 0x7F800000u && (__float_as_uint(y) & 0x7F800000u) != 0x7F800000u &&
           (__float_as_uint(z) & 0x7F800000u) != 0x7F800000u;
}

// minmax[0..2] = min x, y, z; [3..5] = max (ordered encoding; initialised to 0xFFFFFFFF / 0 by the host)
__global__ void __launch_bounds__(256) k_cells_bounds(const float4* __restrict__ sph, uint32_t n, uint32_t* minmax) {
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 s = sph[i];
        if (!finite3(s.x, s.y, s.z)) continue;  // (such a row sorts first and keeps its cell from ever being rejected)
        const uint32_t o[3] = {float_to_ordered(s.x), float_to_ordered(s.y), float_to_ordered(s.z)};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = o[k] < lo[k] ? o[k] : lo[k];
            hi[k] = o[k] > hi[k] ? o[k] : hi[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (uint32_t off = 32u; off; off >>= 1) {
            const uint32_t a = __shfl_xor(lo[k], off, 64), b = __shfl_xor(hi[k], off, 64);
            lo[k] = a < lo[k] ? a : lo[k];
            hi[k] = b > hi[k] ? b : hi[k];
        }
    }
    // one set of atomics per workgroup (the launch is at most 256 of them): every wave's on the same six words took 283 us at 1 M rows
    __shared__ uint32_t red[4][6];
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            red[threadIdx.x >> 6][k] = lo[k];
            red[threadIdx.x >> 6][3 + k] = hi[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6u) {
        const uint32_t k = threadIdx.x;
        uint32_t v = red[0][k];
        for (uint32_t wv = 1; wv < 4u; ++wv) v = k < 3u ? (red[wv][k] < v ? red[wv][k] : v) : (red[wv][k] > v ? red[wv][k] : v);
        if (k < 3u) atomicMin(&minmax[k], v);
        else atomicMax(&minmax[k], v);
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ void __launch_bounds__(256) k_cells_keys(const float4* __restrict__ sph, uint32_t n, const uint32_t* __restrict__ minmax,
                                                     uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 s = sph[i];
    uint32_t key = 0u;
    if (finite3(s.x, s.y, s.z)) {
        const float p[3] = {s.x, s.y, s.z};
        uint32_t q[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float lo = ordered_to_float(minmax[k]), hi = ordered_to_float(minmax[3 + k]);
            const float ext = hi - lo;
            float f = ext > 0.0f ? (p[k] - lo) / ext * 1024.0f : 0.0f;
            f = f < 0.0f ? 0.0f : (f > 1023.0f ? 1023.0f : f);
            q[k] = (uint32_t)f;
        }
        key = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    }
    keys[i] = key;
    vals[i] = i;
}

// ---- the gather: slot order copies and the per-64-slot summaries ----
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (uint32_t off = 32u; off; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (uint32_t off = 32u; off; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ bool wave_uniform(uint32_t v, bool live) {
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);  // (lane 0 is live in every wave that has a live slot)
    return __ballot(live && v != first) == 0ull;
}

__global__ void __launch_bounds__(256) k_cells_gather(Columns c, const float4* __restrict__ sph, const uint32_t* __restrict__ sorted_rows,
                                                       CellsOrder o) {
    __shared__ float4 lds_g[4][192];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * 4u + wv;
    if (w >= o.n_waves) return;
    const uint32_t slot = w * 64u + lane;
    const bool live = slot < c.n;
    const uint32_t row = live ? sorted_rows[slot] : 0xFFFFFFFFu;
    const uint32_t rrow = live ? row : sorted_rows[w * 64u];  // (dead lanes repeat the wave's first row: they agree with it)
    o.perm[slot] = row;
    const float4 sp = sph[rrow];
    const float4* gsrc = reinterpret_cast<const float4*>(c.global) + 3ull * rrow;
    const float4 ga = gsrc[0], gb = gsrc[1], gc = gsrc[2];
    const uint32_t vv = live ? c.view_visibility[rrow] : 0u;
    const uint32_t fl = c.flags[rrow], lm = c.layer_mask[rrow];
    const uint32_t lm_hi = c.layer_mask_hi ? c.layer_mask_hi[rrow] : 0u;
    const float hx = c.aabb_half[3ull * rrow], hy = c.aabb_half[3ull * rrow + 1u], hz = c.aabb_half[3ull * rrow + 2u];
    o.sph_s[slot] = sp;
    o.vv_s[slot] = (uint8_t)vv;
    o.pass_s[slot] = 0u;  // (the first frame over the order starts from zeroed masks)
    // GlobalTransforms: three contiguous 1 KB rows per wave through the wave's own LDS transpose
    float4* lds_wave = lds_g[wv];
    lds_wave[lane * 3u] = ga;
    lds_wave[lane * 3u + 1u] = gb;
    lds_wave[lane * 3u + 2u] = gc;
    MI_WAVE_LDS_SYNC();
    float4* gdst = reinterpret_cast<float4*>(o.g_s) + 192ull * w;
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) gdst[k * 64u + lane] = lds_wave[k * 64u + lane];

    // bounding sphere of the wave's world spheres: centre = middle of the centres' box, radius = max(|c_i - centre| + r_i), rounded
    // UP generously (the frame kernel's margin covers the rest); NaN / Inf anywhere makes the radius NaN: never rejected
    const bool fin = finite3(sp.x, sp.y, sp.z) && (__float_as_uint(sp.w) & 0x7F800000u) != 0x7F800000u;
    const bool all_fin = __ballot(!fin) == 0ull;
    const float cx = 0.5f * (wave_min(sp.x) + wave_max(sp.x)), cy = 0.5f * (wave_min(sp.y) + wave_max(sp.y)), cz = 0.5f * (wave_min(sp.z) + wave_max(sp.z));
    const float dx = sp.x - cx, dy = sp.y - cy, dz = sp.z - cz;
    const float reach = sqrtf(dx * dx + dy * dy + dz * dz) + fabsf(sp.w);
    float R = wave_max(reach) * 1.0001f + 1e-6f * (fabsf(cx) + fabsf(cy) + fabsf(cz));
    if (!all_fin) R = __uint_as_float(0x7FC00000u);
    // what the slots agree on
    const bool uni_flags = wave_uniform(fl, live) && wave_uniform(lm, live) && __ballot(live && lm_hi != 0u) == 0ull;
    const bool uni_half = wave_uniform(__float_as_uint(hx), live) && wave_uniform(__float_as_uint(hy), live) && wave_uniform(__float_as_uint(hz), live);
    // rejectable: every row is in the cull query (no NoCpuCulling), has bounds (Aabb or Sphere) and no NoFrustumCulling -- then
    // "its sphere is outside a view's frustum" means "not visible in that view" (visibility/mod.rs:823-843)
    const bool cullable = !(fl & 0x10u) && (fl & (0x04u | 0x08u)) != 0u && !(fl & 0x02u);
    const bool rejectable = __ballot(live && !cullable) == 0ull;
    const bool vv_zero = __ballot(live && vv != 0u) == 0ull;
    if (lane == 0u) {
        o.sum_a[w] = make_float4(cx, cy, cz, R);
        o.sum_b[w] = make_uint4((fl & 0xFFu) | (uni_flags ? CELLS_UNIFORM_FLAGS : 0u) | (uni_half ? CELLS_UNIFORM_HALF : 0u) | (rejectable ? CELLS_REJECTABLE : 0u),
                                lm, 0u, 0u);
        o.sum_h[w] = make_float4(hx, hy, hz, 0.0f);
        o.state[w] = vv_zero ? 1u : 0u;
    }
}

size_t cells_sort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    uint32_t* nul = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, (size_t)n, 0u, 30u, (hipStream_t) nullptr) != hipSuccess) return 0;
    return bytes;
}

hipError_t launch_cells_build(const Columns& c, const float* sph, const CellsOrder& o, uint32_t* minmax, uint32_t* keys_a, uint32_t* keys_b,
                              uint32_t* vals_a, uint32_t* vals_b, void* sort_temp, size_t sort_temp_bytes, hipStream_t stream) {
    if (c.n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(minmax, 0xFF, 12, stream);  // min x, y, z = 0xFFFFFFFF
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(minmax + 3, 0, 12, stream)) != hipSuccess) return e;  // max x, y, z = 0
    const float4* s4 = reinterpret_cast<const float4*>(sph);
    const uint32_t blocks = (c.n + 255u) / 256u;
    hipLaunchKernelGGL(k_cells_bounds, dim3(blocks < 256u ? blocks : 256u), dim3(256), 0, stream, s4, c.n, minmax);
    hipLaunchKernelGGL(k_cells_keys, dim3(blocks), dim3(256), 0, stream, s4, c.n, (const uint32_t*)minmax, keys_a, vals_a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(sort_temp, sort_temp_bytes, keys_a, keys_b, vals_a, vals_b, (size_t)c.n, 0u, 30u, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cells_gather, dim3((o.n_waves + 3u) / 4u), dim3(256), 0, stream, c, s4, (const uint32_t*)vals_b, o);
    return hipGetLastError();
}

}  // namespace mi
